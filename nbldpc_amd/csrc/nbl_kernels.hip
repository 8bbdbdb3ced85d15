// nbldpc_amd/csrc/nbl_kernels.hip -- HIP kernels of the NB-LDPC message-passing iteration for gfx950.
//
// One iteration of the reference's flooding schedule (NBLDPC.cpp:673-768 / 805-919 / 973-1131) is three launches:
//   vn_kernel   one wave per (codeword, variable): a-posteriori sum, hard decision, variable-to-check messages
//   syn_kernel  one wave per codeword: syndrome over GF(q), first-zero-syndrome freeze of the output
//   cn_*_kernel one wave per (codeword, check): check-node update (EMS / T-EMS / log-QSPA)
// Everything is FP64 and every floating-point expression keeps the reference's association order; the file
// is compiled with -ffp-contract=off so no multiply-add is fused.
#include <hip/hip_runtime.h>
#include "nbl_device.h"
#include "nbl_kernels.h"

// ---------------------------------------------------------------------------------------------------------
// init: host/API layout [B][N][q-1] -> padded [B][N][q] with slot 0 = 0.0; clear c2v; v2c = L_ch
// (NBLDPC.cpp:649-662 / 781-794 / 934-969)
// ---------------------------------------------------------------------------------------------------------
// write_v2c bit 0: also v2c = L_ch (damped methods read it); bit 1: leave c2v alone (fused iterations read the zeros of
// iteration 0 from a buffer of their own)
__global__ void init_kernel(const double *__restrict__ Lin, NblGraphDev g, NblWork w, int B, int write_v2c)
{
	const int q = g.q;
	long long total = (long long)B * g.N * q;
	for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
		int a = (int)(i % q);
		long long bn = i / q;
		if (Lin) w.Lch[i] = a ? Lin[bn * (q - 1) + (a - 1)] : 0.0; // else: demod_kernel has already filled Lch
	}
	long long etotal = (long long)B * g.E * q;
	if (!(write_v2c & 2))
		for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < etotal; i += (long long)gridDim.x * blockDim.x)
			w.c2v[i] = 0.0;
	if (write_v2c & 1) {
		for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < etotal; i += (long long)gridDim.x * blockDim.x) {
			int a = (int)(i % q);
			long long be = i / q;
			int e = (int)(be % g.E);
			long long b = be / g.E;
			// variable of edge e: binary search in voff
			int lo = 0, hi = g.N;
			while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (g.voff[mid] <= e) lo = mid; else hi = mid; }
			w.v2c[i] = a ? (Lin ? Lin[(b * g.N + lo) * (q - 1) + (a - 1)] : w.Lch[(b * g.N + lo) * q + a]) : 0.0;
		}
	}
	for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (long long)gridDim.x * blockDim.x) {
		w.done[i] = 0;
		w.iters[i] = 0;
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) *w.n_done = 0;
}

// ---------------------------------------------------------------------------------------------------------
// soft demodulator: received samples -> symbol LLRs, written straight into the padded Lch layout.
// Expression order is CComm::Demodulate's (Comm.cpp:356, :364-378 for BPSK; :394-395 for q-ary constellations).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void demod_kernel(const double *__restrict__ rx, int L, double sigma_n, int mod_order, int p,
                                                    const double *__restrict__ cons, const int *__restrict__ src, NblGraphDev g,
                                                    NblWork w, int B)
{
	const int lane = lane_id();
	const long long node = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	if (node >= (long long)B * g.N) return;
	const int b = (int)(node / g.N), n = (int)(node % g.N), q = g.q;
	double *dst = w.Lch + ((size_t)b * g.N + n) * q;
	const double *r = rx + (size_t)b * L * 2;
	if (mod_order == 2) {
		double bl[8];
		for (int k = 0; k < 8; k++) {
			const int sidx = (k < p) ? src[n * p + k] : -1;
			bl[k] = (sidx < 0) ? 0.0 : -2 * r[2 * sidx] / (sigma_n * sigma_n);
		}
		for (int a = lane; a < q; a += 64) {
			double acc = 0;
#pragma unroll
			for (int k = 0; k < 8; k++)
				if (k < p && (a & (1 << k)) != 0) acc += bl[k];
			dst[a] = a ? acc : 0.0;
		}
	} else {
		const int sidx = src[n];
		const double c0r = cons[0], c0i = cons[1];
		const double re = sidx < 0 ? 0.0 : r[2 * sidx], im = sidx < 0 ? 0.0 : r[2 * sidx + 1];
		for (int a = lane; a < q; a += 64) {
			const double cr = cons[2 * a], ci = cons[2 * a + 1];
			const double num = (2 * re - c0r - cr) * (cr - c0r) + (2 * im - c0i - ci) * (ci - c0i);
			dst[a] = (a == 0 || sidx < 0) ? 0.0 : num / (2 * sigma_n * sigma_n);
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// variable node: L_post = L_ch + c2v_0 + c2v_1 ... (in that order), decision, v2c = L_post - c2v (+ damping)
// (NBLDPC.cpp:676-691, 718-744 | 808-823, 848-857 | 977-992, 1029-1052)
// ---------------------------------------------------------------------------------------------------------
template <int Q, bool DAMP>
__global__ __launch_bounds__(256) void vn_kernel(NblGraphDev g, NblWork w, NblRun r)
{
	constexpr int NS = Fld<Q>::NS;
	const int lane = lane_id();
	const long long node = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	if (node >= (long long)r.B * g.N) return;
	const int b = nbl_codeword(w, r, (int)(node / g.N)), n = (int)(node % g.N);
	if (b < 0) return;
	if (!r.fixed_iters && w.done[b]) return;

	const int e0 = g.voff[n], dv = g.voff[n + 1] - e0;
	const double *L = w.Lch + ((size_t)b * g.N + n) * Q;
	const double *C = w.c2v + (size_t)b * g.E * Q;
	double *V = w.v2c + ((size_t)b * g.E + e0) * Q;

	double post[NS];
#pragma unroll
	for (int i = 0; i < NS; i++) {
		int a = lane + 64 * i;
		post[i] = (a < Q) ? L[a] : 0.0;
	}
	for (int d = 0; d < dv; d++) {
		const double *Cd = C + (size_t)g.v_cpos[e0 + d] * Q;
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int a = lane + 64 * i;
			if (a < Q) post[i] = post[i] + Cd[a];
		}
	}
	int dec = wave_decide<NS>(post, lane, Q);
	if (lane == 0) w.dec[(size_t)b * g.N + n] = dec;
	if (w.post) {
		double *P = w.post + ((size_t)b * g.N + n) * Q;
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int a = lane + 64 * i;
			if (a < Q) P[a] = post[i];
		}
	}
	for (int d = 0; d < dv; d++) {
		const double *Cd = C + (size_t)g.v_cpos[e0 + d] * Q;
		double *Vd = V + (size_t)d * Q;
		double nv[NS];
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int a = lane + 64 * i;
			nv[i] = (a < Q) ? post[i] - Cd[a] : 0.0;
		}
		if (DAMP) {
			double ov[NS];
#pragma unroll
			for (int i = 0; i < NS; i++) {
				int a = lane + 64 * i;
				ov[i] = (a < Q) ? Vd[a] : 0.0;
			}
			int before = wave_decide<NS>(ov, lane, Q);
			int after = wave_decide<NS>(nv, lane, Q);
			if (before != after) {
#pragma unroll
				for (int i = 0; i < NS; i++) nv[i] = __dadd_rn(__dmul_rn(r.damp_old, ov[i]), __dmul_rn(r.damp_new, nv[i]));
			}
		}
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int a = lane + 64 * i;
			if (a < Q) Vd[a] = (a == 0) ? 0.0 : nv[i];
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// The same variable-node pass for small fields (q <= 32): 64 / q variables per wave, lane = (variable of the wave, symbol).
// One variable per wave leaves 48 of 64 lanes idle at q = 16 and makes the launch a million tiny waves; the arithmetic per
// variable is unchanged (same sums in the same order, same decision rule), only the reductions run inside groups of q lanes.
// ---------------------------------------------------------------------------------------------------------
template <int Q> __device__ __forceinline__ double group_max(double v)
{
#pragma unroll
	for (int mk = 1; mk < Q; mk <<= 1) v = dmax(v, __shfl_xor(v, mk, 64));
	return v;
}
// DecideLLRVector (:1542-1562) inside a group of Q lanes: lowest symbol among the maxima of {0, v}; v of symbol 0 must be <= 0
template <int Q> __device__ __forceinline__ int group_decide(double v, int sub)
{
	const double mx = dmax(group_max<Q>(v), 0.0);
	const uint64_t hit = __ballot(v == mx);
	const unsigned grp = (unsigned)((hit >> (sub * Q)) & ((Q == 32) ? 0xffffffffull : ((1ull << Q) - 1)));
	return (mx > 0.0 && grp) ? __builtin_ctz(grp) : 0;
}

template <int Q, bool DAMP>
__global__ __launch_bounds__(256) void vn_packed_kernel(NblGraphDev g, NblWork w, NblRun r)
{
	constexpr int G = 64 / Q;
	const int lane = lane_id(), sub = lane / Q, a = lane % Q;
	const long long node = ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * G + sub;
	int b = -1, n = 0;
	if (node < (long long)r.B * g.N) {
		b = nbl_codeword(w, r, (int)(node / g.N));
		n = (int)(node % g.N);
		if (b >= 0 && !r.fixed_iters && w.done[b]) b = -1;
	}
	const bool live = b >= 0;
	const int bb = live ? b : 0; // idle groups compute on codeword 0 / variable 0 and store nothing
	const int e0 = g.voff[live ? n : 0], dv = live ? g.voff[n + 1] - e0 : 0;
	const double *L = w.Lch + ((size_t)bb * g.N + (live ? n : 0)) * Q;
	const double *C = w.c2v + (size_t)bb * g.E * Q;
	double *V = w.v2c + ((size_t)bb * g.E + e0) * Q;

	double post = L[a];
	for (int d = 0; d < g.maxdv; d++)
		if (d < dv) post = post + C[(size_t)g.v_cpos[e0 + d] * Q + a];
	const int dec = group_decide<Q>(post, sub);
	if (live && a == 0) w.dec[(size_t)b * g.N + n] = dec;
	if (live && w.post) w.post[((size_t)b * g.N + n) * Q + a] = post;
	for (int d = 0; d < g.maxdv; d++) {
		const bool on = d < dv; // (the group reductions below need every lane of the wave, so the loop bound is the wave's)
		const size_t co = on ? (size_t)g.v_cpos[e0 + d] * Q + a : (size_t)a;
		double nv = post - C[co];
		if (DAMP) {
			const double ov = on ? V[(size_t)d * Q + a] : 0.0;
			const int before = group_decide<Q>(ov, sub), after = group_decide<Q>(nv, sub);
			const double blend = __dadd_rn(__dmul_rn(r.damp_old, ov), __dmul_rn(r.damp_new, nv));
			nv = (before != after) ? blend : nv;
		}
		if (on) V[(size_t)d * Q + a] = (a == 0) ? 0.0 : nv;
	}
}

// ---------------------------------------------------------------------------------------------------------
// syndrome + output freeze: one wave per codeword, lanes over checks (NBLDPC.cpp:693-715 / 826-846 / 999-1026)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void syn_kernel(NblGraphDev g, NblWork w, NblRun r)
{
	const int lane = lane_id();
	const int b = nbl_codeword(w, r, blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
	if (b < 0) return;
	const int frozen = w.done[b];
	if (frozen) return;
	const int *dec = w.dec + (size_t)b * g.N;
	int bad = 0;
	for (int m = lane; m < g.M; m += 64) {
		int s = 0;
		for (int ce = g.coff[m]; ce < g.coff[m + 1]; ce++) s ^= g.mul[g.c_h[ce] * g.q + dec[g.c_var[ce]]];
		bad |= (s != 0);
	}
	const int ok = (__ballot(bad) == 0ull);
	int *out = w.out + (size_t)b * g.N;
	for (int n = lane; n < g.N; n += 64) out[n] = dec[n];
	if (lane == 0) {
		if (ok) {
			w.done[b] = 1;
			w.iters[b] = r.iter;
			atomicAdd(w.n_done, 1);
		} else {
			w.iters[b] = r.iter;
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// EMS check node (NBLDPC.cpp:859-917 with SortLLRVector :1715-1746 and ConstructConf :1748-1786)
//
// The reference enumerates conf(q,1) U conf(nm,nc) by recursion for every output edge.  Here the same maximum
// over the same configuration set is computed by max-plus dynamic programming over the other edges in index
// order, so every candidate value is the same left-to-right sum ((x1+x2)+x3).. (x -> fl(x+c) is monotone, so
// max and the rounded add commute).  What is NOT reproduced is the reference's running add-then-subtract
// residue.
// ---------------------------------------------------------------------------------------------------------

// Top-nm selection under SortLLRVector's order: value descending, among equal values the HIGHER symbol first.
// Wave-level quickselect on ballots: candidate sets live in scalar registers.  Returns member masks per slot and
// the rank-0 element (value + symbol).
template <int NS>
__device__ __forceinline__ void select_top(const double (&v)[NS], int lane, int q, int nm, uint64_t (&member)[NS],
                                           double &top_v, int &top_a)
{
	uint64_t valid[NS], cand[NS];
#pragma unroll
	for (int i = 0; i < NS; i++) {
		valid[i] = __ballot(lane + 64 * i < q);
		cand[i] = valid[i];
		member[i] = valid[i];
	}
	if (nm < q) {
		for (int guard = 0; guard < 4 * 64 + 8; guard++) {
			// pivot: first remaining candidate
			double pv = 0.0;
			int pa = 0;
			bool found = false;
#pragma unroll
			for (int i = 0; i < NS; i++) {
				if (!found && cand[i]) {
					int pl = __builtin_ctzll(cand[i]);
					pv = read_lane_f64(v[i], pl);
					pa = pl + 64 * i;
					found = true;
				}
			}
			if (!found) break; // cannot happen: the nm-th element is always a candidate
			uint64_t gt[NS];
			int G = 1;
#pragma unroll
			for (int i = 0; i < NS; i++) {
				int a = lane + 64 * i;
				gt[i] = __ballot(a < q && (v[i] > pv || (v[i] == pv && a > pa)));
				G += __popcll(gt[i]);
			}
			const int ps = pa >> 6;
			const uint64_t pbit = 1ull << (pa & 63);
			if (G == nm) {
#pragma unroll
				for (int i = 0; i < NS; i++) member[i] = gt[i] | ((i == ps) ? pbit : 0ull);
				break;
			}
			if (G > nm) {
#pragma unroll
				for (int i = 0; i < NS; i++) cand[i] &= gt[i];
			} else {
#pragma unroll
				for (int i = 0; i < NS; i++) cand[i] &= ~gt[i] & ~((i == ps) ? pbit : 0ull);
			}
		}
	}
	// rank 0: maximum under the same order
	double bv = -__builtin_huge_val();
	int ba = -1;
#pragma unroll
	for (int i = 0; i < NS; i++) {
		int a = lane + 64 * i;
		if (a < q && (v[i] > bv || (v[i] == bv && a > ba))) { bv = v[i]; ba = a; }
	}
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) {
		double ob = __shfl_xor(bv, off, 64);
		int oa = __shfl_xor(ba, off, 64);
		if (ob > bv || (ob == bv && oa > ba)) { bv = ob; ba = oa; }
	}
	top_v = uniform_f64(bv);
	top_a = uniform(ba);
}

struct EmsLds {
	double *U;   // [dc][Q]   check-domain input vectors: U[j][h_j*a] = v2c_j[a], U[j][0] = 0
	double *lv;  // [dc][nm]  values of the nm most reliable entries (rank 0 first)
	int *lt;     // [dc][nm]  their check-domain symbols
	double *A;   // [layers][Q] DP ping
	double *Bq;  // [layers][Q] DP pong
	double *Sv;  // [Q]       final configuration-set maxima of the current output edge
};

template <int Q>
__global__ __launch_bounds__(64) void cn_ems_kernel(NblGraphDev g, NblWork w, NblRun r, int layers)
{
	constexpr int NS = Fld<Q>::NS;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int lane = lane_id();
	const int b = nbl_codeword(w, r, blockIdx.x / g.M), m = blockIdx.x % g.M;
	if (b < 0) return;
	if (!r.fixed_iters && w.done[b]) return;
	const int c0 = g.coff[m], dc = g.coff[m + 1] - c0;
	const int nm = r.nm;
	// debug stamps (diagnostic runs only): cycles per section, summed over sampled blocks
	unsigned long long st_t0 = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	const bool st_on = (w.stamps != nullptr) && ((blockIdx.x & 63) == 0);
#define STAMP(i) do { if (st_on) { unsigned long long t1_ = clock64(); st_acc[i] += t1_ - st_t0; st_t0 = t1_; } } while (0)
	if (st_on) st_t0 = clock64();

	EmsLds s;
	s.U = (double *)smem;
	s.A = s.U + g.maxdc * Q;
	s.Bq = s.A + layers * Q;
	s.Sv = s.Bq + layers * Q;
	s.lv = s.Sv + Q;
	s.lt = (int *)(s.lv + g.maxdc * nm);

	const double *V = w.v2c + (size_t)b * g.E * Q;
	double *C = w.c2v + ((size_t)b * g.E + c0) * Q;

	// ---- stage the dc incoming vectors: permute into the check domain, select the nm best -----------------
	for (int j = 0; j < dc; j++) {
		const double *Vj = V + (size_t)g.c_epos[c0 + j] * Q;
		GfMul<Q> mh;
		mh.init(g.c_h[c0 + j], g.poly, lane);
		double v[NS];
		int t[NS];
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int a = lane + 64 * i;
			v[i] = (a < Q && a > 0) ? Vj[a] : 0.0;
			t[i] = mh.at_slot(i);
			if (a < Q) s.U[j * Q + t[i]] = v[i];
		}
		uint64_t member[NS];
		double top_v;
		int top_a;
		STAMP(0);
		select_top<NS>(v, lane, Q, nm, member, top_v, top_a);
		STAMP(1);
		// compact the members into the list, rank 0 swapped to the front
		int base = 0, p0 = 0;
#pragma unroll
		for (int i = 0; i < NS; i++) {
			if (i == (top_a >> 6)) p0 = base + __popcll(member[i] & ((1ull << (top_a & 63)) - 1ull));
			base += __popcll(member[i]);
		}
		base = 0;
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int a = lane + 64 * i;
			if ((member[i] >> lane) & 1ull) {
				int pos = base + prefix_count(member[i]);
				if (a == top_a) pos = 0;
				else if (pos == 0) pos = p0;
				s.lv[j * nm + pos] = v[i];
				s.lt[j * nm + pos] = t[i];
			}
			base += __popcll(member[i]);
		}
		STAMP(2);
	}
	__syncthreads();

	// ---- one output edge at a time ---------------------------------------------------------------------------
	for (int x = 0; x < dc; x++) {
		// the other edges in index order: OTH(l) = l-th edge != x
#define OTH(l) ((l) + ((l) >= x ? 1 : 0))
		const int rn = dc - 1;
		int zall = 0;
		for (int l = 0; l < rn; l++) zall ^= s.lt[OTH(l) * nm];

		double S[NS];
#pragma unroll
		for (int i = 0; i < NS; i++) S[i] = -NBL_DBL_MAX;

		// conf(q,1): at most one edge deviates, to ANY symbol (:894)
		for (int pi = 0; pi < rn; pi++) {
			const int jd = OTH(pi);
			const int shift = zall ^ s.lt[jd * nm];
#pragma unroll
			for (int i = 0; i < NS; i++) {
				int sym = lane + 64 * i;
				if (sym < Q) {
					double u = s.U[jd * Q + (sym ^ shift)];
					double acc = 0.0;
					for (int l = 0; l < rn; l++) acc = acc + ((l == pi) ? u : s.lv[OTH(l) * nm]);
					S[i] = dmax(S[i], acc);
				}
			}
		}

		STAMP(3);
		// conf(nm,nc): at most nc edges deviate, each inside its nm best (:897); conf(nm,0) is the all-rank-0 configuration alone,
		// which conf(q,1) already holds
		if (r.nc < 1) {
		} else if (layers == 1) {
			// nc >= dc-1: no deviation counting needed -> plain truncated max-plus convolution
			double *A = s.A, *Bq = s.Bq;
			__syncthreads();
			for (int sidx = lane; sidx < Q; sidx += 64) A[sidx] = NBL_NEG_INF;
			__syncthreads();
			if (rn == 1) {
				const int j1 = OTH(0);
				for (int k = lane; k < nm; k += 64) A[s.lt[j1 * nm + k]] = 0.0 + s.lv[j1 * nm + k];
			} else {
				const int j1 = OTH(0), j2 = OTH(1);
				for (int idx = lane; idx < nm * nm; idx += 64) {
					int k1 = idx / nm, k2 = idx - k1 * nm;
					double val = (0.0 + s.lv[j1 * nm + k1]) + s.lv[j2 * nm + k2];
					__hip_atomic_fetch_max(&A[s.lt[j1 * nm + k1] ^ s.lt[j2 * nm + k2]], val, __ATOMIC_RELAXED,
					                       __HIP_MEMORY_SCOPE_WORKGROUP);
				}
			}
			__syncthreads();
			STAMP(4);
			for (int l = 2; l < rn; l++) {
				const int jl = OTH(l);
				double acc[NS];
#pragma unroll
				for (int i = 0; i < NS; i++) acc[i] = NBL_NEG_INF;
				for (int k = 0; k < nm; k++) {
					const int tk = uniform(s.lt[jl * nm + k]);
					const double vk = s.lv[jl * nm + k];
#pragma unroll
					for (int i = 0; i < NS; i++) {
						int sym = lane + 64 * i;
						if (sym < Q) acc[i] = dmax(acc[i], A[sym ^ tk] + vk);
					}
				}
				if (l == rn - 1) {
#pragma unroll
					for (int i = 0; i < NS; i++) S[i] = dmax(S[i], acc[i]);
				} else {
#pragma unroll
					for (int i = 0; i < NS; i++) {
						int sym = lane + 64 * i;
						if (sym < Q) Bq[sym] = acc[i];
					}
					__syncthreads();
					double *T = A; A = Bq; Bq = T;
				}
			}
			if (rn <= 2) {
#pragma unroll
				for (int i = 0; i < NS; i++) {
					int sym = lane + 64 * i;
					if (sym < Q) S[i] = dmax(S[i], A[sym]);
				}
			}
		} else {
			// layered DP: A[d][s] = best value reaching check sum s with exactly d deviations
			double *A = s.A, *Bq = s.Bq;
			__syncthreads();
			for (int idx = lane; idx < layers * Q; idx += 64) A[idx] = (idx == 0) ? 0.0 : NBL_NEG_INF;
			__syncthreads();
			for (int l = 0; l < rn; l++) {
				const int jl = OTH(l);
				const int z = s.lt[jl * nm];
				const double mz = s.lv[jl * nm];
				for (int d = 0; d < layers; d++) {
					double acc[NS];
#pragma unroll
					for (int i = 0; i < NS; i++) {
						int sym = lane + 64 * i;
						acc[i] = (sym < Q) ? A[d * Q + (sym ^ z)] + mz : NBL_NEG_INF;
					}
					if (d >= 1) {
						for (int k = 1; k < nm; k++) {
							const int tk = uniform(s.lt[jl * nm + k]);
							const double vk = s.lv[jl * nm + k];
#pragma unroll
							for (int i = 0; i < NS; i++) {
								int sym = lane + 64 * i;
								if (sym < Q) acc[i] = dmax(acc[i], A[(d - 1) * Q + (sym ^ tk)] + vk);
							}
						}
					}
#pragma unroll
					for (int i = 0; i < NS; i++) {
						int sym = lane + 64 * i;
						if (sym < Q) Bq[d * Q + sym] = acc[i];
					}
				}
				__syncthreads();
				double *T = A; A = Bq; Bq = T;
			}
			for (int d = 0; d < layers; d++) {
#pragma unroll
				for (int i = 0; i < NS; i++) {
					int sym = lane + 64 * i;
					if (sym < Q) S[i] = dmax(S[i], A[d * Q + sym]);
				}
			}
		}

		STAMP(5);
		// ---- output: c2v[a] = shape(S[h_x a] - S[0]) (:899-916) ----------------------------------------------
		double *Sv = s.Sv;
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int sym = lane + 64 * i;
			if (sym < Q) Sv[sym] = S[i];
		}
		__syncthreads();
		{
			GfMul<Q> mh;
			mh.init(g.c_h[c0 + x], g.poly, lane);
			const double s0 = Sv[0];
			double *Cx = C + (size_t)x * Q;
#pragma unroll
			for (int i = 0; i < NS; i++) {
				int a = lane + 64 * i;
				if (a < Q) Cx[a] = (a == 0) ? 0.0 : shape_llr(Sv[mh.at_slot(i)] - s0, r.factor, r.offset);
			}
		}
#undef OTH
		STAMP(6);
	}
	if (st_on && lane == 0) {
		for (int i = 0; i < 8; i++) atomicAdd(&w.stamps[i], st_acc[i]);
		atomicAdd(&w.stamps[15], 1ull);
	}
#undef STAMP
}

// ---------------------------------------------------------------------------------------------------------
// state read-back (parity tests): padded device layout -> [.][q-1]
// ---------------------------------------------------------------------------------------------------------
__global__ void unpad_kernel(const double *__restrict__ src, double *__restrict__ dst, const int *__restrict__ map,
                             int rows, int q)
{
	long long total = (long long)rows * (q - 1);
	for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
		int a = (int)(i % (q - 1)) + 1;
		int row = (int)(i / (q - 1));
		int srow = map ? map[row] : row;
		dst[i] = src[(size_t)srow * q + a];
	}
}

// ---------------------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------------------
#define NBL_DISPATCH_Q(q, ...)                                  \
	switch (q) {                                                \
	case 4: { constexpr int QQ = 4; __VA_ARGS__; } break;       \
	case 8: { constexpr int QQ = 8; __VA_ARGS__; } break;       \
	case 16: { constexpr int QQ = 16; __VA_ARGS__; } break;     \
	case 32: { constexpr int QQ = 32; __VA_ARGS__; } break;     \
	case 64: { constexpr int QQ = 64; __VA_ARGS__; } break;     \
	case 128: { constexpr int QQ = 128; __VA_ARGS__; } break;   \
	case 256: { constexpr int QQ = 256; __VA_ARGS__; } break;   \
	default: return hipErrorInvalidValue;                       \
	}

hipError_t nbl_launch_init(const double *d_Lin, const NblGraphDev &g, const NblWork &w, int B, int write_v2c, hipStream_t st)
{
	long long total = (long long)B * g.E * g.q;
	int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
	if (blocks < 1) blocks = 1;
	hipLaunchKernelGGL(init_kernel, dim3(blocks), dim3(256), 0, st, d_Lin, g, w, B, write_v2c);
	return hipGetLastError();
}

hipError_t nbl_launch_demod(const double *d_rx, int L, double sigma, int mod_order, const double *d_cons, const int *d_src,
                            const NblGraphDev &g, const NblWork &w, int B, hipStream_t st)
{
	long long nodes = (long long)B * g.N;
	dim3 grid((unsigned)((nodes + 3) / 4)), block(256);
	demod_kernel<<<grid, block, 0, st>>>(d_rx, L, sigma, mod_order, g.p, d_cons, d_src, g, w, B);
	return hipGetLastError();
}

hipError_t nbl_launch_vn(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool damp, hipStream_t st)
{
	long long nodes = (long long)r.B * g.N;
	dim3 block(256);
	if (g.q <= 32 && !getenv("NBL_VN_UNPACKED")) { // small fields: 64 / q variables per wave
		const long long waves = (nodes + (64 / g.q) - 1) / (64 / g.q);
		dim3 pgrid((unsigned)((waves + 3) / 4));
#define NBL_VNP(QQ) { if (damp) vn_packed_kernel<QQ, true><<<pgrid, block, 0, st>>>(g, w, r); else vn_packed_kernel<QQ, false><<<pgrid, block, 0, st>>>(g, w, r); }
		switch (g.q) {
		case 4: NBL_VNP(4) break;
		case 8: NBL_VNP(8) break;
		case 16: NBL_VNP(16) break;
		case 32: NBL_VNP(32) break;
		default: return hipErrorInvalidValue;
		}
#undef NBL_VNP
		return hipGetLastError();
	}
	dim3 grid((unsigned)((nodes + 3) / 4));
	if (damp) { NBL_DISPATCH_Q(g.q, vn_kernel<QQ, true><<<grid, block, 0, st>>>(g, w, r)) }
	else { NBL_DISPATCH_Q(g.q, vn_kernel<QQ, false><<<grid, block, 0, st>>>(g, w, r)) }
	return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// early exit on large batches: the list of codewords that are still iterating, ascending (one workgroup; B is a few
// thousand).  Grids of the next window of iterations cover this list instead of the whole batch.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void compact_kernel(const uint8_t *__restrict__ done, int B, int *__restrict__ active, int *__restrict__ n_act)
{
	__shared__ int wsum[16];
	__shared__ int base;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	if (threadIdx.x == 0) base = 0;
	__syncthreads();
	for (int b0 = 0; b0 < B; b0 += 1024) {
		const int b = b0 + (int)threadIdx.x;
		const bool live = b < B && !done[b];
		const uint64_t m = __ballot(live);
		if (lane == 0) wsum[wv] = __builtin_popcountll(m);
		__syncthreads();
		int before = base;
		for (int k = 0; k < wv; k++) before += wsum[k];
		if (live) active[before + prefix_count(m)] = b;
		__syncthreads();
		if (threadIdx.x == 0) {
			int t = 0;
			for (int k = 0; k < 16; k++) t += wsum[k];
			base += t;
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) *n_act = base;
}

hipError_t nbl_launch_compact(const uint8_t *done, int B, int *active, int *n_act, hipStream_t st)
{
	hipLaunchKernelGGL(compact_kernel, dim3(1), dim3(1024), 0, st, done, B, active, n_act);
	return hipGetLastError();
}

hipError_t nbl_launch_syn(const NblGraphDev &g, const NblWork &w, const NblRun &r, hipStream_t st)
{
	dim3 grid((unsigned)((r.B + 3) / 4)), block(256);
	hipLaunchKernelGGL(syn_kernel, grid, block, 0, st, g, w, r);
	return hipGetLastError();
}

size_t nbl_ems_lds_bytes(const NblGraphDev &g, int nm, int layers)
{
	size_t doubles = (size_t)g.maxdc * g.q + (2 * (size_t)layers + 1) * g.q + (size_t)g.maxdc * nm;
	return doubles * 8 + (size_t)g.maxdc * nm * 4 + 16;
}

int nbl_ems_layers(const NblGraphDev &g, int nc)
{
	return (nc >= g.maxdc - 1) ? 1 : nc + 1;
}

hipError_t nbl_launch_cn_ems(const NblGraphDev &g, const NblWork &w, const NblRun &r, hipStream_t st)
{
	const int layers = nbl_ems_layers(g, r.nc);
	const size_t lds = nbl_ems_lds_bytes(g, r.nm, layers);
	dim3 grid((unsigned)((long long)r.B * g.M)), block(64);
	if (lds > 160 * 1024) return hipErrorInvalidValue; // (nbl_create refuses such shapes)
	NBL_DISPATCH_Q(g.q, {
		if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)cn_ems_kernel<QQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		cn_ems_kernel<QQ><<<grid, block, lds, st>>>(g, w, r, layers);
	})
	return hipGetLastError();
}

hipError_t nbl_launch_unpad(const double *src, double *dst, const int *map, int rows, int q, hipStream_t st)
{
	long long total = (long long)rows * (q - 1);
	int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
	if (blocks < 1) blocks = 1;
	hipLaunchKernelGGL(unpad_kernel, dim3(blocks), dim3(256), 0, st, src, dst, map, rows, q);
	return hipGetLastError();
}
