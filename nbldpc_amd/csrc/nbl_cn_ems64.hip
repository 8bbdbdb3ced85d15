// nbldpc_amd/csrc/nbl_cn_ems64.hip -- EMS check node for GF(64) (the BDS 576.288 code of BASELINE config 4 decoded with EMS
// instead of T-EMS): NBLDPC.cpp:859-917, SortLLRVector :1715-1746, ConstructConf :1748-1786.
//
// FOUR checks per wave: a group of 16 lanes (one DPP row) per check, lane l of the group owns symbols l, l+16, l+32, l+48 of
// every vector (variable domain when loading and storing, check domain in between).  The general kernel (cn_ems_kernel<64>) gives
// a check a whole wave and spends its time waiting on a chain of short LDS phases; here four such chains share the wave.
// Same arithmetic as the general kernel -- the max-plus dynamic programme over the other edges in index order, every candidate
// the same left-to-right sum -- so the message state is bit-identical to it and to the canonical oracle (tests).
//   * top-nm selection: quickselect on the group's ballots (a 64-bit member mask per check, bit = symbol), SortLLRVector's order
//     (value descending, higher symbol first among equals);
//   * nc >= dc - 1: plain truncated convolution (pair scatter with LDS max atomics + gathers); otherwise one layer per deviation
//     count, the <= 4 layers of a lane's four check sums in registers;
//   * FUSED (variable degrees <= 3): the variable-node pass of every edge is recomputed from the previous iteration's c2v inside
//     this launch (a-posteriori sum in edge order, decision by the check that holds the variable's first edge), two edges' loads
//     in flight at a time; c2v is double-buffered.
// Any check degree 3..8; groups may diverge (different degrees, different quickselect trip counts).
#include <hip/hip_runtime.h>
#include "nbl_device.h"
#include "nbl_kernels.h"

namespace {

constexpr int Q = 64;   // field size
constexpr int LPC = 16; // lanes per check

#define WSYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")

template <int CTRL> __device__ __forceinline__ double row_mov_f64(double x)
{
	const int lo = __builtin_amdgcn_update_dpp(__double2loint(x), __double2loint(x), CTRL, 0xF, 0xF, false);
	const int hi = __builtin_amdgcn_update_dpp(__double2hiint(x), __double2hiint(x), CTRL, 0xF, 0xF, false);
	return __hiloint2double(hi, lo);
}
// maximum over the 16 lanes of the group, delivered to every lane of it
__device__ __forceinline__ double row_max_f64(double v)
{
	v = dmax(v, row_mov_f64<0xB1>(v));  // quad_perm [1,0,3,2]
	v = dmax(v, row_mov_f64<0x4E>(v));  // quad_perm [2,3,0,1]
	v = dmax(v, row_mov_f64<0x141>(v)); // row_half_mirror
	v = dmax(v, row_mov_f64<0x140>(v)); // row_mirror
	return v;
}
// a predicate over the group's 64 symbols as a mask, bit a = symbol a = lane (a & 15), slot (a >> 4)
__device__ __forceinline__ uint64_t sym_mask(const bool (&p)[4], int gi)
{
	uint64_t m = 0;
#pragma unroll
	for (int i = 0; i < 4; i++) m |= ((__ballot(p[i]) >> (16 * gi)) & 0xffffull) << (16 * i);
	return m;
}

// DecideLLRVector (:1542-1562): lowest symbol among the maxima of {0, v}
__device__ __forceinline__ int row_decide(const double (&v)[4], int gi)
{
	const double mx = dmax(row_max_f64(dmax(dmax(v[0], v[1]), dmax(v[2], v[3]))), 0.0);
	const bool eq[4] = {v[0] == mx, v[1] == mx, v[2] == mx, v[3] == mx};
	const uint64_t hit = sym_mask(eq, gi);
	return (mx > 0.0 && hit) ? __builtin_ctzll(hit) : 0;
}

// value of symbol `a` of the group's vector v (a uniform inside the group)
__device__ __forceinline__ double sym_value(const double (&v)[4], int a, int gi)
{
	const int i = a >> 4;
	const double mine = (i == 0) ? v[0] : (i == 1) ? v[1] : (i == 2) ? v[2] : v[3];
	return __shfl(mine, 16 * gi + (a & 15), 64);
}

struct Ent { double v; int t; }; // list entry: value, check-domain symbol (two arrays in LDS)

// LDS per check -- what bounds the occupancy of this kernel (3.25 KB at dc = 4, nm = 16, plain convolution: three waves per SIMD):
// the dc input vectors (which double as the staging of the incoming values), the DP layers (updated in place: every lane reads,
// fence, every lane writes; they also stage the output), the lists as 8 + 4 bytes per entry
__host__ __device__ inline int dp_vectors(int, int layers) { return layers; }
__host__ __device__ inline size_t group_bytes(int mdc, int nm, int layers)
{
	return (((size_t)8 * ((size_t)mdc * Q + (size_t)dp_vectors(mdc, layers) * Q) + (size_t)12 * mdc * nm) + 15) & ~(size_t)15;
}

template <bool FUSED>
__global__ __launch_bounds__(64) void cn_ems_q64_kernel(NblGraphDev g, NblWork w, NblRun r, int layers)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int lane = lane_id(), gi = lane >> 4, sl = lane & 15;
	// (g.M + 3) / 4 workgroups per codeword, all of them on one XCD (nbl_device.h); group gi of workgroup `unit` holds check
	// 4 unit + gi
	const NblXcdSlot xs = nbl_xcd_slot((g.M + 3) >> 2);
	const int m = xs.unit * 4 + gi;
	int b = -1;
	if (m < g.M) {
		b = nbl_codeword(w, r, xs.slot);
		if (b >= 0 && !r.fixed_iters && w.done[b]) b = -1;
	}
	if (b < 0) return; // (a whole group leaves; the others never look at its lanes)
	const int c0 = g.coff[m], dc = g.coff[m + 1] - c0, nm = r.nm, mdc = g.maxdc;
	char *base = smem + (size_t)gi * group_bytes(mdc, nm, layers);
	double *U = (double *)base;              // [mdc][Q]   check-domain input vectors: U[j][h_j*a] = v2c_j[a], U[j][0] = 0
	double *A = U + mdc * Q;                 // [layers][Q] DP layers
	double *Sv = A;                          // [Q]        the maxima of the current output edge (after its DP)
	double *lv = A + dp_vectors(mdc, layers) * Q; // [mdc][nm] values of the nm most reliable entries of every edge (rank 0 first)
	int *lt = (int *)(lv + mdc * nm);        // [mdc][nm]  their check-domain symbols
	auto entry = [&](int idx) {
		Ent en;
		en.v = lv[idx];
		en.t = lt[idx];
		return en;
	};

	double *C = w.c2v + ((size_t)b * g.E + c0) * Q;

	// ---- the dc incoming vectors, two edges at a time (all their loads in flight before the first use); each is permuted into
	// the check domain and its nm best are selected ------------------------------------------------------------------------------
	for (int pb = 0; pb < dc; pb += 2) {
		double *vin = U + pb * Q; // the incoming values wait, in variable-domain order, in the rows they are permuted into below
		if (!FUSED) {
			const double *V = w.v2c + (size_t)b * g.E * Q;
			double v[2][4];
#pragma unroll
			for (int u = 0; u < 2; u++) {
				const double *Vd = V + (size_t)g.c_epos[c0 + ((pb + u < dc) ? pb + u : 0)] * Q;
#pragma unroll
				for (int i = 0; i < 4; i++) v[u][i] = Vd[sl + 16 * i];
			}
#pragma unroll
			for (int u = 0; u < 2; u++)
				if (pb + u < dc) {
#pragma unroll
					for (int i = 0; i < 4; i++) vin[u * Q + sl + 16 * i] = (sl + 16 * i == 0) ? 0.0 : v[u][i];
				}
		} else {
			const double *Cp = w.c2v_prev + (w.c2v_prev_shared ? (size_t)0 : (size_t)b * g.E * Q); // (iteration 1: one shared block of zeros)
			int4 row[2];
			int n[2], e[2];
			double L[2][4], x0[2][4], x1[2][4], x2[2][4];
#pragma unroll
			for (int u = 0; u < 2; u++) {
				const int ce = c0 + ((pb + u < dc) ? pb + u : 0);
				row[u] = ((const int4 *)g.c_nbr)[ce];
				n[u] = g.c_var[ce];
				e[u] = g.c_epos[ce];
			}
#pragma unroll
			for (int u = 0; u < 2; u++) {
				const double *pl = w.Lch + ((size_t)b * g.N + n[u]) * Q;
				const double *p0 = Cp + (size_t)row[u].x * Q, *p1 = Cp + (size_t)row[u].y * Q, *p2 = Cp + (size_t)(row[u].z >= 0 ? row[u].z : row[u].x) * Q;
#pragma unroll
				for (int i = 0; i < 4; i++) {
					L[u][i] = pl[sl + 16 * i];
					x0[u][i] = p0[sl + 16 * i];
					x1[u][i] = p1[sl + 16 * i];
					x2[u][i] = p2[sl + 16 * i];
				}
			}
#pragma unroll
			for (int u = 0; u < 2; u++) {
				if (pb + u < dc) {
					const int ce = c0 + pb + u;
					double post[4], nv[4];
#pragma unroll
					for (int i = 0; i < 4; i++) {
						post[i] = (L[u][i] + x0[u][i]) + x1[u][i];
						post[i] = (row[u].z >= 0) ? post[i] + x2[u][i] : post[i];
						nv[i] = post[i] - ((row[u].x == ce) ? x0[u][i] : (row[u].y == ce) ? x1[u][i] : x2[u][i]);
					}
					if (row[u].w) { // the variable's first edge: this check reports the hard decision (and the a-posteriori vector)
						const int dec = row_decide(post, gi);
						if (sl == 0) w.dec[(size_t)b * g.N + n[u]] = dec;
						if (w.post) {
#pragma unroll
							for (int i = 0; i < 4; i++) w.post[((size_t)b * g.N + n[u]) * Q + sl + 16 * i] = post[i];
						}
					}
#pragma unroll
					for (int i = 0; i < 4; i++) {
						const double x = (sl + 16 * i == 0) ? 0.0 : nv[i];
						if (w.store_v2c) w.v2c[((size_t)b * g.E + e[u]) * Q + sl + 16 * i] = x;
						vin[u * Q + sl + 16 * i] = x;
					}
				}
			}
		}
#pragma unroll 1
		for (int u = 0; u < 2 && pb + u < dc; u++) {
			const int j = pb + u;
			const uint8_t *mrow = g.mul + (size_t)g.c_h[c0 + j] * Q;
			double v[4];
			int t[4];
#pragma unroll
			for (int i = 0; i < 4; i++) {
				v[i] = vin[u * Q + sl + 16 * i];
				t[i] = mrow[sl + 16 * i];
			}
			WSYNC(); // (the row is permuted in place: every lane has read before any lane writes)
#pragma unroll
			for (int i = 0; i < 4; i++) U[j * Q + t[i]] = v[i];
			// rank 0: the maximum, highest symbol among equals (SortLLRVector's order :1731)
			const double mx = row_max_f64(dmax(dmax(v[0], v[1]), dmax(v[2], v[3])));
			const bool eqm[4] = {v[0] == mx, v[1] == mx, v[2] == mx, v[3] == mx};
			const int top_a = 63 - __builtin_clzll(sym_mask(eqm, gi) | 1ull); // (| 1: a mask is never empty unless the input holds NaNs)
			// the nm best: quickselect, pivot = lowest remaining candidate, `above` = everything that sorts before the pivot
			uint64_t member = ~0ull, cand = ~0ull;
			if (nm < Q) {
				for (int guard = 0; guard < Q + 2 && cand; guard++) {
					const int pa = __builtin_ctzll(cand);
					const double pv = sym_value(v, pa, gi);
					bool ab[4];
#pragma unroll
					for (int i = 0; i < 4; i++) ab[i] = v[i] > pv || (v[i] == pv && sl + 16 * i > pa);
					const uint64_t above = sym_mask(ab, gi);
					const int G = 1 + __builtin_popcountll(above);
					if (G == nm) { member = above | (1ull << pa); cand = 0; }
					else if (G > nm) cand &= above;
					else cand &= ~above & ~(1ull << pa);
				}
			}
			// compact the members into the list (symbol order), rank 0 swapped to the front
			const int p0 = __builtin_popcountll(member & ((1ull << top_a) - 1ull));
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const int a = sl + 16 * i;
				if ((member >> a) & 1ull) {
					int pos = __builtin_popcountll(member & ((1ull << a) - 1ull));
					if (a == top_a) pos = 0;
					else if (pos == 0) pos = p0;
					if (pos >= nm) continue; // (only with NaN inputs, where the selection cannot settle)
					lv[j * nm + pos] = v[i];
					lt[j * nm + pos] = t[i];
				}
			}
		}
		WSYNC();
	}

	// ---- one output edge at a time -----------------------------------------------------------------------------------------
	for (int x = 0; x < dc; x++) {
#define OTH(l) ((l) + ((l) >= x ? 1 : 0))
		const int rn = dc - 1;
		int zall = 0;
		for (int l = 0; l < rn; l++) zall ^= lt[OTH(l) * nm];
		double S[4];
#pragma unroll
		for (int i = 0; i < 4; i++) S[i] = -NBL_DBL_MAX;

		// conf(q,1): at most one edge deviates, to ANY symbol (:894)
		for (int pi = 0; pi < rn; pi++) {
			const int jd = OTH(pi);
			const int shift = zall ^ lt[jd * nm];
			double acc[4] = {0.0, 0.0, 0.0, 0.0};
			for (int l = 0; l < rn; l++) {
				if (l == pi) {
#pragma unroll
					for (int i = 0; i < 4; i++) acc[i] = acc[i] + U[jd * Q + ((sl + 16 * i) ^ shift)];
				} else {
					const double tv = lv[OTH(l) * nm];
#pragma unroll
					for (int i = 0; i < 4; i++) acc[i] = acc[i] + tv;
				}
			}
#pragma unroll
			for (int i = 0; i < 4; i++) S[i] = dmax(S[i], acc[i]);
		}

		// conf(nm,nc): at most nc edges deviate, each inside its nm best (:897); conf(nm,0) is the all-rank-0 configuration alone,
		// which conf(q,1) already holds
		WSYNC();
		if (r.nc < 1) {
		} else if (layers == 1) {
			// nc >= dc-1: no deviation counting needed -> plain truncated max-plus convolution
#pragma unroll
			for (int i = 0; i < 4; i++) A[sl + 16 * i] = NBL_NEG_INF;
			WSYNC();
			if (rn == 1) {
				const int j1 = OTH(0);
				for (int k = sl; k < nm; k += LPC) A[lt[j1 * nm + k]] = 0.0 + lv[j1 * nm + k];
			} else {
				const int j1 = OTH(0), j2 = OTH(1);
				const unsigned inv = ((1u << 20) + nm - 1) / nm; // idx / nm for idx < nm * nm <= 4096 without an integer division
				for (int idx = sl; idx < nm * nm; idx += LPC) {
					const int k1 = (int)(((unsigned)idx * inv) >> 20), k2 = idx - k1 * nm;
					const Ent e1 = entry(j1 * nm + k1), e2 = entry(j2 * nm + k2);
					__hip_atomic_fetch_max(&A[e1.t ^ e2.t], (0.0 + e1.v) + e2.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
			}
			WSYNC();
			for (int l = 2; l < rn; l++) {
				const int jl = OTH(l);
				double acc[4] = {NBL_NEG_INF, NBL_NEG_INF, NBL_NEG_INF, NBL_NEG_INF};
				for (int k = 0; k < nm; k++) {
					const Ent en = entry(jl * nm + k);
#pragma unroll
					for (int i = 0; i < 4; i++) acc[i] = dmax(acc[i], A[(sl + 16 * i) ^ en.t] + en.v);
				}
				if (l == rn - 1) {
#pragma unroll
					for (int i = 0; i < 4; i++) S[i] = dmax(S[i], acc[i]);
				} else {
					WSYNC();
#pragma unroll
					for (int i = 0; i < 4; i++) A[sl + 16 * i] = acc[i];
					WSYNC();
				}
			}
			if (rn <= 2) {
#pragma unroll
				for (int i = 0; i < 4; i++) S[i] = dmax(S[i], A[sl + 16 * i]);
			}
		} else {
			// layered DP: A[d][s] = best value reaching check sum s with exactly d deviations.  After the first other edge this is
			// layer 0 = {z: 0 + rank-0 value}, layer 1 = {t_k: 0 + v_k, k >= 1}; after l edges no path has more than l deviations,
			// so layer d is only touched for d <= l + 1.
			for (int d = 0; d < layers; d++)
#pragma unroll
				for (int i = 0; i < 4; i++) A[d * Q + sl + 16 * i] = NBL_NEG_INF;
			WSYNC();
			{
				const int j0 = OTH(0);
				for (int k = sl; k < nm; k += LPC) {
					const Ent en = entry(j0 * nm + k);
					A[(k == 0 ? 0 : Q) + en.t] = 0.0 + en.v;
				}
			}
			WSYNC();
			for (int l = 1; l < rn; l++) {
				const int jl = OTH(l);
				const Ent top = entry(jl * nm);
				double acc[4][4]; // [layer][slot]
#pragma unroll
				for (int d = 0; d < 4; d++)
#pragma unroll
					for (int i = 0; i < 4; i++) acc[d][i] = (d < layers) ? A[d * Q + ((sl + 16 * i) ^ top.t)] + top.v : NBL_NEG_INF;
				for (int k = 1; k < nm; k++) {
					const Ent en = entry(jl * nm + k);
#pragma unroll
					for (int d = 1; d < 4; d++)
						if (d < layers && d <= l + 1) {
#pragma unroll
							for (int i = 0; i < 4; i++) acc[d][i] = dmax(acc[d][i], A[(d - 1) * Q + ((sl + 16 * i) ^ en.t)] + en.v);
						}
				}
				if (l == rn - 1) {
#pragma unroll
					for (int d = 0; d < 4; d++)
						if (d < layers) {
#pragma unroll
							for (int i = 0; i < 4; i++) S[i] = dmax(S[i], acc[d][i]);
						}
				} else {
					WSYNC();
#pragma unroll
					for (int d = 0; d < 4; d++)
						if (d < layers) {
#pragma unroll
							for (int i = 0; i < 4; i++) A[d * Q + sl + 16 * i] = acc[d][i];
						}
					WSYNC();
				}
			}
			if (rn == 1)
				for (int d = 0; d < layers; d++)
#pragma unroll
					for (int i = 0; i < 4; i++) S[i] = dmax(S[i], A[d * Q + sl + 16 * i]);
		}

		// ---- output: c2v[a] = shape(S[h_x a] - S[0]) (:899-916) ------------------------------------------------------------
		WSYNC();
#pragma unroll
		for (int i = 0; i < 4; i++) Sv[sl + 16 * i] = S[i];
		WSYNC();
		{
			const double s0 = Sv[0];
			const uint8_t *mrow = g.mul + (size_t)g.c_h[c0 + x] * Q;
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const int a = sl + 16 * i;
				C[(size_t)x * Q + a] = (a == 0) ? 0.0 : shape_llr(Sv[mrow[a]] - s0, r.factor, r.offset);
			}
		}
#undef OTH
	}
}

} // namespace

// GF(64), check degrees 3..8, at most four deviation-count layers, the four checks of a wave within 64 KB of LDS
bool nbl_ems64_applicable(const NblGraphDev &g, int min_dc, int nm, int nc)
{
	if (g.q != 64 || min_dc < 3 || g.maxdc > NBL_MAXDC) return false;
	const int layers = nbl_ems_layers(g, nc);
	return layers <= 4 && group_bytes(g.maxdc, nm, layers) * 4 <= 64 * 1024;
}

hipError_t nbl_launch_cn_ems64(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool fused, hipStream_t st)
{
	const int layers = nbl_ems_layers(g, r.nc);
	const size_t lds = group_bytes(g.maxdc, r.nm, layers) * 4;
	dim3 grid(nbl_xcd_grid(r.B, (g.M + 3) >> 2)), block(64);
	if (fused) cn_ems_q64_kernel<true><<<grid, block, lds, st>>>(g, w, r, layers);
	else cn_ems_q64_kernel<false><<<grid, block, lds, st>>>(g, w, r, layers);
	return hipGetLastError();
}
