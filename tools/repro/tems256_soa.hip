// tools/repro/tems256_soa.hip -- nbl_cn_tems256.hip with the DP state as six parallel [4] arrays (the layout that yields wrong path
// codes; generated from the product source, see tools/repro/bisect_soa.sh).  NOT part of the product.
// nbldpc_amd/csrc/nbl_cn_tems256.hip -- trellis-EMS check node for GF(256), check degree 4, nc <= 3 (the six (2,4)-regular
// GF(256) codes the reference ships): NBLDPC.cpp:1055-1130, TEMS_Get_Beta :1789, TEMS_Get_deltaU :1814, TEMS_Get_Min :1836,
// TEMS_ConstructConf :1892.  Same arithmetic, comparisons and tie rules as the general kernels in nbl_cn_tems.hip (whose
// header explains why the min-plus dynamic programme over the columns equals the reference's path enumeration) and the same
// plan as the GF(64) kernel in nbl_cn_tems64.hip, with four symbols per lane:
//
//   * lane l owns the delta-domain symbols l, 64+l, 128+l, 192+l: trellis columns, column order, candidate marks and every
//     dynamic-programme state of those four check sums live in registers;
//   * layer 1 is a per-lane minimum (layer 0 is the constant {0 at check sum 0}); layers 2 and 3 gather the 24-byte predecessor
//     record S[s ^ q] from LDS: for the lane's four symbols that is one address (lane ^ q) XOR-ed with the four plane offsets;
//   * LDS per wave: one 10 KB region used in phases (trellis while it is built and again in the output stage, predecessor
//     records + ONE column's candidate list during the programme, one output vector at a time) against 51 KB of the general
//     kernel, which runs less than one wave per SIMD at q = 256.
// FUSED = true additionally runs the variable-node pass of the iteration for the four incoming edges (dv = 2 codes), exactly
// as nbl_cn_tems64.hip does.
#include <hip/hip_runtime.h>
#include "../../nbldpc_amd/csrc/nbl_device.h"
#include "../../nbldpc_amd/csrc/nbl_kernels.h"

namespace {

constexpr int Q = 256, P = 8, DC = 4, NS = 4;

struct __attribute__((aligned(16))) Cand { double u; int q16; unsigned dig; }; // cost, symbol << 4, symbol << digit shift

// smaller cost wins, equal cost: smaller path code (= earlier in the reference's enumeration order): the lexicographic
// comparison as mask arithmetic (three compares, the AND / OR on the scalar unit), then one select for the code; the cost
// itself is the plain minimum either way (as in nbl_cn_tems64.hip: 5 vector instructions per relaxation instead of 8)
__device__ __forceinline__ void relax(double &bv, unsigned &bc, double val, unsigned code)
{
	const bool take = (val < bv) | ((val == bv) & (code < bc));
	bc = take ? code : bc;
	bv = __builtin_fmin(bv, val);
}

__device__ __forceinline__ double pick(const double (&u)[DC], int k)
{
	return k == 0 ? u[0] : k == 1 ? u[1] : k == 2 ? u[2] : u[3];
}

// NC = the deviation budget tems_nc (1..3) as a template parameter: with a run-time nc the layer-3 update sits behind a branch
// in the candidate loop and the 16-byte predecessor loads are split in two
template <bool FUSED, int NC>
__global__ __launch_bounds__(64) void cn_tems_q256_dc4_kernel(NblGraphDev g, NblWork w, NblRun r)
{
	// One 10 KB region, used in two phases (sixteen waves per CU = four per SIMD; the register count allows as many):
	//   programme:  predecessor records Sv (4 KB: 16-byte cost pairs) | Sc (2 KB: 8-byte code pairs) + the candidate list of the
	//               current column (4 KB: at most 255 candidates, padded to a multiple of four)
	//   before / after it:  the trellis dU (8 KB, rebuilt from registers for the output stage) + one output vector Lc (2 KB)
	// Every record is read with the stride of its own size (16-byte costs, 8-byte codes), so a gather touches every LDS bank
	// once: the 8-byte codes used to sit in 16-byte slots, and their reads at a 16-byte stride reached only half of the banks.
	__shared__ __attribute__((aligned(16))) char lds[Q * 16 + Q * 8 + Q * 16];
	char *Sraw = lds;                                      // predecessor records
	double2 *Sv = (double2 *)Sraw;                         // [Q] cost of layers 1, 2 of every check sum before the current column
	uint2 *Sc = (uint2 *)(Sraw + Q * 16);                  // [Q] their path codes (x, y)
	Cand *cl = (Cand *)(lds + Q * 16 + Q * 8);             // [Q] deviation candidates of the current column
	double (*dU)[Q] = (double (*)[Q])lds;                  // [DC][Q] delta-domain trellis (:1814-1834)
	double *Lc = (double *)(lds + DC * Q * 8);             // [Q] extrinsic minima of one output edge at a time (:1075-1102)

	const int lane = lane_id();
	const NblXcdSlot xs = nbl_xcd_slot(g.M); // all checks of a codeword on one XCD (nbl_device.h)
	const int b = nbl_codeword(w, r, xs.slot), m = xs.unit;
	if (b < 0) return;
	if (!r.fixed_iters && w.done[b]) return;
	const int c0 = g.coff[m];
	const int nr = r.nr;
	constexpr int nc = NC;
	double *V = w.v2c + (size_t)b * g.E * Q;
	double *C = w.c2v + ((size_t)b * g.E + c0) * Q;

	// ---- 0/1. (FUSED: variable-node pass) beta, syndrome, dU --------------------------------------------------------------
	int beta[DC], syn = 0;
	GfMul<Q> mh[DC];
#pragma unroll
	for (int d = 0; d < DC; d++) {
		double v[NS];
		if (!FUSED) {
			const double *Vd = V + (size_t)g.c_epos[c0 + d] * Q;
#pragma unroll
			for (int i = 0; i < NS; i++) v[i] = Vd[lane + 64 * i];
			if (lane == 0) v[0] = 0.0;
		} else {
			// post = (L_ch + c2v_0) + c2v_1 of the edge's variable (NBLDPC.cpp:977-992), hard decision by the check that holds the
			// variable's first edge, v2c = post - c2v of this edge, damped 1/4 : 3/4 when its hard decision moves (:1029-1052)
			const double *Cp = w.c2v_prev + (w.c2v_prev_shared ? (size_t)0 : (size_t)b * g.E * Q); // (iteration 1: one shared block of zeros)
			const int n = g.c_var[c0 + d], e = g.c_epos[c0 + d], e0 = g.voff[n];
			const double *pl = w.Lch + ((size_t)b * g.N + n) * Q;
			const double *pa = Cp + (size_t)g.v_cpos[e0] * Q, *pb = Cp + (size_t)g.v_cpos[e0 + 1] * Q;
			const bool ownA = (e == e0);
			double post[NS], ov[NS];
#pragma unroll
			for (int i = 0; i < NS; i++) {
				const int a = lane + 64 * i;
				const double ca = pa[a], cb = pb[a];
				post[i] = (pl[a] + ca) + cb;
				v[i] = post[i] - (ownA ? ca : cb);
				ov[i] = V[(size_t)e * Q + a];
			}
			if (ownA) {
				const int dec = wave_decide<NS>(post, lane, Q);
				if (lane == 0) w.dec[(size_t)b * g.N + n] = dec;
				if (w.post) {
#pragma unroll
					for (int i = 0; i < NS; i++) w.post[((size_t)b * g.N + n) * Q + lane + 64 * i] = post[i];
				}
			}
			if (wave_decide<NS>(ov, lane, Q) != wave_decide<NS>(v, lane, Q)) {
#pragma unroll
				for (int i = 0; i < NS; i++) v[i] = __dadd_rn(__dmul_rn(r.damp_old, ov[i]), __dmul_rn(r.damp_new, v[i]));
			}
			if (lane == 0) v[0] = 0.0;
#pragma unroll
			for (int i = 0; i < NS; i++) V[(size_t)e * Q + lane + 64 * i] = v[i];
		}
		// most reliable symbol: strict '>' over ascending symbols from a running maximum of 0 (:1798-1807) = the lowest symbol
		// that holds the maximum of {0, v[1..]}; symbol 0 (value 0) when nothing is positive
		const double mx = wave_fmax(dmax(dmax(v[0], v[1]), dmax(v[2], v[3])));
		int arg = 0;
#pragma unroll
		for (int i = NS - 1; i >= 0; i--) {
			const uint64_t hit = __ballot(v[i] == mx);
			arg = hit ? 64 * i + __builtin_ctzll(hit) : arg;
		}
		mh[d].init(g.c_h[c0 + d], g.poly, lane);
		int bd = 0;
#pragma unroll
		for (int k = 0; k < P; k++) bd ^= ((arg >> k) & 1) ? mh[d].basis[k] : 0; // beta_d = h * argmax
		bd = uniform(bd);
		beta[d] = bd;
		syn ^= bd;
#pragma unroll
		for (int i = 0; i < NS; i++) dU[d][mh[d].at_slot(i) ^ bd] = mx - v[i]; // dU[d][h a ^ beta] = Lmax - L(a) (:1826-1831)
	}
	__syncthreads();
	double u[NS][DC];
#pragma unroll
	for (int d = 0; d < DC; d++)
#pragma unroll
		for (int i = 0; i < NS; i++) u[i][d] = dU[d][lane + 64 * i];
	__syncthreads(); // the region is about to hold the predecessor records

	// ---- 2. stable ascending order of the four columns per symbol, the nr smallest marked (:1836-1890) --------------------
	int mask[NS], o0[NS], o1[NS];
#pragma unroll
	for (int i = 0; i < NS; i++) {
		mask[i] = 0; o0[i] = 0; o1[i] = 0;
#pragma unroll
		for (int d = 0; d < DC; d++) {
			int rank = 0;
#pragma unroll
			for (int e = 0; e < DC; e++)
				if (e != d) rank += (u[i][e] < u[i][d] || (u[i][e] == u[i][d] && e < d)) ? 1 : 0;
			mask[i] |= (rank < nr) ? (1 << d) : 0;
			o0[i] = (rank == 0) ? d : o0[i];
			o1[i] = (rank == 1) ? d : o1[i];
		}
	}

	// ---- 3. min-plus dynamic programme over the columns; v[l], c[l] = best path with exactly l deviating columns -----------
	const double INF = __builtin_huge_val();
	// (an array of structs with interleaved field types on purpose: six parallel arrays are promoted to <4 x double> / <4 x i32>
	// vector values, and hipcc 7.2's late GVN over those then yields wrong path codes on tie-heavy inputs at -O2 and above)
	double v1[NS], v2[NS], v3[NS];
	unsigned c1[NS], c2[NS], c3[NS];
#pragma unroll
	for (int i = 0; i < NS; i++) {
		v1[i] = v2[i] = v3[i] = INF;
		c1[i] = c2[i] = c3[i] = 0;
		const int s = lane + 64 * i;
		if ((mask[i] & 1) && s > 0) { v1[i] = u[i][0]; c1[i] = (unsigned)s << (P * (DC - 1)); }
	}
	const int lane16 = lane << 4;
#pragma unroll
	for (int d = 1; d < DC; d++) {
		const int sh = P * (DC - 1 - d); // digit of column d in the path code
		// predecessor records and the candidates of this column (non-zero symbols, symbol 0 = "no deviation")
		int base = 0;
#pragma unroll
		for (int i = 0; i < NS; i++) {
			const int s = lane + 64 * i;
			Sv[s] = make_double2(v1[i], v2[i]);
			Sc[s] = make_uint2(c1[i], c2[i]);
			const bool c = ((mask[i] >> d) & 1) && s > 0;
			const uint64_t bal = __ballot(c);
			if (c) {
				Cand e;
				e.u = u[i][d];
				e.q16 = s << 4;
				e.dig = (unsigned)s << sh;
				cl[base + prefix_count(bal)] = e;
			}
			base += uniform(__builtin_popcountll(bal));
		}
		const int n = base, n4 = (n + 3) & ~3;
		if (lane < 4 && n + lane < n4) { // pad to a multiple of four with entries that can never win
			Cand e;
			e.u = INF;
			e.q16 = 0;
			e.dig = 0;
			cl[n + lane] = e;
		}
		__syncthreads();
		if (nc >= 2) {
			for (int k = 0; k < n4; k += 2) {
				Cand e[2];
#pragma unroll
				for (int t = 0; t < 2; t++) e[t] = cl[k + t]; // LDS broadcast
#pragma unroll
				for (int t = 0; t < 2; t++) {
					const int A = lane16 ^ e[t].q16; // byte offset of record (lane ^ q) for the lane's symbol 0; symbol i: XOR i << 10
#pragma unroll
					for (int i = 0; i < NS; i++) {
						const int off = A ^ (i << 10);
						const double2 sv = *(const double2 *)(Sraw + off);
						const uint2 sc = *(const uint2 *)(Sraw + Q * 16 + (off >> 1));
						relax(v2[i], c2[i], sv.x + e[t].u, sc.x + e[t].dig);
						if (d >= 2 && nc >= 3) relax(v3[i], c3[i], sv.y + e[t].u, sc.y + e[t].dig);
					}
				}
			}
		}
#pragma unroll
		for (int i = 0; i < NS; i++) {
			const int s = lane + 64 * i;
			if (((mask[i] >> d) & 1) && s > 0) relax(v1[i], c1[i], u[i][d], (unsigned)s << sh);
		}
		__syncthreads();
	}
	// dW, Eta: best layer of each of the lane's check sums
	double dW[NS];
	unsigned eta[NS];
#pragma unroll
	for (int i = 0; i < NS; i++) {
		dW[i] = INF;
		eta[i] = 0xffffffffu;
		const double v0 = (lane == 0 && i == 0) ? 0.0 : INF;
		if (v0 < dW[i] || (v0 == dW[i] && 0u < eta[i])) { dW[i] = v0; eta[i] = 0u; }
		if (nc >= 1 && (v1[i] < dW[i] || (v1[i] == dW[i] && c1[i] < eta[i]))) { dW[i] = v1[i]; eta[i] = c1[i]; }
		if (nc >= 2 && (v2[i] < dW[i] || (v2[i] == dW[i] && c2[i] < eta[i]))) { dW[i] = v2[i]; eta[i] = c2[i]; }
		if (nc >= 3 && (v3[i] < dW[i] || (v3[i] == dW[i] && c3[i] < eta[i]))) { dW[i] = v3[i]; eta[i] = c3[i]; }
	}

	// ---- 4. outputs, one edge at a time (the trellis goes back into the region: the last barrier above has passed) ----------
#pragma unroll
	for (int d = 0; d < DC; d++)
#pragma unroll
		for (int i = 0; i < NS; i++) dU[d][lane + 64 * i] = u[i][d];
#pragma unroll
	for (int d = 0; d < DC; d++) {
#pragma unroll
		for (int i = 0; i < NS; i++) Lc[lane + 64 * i] = NBL_DBL_MAX;
		__syncthreads();
#pragma unroll
		for (int i = 0; i < NS; i++) {
			const int s = lane + 64 * i;
			const int dev = (int)((eta[i] >> (P * (DC - 1 - d))) & (Q - 1));
			const double cand = dW[i] - dU[d][dev]; // :1088
			__hip_atomic_fetch_min(&Lc[s ^ dev], cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		__syncthreads();
#pragma unroll
		for (int i = 0; i < NS; i++) {
			const int s = lane + 64 * i;
			if (Lc[s] == NBL_DBL_MAX) Lc[s] = (d == o0[i]) ? pick(u[i], o1[i]) : pick(u[i], o0[i]); // never reached (:1095-1102)
		}
		__syncthreads();
		{
			// delta domain -> LLR, un-permute by h (:1105-1127)
			const int bsyn = syn ^ beta[d];
			const double L0 = -1.0 * Lc[bsyn];
#pragma unroll
			for (int i = 0; i < NS; i++) {
				const int a = lane + 64 * i;
				const int e = mh[d].at_slot(i) ^ bsyn; // eta with h^-1 (eta ^ bsyn) = a
				C[(size_t)d * Q + a] = (a == 0) ? 0.0 : shape_llr(-1.0 * Lc[e] - L0, r.factor, r.offset);
			}
		}
		__syncthreads(); // Lc is reused by the next edge
	}
}

} // namespace

bool nbl_tems256_applicable(const NblGraphDev &g, bool all_dc4, int nr, int nc)
{
	return g.q == 256 && all_dc4 && nc >= 1 && nc <= 3 && nr >= 1 && nr <= 4;
}

hipError_t nbl_launch_cn_tems256(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool fused, hipStream_t st)
{
	dim3 grid(nbl_xcd_grid(r.B, g.M)), block(64);
	switch ((fused ? 4 : 0) + r.nc) {
	case 1: cn_tems_q256_dc4_kernel<false, 1><<<grid, block, 0, st>>>(g, w, r); break;
	case 2: cn_tems_q256_dc4_kernel<false, 2><<<grid, block, 0, st>>>(g, w, r); break;
	case 3: cn_tems_q256_dc4_kernel<false, 3><<<grid, block, 0, st>>>(g, w, r); break;
	case 5: cn_tems_q256_dc4_kernel<true, 1><<<grid, block, 0, st>>>(g, w, r); break;
	case 6: cn_tems_q256_dc4_kernel<true, 2><<<grid, block, 0, st>>>(g, w, r); break;
	case 7: cn_tems_q256_dc4_kernel<true, 3><<<grid, block, 0, st>>>(g, w, r); break;
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}
