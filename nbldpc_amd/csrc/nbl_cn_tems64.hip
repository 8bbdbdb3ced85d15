// nbldpc_amd/csrc/nbl_cn_tems64.hip -- trellis-EMS check node for GF(64), check degree 4, nc <= 3 (BASELINE config 4):
// NBLDPC.cpp:1055-1130, TEMS_Get_Beta :1789, TEMS_Get_deltaU :1814, TEMS_Get_Min :1836, TEMS_ConstructConf :1892.
// Same arithmetic, same comparisons and the same tie rules as the general kernels in nbl_cn_tems.hip (whose header
// explains why the min-plus dynamic programme over the columns equals the reference's path enumeration); what changes is
// where the data lives:
//
//   * q = 64 = one delta-domain symbol per lane: the four trellis columns dU[d][lane], the column order, the candidate marks
//     and all dynamic-programme states of the lane's check sum stay in registers;
//   * layer 0 (no deviation) is the constant {0 at check sum 0}, so layer 1 (one deviating column) is a per-lane minimum
//     over the marked columns -- no convolution; only layers 2 and 3 gather a predecessor state `S[s ^ q]` from LDS, and
//     they read just the 24 bytes they need (cost and path code of layers 1, 2) instead of a 48-byte record;
//   * the candidates of a column are read from the registers of their lanes (ballot + v_readlane: no list in LDS), the four
//     outputs are formed together (3 barriers instead of 12), wave reductions run on the DPP network.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "nbl_device.h"
#include "nbl_kernels.h"

namespace {

constexpr int Q = 64, P = 6, DC = 4;


// smaller cost wins, equal cost: smaller path code (= earlier in the reference's enumeration order): the lexicographic
// comparison as mask arithmetic (three compares, the AND / OR on the scalar unit), then one select for the code; the cost
// itself is the plain minimum either way
__device__ __forceinline__ void relax(double &bv, unsigned &bc, double val, unsigned code)
{
	const bool take = (val < bv) | ((val == bv) & (code < bc));
	bc = take ? code : bc;
	bv = __builtin_fmin(bv, val);
}

// the same without the comparison of the path codes: right unless val == bv in some lane, which `tie` records (a wave-wide mask
// kept on the scalar unit) -- the caller then repeats the column with relax().  Costs that stand for "no path" must be NaN
// here, not infinity: inf == inf would report a tie at every unreachable check sum, NaN compares false and fmin drops it.
// What it keeps instead of the path code is `from`, the LDS offset of the winning predecessor: the code is formed once per column
// from it (predecessor's code + this column's digit), not once per candidate.
__device__ __forceinline__ void relax_untied(double &bv, int &from, double val, int off, uint64_t &tie)
{
	tie |= __ballot(val == bv);
	from = (val < bv) ? off : from;
	// (v_min_f64 as such: behind __builtin_fmin the compiler first canonicalises the running minimum, an instruction per chain and trip)
	asm("v_min_f64 %0, %1, %2" : "=v"(bv) : "v"(bv), "v"(val));
}

__device__ __forceinline__ double pick(const double (&u)[DC], int k)
{
	return k == 0 ? u[0] : k == 1 ? u[1] : k == 2 ? u[2] : u[3];
}

// NC = the deviation budget tems_nc (1..3) as a template parameter: with a run-time nc the layer-3 update sits behind a branch
// in the candidate loop and the 16-byte predecessor loads are split in two
template <bool FUSED, int NC>
__global__ __launch_bounds__(64) void cn_tems_q64_dc4_kernel(NblGraphDev g, NblWork w, NblRun r)
{
	__shared__ double dU[DC][Q];                            // delta-domain trellis (:1814-1834)
	// one phased region: during the dynamic programme the predecessor states, afterwards the extrinsic minima of the four
	// outputs (4 KB of LDS per wave in all: the register count, not LDS, sets the waves per SIMD)
	__shared__ __attribute__((aligned(16))) char phased[DC * Q * 8];
	// cost of layers 1, 2 of every check sum before the current column and their path codes: three arrays of 8-byte entries, so
	// that one XOR gives the address of all three gathers (NaN = no path: relax_untied)
	double *const Sv1 = (double *)phased, *const Sv2 = Sv1 + Q; // [Q], [Q]
	uint2 *const Sc = (uint2 *)(phased + Q * 16);           // [Q]
	double (*const Lc)[Q] = (double (*)[Q])phased;          // [DC][Q] extrinsic minima of every output edge (:1075-1102), after the DP
	static_assert(sizeof(phased) >= DC * Q * 8, "Lc must fit the phased region");
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

	// ---- 0. FUSED: the variable-node pass of this iteration for the four incoming edges (NBLDPC.cpp:977-992, :1029-1052) ----
	// post = (L_ch + c2v_0) + c2v_1 of the edge's variable (dv = 2), hard decision by the check that holds the variable's
	// first edge, v2c = post - c2v of this edge, damped 1/4 : 3/4 against the previous v2c when its hard decision moves.
	// The hard decision of the previous v2c is not recomputed from the vector: it is what step 1 below found for that vector in
	// the previous iteration (w.edge_dec), so the previous v2c itself is only loaded when the decision moved (or in iteration 1,
	// where it is L_ch and nothing has been recorded yet).
	double vin[DC], vmax[DC];
	int varg[DC];
	if (FUSED) {
		const double *Cp = w.c2v_prev + (w.c2v_prev_shared ? (size_t)0 : (size_t)b * g.E * Q); // (iteration 1: one shared block of zeros)
		double l[DC], ca[DC], cb[DC], ov[DC];
		int nvar[DC], eidx[DC], before[DC];
		bool ownA[DC];
		const bool first = (r.iter == 1);
#pragma unroll
		for (int d = 0; d < DC; d++) { // every load first
			const int n = g.c_var[c0 + d], e = g.c_epos[c0 + d], e0 = g.voff[n];
			l[d] = w.Lch[((size_t)b * g.N + n) * Q + lane];
			ca[d] = Cp[(size_t)g.v_cpos[e0] * Q + lane];
			cb[d] = Cp[(size_t)g.v_cpos[e0 + 1] * Q + lane];
			ov[d] = first ? V[(size_t)e * Q + lane] : 0.0;
			before[d] = first ? 0 : w.edge_dec[(size_t)b * g.E + e];
			nvar[d] = n; eidx[d] = e; ownA[d] = (e == e0);
		}
#pragma unroll
		for (int d = 0; d < DC; d++) {
			const double post[1] = {(l[d] + ca[d]) + cb[d]};
			if (ownA[d]) {
				const int dec = wave_decide<1>(post, lane, Q);
				if (lane == 0) w.dec[(size_t)b * g.N + nvar[d]] = dec;
				if (w.post) w.post[((size_t)b * g.N + nvar[d]) * Q + lane] = post[0];
			}
			double nv = lane == 0 ? 0.0 : post[0] - (ownA[d] ? ca[d] : cb[d]);
			// DecideLLRVector of the new vector (:1542-1562): lowest symbol among the maxima of {0, nv[1..]}
			// (lane 0 holds 0; nothing positive -- every vector of a converged all-zero codeword -- is settled by one compare: maximum 0 at symbol 0)
			double mx = 0.0;
			int arg = 0;
			uint64_t hit;
			if (__ballot(nv > 0.0)) {
				mx = wave_fmax_nonneg(nv);
				hit = __ballot(nv == mx);
				arg = hit ? __builtin_ctzll(hit) : 0;
			}
			int bef = uniform(before[d]);
			if (first) {
				const double old[1] = {ov[d]};
				bef = wave_decide<1>(old, lane, Q);
			}
			if (bef != arg) {
				const double old = first ? ov[d] : V[(size_t)eidx[d] * Q + lane];
				nv = __dadd_rn(__dmul_rn(r.damp_old, old), __dmul_rn(r.damp_new, nv));
				if (lane == 0) nv = 0.0;
				mx = wave_fmax_nonneg(nv);
				hit = __ballot(nv == mx);
				arg = hit ? __builtin_ctzll(hit) : 0;
			}
			vin[d] = nv;
			vmax[d] = mx;
			varg[d] = arg;
			V[(size_t)eidx[d] * Q + lane] = nv;
			if (lane == 0) w.edge_dec[(size_t)b * g.E + eidx[d]] = arg;
		}
	}

	// ---- 1. beta, syndrome, dU -------------------------------------------------------------------------------------------
	int beta[DC], hmul[DC], syn = 0;
#pragma unroll
	for (int d = 0; d < DC; d++) {
		const double v = FUSED ? vin[d] : (lane > 0 ? V[(size_t)g.c_epos[c0 + d] * Q + lane] : 0.0);
		// most reliable symbol: strict '>' over ascending symbols from a running maximum of 0 (:1798-1807) = the lowest symbol
		// that holds the maximum of {0, v[1..]}; symbol 0 (value 0) when nothing is positive
		double mx;
		int arg;
		if (FUSED) { mx = vmax[d]; arg = varg[d]; } // found by the variable-node stage above
		else {
			mx = wave_fmax_nonneg(v); // (lane 0 holds 0)
			const uint64_t top = __ballot(v == mx);
			arg = top ? __builtin_ctzll(top) : 0;
		}
		// products with the edge coefficient from the 4 KB multiplication table (L1 / scalar-cache resident) instead of GF(2)-linear
		// arithmetic in registers: h * lane per lane, h * argmax for the wave
		const uint8_t *mrow = g.mul + (size_t)g.c_h[c0 + d] * Q;
		const int bd = uniform((int)mrow[arg]);                                // beta_d = h * argmax
		beta[d] = bd;
		syn ^= bd;
		hmul[d] = (int)mrow[lane];
		dU[d][hmul[d] ^ bd] = mx - v; // dU[d][h a ^ beta] = Lmax - L(a), L(0) = 0 (:1826-1831)
	}
	__syncthreads();
	double u[DC];
#pragma unroll
	for (int d = 0; d < DC; d++) u[d] = dU[d][lane];

	// ---- 2. stable ascending order of the four columns per symbol, the nr smallest marked (:1836-1890) --------------------
	int mask = 0, o0 = 0, o1 = 0;
#pragma unroll
	for (int d = 0; d < DC; d++) {
		int rank = 0;
#pragma unroll
		for (int e = 0; e < DC; e++)
			if (e != d) rank += (u[e] < u[d] || (u[e] == u[d] && e < d)) ? 1 : 0;
		mask |= (rank < nr) ? (1 << d) : 0;
		o0 = (rank == 0) ? d : o0;
		o1 = (rank == 1) ? d : o1;
	}
	// ---- 3. min-plus dynamic programme over the columns; v[l], c[l] = best path with exactly l deviating columns -----------
	const double INF = __builtin_huge_val();
	// the cheapest single deviation that reaches the lane's check sum (its column has rank 0 there: always marked)
	const double m1 = (lane == 0) ? 0.0 : dmin(dmin(u[0], u[1]), dmin(u[2], u[3]));
	int lane8 = lane << 3;
	asm("" : "+v"(lane8)); // (opaque: otherwise (lane << 3) ^ (q << 3) becomes (lane ^ q) << 3, a second vector instruction per gather)
	double v1 = INF, v2 = INF, v3 = INF;
	unsigned c1 = 0, c2 = 0, c3 = 0;
	if (((mask >> 0) & 1) && lane > 0) { v1 = u[0]; c1 = (unsigned)lane << (P * (DC - 1)); }
	// one column folded into layers 2 (and 3): every candidate (cost ud at deviation symbol q, digit q << sh) against the
	// predecessor state of check sum s ^ q.  Candidates = the marked non-zero symbols of the column (symbol 0 = "no deviation"),
	// in ascending order: they are taken straight from the registers of their lanes (v_readlane into scalar operands of the
	// adds) -- a candidate list in LDS would cost a 16-byte broadcast read per candidate on top of the 24 bytes of the gather,
	// and the LDS port is as busy as the vector unit in this kernel
	auto fold = [&](auto exact_tag, bool both, uint64_t cm, double ud, int sh, uint64_t &tie) {
		constexpr bool EXACT = decltype(exact_tag)::value;
		int from2 = lane8, from3 = lane8;
		const double v2in = v2, v3in = v3;
		auto one = [&](int q) {
			const double cu = read_lane_f64(ud, q);
			const int off = lane8 ^ (q << 3);
			const double s1 = *(const double *)((const char *)Sv1 + off);
			if (EXACT) {
				const unsigned dig = (unsigned)q << sh;
				const uint2 sc = *(const uint2 *)((const char *)Sc + off);
				relax(v2, c2, s1 + cu, sc.x + dig);
				if (both) relax(v3, c3, *(const double *)((const char *)Sv2 + off) + cu, sc.y + dig);
			} else {
				relax_untied(v2, from2, s1 + cu, off, tie);
				if (both) relax_untied(v3, from3, *(const double *)((const char *)Sv2 + off) + cu, off, tie);
			}
		};
		int left = __builtin_popcountll(cm);
		for (; left >= 4; left -= 4) {
			int q[4];
#pragma unroll
			for (int t = 0; t < 4; t++) { q[t] = __builtin_ctzll(cm); cm &= cm - 1; }
#pragma unroll
			for (int t = 0; t < 4; t++) one(q[t]);
		}
		for (; left > 0; left--) { const int q = __builtin_ctzll(cm); cm &= cm - 1; one(q); }
		if (!EXACT) { // the path codes of the layers this column improved (strictly: an equal cost is a tie and ends in the exact pass)
			const unsigned n2 = Sc[from2 >> 3].x + ((unsigned)((from2 ^ lane8) >> 3) << sh);
			c2 = (v2 < v2in) ? n2 : c2;
			if (both) {
				const unsigned n3 = Sc[from3 >> 3].y + ((unsigned)((from3 ^ lane8) >> 3) << sh);
				c3 = (v3 < v3in) ? n3 : c3;
			}
		}
	};
#pragma unroll
	for (int d = 1; d < DC; d++) {
		const int sh = P * (DC - 1 - d); // digit of column d in the path code
		const double NOPATH = __builtin_nan("");
		Sv1[lane] = (v1 == INF) ? NOPATH : v1;
		Sv2[lane] = (v2 == INF) ? NOPATH : v2;
		Sc[lane] = make_uint2(c1, c2);
		__syncthreads();
		if (nc >= 2) {
			// Candidates that cannot matter are left out: every cost is >= 0, so a path through a deviation of cost u costs at least u,
			// and check sum s is already reached for best[s] = min(cheapest single deviation m1[s], v2[s], v3[s]) -- a bound that only
			// falls from here on.  u > max_s best[s] can therefore neither win nor tie anywhere, directly or as a predecessor.
			// (the bound from m1 alone leaves out next to nothing: it is v2 and v3 that bring it down)
			const double reach = wave_fmax_nonneg(__builtin_fmin(__builtin_fmin(m1, v2), v3));
			// ... and per layer: a layer-2 path through this column costs at least u + min v1, a layer-3 path at least u + min v2 (the sums
			// as the relaxation itself rounds them: rounding is monotone, so every actual value is >= the rounded bound)
			const bool marked = ((mask >> d) & 1) && lane > 0;
			const bool both = (d >= 2 && nc >= 3);
			const double least1 = wave_fmin_nonneg(v1), least2 = both ? wave_fmin_nonneg(v2) : 0.0; // (whole-wave reductions: not inside a lane condition)
			const uint64_t cm2 = __ballot(marked && u[d] + least1 <= reach);
			const uint64_t cm3 = both ? (cm2 & __ballot(u[d] + least2 <= reach)) : 0;
			const double v2s = v2, v3s = v3;
			const unsigned c2s = c2, c3s = c3;
			uint64_t tie = 0;
			if (both) fold(std::false_type{}, true, cm3, u[d], sh, tie);
			fold(std::false_type{}, false, cm2 & ~cm3, u[d], sh, tie);
#ifdef NBL_T64_NOTIE
			tie = 0; // diagnostic (wrong results on ties): never repeat a column
#endif
			if (tie) { // equal costs somewhere: the column again, path codes compared (integer-valued LLRs, dead-zone zeros)
				v2 = v2s; v3 = v3s; c2 = c2s; c3 = c3s;
				if (both) fold(std::true_type{}, true, cm3, u[d], sh, tie);
				fold(std::true_type{}, false, cm2 & ~cm3, u[d], sh, tie);
			}
		}
		if (((mask >> d) & 1) && lane > 0) relax(v1, c1, u[d], (unsigned)lane << sh);
		__syncthreads();
	}
	// dW, Eta: best layer of the lane's check sum
	double dW = INF;
	unsigned eta = 0xffffffffu;
	{
		const double v0 = lane == 0 ? 0.0 : INF;
		if (v0 < dW || (v0 == dW && 0u < eta)) { dW = v0; eta = 0u; }
		if (nc >= 1 && (v1 < dW || (v1 == dW && c1 < eta))) { dW = v1; eta = c1; }
		if (nc >= 2 && (v2 < dW || (v2 == dW && c2 < eta))) { dW = v2; eta = c2; }
		if (nc >= 3 && (v3 < dW || (v3 == dW && c3 < eta))) { dW = v3; eta = c3; }
	}

	// ---- 4. outputs of the four edges --------------------------------------------------------------------------------------
#pragma unroll
	for (int d = 0; d < DC; d++) Lc[d][lane] = NBL_DBL_MAX;
	__syncthreads();
#pragma unroll
	for (int d = 0; d < DC; d++) {
		const int dev = (int)((eta >> (P * (DC - 1 - d))) & (Q - 1));
		const double cand = dW - dU[d][dev]; // :1088
		__hip_atomic_fetch_min(&Lc[d][lane ^ dev], cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
	__syncthreads();
#pragma unroll
	for (int d = 0; d < DC; d++)
		if (Lc[d][lane] == NBL_DBL_MAX) Lc[d][lane] = (d == o0) ? pick(u, o1) : pick(u, o0); // never reached (:1095-1102)
	__syncthreads();
#pragma unroll
	for (int d = 0; d < DC; d++) {
		// delta domain -> LLR, un-permute by h (:1105-1127)
		const int bsyn = syn ^ beta[d];
		const double L0 = -1.0 * Lc[d][bsyn];
		const int e = hmul[d] ^ bsyn; // eta with h^-1 (eta ^ bsyn) = lane
		C[(size_t)d * Q + lane] = (lane == 0) ? 0.0 : shape_llr(-1.0 * Lc[d][e] - L0, r.factor, r.offset);
	}
}

} // namespace

bool nbl_tems64_applicable(const NblGraphDev &g, bool all_dc4, int nr, int nc)
{
	return g.q == 64 && all_dc4 && nc >= 1 && nc <= 3 && nr >= 1 && nr <= 4;
}

hipError_t nbl_launch_cn_tems64(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool fused, hipStream_t st)
{
	dim3 grid(nbl_xcd_grid(r.B, g.M)), block(64);
	switch ((fused ? 4 : 0) + r.nc) {
	case 1: cn_tems_q64_dc4_kernel<false, 1><<<grid, block, 0, st>>>(g, w, r); break;
	case 2: cn_tems_q64_dc4_kernel<false, 2><<<grid, block, 0, st>>>(g, w, r); break;
	case 3: cn_tems_q64_dc4_kernel<false, 3><<<grid, block, 0, st>>>(g, w, r); break;
	case 5: cn_tems_q64_dc4_kernel<true, 1><<<grid, block, 0, st>>>(g, w, r); break;
	case 6: cn_tems_q64_dc4_kernel<true, 2><<<grid, block, 0, st>>>(g, w, r); break;
	case 7: cn_tems_q64_dc4_kernel<true, 3><<<grid, block, 0, st>>>(g, w, r); break;
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}
