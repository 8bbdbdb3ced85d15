// nbldpc_amd/csrc/nbl_cn_tems256.hip -- trellis-EMS check node for GF(256), check degree 4, nc <= 3 (the six (2,4)-regular
// GF(256) codes the reference ships): NBLDPC.cpp:1055-1130, TEMS_Get_Beta :1789, TEMS_Get_deltaU :1814, TEMS_Get_Min :1836,
// TEMS_ConstructConf :1892.  Same arithmetic, comparisons and tie rules as the general kernels in nbl_cn_tems.hip (whose
// header explains why the min-plus dynamic programme over the columns equals the reference's path enumeration) and the same
// plan as the GF(64) kernel in nbl_cn_tems64.hip, with four symbols per lane:
//
//   * lane l owns the delta-domain symbols l, 64+l, 128+l, 192+l: trellis columns, column order, candidate marks and every
//     dynamic-programme state of those four check sums live in registers;
//   * layer 1 is a per-lane minimum (layer 0 is the constant {0 at check sum 0}); layers 2 and 3 gather the predecessor state
//     S[s ^ q] from LDS: for the lane's four symbols that is one address (lane ^ q) XOR-ed with the four plane offsets; the
//     candidates (q, cost) themselves come from the registers of their lanes (ballot + v_readlane), and the path codes are only
//     compared when a cost ties (relax_untied);
//   * LDS per wave: one 10 KB region used in phases (trellis while it is built and again in the output stage, predecessor
//     state during the programme, one output vector at a time) against 51 KB of the general kernel, which runs less than one
//     wave per SIMD at q = 256.
// FUSED = true additionally runs the variable-node pass of the iteration for the four incoming edges (dv = 2 codes), exactly
// as nbl_cn_tems64.hip does.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "nbl_device.h"
#include "nbl_kernels.h"

namespace {

constexpr int Q = 256, P = 8, DC = 4, NS = 4;


// smaller cost wins, equal cost: smaller path code (= earlier in the reference's enumeration order): the lexicographic
// comparison as mask arithmetic (three compares, the AND / OR on the scalar unit), then one select for the code; the cost
// itself is the plain minimum either way (as in nbl_cn_tems64.hip: 5 vector instructions per relaxation instead of 8)
__device__ __forceinline__ void relax(double &bv, unsigned &bc, double val, unsigned code)
{
	const bool take = (val < bv) | ((val == bv) & (code < bc));
	bc = take ? code : bc;
	bv = __builtin_fmin(bv, val);
}

// the same without the comparison of the path codes (nbl_cn_tems64.hip): right unless val == bv in some lane, which `tie` records
// (a wave-wide mask on the scalar unit) -- the caller then repeats the column with relax().  `from` = LDS offset of the winning
// predecessor: the path code is formed from it once per column.  "No path" must be NaN here, not infinity (inf == inf is a tie).
__device__ __forceinline__ void relax_untied(double &bv, int &from, double val, int off, uint64_t &tie)
{
	tie |= __ballot(val == bv);
	from = (val < bv) ? off : from;
	asm("v_min_f64 %0, %1, %2" : "=v"(bv) : "v"(bv), "v"(val)); // (__builtin_fmin canonicalises the running minimum first)
}

__device__ __forceinline__ double pick(const double (&u)[DC], int k)
{
	return k == 0 ? u[0] : k == 1 ? u[1] : k == 2 ? u[2] : u[3];
}

// NC = the deviation budget tems_nc (1..3) as a template parameter: with a run-time nc the layer-3 update sits behind a branch
// in the candidate loop and the 16-byte predecessor loads are split in two
template <bool FUSED, int NC>
__global__ __launch_bounds__(64, 4) void cn_tems_q256_dc4_kernel(NblGraphDev g, NblWork w, NblRun r)
{
	// One 10 KB region, used in two phases (sixteen waves per CU = four per SIMD; the register count allows as many):
	//   programme:  predecessor state Sv1 | Sv2 (2 KB each: costs of layers 1, 2) | Sc (2 KB: 8-byte code pairs) -- three arrays of
	//               8-byte entries, so one offset serves all three gathers and each touches every LDS bank once -- and the
	//               column-start copy of layer 3 (3 KB), which only the exact pass after a tie reads
	//   before / after it:  the trellis dU (8 KB, rebuilt from registers for the output stage) + one output vector Lc (2 KB)
	__shared__ __attribute__((aligned(16))) char lds[Q * 16 + Q * 8 + Q * 16];
	double *Sv1 = (double *)lds, *Sv2 = Sv1 + Q;           // [Q], [Q] cost of layers 1, 2 of every check sum before the current column (NaN = no path)
	uint2 *Sc = (uint2 *)(lds + Q * 16);                   // [Q] their path codes (x, y)
	double *V3s = (double *)(lds + Q * 24);                // [Q] layer 3 at the start of the column, cost
	unsigned *C3s = (unsigned *)(lds + Q * 32);            // [Q] and path code (read back only when the column ties)
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
		// (symbol 0 holds 0; nothing positive -- every vector of a converged all-zero codeword -- is settled by one compare)
		const double lmx = dmax(dmax(v[0], v[1]), dmax(v[2], v[3]));
		double mx = 0.0;
		int arg = 0;
		if (__ballot(lmx > 0.0)) {
			mx = wave_fmax_nonneg(lmx);
#pragma unroll
			for (int i = NS - 1; i >= 0; i--) {
				const uint64_t hit = __ballot(v[i] == mx);
				arg = hit ? 64 * i + __builtin_ctzll(hit) : arg;
			}
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
	struct DpState { double v1; unsigned c1; double v2; unsigned c2; double v3; unsigned c3; } st[NS];
#pragma unroll
	for (int i = 0; i < NS; i++) {
		st[i].v1 = st[i].v2 = st[i].v3 = INF;
		st[i].c1 = st[i].c2 = st[i].c3 = 0;
		const int s = lane + 64 * i;
		if ((mask[i] & 1) && s > 0) { st[i].v1 = u[i][0]; st[i].c1 = (unsigned)s << (P * (DC - 1)); }
	}
	int lane8 = lane << 3;
	asm("" : "+v"(lane8)); // (opaque: otherwise (lane << 3) ^ (q << 3) becomes (lane ^ q) << 3, a second vector instruction per gather)
	// one candidate (cost cu at deviation symbol q) of column d folded into layers 2 (and 3) of the lane's four check sums:
	// the predecessor of check sum s is s ^ q -- one offset (lane ^ q) XOR-ed with the four plane offsets
	auto fold_one = [&](auto exact_tag, bool both, int q, double cu, int sh, int (&from2)[NS], int (&from3)[NS], uint64_t &tie) __attribute__((always_inline)) {
		constexpr bool EXACT = decltype(exact_tag)::value;
		const int A = lane8 ^ (q << 3);
#pragma unroll
		for (int i = 0; i < NS; i++) {
			const int off = A ^ (i << 9);
			const double s1 = *(const double *)((const char *)Sv1 + off);
			if (EXACT) {
				const unsigned dig = (unsigned)q << sh;
				const uint2 sc = *(const uint2 *)((const char *)Sc + off);
				relax(st[i].v2, st[i].c2, s1 + cu, sc.x + dig);
				if (both) relax(st[i].v3, st[i].c3, *(const double *)((const char *)Sv2 + off) + cu, sc.y + dig);
			} else {
				relax_untied(st[i].v2, from2[i], s1 + cu, off, tie);
				if (both) relax_untied(st[i].v3, from3[i], *(const double *)((const char *)Sv2 + off) + cu, off, tie);
			}
		}
	};
	// the whole column: its candidates are the marked non-zero symbols (symbol 0 = "no deviation"), read from the registers of
	// their lanes plane by plane (ballot + v_readlane: no candidate list in LDS)
	auto fold = [&](auto exact_tag, auto both_tag, auto d_tag, uint64_t cm0, uint64_t cm1, uint64_t cm2, uint64_t cm3, uint64_t &tie) __attribute__((always_inline)) {
		constexpr bool EXACT = decltype(exact_tag)::value, both = decltype(both_tag)::value;
		constexpr int D = decltype(d_tag)::value, sh = P * (DC - 1 - D);
		int from2[NS], from3[NS]; // (start: the check sum's own offset, which no candidate produces -- q = 0 is none)
#pragma unroll
		for (int i = 0; i < NS; i++) from2[i] = from3[i] = lane8 ^ (i << 9);
#pragma unroll
		for (int ip = 0; ip < NS; ip++) {
			uint64_t m = ip == 0 ? cm0 : ip == 1 ? cm1 : ip == 2 ? cm2 : cm3;
			int left = __builtin_popcountll(m);
			for (; left >= 2; left -= 2) {
				const int l0 = __builtin_ctzll(m);
				m &= m - 1;
				const int l1 = __builtin_ctzll(m);
				m &= m - 1;
				const double cu0 = read_lane_f64(u[ip][D], l0), cu1 = read_lane_f64(u[ip][D], l1);
				fold_one(exact_tag, both, l0 + 64 * ip, cu0, sh, from2, from3, tie);
				fold_one(exact_tag, both, l1 + 64 * ip, cu1, sh, from2, from3, tie);
			}
			if (left) {
				const int l0 = __builtin_ctzll(m);
				fold_one(exact_tag, both, l0 + 64 * ip, read_lane_f64(u[ip][D], l0), sh, from2, from3, tie);
			}
		}
		if (!EXACT) { // the path codes of the layers this column improved
#pragma unroll
			for (int i = 0; i < NS; i++) {
				const int own = lane8 ^ (i << 9);
				const unsigned n2 = Sc[from2[i] >> 3].x + ((unsigned)((from2[i] ^ own) >> 3) << sh);
				st[i].c2 = (from2[i] != own) ? n2 : st[i].c2;
				if (both) {
					const unsigned n3 = Sc[from3[i] >> 3].y + ((unsigned)((from3[i] ^ own) >> 3) << sh);
					st[i].c3 = (from3[i] != own) ? n3 : st[i].c3;
				}
			}
		}
	};
	auto column = [&](auto d_tag) __attribute__((always_inline)) {
		constexpr int d = decltype(d_tag)::value, sh = P * (DC - 1 - d); // sh = digit of column d in the path code
		const double NOPATH = __builtin_nan("");
		uint64_t cm[NS];
		// Candidates that cannot matter are left out (nbl_cn_tems64.hip): a path through a deviation of cost u costs at least u, and
		// check sum s is already reached for best[s] = min(cheapest single deviation, v2[s], v3[s]); u > max_s best[s] neither wins nor ties
		// -- and per layer: a layer-2 path through this column costs at least u + min v1, a layer-3 path at least u + min v2 (the sums as
		// the relaxation itself rounds them: rounding is monotone, so every actual value is >= the rounded bound)
		constexpr bool both = (d >= 2 && nc >= 3);
		double reach = 0.0, least1 = INF, least2 = INF;
#pragma unroll
		for (int i = 0; i < NS; i++) {
			const double m1 = (lane + 64 * i == 0) ? 0.0 : dmin(dmin(u[i][0], u[i][1]), dmin(u[i][2], u[i][3]));
			reach = dmax(reach, __builtin_fmin(__builtin_fmin(m1, st[i].v2), st[i].v3));
			least1 = dmin(least1, st[i].v1);
			least2 = dmin(least2, st[i].v2);
		}
		reach = wave_fmax_nonneg(reach);
		least1 = wave_fmin_nonneg(least1);
		least2 = both ? wave_fmin_nonneg(least2) : 0.0;
		uint64_t cm3[NS];
#pragma unroll
		for (int i = 0; i < NS; i++) {
			const int s = lane + 64 * i;
			Sv1[s] = (st[i].v1 == INF) ? NOPATH : st[i].v1;
			Sv2[s] = (st[i].v2 == INF) ? NOPATH : st[i].v2;
			Sc[s] = make_uint2(st[i].c1, st[i].c2);
			V3s[s] = st[i].v3; // (what the exact pass starts from, should this column tie)
			C3s[s] = st[i].c3;
			cm[i] = __ballot(((mask[i] >> d) & 1) && s > 0 && u[i][d] + least1 <= reach);
			cm3[i] = both ? (cm[i] & __ballot(u[i][d] + least2 <= reach)) : 0;
			cm[i] &= ~cm3[i]; // layer 2 only
		}
		__syncthreads();
		if (nc >= 2) {
			uint64_t tie = 0;
			if (both) fold(std::false_type{}, std::true_type{}, std::integral_constant<int, d>{}, cm3[0], cm3[1], cm3[2], cm3[3], tie);
			fold(std::false_type{}, std::false_type{}, std::integral_constant<int, d>{}, cm[0], cm[1], cm[2], cm[3], tie);
			if (tie) { // equal costs somewhere: the column again from its start, path codes compared
#pragma unroll
				for (int i = 0; i < NS; i++) {
					const int s = lane + 64 * i;
					const double v2s = Sv2[s];
					st[i].v2 = (v2s != v2s) ? INF : v2s;
					st[i].c2 = Sc[s].y;
					st[i].v3 = V3s[s];
					st[i].c3 = C3s[s];
				}
				if (both) fold(std::true_type{}, std::true_type{}, std::integral_constant<int, d>{}, cm3[0], cm3[1], cm3[2], cm3[3], tie);
				fold(std::true_type{}, std::false_type{}, std::integral_constant<int, d>{}, cm[0], cm[1], cm[2], cm[3], tie);
			}
		}
#pragma unroll
		for (int i = 0; i < NS; i++) {
			const int s = lane + 64 * i;
			if (((mask[i] >> d) & 1) && s > 0) relax(st[i].v1, st[i].c1, u[i][d], (unsigned)s << sh);
		}
		__syncthreads();
	};
	column(std::integral_constant<int, 1>{});
	column(std::integral_constant<int, 2>{});
	column(std::integral_constant<int, 3>{});
	// dW, Eta: best layer of each of the lane's check sums
	double dW[NS];
	unsigned eta[NS];
#pragma unroll
	for (int i = 0; i < NS; i++) {
		dW[i] = INF;
		eta[i] = 0xffffffffu;
		const double v0 = (lane == 0 && i == 0) ? 0.0 : INF;
		if (v0 < dW[i] || (v0 == dW[i] && 0u < eta[i])) { dW[i] = v0; eta[i] = 0u; }
		if (nc >= 1 && (st[i].v1 < dW[i] || (st[i].v1 == dW[i] && st[i].c1 < eta[i]))) { dW[i] = st[i].v1; eta[i] = st[i].c1; }
		if (nc >= 2 && (st[i].v2 < dW[i] || (st[i].v2 == dW[i] && st[i].c2 < eta[i]))) { dW[i] = st[i].v2; eta[i] = st[i].c2; }
		if (nc >= 3 && (st[i].v3 < dW[i] || (st[i].v3 == dW[i] && st[i].c3 < eta[i]))) { dW[i] = st[i].v3; eta[i] = st[i].c3; }
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
