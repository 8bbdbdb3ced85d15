// nbldpc_amd/csrc/nbl_cn_small.hip -- check-node kernels for small fields (q <= 32: the three shipped GF(16) codes, BASELINE
// config 1): EMS (NBLDPC.cpp:859-917), T-EMS (:1055-1130) and log-QSPA (:747-767).  (The templates also compile for q = 64, the
// degenerate case of one check per wave; only log-QSPA uses that -- irregular GF(64) codes -- see nbl_small_applicable.)
//
// The general kernels (nbl_kernels.hip, nbl_cn_tems.hip, nbl_cn_bp.hip) give every check a whole wave, lane = symbol: at q = 16
// three quarters of the lanes idle and every step of the (short, latency-bound) chain of LDS phases is paid per check.  Here a
// wave holds 64 / q checks: lane = (check of the wave, symbol), every group of q lanes works on its own LDS region, reductions
// run inside the group (DPP row operations: a group never spans a DPP row boundary it does not own).  The arithmetic per check
// is the general kernels' -- the same candidates, the same left-to-right sums, the same tie rules -- so the results are
// bit-identical to theirs for EMS and T-EMS (tests/test_gpu_parity.py runs both and the oracle); log-QSPA keeps vectors as
// mantissa x 2^exponent like nbl_cn_bp256.hip (LLRs within 1e-9 of the oracle's FP64 restatement, identical decisions).
//
// Groups of one wave are independent: checks of different degree (the GF(16) codes mix 4 and 5), different candidate counts or
// different numeric ranges simply diverge; no lane of one group ever reads a register of another.  One wave per workgroup, so
// the phases only need the compiler to keep the order of the LDS operations (WSYNC), not a barrier.
#include <hip/hip_runtime.h>
#include "nbl_device.h"
#include "nbl_fastmath.h"
#include "nbl_kernels.h"

namespace {

#define WSYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")

template <int CTRL> __device__ __forceinline__ double dppx_f64(double x)
{
	const int lo = __builtin_amdgcn_update_dpp(__double2loint(x), __double2loint(x), CTRL, 0xF, 0xF, false);
	const int hi = __builtin_amdgcn_update_dpp(__double2hiint(x), __double2hiint(x), CTRL, 0xF, 0xF, false);
	return __hiloint2double(hi, lo);
}
// maximum over the Q lanes of the group, delivered to every lane of it (butterfly: xor 1, xor 2, mirror 8, mirror 16, xor 16)
template <int Q> __device__ __forceinline__ double gmax_f64(double v)
{
	if (Q >= 2) v = dmax(v, dppx_f64<0xB1>(v));   // quad_perm [1,0,3,2]
	if (Q >= 4) v = dmax(v, dppx_f64<0x4E>(v));   // quad_perm [2,3,0,1]
	if (Q >= 8) v = dmax(v, dppx_f64<0x141>(v));  // row_half_mirror
	if (Q >= 16) v = dmax(v, dppx_f64<0x140>(v)); // row_mirror
	if (Q >= 32) v = dmax(v, __shfl_xor(v, 16, 64));
	if (Q >= 64) v = dmax(v, __shfl_xor(v, 32, 64));
	return v;
}
template <int Q> __device__ __forceinline__ int gmax_i32(int v)
{
	constexpr int ID = (int)0x80000000;
	if (Q >= 2) v = max(v, __builtin_amdgcn_update_dpp(ID, v, 0xB1, 0xF, 0xF, false));
	if (Q >= 4) v = max(v, __builtin_amdgcn_update_dpp(ID, v, 0x4E, 0xF, 0xF, false));
	if (Q >= 8) v = max(v, __builtin_amdgcn_update_dpp(ID, v, 0x141, 0xF, 0xF, false));
	if (Q >= 16) v = max(v, __builtin_amdgcn_update_dpp(ID, v, 0x140, 0xF, 0xF, false));
	if (Q >= 32) v = max(v, __shfl_xor(v, 16, 64));
	if (Q >= 64) v = max(v, __shfl_xor(v, 32, 64));
	return v;
}
// the group's bits of a wave ballot, bit i = lane i of the group
template <int Q> __device__ __forceinline__ uint64_t gballot(bool p, int gi)
{
	const uint64_t m = __ballot(p);
	return (Q == 64) ? m : (m >> (gi * (Q & 63))) & ((1ull << (Q & 63)) - 1ull);
}

// which (codeword, check) the lane's group works on
template <int Q> struct Ctx { int gi, sl, b, c0, dc; bool live; };
template <int Q> __device__ __forceinline__ Ctx<Q> ctx_init(const NblGraphDev &g, const NblWork &w, const NblRun &r)
{
	constexpr int P = Fld<Q>::P, G = 64 / Q;
	Ctx<Q> c;
	const int lane = lane_id();
	c.gi = lane >> P;
	c.sl = lane & (Q - 1);
	// (g.M + G - 1) / G workgroups per codeword, all of them on one XCD (nbl_device.h); group gi of workgroup `unit` holds check
	// G unit + gi
	const NblXcdSlot xs = nbl_xcd_slot((g.M + G - 1) / G);
	const int m = xs.unit * G + c.gi;
	c.b = -1;
	if (m < g.M) {
		c.b = nbl_codeword(w, r, xs.slot);
		if (c.b >= 0 && !r.fixed_iters && w.done[c.b]) c.b = -1;
	}
	c.live = c.b >= 0;
	c.c0 = c.live ? g.coff[m] : 0;
	c.dc = c.live ? g.coff[m + 1] - c.c0 : 0;
	return c;
}

// DecideLLRVector (:1542-1562) inside the group: lowest symbol among the maxima of {0, v}; v of symbol 0 must be <= 0
template <int Q> __device__ __forceinline__ int gdecide(double v, int gi)
{
	const double mx = dmax(gmax_f64<Q>(v), 0.0);
	const uint64_t hit = gballot<Q>(v == mx, gi);
	return (mx > 0.0 && hit) ? __builtin_ctzll(hit) : 0;
}

// The dc incoming vectors of the group's check, handed to `stage(d, v, vec)` edge by edge (v = the lane's symbol of v2c_d, 0 for
// symbol 0; vec = the whole vector in LDS, variable-domain order).  CH EDGES AT A TIME: every global load of a chunk -- indices
// first, then data -- is issued before the first value is used, because the stage bodies are chains of LDS phases behind
// fences that no load can be hoisted over.
//
// FUSED (one launch = a whole flooding iteration, c2v double-buffered, variable degrees <= 3): the variable-node pass of a
// check-major edge is recomputed from the previous iteration's c2v by the group that needs the message -- a-posteriori sum
// L_ch + c2v_0 + c2v_1 (+ c2v_2) in edge order (:676-691 / :808-823 / :977-992), hard decision written by the check that holds
// the variable's first edge, v2c = post - c2v of this edge, and for the damped methods (DAMP: log-QSPA 1/2 : 1/2 :730-741,
// T-EMS 1/4 : 3/4 :1029-1052) the blend with the previous v2c when the message's hard decision moved.  The previous decision is
// what this stage recorded one iteration ago (w.edge_dec), as in the GF(256) kernels; g.c_nbr holds the c2v slots of the
// variable's edges, so the data loads are one index load deep.
// The values of a chunk wait in `vin` ([CH][Q] doubles of the group's LDS region), so the stage body exists once in the code.
// CH trades loads in flight against registers (and registers against waves per SIMD: these kernels are latency-bound, their
// throughput follows the occupancy).
template <int Q, bool FUSED, bool DAMP, int CH, class Stage>
__device__ __forceinline__ void for_each_input(const NblGraphDev &g, const NblWork &w, const NblRun &r, const Ctx<Q> &c, double *vin, Stage &&stage)
{
	const int sl = c.sl, dc = c.dc, c0 = c.c0;
	for (int base = 0; base < dc; base += CH) {
		if (!FUSED) {
			const double *V = w.v2c + (size_t)c.b * g.E * Q;
			double v[CH];
#pragma unroll
			for (int u = 0; u < CH; u++) {
				const int d = (base + u < dc) ? base + u : 0;
				v[u] = V[(size_t)g.c_epos[c0 + d] * Q + sl];
			}
#pragma unroll
			for (int u = 0; u < CH; u++) vin[u * Q + sl] = (sl > 0) ? v[u] : 0.0;
		} else {
			const double *Cp = w.c2v_prev + (w.c2v_prev_shared ? (size_t)0 : (size_t)c.b * g.E * Q); // (iteration 1: one shared block of zeros)
			int4 row[CH];
			int n[CH], e[CH], before[CH];
			double L[CH], x0[CH], x1[CH], x2[CH], ov[CH];
#pragma unroll
			for (int u = 0; u < CH; u++) {
				const int ce = c0 + ((base + u < dc) ? base + u : 0);
				row[u] = ((const int4 *)g.c_nbr)[ce];
				n[u] = g.c_var[ce];
				e[u] = g.c_epos[ce];
			}
#pragma unroll
			for (int u = 0; u < CH; u++) {
				L[u] = w.Lch[((size_t)c.b * g.N + n[u]) * Q + sl];
				x0[u] = Cp[(size_t)row[u].x * Q + sl];
				x1[u] = Cp[(size_t)row[u].y * Q + sl];
				x2[u] = Cp[(size_t)(row[u].z >= 0 ? row[u].z : row[u].x) * Q + sl];
				if (DAMP) {
					ov[u] = w.v2c[((size_t)c.b * g.E + e[u]) * Q + sl];
					before[u] = (r.iter > 1) ? w.edge_dec[(size_t)c.b * g.E + e[u]] : 0;
				}
			}
#pragma unroll
			for (int u = 0; u < CH; u++) {
				if (base + u < dc) {
					const int ce = c0 + base + u;
					double post = (L[u] + x0[u]) + x1[u];
					post = (row[u].z >= 0) ? post + x2[u] : post;
					double nv = post - ((row[u].x == ce) ? x0[u] : (row[u].y == ce) ? x1[u] : x2[u]);
					if (row[u].w) { // the variable's first edge: this check reports the hard decision (and the a-posteriori vector)
						const int dec = gdecide<Q>(post, c.gi);
						if (sl == 0) w.dec[(size_t)c.b * g.N + n[u]] = dec;
						if (w.post) w.post[((size_t)c.b * g.N + n[u]) * Q + sl] = post;
					}
					if (DAMP) {
						// iteration 1: the previous v2c is L_ch (init kernel), nothing has been recorded yet
						const int bf = (r.iter == 1) ? gdecide<Q>(ov[u], c.gi) : before[u];
						int after = gdecide<Q>(nv, c.gi);
						if (bf != after) {
							nv = __dadd_rn(__dmul_rn(r.damp_old, ov[u]), __dmul_rn(r.damp_new, nv));
							after = gdecide<Q>((sl == 0) ? 0.0 : nv, c.gi);
						}
						if (sl == 0) w.edge_dec[(size_t)c.b * g.E + e[u]] = after;
						w.v2c[((size_t)c.b * g.E + e[u]) * Q + sl] = (sl == 0) ? 0.0 : nv;
					} else if (w.store_v2c) {
						w.v2c[((size_t)c.b * g.E + e[u]) * Q + sl] = (sl == 0) ? 0.0 : nv;
					}
					vin[u * Q + sl] = (sl == 0) ? 0.0 : nv;
				}
			}
		}
		WSYNC(); // (the stages read other lanes' staged values)
#pragma unroll 1
		for (int u = 0; u < CH && base + u < dc; u++) stage(base + u, vin[u * Q + sl], vin + u * Q);
	}
}

// =====================================================================================================================
// T-EMS (the steps and their reference lines are those of cn_tems_fast_kernel, nbl_cn_tems.hip; nc <= 3)
// =====================================================================================================================
struct __attribute__((aligned(16))) TState { double v[4]; unsigned c[4]; };

__host__ __device__ inline size_t tems_small_group_bytes(int q, int mdc)
{
	const size_t n = (size_t)8 * (mdc * q + q) + (size_t)q * sizeof(TState);
	return (n + 15) & ~(size_t)15;
}

template <int Q, bool FUSED>
__global__ __launch_bounds__(64, 6) void cn_tems_small_kernel(NblGraphDev g, NblWork w, NblRun r)
{
	constexpr int P = Fld<Q>::P, TCH = 2; // (TState is 48 bytes per symbol: room for 6 staged vectors)
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const Ctx<Q> c = ctx_init<Q>(g, w, r);
	if (!c.live) return;
	const int sl = c.sl, dc = c.dc, c0 = c.c0, nr = r.nr, nc = r.nc, mdc = g.maxdc;
	char *base = smem + (size_t)c.gi * tems_small_group_bytes(Q, mdc);
	double *dU = (double *)base;                 // [mdc][Q]
	double *Lc = dU + mdc * Q;                   // [Q]
	TState *A = (TState *)(Lc + Q);              // [Q] DP states, updated in place (every lane reads, fence, every lane writes)
	double *vin = (double *)A;                   // [TCH][Q] the incoming values of a chunk of edges (before the DP starts)

	double *C = w.c2v + ((size_t)c.b * g.E + c0) * Q;

	// ---- 1. beta, syndrome, dU (TEMS_Get_Beta :1789-1812, TEMS_Get_deltaU :1814-1834) -------------------------------
	int syn = 0;
	unsigned long long betas = 0; // beta_d, 8 bits each (group-uniform)
	for_each_input<Q, FUSED, true, TCH>(g, w, r, c, vin, [&](int d, double v, const double *) {
		// most reliable symbol: strict '>' over ascending symbols from a running maximum of 0 (:1798-1807)
		const double mx = dmax(gmax_f64<Q>(v), 0.0);
		const uint64_t hit = gballot<Q>(v == mx, c.gi);
		const int arg = (mx > 0.0 && hit) ? __builtin_ctzll(hit) : 0;
		const uint8_t *mrow = g.mul + (size_t)g.c_h[c0 + d] * Q;
		const int bd = mrow[arg]; // beta_d = h * argmax (0 if nothing positive)
		betas |= (unsigned long long)bd << (8 * d);
		syn ^= bd;
		dU[d * Q + (mrow[sl] ^ bd)] = mx - v; // dU[d][h a ^ beta] = mx - L(a), L(0) = 0  (:1826-1831)
	});
	WSYNC();

	// ---- 2. per deviation symbol: stable ascending order of the columns, Nr smallest marked (TEMS_Get_Min :1836-1890) ----
	// the lane keeps what belongs to its own symbol: `mask` bit d = column d may deviate to it, o0 / o1 = first two columns
	int mask = 0, o0 = 0, o1 = 0;
	for (int d = 0; d < dc; d++) {
		const double ud = dU[d * Q + sl];
		int rank = 0;
		for (int e = 0; e < dc; e++) {
			const double ue = dU[e * Q + sl];
			rank += (ue < ud || (ue == ud && e < d)) ? 1 : 0;
		}
		if (rank < nr) mask |= 1 << d;
		if (rank == 0) o0 = d;
		if (rank == 1) o1 = d;
	}
	if (sl == 0) mask = 0; // symbol 0 = "no deviation" is handled apart
	// the cheapest single deviation that reaches the lane's check sum (its column has rank 0 there: always marked)
	double m1 = 0.0;
	if (sl > 0) {
		m1 = dU[sl];
		for (int d = 1; d < dc; d++) m1 = dmin(m1, dU[d * Q + sl]);
	}

	// ---- 3. min-plus DP over the columns; all deviation-count layers advance together (TEMS_ConstructConf :1892-1944) ----
	{
		TState z;
#pragma unroll
		for (int l = 0; l < 4; l++) { z.v[l] = __builtin_huge_val(); z.c[l] = 0; }
		if (sl == 0) z.v[0] = 0.0;
		A[sl] = z;
	}
	WSYNC();
	for (int d = 0; d < dc; d++) {
		// the deviation symbols of column d as a bit set of the group (every order gives the same minimum and the same path code)
		TState b = A[sl]; // q_d = 0: dU[d][0] = 0 (:1826), cost unchanged
		// Candidates that cannot matter are left out (nbl_cn_tems64.hip): every cost is >= 0, so a path through a deviation of cost u
		// costs at least u, and check sum s is already reached for best[s] = min(m1[s], the layers of s so far) -- a bound that only
		// falls from here on; u > max_s best[s] can neither win nor tie anywhere, directly or as a predecessor.  (nc >= 1 here.)
		const double reach = gmax_f64<Q>(dmin(dmin(m1, b.v[0]), dmin(b.v[1], dmin(b.v[2], b.v[3]))));
		uint64_t cand = gballot<Q>(((mask >> d) & 1) && dU[d * Q + sl] <= reach, c.gi);
#pragma unroll
		for (int l = 0; l < 4; l++) { b.v[l] = b.v[l] + 0.0; b.c[l] = b.c[l] * Q; }
		while (cand) {
			const int q = __builtin_ctzll(cand);
			cand &= cand - 1;
			const double u = dU[d * Q + q];
			const TState src = A[sl ^ q];
#pragma unroll
			for (int l = 1; l < 4; l++) {
				if (l <= nc && l <= d + 1) { // a path through d+1 columns has at most d+1 deviations
					const double val = src.v[l - 1] + u;
					const unsigned code = src.c[l - 1] * Q + q;
					// smaller cost wins, equal cost: smaller path code
					const bool lt = val < b.v[l];
					b.v[l] = lt ? val : b.v[l];
					b.c[l] = lt ? code : b.c[l];
					const unsigned cm = code < b.c[l] ? code : b.c[l];
					b.c[l] = (val == b.v[l]) ? cm : b.c[l];
				}
			}
		}
		WSYNC();
		A[sl] = b;
		WSYNC();
	}
	// dW, Eta: best layer per check sum
	double dW = __builtin_huge_val();
	unsigned eta = 0xffffffffu;
	{
		const TState f = A[sl];
#pragma unroll
		for (int l = 0; l < 4; l++)
			if (l <= nc && (f.v[l] < dW || (f.v[l] == dW && f.c[l] < eta))) { dW = f.v[l]; eta = f.c[l]; }
	}

	// ---- 4. outputs (:1075-1129) -----------------------------------------------------------------------------------------
	int pshift = (dc - 1) * P; // digit of column d sits at bit P*(dc-1-d) of the path code
	for (int d = 0; d < dc; d++) {
		WSYNC();
		Lc[sl] = NBL_DBL_MAX;
		WSYNC();
		{
			const int dev = (int)((eta >> pshift) & (Q - 1));
			const double cand = dW - dU[d * Q + dev];                            // :1088
			__hip_atomic_fetch_min(&Lc[sl ^ dev], cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		WSYNC();
		if (Lc[sl] == NBL_DBL_MAX) Lc[sl] = (d == o0) ? dU[o1 * Q + sl] : dU[o0 * Q + sl]; // never reached (:1095-1102)
		WSYNC();
		// delta domain -> LLR, un-permute by h (:1105-1127)
		const int bsyn = syn ^ (int)((betas >> (8 * d)) & 255);
		const double L0 = -1.0 * Lc[bsyn];
		const uint8_t *mrow = g.mul + (size_t)g.c_h[c0 + d] * Q;
		const int e = mrow[sl] ^ bsyn; // eta with h^-1 (eta ^ bsyn) = a
		C[(size_t)d * Q + sl] = (sl == 0) ? 0.0 : shape_llr(-1.0 * Lc[e] - L0, r.factor, r.offset);
		pshift -= P;
	}
}

// =====================================================================================================================
// EMS (max-plus dynamic programme of cn_ems_kernel, nbl_kernels.hip; at most four deviation-count layers)
// =====================================================================================================================
struct __attribute__((aligned(16))) EmsEnt { double v; int t; int pad; }; // list entry: value, check-domain symbol

__host__ __device__ inline size_t ems_small_group_bytes(int q, int mdc, int nm, int layers)
{
	const size_t dp = layers > 2 ? layers : 2; // DP layers (updated in place); the staging of a chunk of two inputs aliases them
	const size_t n = (size_t)8 * ((size_t)mdc * q + dp * q) + (size_t)16 * mdc * nm;
	return (n + 15) & ~(size_t)15;
}

// NMT: ems_nm as a compile-time constant (8: the list loops unroll and their LDS reads are issued together) or 0 = run-time
template <int Q, bool FUSED, int NMT>
__global__ __launch_bounds__(64) void cn_ems_small_kernel(NblGraphDev g, NblWork w, NblRun r, int layers)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const Ctx<Q> c = ctx_init<Q>(g, w, r);
	if (!c.live) return;
	const int sl = c.sl, dc = c.dc, c0 = c.c0, nm = NMT ? NMT : r.nm, mdc = g.maxdc;
	char *base = smem + (size_t)c.gi * ems_small_group_bytes(Q, mdc, nm, layers);
	double *U = (double *)base;        // [mdc][Q]   check-domain input vectors: U[j][h_j*a] = v2c_j[a], U[j][0] = 0
	double *A = U + mdc * Q;           // [layers][Q] DP layers, updated in place (every lane reads, fence, every lane writes)
	double *vin = A;                   // [2][Q]     the incoming values of a chunk of edges (before the DP starts; >= 2 vectors there)
	double *Sv = A;                    // [Q]        the maxima of the current output edge (after its DP)
	EmsEnt *ls = (EmsEnt *)(A + (layers > 2 ? layers : 2) * Q); // [mdc][nm] the nm most reliable entries of every edge, by rank (rank 0 first)

	double *C = w.c2v + ((size_t)c.b * g.E + c0) * Q;
	// list entries k0 .. nm-1: unrolled when nm is a compile-time constant
	auto for_entries = [&](int k0, auto &&body) {
		if constexpr (NMT > 0) {
#pragma unroll
			for (int k = k0; k < NMT; k++) body(k);
		} else {
			for (int k = k0; k < nm; k++) body(k);
		}
	};

	// ---- stage the dc incoming vectors: permute into the check domain, rank, keep the nm best ---------------------------
	for_each_input<Q, FUSED, false, 2>(g, w, r, c, vin, [&](int j, double v, const double *vec) {
		const int t = g.mul[(size_t)g.c_h[c0 + j] * Q + sl];
		U[j * Q + t] = v;
		// rank under SortLLRVector's order (:1715-1746): value descending, among equal values the HIGHER symbol first
		int rank = 0;
#pragma unroll 4
		for (int x = 0; x < Q; x++) {
			const double vo = vec[x];
			rank += (vo > v || (vo == v && x > sl)) ? 1 : 0;
		}
		if (rank < nm) {
			EmsEnt e;
			e.v = v;
			e.t = t;
			e.pad = 0;
			ls[j * nm + rank] = e;
		}
	});
	WSYNC();

	// ---- one output edge at a time -----------------------------------------------------------------------------------------
	for (int x = 0; x < dc; x++) {
#define OTH(l) ((l) + ((l) >= x ? 1 : 0))
		const int rn = dc - 1;
		int zall = 0;
		for (int l = 0; l < rn; l++) zall ^= ls[OTH(l) * nm].t;
		double S = -NBL_DBL_MAX;

		// conf(q,1): at most one edge deviates, to ANY symbol (:894)
		for (int pi = 0; pi < rn; pi++) {
			const int jd = OTH(pi);
			const int shift = zall ^ ls[jd * nm].t;
			const double u = U[jd * Q + (sl ^ shift)];
			double acc = 0.0;
			for (int l = 0; l < rn; l++) acc = acc + ((l == pi) ? u : ls[OTH(l) * nm].v);
			S = dmax(S, acc);
		}

		// conf(nm,nc): at most nc edges deviate, each inside its nm best (:897); conf(nm,0) is the all-rank-0 configuration alone,
		// which conf(q,1) already holds
		WSYNC();
		if (r.nc < 1) {
		} else if (layers == 1) {
			// nc >= dc-1: no deviation counting needed -> plain truncated max-plus convolution
			A[sl] = NBL_NEG_INF;
			WSYNC();
			if (rn == 1) {
				const int j1 = OTH(0);
				for (int k = sl; k < nm; k += Q) A[ls[j1 * nm + k].t] = 0.0 + ls[j1 * nm + k].v;
			} else {
				const int j1 = OTH(0), j2 = OTH(1);
				const unsigned inv = ((1u << 20) + nm - 1) / nm; // idx / nm for idx < nm * nm <= 4096 without an integer division
				for (int idx = sl; idx < nm * nm; idx += Q) {
					const int k1 = (int)(((unsigned)idx * inv) >> 20), k2 = idx - k1 * nm;
					const EmsEnt e1 = ls[j1 * nm + k1], e2 = ls[j2 * nm + k2];
					__hip_atomic_fetch_max(&A[e1.t ^ e2.t], (0.0 + e1.v) + e2.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
			}
			WSYNC();
			for (int l = 2; l < rn; l++) {
				const int jl = OTH(l);
				double acc = NBL_NEG_INF;
				for_entries(0, [&](int k) {
					const EmsEnt e = ls[jl * nm + k];
					acc = dmax(acc, A[sl ^ e.t] + e.v);
				});
				if (l == rn - 1) S = dmax(S, acc);
				else {
					WSYNC();
					A[sl] = acc;
					WSYNC();
				}
			}
			if (rn <= 2) S = dmax(S, A[sl]);
		} else {
			// layered DP: A[d][s] = best value reaching check sum s with exactly d deviations.  After the first other edge this is
			// layer 0 = {z: 0 + rank-0 value}, layer 1 = {t_k: 0 + v_k, k >= 1} (what the general kernel's first step computes from
			// the start state {0: 0}); after l edges no path has more than l deviations, so layer d is only touched for d <= l + 1.
			for (int d = 0; d < layers; d++) A[d * Q + sl] = NBL_NEG_INF;
			WSYNC();
			{
				const int j0 = OTH(0);
				for (int k = sl; k < nm; k += Q) {
					const EmsEnt e = ls[j0 * nm + k];
					A[(k == 0 ? 0 : Q) + e.t] = 0.0 + e.v;
				}
			}
			WSYNC();
			for (int l = 1; l < rn; l++) {
				const int jl = OTH(l);
				const EmsEnt top = ls[jl * nm];
				double acc[4];
#pragma unroll
				for (int d = 0; d < 4; d++) acc[d] = (d < layers) ? A[d * Q + (sl ^ top.t)] + top.v : NBL_NEG_INF;
				for_entries(1, [&](int k) {
					const EmsEnt e = ls[jl * nm + k];
#pragma unroll
					for (int d = 1; d < 4; d++)
						if (d < layers && d <= l + 1) acc[d] = dmax(acc[d], A[(d - 1) * Q + (sl ^ e.t)] + e.v);
				});
				if (l == rn - 1) {
#pragma unroll
					for (int d = 0; d < 4; d++)
						if (d < layers) S = dmax(S, acc[d]);
				} else {
					WSYNC();
#pragma unroll
					for (int d = 0; d < 4; d++)
						if (d < layers) A[d * Q + sl] = acc[d];
					WSYNC();
				}
			}
			if (rn == 1)
				for (int d = 0; d < layers; d++) S = dmax(S, A[d * Q + sl]);
		}

		// ---- output: c2v[a] = shape(S[h_x a] - S[0]) (:899-916) ------------------------------------------------------------
		WSYNC();
		Sv[sl] = S;
		WSYNC();
		{
			const double s0 = Sv[0];
			const int t = g.mul[(size_t)g.c_h[c0 + x] * Q + sl];
			C[(size_t)x * Q + sl] = (sl == 0) ? 0.0 : shape_llr(Sv[t] - s0, r.factor, r.offset);
		}
#undef OTH
	}
}

// =====================================================================================================================
// log-QSPA (check-domain XOR convolutions in the log-sum-exp semiring, forward / backward partials: nbl_cn_bp.hip;
// vectors as probabilities relative to their maximum, mantissa x 2^exponent: nbl_cn_bp256.hip)
// =====================================================================================================================
constexpr double LOG2E = 1.4426950408889634, LN2 = 0.6931471805599453;

struct SVec {    // the lane's symbol of a vector
	double m;    // mantissa in [1,2]
	int e;       // exponent, <= 0 up to the float rounding of the reference, clamped at -1e9
	double mx;   // group-uniform: the (near-)maximum that was divided out (log domain)
	double rng;  // group-uniform: upper bound of max - min (log domain)
};

template <int Q> __device__ __forceinline__ SVec to_svec(double L)
{
	SVec r;
	// reference and range through order-preserving 32-bit keys (see to_xvec, nbl_cn_bp256.hip)
	const int k = nbl_key32(L);
	r.mx = (double)nbl_unkey32(gmax_i32<Q>(k));
	const double mn = (double)nbl_unkey32(-gmax_i32<Q>(-k));
	r.rng = (r.mx - mn) * (1.0 + 0x1p-20) + 0x1p-100;
	const double y = dmax((L - r.mx) * LOG2E, -1.0e9);
	const double f = floor(y);
	r.m = nbl_exp2_frac(y - f);
	r.e = (int)f;
	return r;
}

struct BpLds {
	double *Vm;  // [NV][Q] mantissas of the stored vectors
	int *Ve;     // [NV][Q] exponents
	double *Vx;  // [NV][2] {mx, rng}
	double *Pa, *Pb; // [Q] operands of the running convolution (plain probabilities or mantissas)
	int *Ea, *Eb;    // [Q] their exponents (wide path)
	double *T;   // [Q] staging of an output vector
};

// vector slots of a check: p_0 .. p_mdc-1, then F_2 .. F_dc-2 (at least two slots: a chunk of two inputs is staged there)
__host__ __device__ inline int bp_small_vectors(int mdc) { return mdc + (mdc - 3 > 2 ? mdc - 3 : 2); }
__host__ __device__ inline size_t bp_small_group_bytes(int q, int mdc)
{
	const int nv = bp_small_vectors(mdc);
	const size_t n = (size_t)nv * q * 12 + (size_t)nv * 16 + (size_t)q * 24 + (size_t)q * 8;
	return (n + 15) & ~(size_t)15;
}

template <int Q> __device__ __forceinline__ void put_vec(const BpLds &s, int slot, const SVec &v, int sl)
{
	s.Vm[slot * Q + sl] = v.m;
	s.Ve[slot * Q + sl] = v.e;
	if (sl == 0) { s.Vx[2 * slot] = v.mx; s.Vx[2 * slot + 1] = v.rng; }
}
template <int Q> __device__ __forceinline__ SVec get_vec(const BpLds &s, int slot, int sl)
{
	SVec v;
	v.m = s.Vm[slot * Q + sl];
	v.e = s.Ve[slot * Q + sl];
	v.mx = s.Vx[2 * slot];
	v.rng = s.Vx[2 * slot + 1];
	return v;
}

// The sums of one convolution for the lane's symbol z: sum_x A[x] B[z ^ x] = acc 2^ex exp(A.mx + B.mx)
struct ConvSum { double acc; int ex; double base_a, base_b; };
template <int Q> __device__ __forceinline__ ConvSum conv_core_small(const SVec &A, const SVec &B, const BpLds &s, int sl)
{
	constexpr int SH = 500; // see lse_conv, nbl_cn_bp256.hip
	ConvSum c;
	c.base_a = A.mx;
	c.base_b = B.mx;
	WSYNC();
	if (fmin(A.rng, B.rng) < 1000.0) { // (uniform inside the group)
		s.Pa[sl] = ldexp(A.m, A.e + SH);
		s.Pb[sl] = ldexp(B.m, B.e + SH);
		WSYNC();
		double acc = 0.0;
#pragma unroll 4
		for (int x = 0; x < Q; x++) acc = __fma_rn(s.Pa[x], s.Pb[sl ^ x], acc);
		c.acc = acc;
		c.ex = -2 * SH;
	} else {
		s.Pa[sl] = A.m;
		s.Pb[sl] = B.m;
		s.Ea[sl] = A.e;
		s.Eb[sl] = B.e;
		WSYNC();
		// exact top exponent of the output, then every term scaled exactly against it (far below the range gives 0)
		int ex = INT_MIN;
#pragma unroll 4
		for (int x = 0; x < Q; x++) ex = max(ex, s.Ea[x] + s.Eb[sl ^ x]);
		double acc = 0.0;
#pragma unroll 4
		for (int x = 0; x < Q; x++) {
			const int d = s.Ea[x] + s.Eb[sl ^ x] - ex;
			acc = __fma_rn(s.Pa[x], ldexp(s.Pb[sl ^ x], d), acc);
		}
		c.acc = acc;
		c.ex = ex;
	}
	return c;
}
// ... as a message: out[z] = LSE_x(A[x] + B[z^x]) - LSE_x(A[x] + B[x]) for the lane's symbol z (log domain, out[0] = 0)
template <int Q> __device__ __forceinline__ double conv_log_small(const ConvSum &c, int sl, int gi)
{
	const double lse = ((nbl_log_pos(c.acc) + (double)c.ex * LN2) + c.base_a) + c.base_b;
	const double norm = __shfl(lse, gi * Q, 64); // z = 0
	return (sl == 0) ? 0.0 : lse - norm;
}
// ... as the operand of the next convolution (conv_xvec, nbl_cn_bp64.hip): the sums already are probabilities, so mantissa and
// exponent come from the value itself -- no logarithm per symbol, no exponential; reference = the power of two of the group's
// largest entry, whose log-domain level relative to symbol 0 (= LLR 0, like every vector here) is E_max ln 2 - ln(sum of symbol 0)
template <int Q> __device__ __forceinline__ SVec conv_svec_small(const ConvSum &c, int gi)
{
	SVec r;
	int fe;
	r.m = 2.0 * frexp(c.acc, &fe); // acc = (m / 2) 2^fe
	const int e = fe - 1 + c.ex;
	const int emax = gmax_i32<Q>(e), nemin = gmax_i32<Q>(-e);
	r.e = e - emax;
	const double acc0 = __shfl(c.acc, gi * Q, 64);
	const int ex0 = __shfl(c.ex, gi * Q, 64);
	r.mx = (double)(emax - ex0) * LN2 - nbl_log_pos(acc0);
	r.rng = (double)(emax + nemin + 1) * LN2;
	return r;
}
template <int Q> __device__ __forceinline__ double lse_conv_small(const SVec &A, const SVec &B, const BpLds &s, int sl, int gi)
{
	return conv_log_small<Q>(conv_core_small<Q>(A, B, s, sl), sl, gi);
}

template <int Q, bool FUSED>
__global__ __launch_bounds__(64) void cn_bp_small_kernel(NblGraphDev g, NblWork w, NblRun r)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const Ctx<Q> c = ctx_init<Q>(g, w, r);
	if (!c.live) return;
	const int sl = c.sl, gi = c.gi, dc = c.dc, c0 = c.c0, mdc = g.maxdc, nv = bp_small_vectors(mdc);
	char *base = smem + (size_t)gi * bp_small_group_bytes(Q, mdc);
	BpLds s;
	s.Vm = (double *)base;
	s.Vx = s.Vm + nv * Q;
	s.Pa = s.Vx + 2 * nv;
	s.Pb = s.Pa + Q;
	s.T = s.Pb + Q;
	s.Ve = (int *)(s.T + Q);
	double *vin = s.Vm + mdc * Q; // [2][Q] the incoming values of a chunk of edges (the F slots are empty until all inputs are in)
	s.Ea = s.Ve + nv * Q;
	s.Eb = s.Ea + Q;

	double *C = w.c2v + ((size_t)c.b * g.E + c0) * Q;

	// inputs into the check domain: p_d[h_d a] = v2c_d[a], symbol 0 = LLR 0 (the A1 == 0 branch of LLR_BoxPlus, :1623-1632);
	// vector slots: p_d -> d, F_k (k = 2..dc-2) -> mdc + k - 2 (the running R stays in registers)
	for_each_input<Q, FUSED, true, 2>(g, w, r, c, vin, [&](int d, double v, const double *) {
		const int t = g.mul[(size_t)g.c_h[c0 + d] * Q + sl];
		WSYNC();
		s.T[t] = v;
		WSYNC();
		put_vec<Q>(s, d, to_svec<Q>(s.T[sl]), sl);
	});
	WSYNC();

	auto emit = [&](double o, int d) {
		WSYNC();
		s.T[sl] = o;
		WSYNC();
		const int t = g.mul[(size_t)g.c_h[c0 + d] * Q + sl];
		C[(size_t)d * Q + sl] = (sl == 0) ? 0.0 : s.T[t];
	};

	// forward partials F_1 = p_0, F_k+1 = F_k [+] p_k; the last one, F_dc-1, is output dc-1 (A2 == 0 case, :761-764)
	{
		SVec F = get_vec<Q>(s, 0, sl);
		for (int k = 1; k <= dc - 2; k++) {
			const ConvSum cs = conv_core_small<Q>(F, get_vec<Q>(s, k, sl), s, sl);
			if (k + 1 == dc - 1) emit(conv_log_small<Q>(cs, sl, gi), dc - 1);
			else {
				F = conv_svec_small<Q>(cs, gi); // (chained: mantissa and exponent straight from the sums)
				put_vec<Q>(s, mdc + k - 1, F, sl); // F_k+1
			}
		}
	}
	// backward: R_dc-2 = p_dc-1; output d = F_d [+] R_d, R_d-1 = R_d [+] p_d; R_0 is output 0 (A1 == 0 case, :757-760)
	{
		SVec R = get_vec<Q>(s, dc - 1, sl);
		for (int d = dc - 2; d >= 1; d--) {
			WSYNC();
			const SVec F = get_vec<Q>(s, (d == 1) ? 0 : mdc + d - 2, sl);
			emit(lse_conv_small<Q>(F, R, s, sl, gi), d);
			const ConvSum cs = conv_core_small<Q>(R, get_vec<Q>(s, d, sl), s, sl);
			if (d == 1) emit(conv_log_small<Q>(cs, sl, gi), 0);
			else R = conv_svec_small<Q>(cs, gi);
		}
	}
}

template <int Q> struct SmallLaunch {
	static constexpr int G = 64 / Q;
	static dim3 grid(const NblGraphDev &g, const NblRun &r) { return dim3(nbl_xcd_grid(r.B, (g.M + G - 1) / G)); }
	static hipError_t tems(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool fused, hipStream_t st)
	{
		const size_t lds = tems_small_group_bytes(Q, g.maxdc) * G;
		if (fused) cn_tems_small_kernel<Q, true><<<grid(g, r), dim3(64), lds, st>>>(g, w, r);
		else cn_tems_small_kernel<Q, false><<<grid(g, r), dim3(64), lds, st>>>(g, w, r);
		return hipGetLastError();
	}
	static hipError_t ems(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool fused, hipStream_t st)
	{
		const int layers = nbl_ems_layers(g, r.nc);
		const size_t lds = ems_small_group_bytes(Q, g.maxdc, r.nm, layers) * G;
		if (r.nm == 8 && Q >= 8) {
			if (fused) cn_ems_small_kernel<Q, true, (Q >= 8 ? 8 : 0)><<<grid(g, r), dim3(64), lds, st>>>(g, w, r, layers);
			else cn_ems_small_kernel<Q, false, (Q >= 8 ? 8 : 0)><<<grid(g, r), dim3(64), lds, st>>>(g, w, r, layers);
		} else {
			if (fused) cn_ems_small_kernel<Q, true, 0><<<grid(g, r), dim3(64), lds, st>>>(g, w, r, layers);
			else cn_ems_small_kernel<Q, false, 0><<<grid(g, r), dim3(64), lds, st>>>(g, w, r, layers);
		}
		return hipGetLastError();
	}
	static hipError_t bp(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool fused, hipStream_t st)
	{
		const size_t lds = bp_small_group_bytes(Q, g.maxdc) * G;
		if (fused) cn_bp_small_kernel<Q, true><<<grid(g, r), dim3(64), lds, st>>>(g, w, r);
		else cn_bp_small_kernel<Q, false><<<grid(g, r), dim3(64), lds, st>>>(g, w, r);
		return hipGetLastError();
	}
};

} // namespace

// (fused launches additionally need g.c_nbr: variable degrees <= 3 -- nbl_api.cpp)
// q <= 64, every check of degree >= 3 (the log-QSPA schedule below needs a middle edge), the LDS of a wave's 64 / q checks
// within the 64 KB a launch gets without opting in; T-EMS: nc <= 3 and a path code of p * maxdc <= 32 bits as in the general kernel
bool nbl_small_applicable(const NblGraphDev &g, int method, int min_dc, int nm, int nc)
{
	// (q = 64, one check per wave: measured on the BDS code, B = 4096 -- log-QSPA gains 23 % from the fused iteration, EMS loses
	//  11 % to the 64-step ranking loop; T-EMS not measured on an irregular GF(64) code, so it stays with the general kernel too)
	if (g.q > (method == 1 ? 64 : 32) || g.q < 4 || min_dc < 3 || g.maxdc > NBL_MAXDC) return false;
	const int G = 64 / g.q;
	if (method == 2) return nbl_ems_layers(g, nc) <= 4 && ems_small_group_bytes(g.q, g.maxdc, nm, nbl_ems_layers(g, nc)) * G <= 64 * 1024;
	if (method == 4) return nc <= 3 && g.p * g.maxdc <= 32 && tems_small_group_bytes(g.q, g.maxdc) * G <= 64 * 1024;
	if (method == 1) return bp_small_group_bytes(g.q, g.maxdc) * G <= 64 * 1024;
	return false;
}

#define NBL_DISPATCH_SMALL(q, CALL)                    \
	switch (q) {                                       \
	case 4: return SmallLaunch<4>::CALL(g, w, r, fused, st);  \
	case 8: return SmallLaunch<8>::CALL(g, w, r, fused, st);  \
	case 16: return SmallLaunch<16>::CALL(g, w, r, fused, st);\
	case 32: return SmallLaunch<32>::CALL(g, w, r, fused, st);\
	case 64: return SmallLaunch<64>::CALL(g, w, r, fused, st);\
	default: return hipErrorInvalidValue;              \
	}

hipError_t nbl_launch_cn_ems_small(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool fused, hipStream_t st) { NBL_DISPATCH_SMALL(g.q, ems) }
hipError_t nbl_launch_cn_tems_small(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool fused, hipStream_t st) { NBL_DISPATCH_SMALL(g.q, tems) }
hipError_t nbl_launch_cn_bp_small(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool fused, hipStream_t st) { NBL_DISPATCH_SMALL(g.q, bp) }
