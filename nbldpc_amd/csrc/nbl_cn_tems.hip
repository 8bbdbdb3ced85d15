// nbldpc_amd/csrc/nbl_cn_tems.hip -- trellis-EMS check node (NBLDPC.cpp:1055-1130, helpers :1789-1944).
//
// One wave per (codeword, check); lane l owns symbols l + 64 i.  Steps, each citing what it restates:
//   1. beta_d, syndrome, delta-domain trellis dU[d][eta]      TEMS_Get_Beta :1789-1812, TEMS_Get_deltaU :1814-1834
//   2. per deviation symbol: columns ordered by dU, the Nr smallest marked    TEMS_Get_Min :1836-1890
//   3. dW[s] = min over paths, Eta[s] = the first minimal path in enumeration order   TEMS_ConstructConf :1892-1944
//   4. extrinsic output per edge, fill of unreached symbols, back to the normal domain          :1075-1129
//
// Step 3 is the expensive one.  The reference enumerates every path (<= Nc deviating columns, deviation symbols taken from
// the marked set, a non-zero symbol used by one column only) and keeps the first strict minimum.  Here it is a min-plus
// dynamic programme over the columns in order with one layer per deviation count:
//   * dropping the "one column per symbol" rule changes nothing: a path that uses symbol q in two columns has the same
//     check sum as the path with both set to 0, whose cost is not larger (every dU >= 0, fl(x+u) is monotone) and which
//     comes earlier in enumeration order, so a duplicate path never sets dW or Eta;
//   * enumeration order is lexicographic in (q_0 .. q_dc-1), so "first strict minimum" = smallest (cost, path code);
//   * costs are left-to-right sums (((0+u0)+u1)+u2).., and min commutes with the rounded add.  (The one case a dynamic
//     programme cannot follow: two prefixes whose costs differ by an ulp and whose extensions round to the SAME sum -- the
//     reference then keeps the path that comes first in enumeration order, the programme the one with the cheaper prefix.
//     dW is identical, Eta may differ.  It takes manufactured inputs (tests/test_gpu_parity.py) to see it.)
// The residue of the reference's running add/subtract (sumNonLLR += / -=, :1917/:1921) is not reproduced.
#include <hip/hip_runtime.h>
#include "nbl_device.h"
#include "nbl_kernels.h"
#include <cstdlib>

template <int Q>
__global__ __launch_bounds__(64) void cn_tems_kernel(NblGraphDev g, NblWork w, NblRun r)
{
	constexpr int NS = Fld<Q>::NS;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int lane = lane_id();
	const int b = nbl_codeword(w, r, blockIdx.x / g.M), m = blockIdx.x % g.M;
	if (b < 0) return;
	if (!r.fixed_iters && w.done[b]) return;
	const int c0 = g.coff[m], dc = g.coff[m + 1] - c0;
	const int nr = r.nr, nc = r.nc, layers = nc + 1, mdc = g.maxdc;

	double *dU = (double *)smem;                 // [mdc][Q]
	double *Fv = dU + mdc * Q;                   // [2][layers][Q] path cost
	double *Lc = Fv + 2 * layers * Q;            // [Q]
	unsigned *Fc = (unsigned *)(Lc + Q);         // [2][layers][Q] path code, base-Q digits, column 0 most significant
	int *ord01 = (int *)(Fc + 2 * layers * Q);   // [Q] first two columns of the per-symbol order (lo byte, next byte)
	int *cmask = ord01 + Q;                      // [Q] bit d set: column d may deviate to this symbol
	int *clist = cmask + Q;                      // [mdc][Q] deviation symbols allowed per column
	int *ccount = clist + mdc * Q;               // [mdc]
	int *beta = ccount + mdc;                    // [mdc]

	const double *V = w.v2c + (size_t)b * g.E * Q;
	double *C = w.c2v + ((size_t)b * g.E + c0) * Q;

	// ---- 1. beta, syndrome, dU ---------------------------------------------------------------------------------------
	int syn = 0;
	for (int d = 0; d < dc; d++) {
		const double *Vd = V + (size_t)g.c_epos[c0 + d] * Q;
		double v[NS];
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int a = lane + 64 * i;
			v[i] = (a < Q && a > 0) ? Vd[a] : 0.0;
		}
		// most reliable symbol: strict '>' over ascending symbols from a running maximum of 0 (:1798-1807)
		double best = 0.0;
		int arg = 0;
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int a = lane + 64 * i;
			if (a < Q && v[i] > best) { best = v[i]; arg = a; }
		}
#pragma unroll
		for (int off = 32; off >= 1; off >>= 1) {
			double ob = __shfl_xor(best, off, 64);
			int oa = __shfl_xor(arg, off, 64);
			if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
		}
		GfMul<Q> mh;
		mh.init(g.c_h[c0 + d], g.poly, lane);
		// beta_d = h * argmax (0 if nothing positive)
		int bd = 0;
		{
			int x = g.c_h[c0 + d];
			for (int k = 0; k < 8; k++) {
				if ((arg >> k) & 1) bd ^= x;
				x <<= 1;
				if (x & Q) x ^= g.poly;
			}
		}
		bd = uniform(bd);
		const double mx = uniform_f64(best); // = L(argmax), or 0 when beta = 0 (:1824)
		if (lane == 0) beta[d] = bd;
		syn ^= bd;
		// dU[d][h a ^ beta] = mx - L(a), L(0) = 0  (:1826-1831)
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int a = lane + 64 * i;
			if (a < Q) dU[d * Q + (mh.at_slot(i) ^ bd)] = mx - v[i];
		}
	}
	__syncthreads();

	// ---- 2. per deviation symbol: stable ascending order of the columns, Nr smallest marked ---------------------------
	for (int d = lane; d < dc; d += 64) ccount[d] = 0;
#pragma unroll
	for (int i = 0; i < NS; i++) {
		int s = lane + 64 * i;
		if (s < Q) {
			// position of column d in the stable ascending order = number of columns that sort before it (:1851-1866);
			// no per-lane index array, so nothing is indexed dynamically
			int mask = 0, o0 = 0, o1 = 0;
			for (int d = 0; d < dc; d++) {
				const double ud = dU[d * Q + s];
				int rank = 0;
				for (int e = 0; e < dc; e++) {
					const double ue = dU[e * Q + s];
					rank += (ue < ud || (ue == ud && e < d)) ? 1 : 0;
				}
				if (rank < nr) mask |= 1 << d;
				if (rank == 0) o0 = d;
				if (rank == 1) o1 = d;
			}
			if (s == 0) mask = (1 << dc) - 1;
			cmask[s] = mask;
			ord01[s] = o0 | (o1 << 8);
		}
	}
	__syncthreads();
	// candidate lists per column (non-zero symbols only; symbol 0 = "no deviation" is handled apart)
#pragma unroll
	for (int i = 0; i < NS; i++) {
		int s = lane + 64 * i;
		if (s < Q && s > 0) {
			int mask = cmask[s];
			for (int d = 0; d < dc; d++)
				if ((mask >> d) & 1) clist[d * Q + atomicAdd(&ccount[d], 1)] = s;
		}
	}
	__syncthreads();

	// ---- 3. min-plus DP over the columns, one layer per number of deviating columns ------------------------------------
	double *Av = Fv, *Bv = Fv + layers * Q;
	unsigned *Ac = Fc, *Bc = Fc + layers * Q;
	for (int idx = lane; idx < layers * Q; idx += 64) { Av[idx] = (idx == 0) ? 0.0 : __builtin_huge_val(); Ac[idx] = 0; }
	__syncthreads();
	for (int d = 0; d < dc; d++) {
		const int n = ccount[d];
		for (int l = 0; l < layers; l++) {
			double bv[NS];
			unsigned bc[NS];
#pragma unroll
			for (int i = 0; i < NS; i++) {
				int s = lane + 64 * i;
				bv[i] = (s < Q) ? Av[l * Q + s] + 0.0 : __builtin_huge_val(); // q_d = 0: dU[d][0] = 0 (:1826)
				bc[i] = (s < Q) ? Ac[l * Q + s] * Q : 0;
			}
			if (l >= 1) {
				for (int k = 0; k < n; k++) {
					const int q = uniform(clist[d * Q + k]);
					const double u = dU[d * Q + q];
#pragma unroll
					for (int i = 0; i < NS; i++) {
						int s = lane + 64 * i;
						if (s < Q) {
							const double val = Av[(l - 1) * Q + (s ^ q)] + u;
							const unsigned code = Ac[(l - 1) * Q + (s ^ q)] * Q + q;
							if (val < bv[i] || (val == bv[i] && code < bc[i])) { bv[i] = val; bc[i] = code; }
						}
					}
				}
			}
#pragma unroll
			for (int i = 0; i < NS; i++) {
				int s = lane + 64 * i;
				if (s < Q) { Bv[l * Q + s] = bv[i]; Bc[l * Q + s] = bc[i]; }
			}
		}
		__syncthreads();
		double *tv = Av; Av = Bv; Bv = tv;
		unsigned *tc = Ac; Ac = Bc; Bc = tc;
	}
	// dW, Eta: best layer per check sum
	double dW[NS];
	unsigned eta[NS];
#pragma unroll
	for (int i = 0; i < NS; i++) {
		int s = lane + 64 * i;
		dW[i] = __builtin_huge_val();
		eta[i] = 0xffffffffu;
		if (s < Q)
			for (int l = 0; l < layers; l++) {
				const double val = Av[l * Q + s];
				const unsigned code = Ac[l * Q + s];
				if (val < dW[i] || (val == dW[i] && code < eta[i])) { dW[i] = val; eta[i] = code; }
			}
	}

	// ---- 4. outputs ------------------------------------------------------------------------------------------------------
	int pshift = (dc - 1) * Fld<Q>::P; // digit of column d sits at bit P*(dc-1-d) of the path code
	for (int d = 0; d < dc; d++) {
		__syncthreads();
		for (int s = lane; s < Q; s += 64) Lc[s] = NBL_DBL_MAX;
		__syncthreads();
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int s = lane + 64 * i;
			if (s < Q) {
				const int dev = (int)((eta[i] >> pshift) & (Q - 1));
				const double cand = dW[i] - dU[d * Q + dev];                      // :1088
				__hip_atomic_fetch_min(&Lc[s ^ dev], cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		}
		__syncthreads();
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int s = lane + 64 * i;
			if (s < Q && Lc[s] == NBL_DBL_MAX) {                                 // never reached (:1095-1102)
				const int o0 = ord01[s] & 255, o1 = (ord01[s] >> 8) & 255;
				Lc[s] = (d == o0) ? dU[o1 * Q + s] : dU[o0 * Q + s];
			}
		}
		__syncthreads();
		// delta domain -> LLR, un-permute by h (:1105-1127)
		const int bsyn = syn ^ beta[d];
		const double L0 = -1.0 * Lc[bsyn];
		GfMul<Q> mh;
		mh.init(g.c_h[c0 + d], g.poly, lane);
		double *Cd = C + (size_t)d * Q;
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int a = lane + 64 * i;
			if (a < Q) {
				const int e = mh.at_slot(i) ^ bsyn; // eta with h^-1 (eta ^ bsyn) = a
				Cd[a] = (a == 0) ? 0.0 : shape_llr(-1.0 * Lc[e] - L0, r.factor, r.offset);
			}
		}
		pshift -= Fld<Q>::P;
	}
}

// fast variant (nc <= 3): all four deviation-count layers of one check sum in one 48-byte LDS record, candidates as packed
// {dU, symbol} entries read as LDS broadcasts, candidate loop unrolled by four
struct __attribute__((aligned(16))) TState { double v[4]; unsigned c[4]; };
struct __attribute__((aligned(16))) TCand { double u; int q; int pad; };

template <int Q>
__global__ __launch_bounds__(64) void cn_tems_fast_kernel(NblGraphDev g, NblWork w, NblRun r)
{
	constexpr int NS = Fld<Q>::NS;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int lane = lane_id();
	const int b = nbl_codeword(w, r, blockIdx.x / g.M), m = blockIdx.x % g.M;
	if (b < 0) return;
	if (!r.fixed_iters && w.done[b]) return;
	const int c0 = g.coff[m], dc = g.coff[m + 1] - c0;
	const int nr = r.nr, nc = r.nc, layers = nc + 1, mdc = g.maxdc;

	double *dU = (double *)smem;                 // [mdc][Q]
	double *Lc = dU + mdc * Q;                   // [Q]
	TState *st = (TState *)(Lc + Q);             // [2][Q] DP states (ping-pong)
	TCand *cl = (TCand *)(st + 2 * Q);           // [mdc][Q+4] deviation candidates per column
	int *ord01 = (int *)(cl + mdc * (Q + 4));    // [Q] first two columns of the per-symbol order
	int *cmask = ord01 + Q;                      // [Q] bit d set: column d may deviate to this symbol
	int *ccount = cmask + Q;                     // [mdc]
	int *beta = ccount + mdc;                    // [mdc]
	(void)layers;

	const double *V = w.v2c + (size_t)b * g.E * Q;
	double *C = w.c2v + ((size_t)b * g.E + c0) * Q;

	// ---- 1. beta, syndrome, dU ---------------------------------------------------------------------------------------
	int syn = 0;
	for (int d = 0; d < dc; d++) {
		const double *Vd = V + (size_t)g.c_epos[c0 + d] * Q;
		double v[NS];
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int a = lane + 64 * i;
			v[i] = (a < Q && a > 0) ? Vd[a] : 0.0;
		}
		// most reliable symbol: strict '>' over ascending symbols from a running maximum of 0 (:1798-1807)
		double best = 0.0;
		int arg = 0;
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int a = lane + 64 * i;
			if (a < Q && v[i] > best) { best = v[i]; arg = a; }
		}
#pragma unroll
		for (int off = 32; off >= 1; off >>= 1) {
			double ob = __shfl_xor(best, off, 64);
			int oa = __shfl_xor(arg, off, 64);
			if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
		}
		GfMul<Q> mh;
		mh.init(g.c_h[c0 + d], g.poly, lane);
		// beta_d = h * argmax (0 if nothing positive)
		int bd = 0;
		{
			int x = g.c_h[c0 + d];
			for (int k = 0; k < 8; k++) {
				if ((arg >> k) & 1) bd ^= x;
				x <<= 1;
				if (x & Q) x ^= g.poly;
			}
		}
		bd = uniform(bd);
		const double mx = uniform_f64(best); // = L(argmax), or 0 when beta = 0 (:1824)
		if (lane == 0) beta[d] = bd;
		syn ^= bd;
		// dU[d][h a ^ beta] = mx - L(a), L(0) = 0  (:1826-1831)
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int a = lane + 64 * i;
			if (a < Q) dU[d * Q + (mh.at_slot(i) ^ bd)] = mx - v[i];
		}
	}
	__syncthreads();

	// ---- 2. per deviation symbol: stable ascending order of the columns, Nr smallest marked ---------------------------
	for (int d = lane; d < dc; d += 64) ccount[d] = 0;
#pragma unroll
	for (int i = 0; i < NS; i++) {
		int s = lane + 64 * i;
		if (s < Q) {
			// position of column d in the stable ascending order = number of columns that sort before it (:1851-1866);
			// no per-lane index array, so nothing is indexed dynamically
			int mask = 0, o0 = 0, o1 = 0;
			for (int d = 0; d < dc; d++) {
				const double ud = dU[d * Q + s];
				int rank = 0;
				for (int e = 0; e < dc; e++) {
					const double ue = dU[e * Q + s];
					rank += (ue < ud || (ue == ud && e < d)) ? 1 : 0;
				}
				if (rank < nr) mask |= 1 << d;
				if (rank == 0) o0 = d;
				if (rank == 1) o1 = d;
			}
			if (s == 0) mask = (1 << dc) - 1;
			cmask[s] = mask;
			ord01[s] = o0 | (o1 << 8);
		}
	}
	__syncthreads();
	// candidate lists per column (non-zero symbols only; symbol 0 = "no deviation" is handled apart), packed {dU, symbol}
#pragma unroll
	for (int i = 0; i < NS; i++) {
		int s = lane + 64 * i;
		if (s < Q && s > 0) {
			int mask = cmask[s];
			for (int d = 0; d < dc; d++)
				if ((mask >> d) & 1) {
					TCand e;
					e.u = dU[d * Q + s];
					e.q = s;
					e.pad = 0;
					cl[d * (Q + 4) + atomicAdd(&ccount[d], 1)] = e;
				}
		}
	}
	__syncthreads();
	if (lane < 4) { // pad every list to a multiple of four with entries that can never win (cost +inf)
		for (int d = 0; d < dc; d++) {
			const int n = ccount[d];
			if (n + lane < ((n + 3) & ~3)) {
				TCand e;
				e.u = __builtin_huge_val();
				e.q = 0;
				e.pad = 0;
				cl[d * (Q + 4) + n + lane] = e;
			}
		}
	}

	// ---- 3. min-plus DP over the columns; all deviation-count layers advance together -------------------------------
	// state of check sum s: cost v[l] and path code c[l] of the best path with exactly l deviating columns (l = 0..3)
	TState *A = st, *Bs = st + Q;
	for (int s = lane; s < Q; s += 64) {
		TState z;
#pragma unroll
		for (int l = 0; l < 4; l++) { z.v[l] = __builtin_huge_val(); z.c[l] = 0; }
		if (s == 0) z.v[0] = 0.0;
		A[s] = z;
	}
	__syncthreads();
	for (int d = 0; d < dc; d++) {
		const int n4 = (ccount[d] + 3) & ~3;
		const TCand *L = cl + d * (Q + 4);
		TState b[NS];
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int s = lane + 64 * i;
			if (s < Q) {
				b[i] = A[s]; // q_d = 0: dU[d][0] = 0 (:1826), cost unchanged
#pragma unroll
				for (int l = 0; l < 4; l++) { b[i].v[l] = b[i].v[l] + 0.0; b[i].c[l] = b[i].c[l] * Q; }
			}
		}
		for (int k = 0; k < n4; k += 4) {
			TCand e[4];
#pragma unroll
			for (int u = 0; u < 4; u++) e[u] = L[k + u]; // LDS broadcast
#pragma unroll
			for (int u = 0; u < 4; u++) {
#pragma unroll
				for (int i = 0; i < NS; i++) {
					int s = lane + 64 * i;
					if (s < Q) {
						const TState src = A[s ^ e[u].q];
#pragma unroll
						for (int l = 1; l < 4; l++) {
							if (l <= nc && l <= d + 1) { // a path through d+1 columns has at most d+1 deviations
								const double val = src.v[l - 1] + e[u].u;
								const unsigned code = src.c[l - 1] * Q + e[u].q;
								// smaller cost wins, equal cost: smaller path code (two predicated steps, no mask logic)
								const bool lt = val < b[i].v[l];
								b[i].v[l] = lt ? val : b[i].v[l];
								b[i].c[l] = lt ? code : b[i].c[l];
								const unsigned cm = code < b[i].c[l] ? code : b[i].c[l];
								b[i].c[l] = (val == b[i].v[l]) ? cm : b[i].c[l];
							}
						}
					}
				}
			}
		}
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int s = lane + 64 * i;
			if (s < Q) Bs[s] = b[i];
		}
		__syncthreads();
		TState *tsw = A; A = Bs; Bs = tsw;
	}
	// dW, Eta: best layer per check sum
	double dW[NS];
	unsigned eta[NS];
#pragma unroll
	for (int i = 0; i < NS; i++) {
		int s = lane + 64 * i;
		dW[i] = __builtin_huge_val();
		eta[i] = 0xffffffffu;
		if (s < Q) {
			const TState f = A[s];
#pragma unroll
			for (int l = 0; l < 4; l++) {
				if (l <= nc && (f.v[l] < dW[i] || (f.v[l] == dW[i] && f.c[l] < eta[i]))) { dW[i] = f.v[l]; eta[i] = f.c[l]; }
			}
		}
	}

	// ---- 4. outputs ------------------------------------------------------------------------------------------------------
	int pshift = (dc - 1) * Fld<Q>::P; // digit of column d sits at bit P*(dc-1-d) of the path code
	for (int d = 0; d < dc; d++) {
		__syncthreads();
		for (int s = lane; s < Q; s += 64) Lc[s] = NBL_DBL_MAX;
		__syncthreads();
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int s = lane + 64 * i;
			if (s < Q) {
				const int dev = (int)((eta[i] >> pshift) & (Q - 1));
				const double cand = dW[i] - dU[d * Q + dev];                      // :1088
				__hip_atomic_fetch_min(&Lc[s ^ dev], cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		}
		__syncthreads();
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int s = lane + 64 * i;
			if (s < Q && Lc[s] == NBL_DBL_MAX) {                                 // never reached (:1095-1102)
				const int o0 = ord01[s] & 255, o1 = (ord01[s] >> 8) & 255;
				Lc[s] = (d == o0) ? dU[o1 * Q + s] : dU[o0 * Q + s];
			}
		}
		__syncthreads();
		// delta domain -> LLR, un-permute by h (:1105-1127)
		const int bsyn = syn ^ beta[d];
		const double L0 = -1.0 * Lc[bsyn];
		GfMul<Q> mh;
		mh.init(g.c_h[c0 + d], g.poly, lane);
		double *Cd = C + (size_t)d * Q;
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int a = lane + 64 * i;
			if (a < Q) {
				const int e = mh.at_slot(i) ^ bsyn; // eta with h^-1 (eta ^ bsyn) = a
				Cd[a] = (a == 0) ? 0.0 : shape_llr(-1.0 * Lc[e] - L0, r.factor, r.offset);
			}
		}
		pshift -= Fld<Q>::P;
	}
}

static size_t tems_lds_bytes(const NblGraphDev &g, int layers)
{
	const size_t q = g.q, mdc = g.maxdc;
	return (mdc * q + 2 * layers * q + q) * 8 + (2 * layers * q + q + q + mdc * q + 2 * mdc) * 4 + 64;
}

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

hipError_t nbl_launch_cn_tems(const NblGraphDev &g, const NblWork &w, const NblRun &r, hipStream_t st)
{
	const int layers = r.nc + 1;
	const size_t lds = tems_lds_bytes(g, layers);
	if (lds > 160 * 1024) return hipErrorInvalidValue;
	if ((double)g.p * g.maxdc > 32.0) return hipErrorInvalidValue; // path code must fit 32 bits
	dim3 grid((unsigned)((long long)r.B * g.M)), block(64);
	if (r.nc <= 3 && !getenv("NBL_TEMS_GENERIC")) {
		const size_t q = g.q, mdc = g.maxdc;
		const size_t fl = (mdc * q + q) * 8 + 2 * q * 48 + mdc * (q + 4) * 16 + (2 * q + 2 * mdc) * 4 + 64;
		NBL_DISPATCH_Q(g.q, {
			if (fl > 64 * 1024) (void)hipFuncSetAttribute((const void *)cn_tems_fast_kernel<QQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fl);
			cn_tems_fast_kernel<QQ><<<grid, block, fl, st>>>(g, w, r);
		})
		return hipGetLastError();
	}
	NBL_DISPATCH_Q(g.q, {
		if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)cn_tems_kernel<QQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		cn_tems_kernel<QQ><<<grid, block, lds, st>>>(g, w, r);
	})
	return hipGetLastError();
}
