// nbldpc_amd/csrc/nbl_cn_bp.hip -- exact log-domain QSPA check node (NBLDPC.cpp:747-767, L_Back :1565, L_Forward :1593,
// LLR_BoxPlus :1621-1712).
//
// Everything is moved into the CHECK domain first: p_d[y] = v2c_d[h_d^-1 y] (the A1==0 branch of LLR_BoxPlus, :1623-1632).
// With symbol 0 stored explicitly as LLR 0, the full branch of LLR_BoxPlus(L, L1, L2, 1, h) becomes a plain XOR
// convolution in the log-sum-exp semiring,
//      conv(L1, L2)[b] = LSE_x ( L1[x] + L2p[b ^ x] )  -  LSE_x ( L1[x] + L2p[x] ),            L1[0] = L2p[0] = 0,
// because h^-1 (b ^ x) = h^-1 b ^ h^-1 x (GF multiplication is GF(2)-linear); the reference's special cases v1==0 / v2==0
// (:1661, :1681-1688, :1694) are exactly the x = 0 and x = b terms.  The forward partials F_k = p_0 [+] .. [+] p_k-1 and the
// backward partials R_k = p_dc-1 [+] .. [+] p_k+1 are computed ONCE per check (the reference re-derives them for every
// output edge, :751-754, with identical values), output d is conv(F_d, R_d) read at h_d a (:756-765).
//
// Numerics: the reference accumulates the log-sum-exp sequentially, pairwise, in 80-bit long double under g++
// (64-bit under its original MSVC).  A GPU has neither 80-bit registers nor glibc's expl/logl, so bit parity of the LLRs
// is impossible by construction (SURVEY 8c hazard 3); here each LSE is evaluated in FP64 as described at lse_conv below.  Parity for this method is defined on hard decisions, convergence flags and FER, with LLRs within 1e-9
// of the oracle's FP64 restatement (tests/test_gpu_parity.py).
#include <hip/hip_runtime.h>
#include "nbl_device.h"
#include "nbl_kernels.h"

// out[b] = LSE_x(L1[x] + L2[b^x]) - LSE_x(L1[x] + L2[x]) for the lane's symbols b; L1, L2, out: Q doubles in LDS, [0] = 0.
//
// Evaluation: each input is exponentiated ONCE (2Q exponentials per convolution instead of Q^2), the XOR convolution itself is
// a sum of products (all terms positive: no cancellation), one logarithm per output.
//   * narrow inputs (min(range(L1), range(L2)) < 650): plain doubles, e^(L-max), one FMA per term;
//   * wide inputs (LLRs thousands apart, e.g. high-order QAM at high SNR): every probability is kept as mantissa x 2^E with a
//     separate 32-bit exponent, the running sum carries its own exponent (v_ldexp_f64 rescales exactly), so no term is ever
//     lost to underflow -- the reference's log-domain formulation has unlimited range and so has this one.
struct __attribute__((aligned(16))) XP { double m; int e; int pad; }; // value = m * 2^e, m in [1,2)

template <int Q>
__device__ __forceinline__ void lse_conv(const double *L1, const double *L2, double *out, XP *PA, XP *PB, int lane)
{
	constexpr int NS = Fld<Q>::NS;
	double m1 = NBL_NEG_INF, m2 = NBL_NEG_INF, n1 = __builtin_huge_val(), n2 = __builtin_huge_val();
	for (int s = lane; s < Q; s += 64) {
		m1 = dmax(m1, L1[s]); m2 = dmax(m2, L2[s]);
		n1 = dmin(n1, L1[s]); n2 = dmin(n2, L2[s]);
	}
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) {
		m1 = dmax(m1, __shfl_xor(m1, off, 64)); m2 = dmax(m2, __shfl_xor(m2, off, 64));
		n1 = dmin(n1, __shfl_xor(n1, off, 64)); n2 = dmin(n2, __shfl_xor(n2, off, 64));
	}
	// every output has a term >= e^-min(range): with the smaller range below 650 nats whatever underflows is < 2^-76 of the result
	const bool narrow = fmin(m1 - n1, m2 - n2) < 650.0; // wave-uniform
	double lse[NS];
	if (narrow) {
		double *A = (double *)PA, *B = (double *)PB;
		for (int s = lane; s < Q; s += 64) { A[s] = exp(L1[s] - m1); B[s] = exp(L2[s] - m2); }
		__syncthreads();
		double acc[NS];
#pragma unroll
		for (int i = 0; i < NS; i++) acc[i] = 0.0;
#pragma unroll 4
		for (int x = 0; x < Q; x++) {
			const double a = A[x];
#pragma unroll
			for (int i = 0; i < NS; i++) {
				int bsym = lane + 64 * i;
				if (bsym < Q) acc[i] = __fma_rn(a, B[bsym ^ x], acc[i]);
			}
		}
#pragma unroll
		for (int i = 0; i < NS; i++) lse[i] = (log(acc[i]) + m1) + m2;
	} else {
		const double LOG2E = 1.4426950408889634, LN2 = 0.6931471805599453;
		for (int s = lane; s < Q; s += 64) {
			const double y1 = (L1[s] - m1) * LOG2E, y2 = (L2[s] - m2) * LOG2E;
			const double f1 = floor(y1), f2 = floor(y2);
			XP a, b;
			a.m = exp2(y1 - f1); a.e = (int)dmax(f1, -1.0e9); a.pad = 0;
			b.m = exp2(y2 - f2); b.e = (int)dmax(f2, -1.0e9); b.pad = 0;
			PA[s] = a;
			PB[s] = b;
		}
		__syncthreads();
		double acc[NS];
		int ex[NS];
#pragma unroll
		for (int i = 0; i < NS; i++) { acc[i] = 0.0; ex[i] = -2000000000; }
#pragma unroll 2
		for (int x = 0; x < Q; x++) {
			const XP a = PA[x];
#pragma unroll
			for (int i = 0; i < NS; i++) {
				int bsym = lane + 64 * i;
				if (bsym < Q) {
					const XP b = PB[bsym ^ x];
					const int e = a.e + b.e;
					const int top = e > ex[i] ? e : ex[i];
					// both rescalings are exact (power of two); differences beyond the double range give 0
					const int d0 = ex[i] - top, d1 = e - top;
					acc[i] = ldexp(acc[i], d0 < -2000 ? -2000 : d0) + ldexp(a.m * b.m, d1 < -2000 ? -2000 : d1);
					ex[i] = top;
				}
			}
		}
#pragma unroll
		for (int i = 0; i < NS; i++) lse[i] = ((log(acc[i]) + (double)ex[i] * LN2) + m1) + m2;
	}
	const double norm = read_lane_f64(lse[0], 0); // b = 0
	__syncthreads();
#pragma unroll
	for (int i = 0; i < NS; i++) {
		int bsym = lane + 64 * i;
		if (bsym < Q) out[bsym] = (bsym == 0) ? 0.0 : lse[i] - norm;
	}
	__syncthreads();
}

template <int Q>
__global__ __launch_bounds__(64) void cn_bp_kernel(NblGraphDev g, NblWork w, NblRun r)
{
	constexpr int NS = Fld<Q>::NS;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int lane = lane_id();
	const int b = nbl_codeword(w, r, blockIdx.x / g.M), m = blockIdx.x % g.M;
	if (b < 0) return;
	if (!r.fixed_iters && w.done[b]) return;
	const int c0 = g.coff[m], dc = g.coff[m + 1] - c0, mdc = g.maxdc;

	double *Pp = (double *)smem;   // [mdc][Q] inputs in the check domain
	double *F = Pp + mdc * Q;      // [mdc][Q] F[k] = edges 0..k-1   (k = 1..dc-1)
	double *R = F + mdc * Q;       // [mdc][Q] R[k] = edges dc-1..k+1 (k = dc-2..0)
	double *T = R + mdc * Q;       // [Q] scratch
	XP *PA = (XP *)(T + Q);        // [Q] e^(L1 - max), plain double or mantissa/exponent
	XP *PB = PA + Q;               // [Q] e^(L2 - max)

	const double *V = w.v2c + (size_t)b * g.E * Q;
	double *C = w.c2v + ((size_t)b * g.E + c0) * Q;

	for (int d = 0; d < dc; d++) {
		const double *Vd = V + (size_t)g.c_epos[c0 + d] * Q;
		GfMul<Q> mh;
		mh.init(g.c_h[c0 + d], g.poly, lane);
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int a = lane + 64 * i;
			if (a < Q) Pp[d * Q + mh.at_slot(i)] = (a == 0) ? 0.0 : Vd[a];
		}
	}
	__syncthreads();
	// forward partials: F[1] = p_0, F[k+1] = conv(F[k], p_k)
	for (int s = lane; s < Q; s += 64) { F[1 * Q + s] = Pp[s]; R[(dc - 2) * Q + s] = Pp[(dc - 1) * Q + s]; }
	__syncthreads();
	for (int k = 1; k + 1 <= dc - 1; k++) lse_conv<Q>(F + k * Q, Pp + k * Q, F + (k + 1) * Q, PA, PB, lane);
	// backward partials: R[dc-2] = p_dc-1, R[k-1] = conv(R[k], p_k)
	for (int k = dc - 2; k >= 1; k--) lse_conv<Q>(R + k * Q, Pp + k * Q, R + (k - 1) * Q, PA, PB, lane);
	// outputs
	for (int d = 0; d < dc; d++) {
		const double *src;
		if (d == 0) src = R;                            // A1 = 0 (:757-760)
		else if (d == dc - 1) src = F + (dc - 1) * Q;   // A2 = 0 (:761-764)
		else { lse_conv<Q>(F + d * Q, R + d * Q, T, PA, PB, lane); src = T; }
		GfMul<Q> mh;
		mh.init(g.c_h[c0 + d], g.poly, lane);
		double *Cd = C + (size_t)d * Q;
#pragma unroll
		for (int i = 0; i < NS; i++) {
			int a = lane + 64 * i;
			if (a < Q) Cd[a] = (a == 0) ? 0.0 : src[mh.at_slot(i)];
		}
		__syncthreads();
	}
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

hipError_t nbl_launch_cn_bp(const NblGraphDev &g, const NblWork &w, const NblRun &r, hipStream_t st)
{
	const size_t lds = ((size_t)3 * g.maxdc + 5) * g.q * 8;
	dim3 grid((unsigned)((long long)r.B * g.M)), block(64);
	NBL_DISPATCH_Q(g.q, {
		if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)cn_bp_kernel<QQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		cn_bp_kernel<QQ><<<grid, block, lds, st>>>(g, w, r);
	})
	return hipGetLastError();
}
