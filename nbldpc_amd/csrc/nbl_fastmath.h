// nbldpc_amd/csrc/nbl_fastmath.h -- short FP64 log / exp2 for the log-QSPA kernels (host + device).
//
// The check node needs  exp2(x), x in [0, 1)  (mantissa of a probability) and  log(a), a > 0 finite  (a sum of products) four
// times per vector.  The device library's routines handle every special case and cost ~35 / ~60 instructions; these two do
// the job in ~17 / ~30 with the same ~1 ulp accuracy on their (restricted) domains -- tests/test_ddmath.py measures both against
// glibc.  Not for general use: no NaN / infinity / zero / denormal handling beyond what the kernels can produce
// (log(+0) = -inf is kept: an all-zero sum can only come from underflow and must stay "impossible").
#pragma once
#include <math.h>
#if defined(__HIPCC__)
#define NBL_FM __host__ __device__ __forceinline__
#else
#define NBL_FM static inline
#endif

// The polynomial coefficients.  A 64-bit constant cannot be an immediate of a vector instruction, and hipcc materialises each one
// in a VGPR pair (two v_mov_b32) next to its use: 20 + 24 vector instructions per exp2 / log, a third of both routines, 750 per
// check of the GF(256) log-QSPA kernel.  On the device the coefficients therefore live in a (deliberately non-const: no folding
// back into literals) __constant__ table: uniform addresses, so they arrive through scalar loads in SGPR pairs, which the FMAs
// take as their one scalar operand.  The host build (tests/test_ddmath.py) uses the same values as literals.
#define NBL_FM_EXP_COEFFS {1.0 / 87178291200.0, 1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, \
	                       1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 0.6931471805599453094, \
	                       1.4142135623730951455, -9.667293313452913451e-17}
#define NBL_FM_LOG_COEFFS {1.0 / 23.0, 1.0 / 21.0, 1.0 / 19.0, 1.0 / 17.0, 1.0 / 15.0, 1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0, 1.0 / 7.0, 1.0 / 5.0, 1.0 / 3.0, \
	                       0.6931471805599453094, 2.319046813846299558e-17}
#if defined(__HIP_DEVICE_COMPILE__)
static __constant__ double nbl_fm_exp_c[16] = NBL_FM_EXP_COEFFS;
static __constant__ double nbl_fm_log_c[13] = NBL_FM_LOG_COEFFS;
#define NBL_FM_EC(k) nbl_fm_exp_c[k]
#define NBL_FM_LC(k) nbl_fm_log_c[k]
// a * b + c with the (wave-uniform) addend c read from its SGPR pair: left to itself hipcc picks the two-address v_fmac_f64, whose
// addend must be the destination VGPR -- and copies every coefficient into one first (two v_mov_b32 per Horner step)
__device__ __forceinline__ double nbl_fma_sc(double a, double b, double c)
{
	double r;
	asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
	return r;
}
// c * a + b with the (wave-uniform) factor c read from its SGPR pair
__device__ __forceinline__ double nbl_fma_cs(double c, double a, double b)
{
	double r;
	asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "s"(c), "v"(a), "v"(b));
	return r;
}
#define NBL_FMA_SC(a, b, c) nbl_fma_sc(a, b, c)
#define NBL_FMA_CS(c, a, b) nbl_fma_cs(c, a, b)
#else
#define NBL_FMA_SC(a, b, c) fma(a, b, c)
#define NBL_FMA_CS(c, a, b) fma(c, a, b)
static const double nbl_fm_exp_c[16] = NBL_FM_EXP_COEFFS;
static const double nbl_fm_log_c[13] = NBL_FM_LOG_COEFFS;
#define NBL_FM_EC(k) nbl_fm_exp_c[k]
#define NBL_FM_LC(k) nbl_fm_log_c[k]
#endif

// 2^x for 0 <= x < 1: 2^x = sqrt(2) * e^(t ln 2), t = x - 1/2 in [-1/2, 1/2): Taylor series of degree 14 in u = t ln 2 (|u| <= 0.347)
NBL_FM double nbl_exp2_frac(double x)
{
	const double u = (x - 0.5) * NBL_FM_EC(13);
	double p = NBL_FM_EC(0);                   // 1/14!
#pragma unroll
	for (int k = 1; k <= 12; k++) p = NBL_FMA_SC(p, u, NBL_FM_EC(k)); // 1/13! .. 1/2!
	// e^u = 1 + (u + u^2 p): the small part first, then sqrt(2) = hi + lo times (1 + small) with one rounding of the large term
	const double em1 = fma(u * u, p, u);
	const double sq2 = NBL_FM_EC(14);
	const double r = fma(sq2, em1, sq2);
	return NBL_FMA_CS(NBL_FM_EC(15), 1.0 + em1, r);
}

// log(a) for finite a > 0 (log(0) = -inf): a = m 2^e with m in [3/4, 3/2); log m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 1/5
NBL_FM double nbl_log_pos(double a)
{
	if (a == 0.0) return -HUGE_VAL;
	int e;
	double m = frexp(a, &e);                   // [1/2, 1)
	if (m < 0.75) { m *= 2.0; e -= 1; }        // [3/4, 3/2)
	const double f = m - 1.0, d = m + 1.0;
	// s = f / d by reciprocal + one Newton correction of the quotient (d in [1.75, 2.5]).  The reciprocal only has to be good to
	// ~2^-24: the correction squares its error and s_lo below picks up what is left, so the device takes the hardware estimate
	// (v_rcp_f64, one instruction; an IEEE division is ~11) and the host build -- which exists to MEASURE this routine against
	// libm, tests/test_ddmath.py -- rounds its reciprocal to float, the worst a conforming estimate may be
#if defined(__HIP_DEVICE_COMPILE__)
	const double r = __builtin_amdgcn_rcp(d);
#else
	const double r = (double)(float)(1.0 / d);
#endif
	double s = f * r;
	s = fma(fma(-s, d, f), r, s);
	const double s_lo = fma(-s, d, f) * r;     // what is still missing of the quotient
	const double z = s * s;
	double p = NBL_FM_LC(0);                   // 1/23
#pragma unroll
	for (int k = 1; k <= 10; k++) p = NBL_FMA_SC(p, z, NBL_FM_LC(k)); // 1/21 .. 1/3
	// log m = 2 s + (2 s z p + 2 s_lo);  result = e ln2 (hi + lo) + log m
	const double t = 2.0 * s;
	const double lm = t + fma(t * z, p, 2.0 * s_lo);
	const double ed = (double)e;
	return NBL_FMA_CS(NBL_FM_LC(11), ed, NBL_FMA_CS(NBL_FM_LC(12), ed, lm));
}
