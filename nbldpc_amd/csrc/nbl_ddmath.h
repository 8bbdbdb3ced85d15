// nbldpc_amd/csrc/nbl_ddmath.h -- log and cos evaluated in double-double arithmetic, with a rounding verdict.
//
// Why: the reference's noise samples are  sigma * cos(2 pi u2) * sqrt(-2 log(1 - u1))  with glibc's cos / log
// (Rand.cpp:31-37).  FER parity needs those samples bit for bit, and a GPU has no glibc.  glibc documents its results to
// be within 0.5 + b ulp of the true value (log: 0.519 ulp, e_log.c; cos: 0.55 ulp, s_sin.c since 2.28).  So whenever the
// TRUE value lies farther than b ulp from the midpoint between two neighbouring doubles, every result within that bound --
// glibc's included -- is the correctly rounded one.  These routines compute the value to ~2^-100 relative, return the
// correctly rounded double, and tell the caller whether it is `certain` (away from a midpoint by more than the band); the
// uncertain few per cent are evaluated by the host's own libm (nbl_api.cpp).  sqrt and the divisions of CRand are IEEE
// operations and need no such care.
//
// Host and device compile the same code (the host build is how tests/test_ddmath.py compares it with glibc without a GPU).
#pragma once
#include <math.h>
#include <stdint.h>
#include "nbl_dd_constants.h"

#if defined(__HIPCC__)
#define NBL_HD __host__ __device__ __forceinline__
#else
#define NBL_HD static inline
#endif

struct nbl_dd { double hi, lo; };

NBL_HD nbl_dd dd_two_sum(double a, double b)
{
	const double s = a + b, bb = s - a;
	return nbl_dd{s, (a - (s - bb)) + (b - bb)};
}
NBL_HD nbl_dd dd_quick_two_sum(double a, double b) // |a| >= |b|
{
	const double s = a + b;
	return nbl_dd{s, b - (s - a)};
}
NBL_HD nbl_dd dd_two_prod(double a, double b)
{
	const double p = a * b;
	return nbl_dd{p, fma(a, b, -p)};
}
NBL_HD nbl_dd dd_add(nbl_dd a, nbl_dd b)
{
	nbl_dd s = dd_two_sum(a.hi, b.hi);
	const nbl_dd t = dd_two_sum(a.lo, b.lo);
	s.lo += t.hi;
	s = dd_quick_two_sum(s.hi, s.lo);
	s.lo += t.lo;
	return dd_quick_two_sum(s.hi, s.lo);
}
NBL_HD nbl_dd dd_add_d(nbl_dd a, double b)
{
	nbl_dd s = dd_two_sum(a.hi, b);
	s.lo += a.lo;
	return dd_quick_two_sum(s.hi, s.lo);
}
NBL_HD nbl_dd dd_mul(nbl_dd a, nbl_dd b)
{
	nbl_dd p = dd_two_prod(a.hi, b.hi);
	p.lo += a.hi * b.lo + a.lo * b.hi;
	return dd_quick_two_sum(p.hi, p.lo);
}
NBL_HD nbl_dd dd_mul_d(nbl_dd a, double b)
{
	nbl_dd p = dd_two_prod(a.hi, b);
	p.lo += a.lo * b;
	return dd_quick_two_sum(p.hi, p.lo);
}
NBL_HD nbl_dd dd_div(nbl_dd a, nbl_dd b)
{
	const double q1 = a.hi / b.hi;
	nbl_dd r = dd_add(a, dd_mul_d(b, -q1));
	const double q2 = r.hi / b.hi;
	r = dd_add(r, dd_mul_d(b, -q2));
	const double q3 = r.hi / b.hi;
	nbl_dd q = dd_quick_two_sum(q1, q2);
	return dd_add_d(q, q3);
}

// Rounding verdict for a value known as hi + lo (|lo| <= ulp(hi)/2, error far below the band): the nearest double is hi; it is
// `certain` -- every result within (0.5 + band) ulp of the true value equals hi -- unless |lo| comes within band * ulp of ulp/2.
// Powers of two (the ulp changes at hi) are reported uncertain.
NBL_HD bool dd_certain(nbl_dd v, double band)
{
	int e;
	const double m = frexp(v.hi, &e); // |m| in [0.5, 1)
	if (m == 0.5 || m == -0.5 || v.hi == 0.0) return false;
	const double ulp = ldexp(1.0, e - 53);
	return fabs(v.lo) < (0.5 - band) * ulp;
}

// log(v) for 0 < v <= 1 (normal doubles): v = m 2^e, m in [sqrt(1/2), sqrt(2)); log m = 2 atanh(s), s = (m-1)/(m+1)
NBL_HD nbl_dd dd_log(double v)
{
	int e;
	double m = frexp(v, &e); // [0.5, 1)
	if (m < 0.70710678118654752) { m *= 2.0; e -= 1; }
	const nbl_dd num = nbl_dd{m - 1.0, 0.0};               // exact (m within [1/2, 2])
	const nbl_dd den = dd_two_sum(m, 1.0);
	const nbl_dd s = dd_div(num, den);
	const nbl_dd s2 = dd_mul(s, s);
	nbl_dd acc = nbl_dd{nbl_dd_inv_odd[NBL_DD_NLOG - 1][0], nbl_dd_inv_odd[NBL_DD_NLOG - 1][1]};
#pragma unroll
	for (int k = NBL_DD_NLOG - 2; k >= 0; k--)
		acc = dd_add(dd_mul(acc, s2), nbl_dd{nbl_dd_inv_odd[k][0], nbl_dd_inv_odd[k][1]});
	nbl_dd r = dd_mul(acc, s);
	r.hi *= 2.0; r.lo *= 2.0;
	if (e != 0) r = dd_add(dd_mul_d(nbl_dd{NBL_DD_LN2_HI, NBL_DD_LN2_LO}, (double)e), r);
	return r;
}

// cos(x) for 0 <= x < 8: x = k pi/2 + r with pi/2 to 160 bits, then the Taylor series of sin / cos on |r| <= pi/4
NBL_HD nbl_dd dd_cos(double x)
{
	const double kf = rint(x * NBL_DD_2OPI);
	const int k = (int)kf;
	nbl_dd r;
	if (k == 0) r = nbl_dd{x, 0.0};
	else {
		const nbl_dd t1 = dd_two_prod(kf, NBL_DD_PIO2_1);
		const double r1 = x - t1.hi;                          // exact: x and k pi/2 agree to within a factor of two
		r = dd_two_sum(r1, -t1.lo);
		const nbl_dd t2 = dd_two_prod(kf, NBL_DD_PIO2_2);
		r = dd_add(r, nbl_dd{-t2.hi, -t2.lo});
		r = dd_add_d(r, -(kf * NBL_DD_PIO2_3));
	}
	const nbl_dd r2 = dd_mul(r, r);
	const bool use_sin = k & 1;
	nbl_dd acc;
	if (use_sin) {
		acc = nbl_dd{nbl_dd_sin_c[NBL_DD_NTRIG - 1][0], nbl_dd_sin_c[NBL_DD_NTRIG - 1][1]};
#pragma unroll
		for (int i = NBL_DD_NTRIG - 2; i >= 0; i--) acc = dd_add(dd_mul(acc, r2), nbl_dd{nbl_dd_sin_c[i][0], nbl_dd_sin_c[i][1]});
		acc = dd_mul(acc, r);
	} else {
		acc = nbl_dd{nbl_dd_cos_c[NBL_DD_NTRIG - 1][0], nbl_dd_cos_c[NBL_DD_NTRIG - 1][1]};
#pragma unroll
		for (int i = NBL_DD_NTRIG - 2; i >= 0; i--) acc = dd_add(dd_mul(acc, r2), nbl_dd{nbl_dd_cos_c[i][0], nbl_dd_cos_c[i][1]});
	}
	// cos(k pi/2 + r): k = 0: cos r, 1: -sin r, 2: -cos r, 3: sin r, 4: cos r
	const bool neg = ((k & 3) == 1) || ((k & 3) == 2);
	if (neg) { acc.hi = -acc.hi; acc.lo = -acc.lo; }
	return acc;
}

// One uniform draw of CRand (Rand.cpp:17-28): the three congruential generators advance, the sum of the three quotients modulo 1
NBL_HD double nbl_rand_uniform(uint32_t &ix, uint32_t &iy, uint32_t &iz)
{
	ix = (uint32_t)(((uint64_t)ix * 249u) % 61967u);
	iy = (uint32_t)(((uint64_t)iy * 251u) % 63443u);
	iz = (uint32_t)(((uint64_t)iz * 252u) % 63599u);
	double t = (((double)ix) / ((double)61967)) + (((double)iy) / ((double)63443)) + (((double)iz) / ((double)63599));
	t -= (int)t;
	return t;
}

// bands: glibc's documented worst-case errors beyond half an ulp, with a margin (e_log.c: 0.519 ulp; s_sin.c: 0.55 ulp)
#define NBL_BAND_LOG 0.025
#define NBL_BAND_COS 0.06
