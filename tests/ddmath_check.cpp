// tests/ddmath_check.cpp -- host build of nbldpc_amd/csrc/nbl_ddmath.h against this machine's libm (tests/test_ddmath.py).
// Draws (u1, u2) pairs from the reference's generator and compares the double-double log / cos with glibc's:
//   * every value the verdict calls `certain` must equal glibc's result bit for bit,
//   * the uncertain fraction (what the device hands to the host) is reported.
// Also checks CRand::Skip (nbldpc_amd/host/rand.h) against drawing one by one.
#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#include <cmath>
#include "nbl_ddmath.h"
#include "rand.h"
int main(int argc, char **argv)
{
	const long n = argc > 1 ? atol(argv[1]) : 1000000;
	const uint32_t seed = argc > 2 ? (uint32_t)atoi(argv[2]) : 173;
	uint32_t ix = seed, iy = seed, iz = seed;
	CRand ref;
	ref.IX = ref.IY = ref.IZ = seed;
	long flag_log = 0, flag_cos = 0, bad = 0, mism_log = 0, mism_cos = 0, bad_uniform = 0;
	for (long i = 0; i < n; i++) {
		const double u1 = nbl_rand_uniform(ix, iy, iz), u2 = nbl_rand_uniform(ix, iy, iz);
		if (u1 != ref.Rand_Uniform() || u2 != ref.Rand_Uniform()) bad_uniform++;
		const double v = 1.0 - u1, x = 2 * acos(-1.0) * u2;
		const nbl_dd l = dd_log(v), c = dd_cos(x);
		const double gl = log(v), gc = cos(x);
		const bool cl = dd_certain(l, NBL_BAND_LOG), cc = dd_certain(c, NBL_BAND_COS);
		flag_log += !cl; flag_cos += !cc;
		mism_log += gl != l.hi; mism_cos += gc != c.hi;
		if ((cl && gl != l.hi) || (cc && gc != c.hi)) bad++;
	}
	// skip-ahead
	long bad_skip = 0;
	for (unsigned long k : {0ul, 1ul, 2ul, 2048ul, 4099ul, 1000003ul}) {
		CRand a, b;
		a.IX = b.IX = 173 + 61967ul * 3; a.IY = b.IY = 999; a.IZ = b.IZ = 63598;
		for (unsigned long i = 0; i < k; i++) a.Rand_Uniform();
		b.Skip(k);
		if (a.Rand_Uniform() != b.Rand_Uniform() || a.IX != b.IX || a.IY != b.IY || a.IZ != b.IZ) bad_skip++;
	}
	printf("{\"n\": %ld, \"flag_log\": %.6f, \"flag_cos\": %.6f, \"glibc_not_rounded_log\": %ld, \"glibc_not_rounded_cos\": %ld, "
	       "\"certain_but_different\": %ld, \"uniform_mismatch\": %ld, \"skip_mismatch\": %ld}\n",
	       n, (double)flag_log / n, (double)flag_cos / n, mism_log, mism_cos, bad, bad_uniform, bad_skip);
	return 0;
}
