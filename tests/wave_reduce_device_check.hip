// tests/wave_reduce_device_check.hip -- nbl_device.h's wave_fmax_nonneg (exact maximum of max(x, 0) over a wave through two
// unsigned 32-bit DPP reductions) against a host evaluation and against the FP64 reduction wave_fmax, bit for bit, on the GPU
// (tests/test_gpu_noise.py builds and runs it); likewise wave_fmin_nonneg on the absolute values.  Inputs: random doubles of mixed sign and magnitude, rows with equal high words
// (the low-word reduction decides), all-negative rows (result +0), zeros of both signs, denormals, infinities, one-lane maxima.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "nbl_device.h"

__global__ void eval(const double *x, double *a, double *b, double *c)
{
	const int i = blockIdx.x * 64 + threadIdx.x;
	const double v = x[i];
	const double m1 = wave_fmax_nonneg(v), m2 = wave_fmax(dmax(v, 0.0));
	const double n1 = wave_fmin_nonneg(fabs(v)); // (non-negative inputs, +inf allowed)
	if (threadIdx.x == 0) { a[blockIdx.x] = m1; b[blockIdx.x] = m2; c[blockIdx.x] = n1; }
}
static uint64_t bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
int main()
{
	const int rows = 200000, n = rows * 64;
	std::vector<double> x(n), a(rows), b(rows), c(rows);
	srand(11);
	auto rnd = [] { return (double)rand() / ((double)RAND_MAX + 1.0); };
	for (int r = 0; r < rows; r++) {
		const int kind = r % 8;
		for (int l = 0; l < 64; l++) {
			double v = (rnd() - 0.5) * ldexp(1.0, (rand() % 80) - 40);
			if (kind == 1) v = 1.0 + rnd() * 0x1p-21;                       // equal high words, the low words decide
			if (kind == 2) v = -fabs(v);                                    // nothing positive: +0
			if (kind == 3) v = (l & 1) ? -0.0 : 0.0;
			if (kind == 4) v = ldexp(rnd(), -1060);                         // denormals
			if (kind == 5) v = (l == rand() % 64) ? fabs(v) + 1e3 : -1.0;   // one lane holds the maximum
			if (kind == 6 && l == 7) v = HUGE_VAL;
			if (kind == 7) v = floor(v);                                    // integers: many exact ties
			x[r * 64 + l] = v;
		}
	}
	double *dx, *da, *db, *dc;
	if (hipMalloc(&dx, (size_t)n * 8) || hipMalloc(&da, rows * 8) || hipMalloc(&db, rows * 8) || hipMalloc(&dc, rows * 8)) return 2;
	if (hipMemcpy(dx, x.data(), (size_t)n * 8, hipMemcpyHostToDevice)) return 2;
	eval<<<rows, 64>>>(dx, da, db, dc);
	if (hipMemcpy(a.data(), da, rows * 8, hipMemcpyDeviceToHost) || hipMemcpy(b.data(), db, rows * 8, hipMemcpyDeviceToHost) ||
	    hipMemcpy(c.data(), dc, rows * 8, hipMemcpyDeviceToHost))
		return 2;
	long bad_host = 0, bad_f64 = 0, bad_min = 0;
	for (int r = 0; r < rows; r++) {
		double m = 0.0, mn = HUGE_VAL;
		for (int l = 0; l < 64; l++) {
			m = x[r * 64 + l] > m ? x[r * 64 + l] : m;
			mn = fabs(x[r * 64 + l]) < mn ? fabs(x[r * 64 + l]) : mn;
		}
		bad_host += bits(a[r]) != bits(m);
		bad_f64 += bits(a[r]) != bits(b[r]);
		bad_min += bits(c[r]) != bits(mn);
	}
	printf("{\"rows\": %d, \"differ_from_host\": %ld, \"differ_from_fp64_reduction\": %ld, \"min_differs_from_host\": %ld}\n", rows, bad_host, bad_f64, bad_min);
	return 0;
}
