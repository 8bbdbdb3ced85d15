// tests/fastmath_device_check.hip -- the DEVICE build of nbldpc_amd/csrc/nbl_fastmath.h (hardware reciprocal estimate in the short
// logarithm) against this machine's libm, on the GPU: worst difference in ulps (tests/test_gpu_noise.py builds and runs it).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "nbl_fastmath.h"

__global__ void eval(const double *x, const double *a, double *e, double *l, int n)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { e[i] = nbl_exp2_frac(x[i]); l[i] = nbl_log_pos(a[i]); }
}
static double ulp_err(double got, double ref) { if (ref == got) return 0; int e; frexp(ref, &e); return fabs(got - ref) / ldexp(1.0, e - 53); }
int main()
{
	const int n = 4000000;
	std::vector<double> x(n), a(n), e(n), l(n);
	srand(7);
	for (int i = 0; i < n; i++) {
		x[i] = (double)rand() / ((double)RAND_MAX + 1.0);
		a[i] = ldexp(0.5 + x[i], (rand() % 2000) - 1000);
		if (i % 3 == 0) a[i] = 1.0 + (x[i] - 0.5) * ldexp(1.0, -(rand() % 40));
	}
	double *dx, *da, *de, *dl;
	if (hipMalloc(&dx, n * 8) || hipMalloc(&da, n * 8) || hipMalloc(&de, n * 8) || hipMalloc(&dl, n * 8)) return 2;
	hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
	hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
	eval<<<(n + 255) / 256, 256>>>(dx, da, de, dl, n);
	if (hipMemcpy(e.data(), de, n * 8, hipMemcpyDeviceToHost) || hipMemcpy(l.data(), dl, n * 8, hipMemcpyDeviceToHost)) return 2;
	double we = 0, wl = 0;
	for (int i = 0; i < n; i++) {
		const double e1 = ulp_err(e[i], exp2(x[i])), e2 = ulp_err(l[i], log(a[i]));
		we = e1 > we ? e1 : we;
		wl = e2 > wl ? e2 : wl;
	}
	printf("{\"exp2_worst_ulp\": %.3f, \"log_worst_ulp\": %.3f}\n", we, wl);
	return 0;
}
