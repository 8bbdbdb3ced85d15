// tools/bp256_budget.hip -- the building blocks of the GF(256) log-QSPA check node as separate (noinline) device functions, so that
// tools/bp256_budget.py can count the vector instructions of each from the ISA (the kernel inlines them all).
#include <hip/hip_runtime.h>
#include "../nbldpc_amd/csrc/nbl_device.h"
#include "../nbldpc_amd/csrc/nbl_fastmath.h"

extern "C" {
__device__ __noinline__ double piece_log(double a) { return nbl_log_pos(a); }
__device__ __noinline__ double piece_exp2(double x) { return nbl_exp2_frac(x); }
__device__ __noinline__ int piece_decide(double a, double b, double c, double d)
{
	const double v[4] = {a, b, c, d};
	return wave_decide_keyed<4>(v, lane_id(), 256);
}
__device__ __noinline__ int piece_decide_fp64(double a, double b, double c, double d)
{
	const double v[4] = {a, b, c, d};
	return wave_decide<4>(v, lane_id(), 256);
}
// to_xvec of nbl_cn_bp256.hip without the four exponentials (counted by piece_exp2): reference, range, split into floor + fraction
__device__ __noinline__ double piece_xvec_rest(double a, double b, double c, double d, int *e)
{
	const double L[4] = {a, b, c, d};
	const int kx = nbl_key32(dmax(dmax(L[0], L[1]), dmax(L[2], L[3]))), kn = nbl_key32(dmin(dmin(L[0], L[1]), dmin(L[2], L[3])));
	const double mx = (double)nbl_unkey32(wave_imax_id(kx));
	const double mn = (double)nbl_unkey32(-wave_imax_id(-kn));
	double acc = (mx - mn) * (1.0 + 0x1p-20) + 0x1p-100;
#pragma unroll
	for (int i = 0; i < 4; i++) {
		const double y = dmax((L[i] - mx) * 1.4426950408889634, -1.0e9);
		const double f = floor(y);
		acc += y - f;
		e[i] = (int)f;
	}
	return acc;
}
}
__global__ void keep(double *o, int *e)
{
	const double x = o[threadIdx.x];
	o[threadIdx.x] = piece_log(x) + piece_exp2(x) + piece_decide(x, x + 1, x + 2, x + 3) + piece_decide_fp64(x, x + 1, x + 2, x + 3) + piece_xvec_rest(x, x + 1, x + 2, x + 3, e);
}
