// tools/ubench/fp64_power.hip -- the same FP64 multiply-add rate from the vector unit (v_fma_f64) and from the matrix unit
// (v_mfma_f64_16x16x4_f64), every SIMD of the chip busy, for a few seconds each: run it under tools/ubench/fp64_power.sh, which
// samples shader clock and socket power meanwhile.  Question: at the board's power limit, does one of the two units deliver
// more FP64 multiply-adds per second (i.e. sustain a higher clock)?
//   fp64_power valu|mfma [seconds]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_valu(double *out, int trips)
{
	double a[8], x = 1.0 + 1e-9 * threadIdx.x, y = 1e-12;
	for (int i = 0; i < 8; i++) a[i] = i;
	for (int t = 0; t < trips; t++) {
#pragma unroll
		for (int r = 0; r < 16; r++)
#pragma unroll
			for (int i = 0; i < 8; i++) a[i] = __fma_rn(a[i], x, y);
	}
	double s = 0;
	for (int i = 0; i < 8; i++) s += a[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_mfma(double *out, int trips)
{
	double4_t c[4];
	for (int i = 0; i < 4; i++) c[i] = double4_t{0, 0, 0, 0};
	const double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-3;
	for (int t = 0; t < trips; t++) {
#pragma unroll
		for (int r = 0; r < 2; r++)
#pragma unroll
			for (int i = 0; i < 4; i++) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
	}
	double s = 0;
	for (int i = 0; i < 4; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main(int argc, char **argv)
{
	const bool mfma = argc > 1 && !strcmp(argv[1], "mfma");
	const double secs = argc > 2 ? atof(argv[2]) : 6.0;
	double *out;
	const int blocks = 256 * 4; // four waves per SIMD
	if (hipMalloc(&out, (size_t)blocks * 256 * 8)) return 2;
	// per trip and wave: valu 128 v_fma_f64 = 8192 lane-FMAs; mfma 8 instructions x 1024 = 8192 FMAs
	const int trips = 20000;
	double fmas = 0;
	const auto t0 = std::chrono::steady_clock::now();
	int launches = 0;
	while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
		if (mfma) k_mfma<<<blocks, 256>>>(out, trips);
		else k_valu<<<blocks, 256>>>(out, trips);
		if (hipDeviceSynchronize()) return 3;
		launches++;
		fmas += (double)blocks * 4 * trips * 8192.0;
	}
	const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	printf("%s: %d launches, %.2f s, %.1f TFLOP/s FP64 (2 flops per multiply-add)\n", mfma ? "v_mfma_f64_16x16x4_f64" : "v_fma_f64", launches, dt, 2 * fmas / dt / 1e12);
	return 0;
}
