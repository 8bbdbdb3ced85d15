// tools/ubench/mfma_f64_adder.hip -- can the FP64 matrix pipe take the ADDS of the max-plus gather off the vector ALU?
//
// D = A x B + C with A = [1 0 0 0] (rows) and B[0][j] = v gives D[i][j] = fl(C[i][j] + v): one v_mfma_f64_16x16x4_f64 adds one
// wave-uniform value to four doubles per lane (a 16x16 tile), (the 4x4x4 four-block form does NOT add with this operand pattern: kept as a negative check).  This
// measures (1) that the sums are the IEEE sums bit for bit (random, infinite, subnormal and cancelling operands), (2) the
// issue rate of the two instructions, alone and interleaved with v_max_f64 / v_add_f64 at 1, 2 and 4 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_f64_adder.hip -o tools/ubench/mfma_f64_adder ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

// exactness: out[l][r] = C[l][r] (+) v through the 16x16x4 form
__global__ void exact16(const double *C, const double *v, double *out, int n)
{
	const int lane = threadIdx.x;
	const double a = lane < 16 ? 1.0 : 0.0; // A[i][k]: lane = i + 16 k
	for (int t = 0; t < n; t++) {
		double4_t c;
		for (int r = 0; r < 4; r++) c[r] = C[((size_t)t * 64 + lane) * 4 + r];
		const double b = v[t]; // B[k][j]: lane = j + 16 k; rows k > 0 meet A = 0 (any finite value)
		double4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
		for (int r = 0; r < 4; r++) out[((size_t)t * 64 + lane) * 4 + r] = d[r];
	}
}
__global__ void exact4(const double *C, const double *v, double *out, int n)
{
	const int lane = threadIdx.x;
	const double a = (lane & 3) == 0 ? 1.0 : 0.0; // 4x4x4, 4 blocks: A[b][i][k]: lane = 16 b + 4 i + k?  (checked by the result)
	for (int t = 0; t < n; t++) {
		const double c = C[(size_t)t * 64 + lane], b = v[t];
		const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
		out[(size_t)t * 64 + lane] = d;
	}
}

#define REP 2048
__device__ __forceinline__ double vmax(double a, double b)
{
	double r;
	asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
__device__ __forceinline__ double vadd(double a, double b)
{
	double r;
	asm("v_add_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
// One trip = 4 groups; a group is what the gather does per (list entry, convolution): four gathered doubles per lane (here: a
// register set standing in for the ds_read results, made opaque every trip) + the entry's value, then the running maximum.
// MODE 0: the sums by 4 v_add_f64 (today)        MODE 1: the sums by one v_mfma_f64_16x16x4_f64
// MODE 6: every second group on the matrix pipe   MODE 7: every fourth group on the matrix pipe
// MODE 2: 4 mfma alone                            MODE 3: 16 v_max_f64 alone
template <int MODE> __global__ void rate(double *out, long long *cyc, const double *in)
{
	const int lane = threadIdx.x & 63;
	const double a = lane < 16 ? 1.0 : 0.0;
	double b = in[lane & 7];
	double4_t src[4], m[4];
	for (int i = 0; i < 4; i++)
		for (int r = 0; r < 4; r++) { src[i][r] = in[8 + i * 4 + r] + lane; m[i][r] = -1e300; }
	const long long t0 = clock64();
	for (int it = 0; it < REP; it++) {
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const bool on_matrix = MODE == 1 || MODE == 2 || (MODE == 6 && (i & 1)) || (MODE == 7 && i == 3);
#pragma unroll
			for (int r = 0; r < 4; r++) asm volatile("" : "+v"(src[i][r])); // (a fresh gather result as far as the compiler knows)
			// (in place: the gathered values are dead after the sum, as in the kernel)
			if (MODE == 3) {
			} else if (on_matrix) {
				src[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, src[i], 0, 0, 0);
			} else {
#pragma unroll
				for (int r = 0; r < 4; r++) src[i][r] = vadd(src[i][r], b);
			}
			if (MODE != 2) {
#pragma unroll
				for (int r = 0; r < 4; r++) m[i][r] = vmax(m[i][r], src[i][r]);
			}
		}
		asm volatile("" : "+v"(b)); // (the next entry's value)
	}
	const long long t1 = clock64();
	double s = 0;
	for (int i = 0; i < 4; i++)
		for (int r = 0; r < 4; r++) s += m[i][r] + src[i][r];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

template <int MODE> static int run_rate(const char *what, double *dout, long long *dcyc, const double *din, int waves_per_simd)
{
	// 256-thread workgroups (one wave per SIMD each), waves_per_simd of them per CU, 256 CUs
	long long cyc = 0;
	rate<MODE><<<256 * waves_per_simd, 256>>>(dout, dcyc, din);
	CK(hipDeviceSynchronize());
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	CK(hipEventRecord(e0));
	rate<MODE><<<256 * waves_per_simd, 256>>>(dout, dcyc, din);
	CK(hipEventRecord(e1));
	CK(hipDeviceSynchronize());
	float ms = 0;
	CK(hipEventElapsedTime(&ms, e0, e1));
	CK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
	printf("  %-46s %d waves/SIMD: %7.1f cycles per trip of 4 groups (one wave's clock64), %.3f ms for %d trips\n", what, waves_per_simd, (double)cyc / REP, ms, REP);
	return 0;
}

int main()
{
	// ---- exactness ----
	const int n = 4096;
	std::vector<double> C((size_t)n * 256), v(n), out16(C.size()), out4((size_t)n * 64);
	uint64_t x = 88172645463325252ull;
	auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
	auto rd = [&]() -> double {
		const unsigned k = rnd() % 16;
		double d;
		uint64_t bits = rnd();
		if (k == 0) return -(double)INFINITY;
		if (k == 1) { bits &= 0x800fffffffffffffull; memcpy(&d, &bits, 8); return d; }     // subnormal
		if (k == 2) return (double)((long long)(rnd() % 2001) - 1000);                       // small integers (cancellation)
		bits = (bits & 0x800fffffffffffffull) | ((uint64_t)(1023 - 40 + rnd() % 80) << 52);  // 2^-40 .. 2^40
		memcpy(&d, &bits, 8);
		return d;
	};
	for (auto &c : C) c = rd();
	for (auto &b : v) { do b = rd(); while (std::isinf(b)); }
	double *dC, *dv, *dout;
	long long *dcyc;
	CK(hipMalloc(&dC, C.size() * 8)); CK(hipMalloc(&dv, n * 8)); CK(hipMalloc(&dout, 8 * 256 * 1024)); CK(hipMalloc(&dcyc, 8));
	CK(hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice));
	CK(hipMemcpy(dv, v.data(), n * 8, hipMemcpyHostToDevice));
	double *do16;
	CK(hipMalloc(&do16, C.size() * 8));
	exact16<<<1, 64>>>(dC, dv, do16, n);
	CK(hipDeviceSynchronize());
	CK(hipMemcpy(out16.data(), do16, C.size() * 8, hipMemcpyDeviceToHost));
	long bad = 0, zsign = 0;
	for (int t = 0; t < n; t++)
		for (int i = 0; i < 256; i++) {
			const double want = C[(size_t)t * 256 + i] + v[t], got = out16[(size_t)t * 256 + i];
			if (memcmp(&want, &got, 8) != 0) {
				if (want == 0.0 && got == 0.0) zsign++;
				else if (bad++ < 5) printf("  16x16x4: C=%a v=%a want %a got %a\n", C[(size_t)t * 256 + i], v[t], want, got);
			}
		}
	printf("v_mfma_f64_16x16x4_f64 as adder: %ld of %d sums differ from the IEEE sum (%ld differ only in the sign of zero)\n", bad, n * 256, zsign);
	exact4<<<1, 64>>>(dC, dv, do16, n);
	CK(hipDeviceSynchronize());
	CK(hipMemcpy(out4.data(), do16, out4.size() * 8, hipMemcpyDeviceToHost));
	long bad4 = 0, z4 = 0;
	for (int t = 0; t < n; t++)
		for (int i = 0; i < 64; i++) {
			const double want = C[(size_t)t * 64 + i] + v[t], got = out4[(size_t)t * 64 + i];
			if (memcmp(&want, &got, 8) != 0) {
				if (want == 0.0 && got == 0.0) z4++;
				else if (bad4++ < 5) printf("  4x4x4: lane %d C=%a v=%a want %a got %a\n", i, C[(size_t)t * 64 + i], v[t], want, got);
			}
		}
	printf("v_mfma_f64_4x4x4_4b_f64 as adder: %ld of %d sums differ (%ld only in the sign of zero)\n", bad4, n * 64, z4);
	// ---- rates ----
	std::vector<double> in(64);
	for (int i = 0; i < 64; i++) in[i] = 1.0 + i * 0.37;
	double *din;
	CK(hipMalloc(&din, 64 * 8));
	CK(hipMemcpy(din, in.data(), 64 * 8, hipMemcpyHostToDevice));
	for (int w = 1; w <= 4; w++) {
		if (w == 3) continue;
		run_rate<0>("4 groups, sums by v_add_f64 [today]", dout, dcyc, din, w);
		run_rate<7>("1 of 4 groups' sums on the matrix pipe", dout, dcyc, din, w);
		run_rate<6>("2 of 4 groups' sums on the matrix pipe", dout, dcyc, din, w);
		run_rate<1>("4 of 4 groups' sums on the matrix pipe", dout, dcyc, din, w);
		run_rate<2>("4 x v_mfma_f64_16x16x4_f64 alone", dout, dcyc, din, w);
		run_rate<3>("16 x v_max_f64 alone", dout, dcyc, din, w);
	}
	return (bad) ? 1 : 0;
}
