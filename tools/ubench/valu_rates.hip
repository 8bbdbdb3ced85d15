// tools/ubench/valu_rates.hip -- issue rates of the FP64 / integer VALU instructions the decoder kernels lean on (gfx950).
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rates.hip -o tools/ubench/valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 4096
template <int OP> __global__ void k(double *out, int n)
{
	double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	int e = n;
	unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
	for (int i = 0; i < REP; i++) {
#define EIGHT(X) X(a0, u0) X(a1, u1) X(a2, u2) X(a3, u3) X(a4, u4) X(a5, u5) X(a6, u6) X(a7, u7)
		if (OP == 0) {
#define X(a, u) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a) : "v"(a0));
			EIGHT(X)
#undef X
		} else if (OP == 1) {
#define X(a, u) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a) : "v"(e));
			EIGHT(X)
#undef X
		} else if (OP == 2) {
#define X(a, u) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(a0));
			EIGHT(X)
#undef X
		} else if (OP == 3) {
#define X(a, u) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a) : "v"(a0));
			EIGHT(X)
#undef X
		} else if (OP == 4) {
#define X(a, u) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(u) : "v"(u0));
			EIGHT(X)
#undef X
		} else if (OP == 5) {
#define X(a, u) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u) : "v"(u0));
			EIGHT(X)
#undef X
		} else if (OP == 6) {
#define X(a, u) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a), "v"(a0) : "vcc");
			EIGHT(X)
#undef X
		} else if (OP == 7) {
#define X(a, u) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(a0));
			EIGHT(X)
#undef X
		} else if (OP == 8) {
#define X(a, u) asm volatile("v_med3_i32 %0, %0, %1, %1" : "+v"(u) : "v"(u0));
			EIGHT(X)
#undef X
		} else if (OP == 10) {
#define X(a, u) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(u) : "v"(u0));
			EIGHT(X)
#undef X
		} else if (OP == 11) {
#define X(a, u) asm volatile("v_mov_b32 %0, %1" : "=v"(u) : "v"(u0));
			EIGHT(X)
#undef X
		} else if (OP == 12) {
#define X(a, u) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u) : "v"(u0));
			EIGHT(X)
#undef X
		} else if (OP == 13) {
#define X(a, u) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(u) : "v"(a), "v"(a0), "v"(u0) : "vcc");
			EIGHT(X)
#undef X
		} else if (OP == 14) {
#define X(a, u) asm volatile("v_cmp_lt_f64 s[10:11], %1, %2\n\tv_cndmask_b32_e64 %0, %0, %3, s[10:11]" : "+v"(u) : "v"(a), "v"(a0), "v"(u0) : "s10", "s11");
			EIGHT(X)
#undef X
		} else if (OP == 9) {
#define X(a, u) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(u) : "v"(u0));
			EIGHT(X)
#undef X
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7;
}

template <int OP> static void run(const char *name, double *d)
{
	const int blocks = 256 * 4 * 8, threads = 64; // 8 waves per SIMD
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	k<OP><<<blocks, threads>>>(d, 1);
	hipDeviceSynchronize();
	hipEventRecord(a);
	k<OP><<<blocks, threads>>>(d, 1);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms = 0;
	hipEventElapsedTime(&ms, a, b);
	int clk = 0;
	hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
	const double wave_instr = (double)blocks * REP * 8;           // wave-level instructions issued
	const double per_simd = wave_instr / (256.0 * 4.0);
	printf("%-16s %8.3f ms  -> %.2f cycles per wave instruction per SIMD (clock %.0f MHz)\n", name, ms, ms * 1e-3 * clk * 1e3 / per_simd, clk / 1e3);
}

int main()
{
	double *d;
	hipMalloc(&d, 256 * 4 * 8 * 64 * 8);
	run<0>("v_fma_f64", d);
	run<7>("v_mul_f64", d);
	run<2>("v_add_f64", d);
	run<3>("v_max_f64", d);
	run<1>("v_ldexp_f64", d);
	run<6>("v_cmp_lt_f64", d);
	run<4>("v_add3_u32", d);
	run<8>("v_med3_i32", d);
	run<9>("v_lshl_add_u32", d);
	run<5>("v_cndmask_b32 vcc", d);
	run<10>("v_cndmask e64 sgpr", d);
	run<11>("v_mov_b32", d);
	run<12>("v_min_u32", d);
	run<13>("cmp+cndmask vcc (2 instr)", d);
	run<14>("cmp+cndmask sgpr (2 instr)", d);
	return 0;
}
