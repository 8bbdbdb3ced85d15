// tools/ubench/lds_atomic_rates.hip -- what does an LDS ds_max_f64 cost on gfx950, and what would replace it?
// One wave per workgroup, four workgroups per SIMD (the headline kernel's residency); every wave issues REP x 48 operations on
// a 6 KB region (three 2 KB vectors), addresses per lane: linear (conflict-free), random per round (the pair convolution's
// pattern: XOR of two random symbols), or all equal.  Times: wall clock of the launch (all waves), per operation per wave.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/lds_atomic_rates.hip -o tools/ubench/lds_atomic_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
// OP 0: ds_max_f64 (returnless)   1: ds_max_u64   2: ds_max_f32 (32-bit)   3: ds_write_b64 (plain store, wrong result: rate only)
// OP 4: read-max-write through registers (ds_read_b64, v_max_f64, ds_write_b64)   5: ds_read_b64 only
template <int OP, int PATTERN> __global__ __launch_bounds__(64) void k(const int *addr, double *out)
{
	__shared__ __attribute__((aligned(16))) double buf[768];
	const int lane = threadIdx.x;
	for (int i = lane; i < 768; i += 64) buf[i] = -1e300;
	int a[16];
#pragma unroll
	for (int r = 0; r < 16; r++) a[r] = PATTERN == 0 ? lane * 4 % 256 : PATTERN == 1 ? addr[(blockIdx.x * 16 + r) * 64 + lane] & 255 : 7;
	double v = lane * 0.25, acc = 0;
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	for (int it = 0; it < REP; it++) {
#pragma unroll
		for (int r = 0; r < 16; r++) {
#pragma unroll
			for (int b = 0; b < 3; b++) {
				double *p = &buf[b * 256 + (a[r] ^ (b * 37 & 255))];
				if (OP == 0) __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				else if (OP == 1) __hip_atomic_fetch_max((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				else if (OP == 2) __hip_atomic_fetch_max((float *)p, (float)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				else if (OP == 3) *(volatile double *)p = v;
				else if (OP == 4) { const double o = *(volatile double *)p; *(volatile double *)p = o > v ? o : v; }
				else acc += *(volatile double *)p;
			}
			v += 1.0;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	}
	out[blockIdx.x * 64 + lane] = buf[lane] + buf[256 + lane] + buf[512 + lane] + acc;
}

template <int OP, int PATTERN> static void run(const char *name, const int *daddr, double *dout)
{
	const int blocks = 256 * 16; // 16 waves per CU = 4 per SIMD
	k<OP, PATTERN><<<blocks, 64>>>(daddr, dout);
	hipDeviceSynchronize();
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e9;
	for (int t = 0; t < 5; t++) { // (repeat: short kernels see the clock ramp)
		hipEventRecord(e0);
		k<OP, PATTERN><<<blocks, 64>>>(daddr, dout);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		best = ms < best ? ms : best;
	}
	// per CU: 16 waves x REP x 48 ops share one LDS
	const double ns_per_op_cu = best * 1e6 / (16.0 * REP * 48);
	printf("  %-44s %8.3f ms  -> %6.1f ns per wave-operation of the CU's LDS (%.0f cycles at 2.4 GHz)\n", name, best, ns_per_op_cu, ns_per_op_cu * 2.4);
}

int main()
{
	std::vector<int> h(256 * 16 * 16 * 64);
	unsigned x = 2463534242u;
	for (auto &v : h) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; v = (int)(x & 255); }
	int *daddr;
	double *dout;
	hipMalloc(&daddr, h.size() * 4);
	hipMalloc(&dout, 256 * 16 * 64 * 8);
	hipMemcpy(daddr, h.data(), h.size() * 4, hipMemcpyHostToDevice);
#define ALL(P, pn) \
	run<0, P>("ds_max_f64, " pn, daddr, dout); run<1, P>("ds_max_u64, " pn, daddr, dout); run<2, P>("ds_max_f32, " pn, daddr, dout); \
	run<3, P>("ds_write_b64, " pn, daddr, dout); run<4, P>("read + v_max_f64 + write, " pn, daddr, dout); run<5, P>("ds_read_b64, " pn, daddr, dout);
	ALL(0, "linear addresses")
	ALL(1, "random addresses")
	ALL(2, "one address")
	return 0;
}
