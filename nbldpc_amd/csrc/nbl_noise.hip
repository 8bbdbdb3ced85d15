// nbldpc_amd/csrc/nbl_noise.hip -- the reference's AWGN channel on the device.
//
//   CComm::Channel_AWGN   Comm.cpp:328-337   RX = TX + Rand_Norm(0, sigma) on the real AND the imaginary part of every symbol
//   CRand::Rand_Norm      Rand.cpp:31-37     mu + sigma * cos(2 acos(-1) u2) * sqrt(-2 log(1 - u1))
//   CRand::Rand_Uniform   Rand.cpp:17-28     three multiplicative congruential generators, sum of the quotients modulo 1
//
// The generators are integer recurrences: the state in front of uniform draw k is X0 * A^k mod m, so every (lane, symbol)
// starts from the lane's state times a tabulated power.  Divisions, products and sqrt are IEEE operations; log and cos are
// evaluated in double-double (nbl_ddmath.h) and come with a verdict: `certain` results equal glibc's by glibc's own error
// bound, the rest (about one value in six) is listed for the host, which evaluates exactly those with its libm
// (nbl_decode_batch_noise in nbl_api.cpp) before the samples are formed.  Three kernels:
//   noise_gen_kernel     one thread per normal draw: u1, u2, log(1 - u1), cos(2 pi u2) -> fn[draw] = {lg, cs}; uncertain ones listed
//   noise_patch_kernel   host-evaluated values written over the listed entries
//   noise_finish_kernel  RX = TX + (0 + sigma * cs * sqrt(-2 lg)) in the reference's expression order
#include <hip/hip_runtime.h>
#include "nbl_ddmath.h"
#include "nbl_device.h"
#include "nbl_kernels.h"

__global__ __launch_bounds__(256) void noise_gen_kernel(const uint32_t *__restrict__ state, const uint32_t *__restrict__ jump, int L,
                                                        long long total, double2 *__restrict__ fn, uint32_t *__restrict__ flag_idx,
                                                        double *__restrict__ flag_arg, unsigned *__restrict__ flag_count, unsigned cap)
{
	const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (tid >= total) return;
	const int comp = (int)(tid & 1);
	const long long bs = tid >> 1;
	const int s = (int)(bs % L);
	const long long b = bs / L;
	// state in front of this symbol's first draw: X0 * A^(4 s) mod m
	uint32_t ix = (uint32_t)(((uint64_t)(state[b * 3 + 0] % 61967u) * jump[s]) % 61967u);
	uint32_t iy = (uint32_t)(((uint64_t)(state[b * 3 + 1] % 63443u) * jump[L + s]) % 63443u);
	uint32_t iz = (uint32_t)(((uint64_t)(state[b * 3 + 2] % 63599u) * jump[2 * L + s]) % 63599u);
	if (comp) { // the imaginary part comes after the two draws of the real part: A^2 mod m
		ix = (uint32_t)(((uint64_t)ix * ((249u * 249u) % 61967u)) % 61967u);
		iy = (uint32_t)(((uint64_t)iy * ((251u * 251u) % 63443u)) % 63443u);
		iz = (uint32_t)(((uint64_t)iz * ((252u * 252u) % 63599u)) % 63599u);
	}
	const double u1 = nbl_rand_uniform(ix, iy, iz);
	const double u2 = nbl_rand_uniform(ix, iy, iz);
	const double v = 1.0 - u1;
	const double x = 2 * 0x1.921fb54442d18p+1 * u2; // 2 * acos(-1.0) * u2, Rand.cpp:35
	const nbl_dd lg = dd_log(v);
	const nbl_dd cs = dd_cos(x);
	double2 r;
	r.x = lg.hi;
	r.y = cs.hi;
	fn[tid] = r;
	// uncertain values go to the list: one atomic per wave (ballots + prefix counts), not one per value
	const bool ul = !dd_certain(lg, NBL_BAND_LOG), uc = !dd_certain(cs, NBL_BAND_COS);
	const uint64_t ml = __ballot(ul), mc = __ballot(uc);
	const int nl = __builtin_popcountll(ml), ncs = __builtin_popcountll(mc);
	if (nl + ncs) {
		unsigned base = 0;
		if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(ml | mc)) base = atomicAdd(flag_count, (unsigned)(nl + ncs));
		base = (unsigned)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(ml | mc));
		if (ul) {
			const unsigned k = base + (unsigned)prefix_count(ml);
			if (k < cap) { flag_idx[k] = (uint32_t)(tid * 2); flag_arg[k] = v; }
		}
		if (uc) {
			const unsigned k = base + (unsigned)nl + (unsigned)prefix_count(mc);
			if (k < cap) { flag_idx[k] = (uint32_t)(tid * 2 + 1); flag_arg[k] = -x; } // sign bit = "cosine" (x >= 0; -0.0 for x = 0)
		}
	}
}

__global__ void noise_patch_kernel(double *__restrict__ fn, const uint32_t *__restrict__ flag_idx, const double *__restrict__ val, unsigned n)
{
	const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k < n) fn[flag_idx[k]] = val[k];
}

__global__ __launch_bounds__(256) void noise_finish_kernel(const double2 *__restrict__ fn, const uint8_t *__restrict__ tx_index,
                                                           const double *__restrict__ cons, double sigma, double mu, long long nsym,
                                                           double2 *__restrict__ rx)
{
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nsym) return;
	const int t = tx_index[i];
	const double2 fr = fn[2 * i], fi = fn[2 * i + 1];
	// Rand_Norm: u = mu + sigma * cos(..) * sqrt(-2.0 * log(..)), left to right (Rand.cpp:35); mu = 0 arrives as an argument so
	// that the addition stays an addition
	const double nr = mu + sigma * fr.y * sqrt(-2.0 * fr.x);
	const double ni = mu + sigma * fi.y * sqrt(-2.0 * fi.x);
	double2 o;
	o.x = cons[2 * t] + nr;
	o.y = cons[2 * t + 1] + ni;
	rx[i] = o;
}

hipError_t nbl_launch_noise_gen(const uint32_t *state, const uint32_t *jump, int L, int B, double *fn, uint32_t *flag_idx, double *flag_arg,
                                unsigned *flag_count, unsigned cap, hipStream_t st)
{
	const long long total = (long long)B * L * 2;
	noise_gen_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(state, jump, L, total, (double2 *)fn, flag_idx, flag_arg, flag_count, cap);
	return hipGetLastError();
}

hipError_t nbl_launch_noise_patch(double *fn, const uint32_t *flag_idx, const double *val, unsigned n, hipStream_t st)
{
	if (n == 0) return hipSuccess;
	noise_patch_kernel<<<dim3((n + 255) / 256), dim3(256), 0, st>>>(fn, flag_idx, val, n);
	return hipGetLastError();
}

hipError_t nbl_launch_noise_finish(const double *fn, const uint8_t *tx_index, const double *cons, double sigma, int L, int B, double *rx, hipStream_t st)
{
	const long long nsym = (long long)B * L;
	noise_finish_kernel<<<dim3((unsigned)((nsym + 255) / 256)), dim3(256), 0, st>>>((const double2 *)fn, tx_index, cons, sigma, 0.0, nsym, (double2 *)rx);
	return hipGetLastError();
}
