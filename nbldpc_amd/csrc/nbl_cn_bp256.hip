// nbldpc_amd/csrc/nbl_cn_bp256.hip -- exact log-domain QSPA check node for GF(256), check degree 4 (BASELINE config 5):
// NBLDPC.cpp:747-767, L_Back :1565, L_Forward :1593, LLR_BoxPlus :1621-1712.  Same mathematics as the generic kernel in
// nbl_cn_bp.hip (check-domain XOR convolution in the log-sum-exp semiring, forward/backward partials computed once, six
// convolutions per check), re-laid for the machine:
//
//   * one check per wave; lane l owns the four consecutive check-domain symbols 4l..4l+3 of every vector, so a 4x4 block of
//     the XOR convolution  out[z] += A[x] * B[z ^ x]  (x = 4g..4g+3) needs four broadcast values of A and the four values of
//     B at chunk l ^ g: 16 FMAs per two 16-byte LDS gathers instead of one LDS read per FMA;
//   * every vector is held as probabilities relative to its own maximum, mantissa in [1,2] x 2^e with a separate 32-bit
//     exponent, IN REGISTERS (12 VGPRs per vector); LDS holds only the two operands of the running convolution (6 KB per wave,
//     the generic kernel needs 35 KB and runs one wave per SIMD);
//   * narrow convolutions (the smaller of the two input ranges below 1000 nats; both operands pre-scaled by 2^500, see lse_conv)
//     run on plain doubles, one FMA per term;
//   * wide convolutions (LLRs thousands of nats apart, the normal state after a few iterations of a converged frame) take
//     the per-output top exponent first (integer max-plus pass), then one v_add3 + v_ldexp_f64 + v_fma_f64 per term: every
//     term is scaled exactly, nothing is lost to underflow -- the reference's log-domain recursion has unlimited range and
//     so has this.
//
// Numerics: as for the generic kernel, LLRs are not bit-identical to the reference's 80-bit long-double recursion (SURVEY 8c
// hazard 3); parity is on hard decisions, flags and FER, LLRs within 1e-9 of the oracle's FP64 restatement.
#include <hip/hip_runtime.h>
#include "nbl_device.h"
#include "nbl_fastmath.h"
#include "nbl_kernels.h"

// section marks for a static instruction budget of the ISA (-DNBL_EMS_MARKS: a comment line per mark, nothing else changes)
#ifdef NBL_EMS_MARKS
#define BPMARK(i) asm volatile("; NBLMARK " #i)
#else
#define BPMARK(i) do { } while (0)
#endif

namespace {

constexpr int Q = 256;
constexpr int TOPK = 2;
constexpr double LOG2E = 1.4426950408889634, LN2 = 0.6931471805599453;

struct XVec {      // probabilities of the lane's four symbols relative to the vector's maximum
	double m[4];   // mantissa in [1,2]
	int e[4];      // exponent, <= 0, clamped at -1e9
	double mx;     // wave-uniform: the maximum that was divided out (log domain)
	double rng;    // wave-uniform: max - min (log domain)
};

// the TOPK largest entries of a vector and their exponents, wave-uniform (seeds of the per-output exponent estimate of the wide
// path; computed only there -- the narrow path, which is what runs at every realistic operating point, has no use for them);
// ties in any order
struct TopK { int tsym[TOPK], te[TOPK]; };
__device__ __forceinline__ TopK top_entries(const XVec &r, int lane)
{
	TopK t;
	int taken = 0;
#pragma unroll
	for (int k = 0; k < TOPK; k++) {
		int key = -1;
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const int ki = ((max(r.e[i], -(1 << 22)) + (1 << 22)) << 8) | (4 * lane + i);
			key = ((taken >> i) & 1) ? key : max(key, ki);
		}
		const int best = wave_imax(key);
		t.tsym[k] = best & 255;
		t.te[k] = (best >> 8) - (1 << 22);
		if (key == best) taken |= 1 << (best & 3);
	}
	return t;
}

__device__ __forceinline__ XVec to_xvec(const double (&L)[4], int lane)
{
	XVec r;
	// The reference that is divided out only has to be CLOSE to the maximum (it is added back after the convolution; an entry a
	// float ulp above it just has mantissa 1 + 1e-7), and the range only has to be an upper bound: both reductions run on
	// order-preserving 32-bit keys of the values rounded to float (one DPP instruction per step instead of three).
	const int kx = nbl_key32(dmax(dmax(L[0], L[1]), dmax(L[2], L[3]))), kn = nbl_key32(dmin(dmin(L[0], L[1]), dmin(L[2], L[3])));
	r.mx = (double)nbl_unkey32(wave_imax_id(kx));
	const double mn = (double)nbl_unkey32(-wave_imax_id(-kn));
	r.rng = (r.mx - mn) * (1.0 + 0x1p-20) + 0x1p-100; // (covers the float rounding of both ends)
#pragma unroll
	for (int i = 0; i < 4; i++) {
		const double y = dmax((L[i] - r.mx) * LOG2E, -1.0e9);
		const double f = floor(y);
		r.m[i] = nbl_exp2_frac(y - f);
		r.e[i] = (int)f;
	}
	return r;
}

struct Lds {
	double2 *Am01, *Am23, *Bm01, *Bm23; // [64] mantissas (or plain probabilities) of symbols (4c, 4c+1) / (4c+2, 4c+3)
	int4 *Ae, *Be;                      // [64] exponents of symbols 4c..4c+3
};

// The XOR convolution of two vectors for the lane's symbols z = 4 lane + i: sum_x A[x] B[z^x] = acc[i] 2^ex[i] e^(A.mx + B.mx)
struct ConvAcc { double acc[4]; int ex[4]; double base_a, base_b; };
__device__ __forceinline__ ConvAcc conv_core(const XVec &A, const XVec &B, const Lds &s, int lane, unsigned long long *stamps)
{
	ConvAcc c;
	c.base_a = A.mx;
	c.base_b = B.mx;
	// Both operands are scaled by 2^500 for the plain-double path: an entry then survives down to e^-1054 of its vector's maximum
	// and a product down to e^-1400 of the largest product.  Every output has a term >= e^-min(range) (the maximum of one vector
	// times any entry of the other), and a term within e^-40 of it has both factors within e^-(min(range)+40) of their maxima:
	// with min(range) < 1000 nats nothing that matters underflows, and 256 products of at most 2^1000 cannot overflow.
	constexpr int SH = 500;
	const bool narrow = fmin(A.rng, B.rng) < 1000.0; // wave-uniform
#ifdef NBL_EMS_STAMPS
	// diagnostic build only (tools/bp_split.py): how many convolutions take the plain-double path / the mantissa-exponent path
	if (stamps && lane == 0) atomicAdd(&stamps[narrow ? 0 : 1], 1ull);
#endif
	if (narrow) {
		s.Am01[lane] = make_double2(ldexp(A.m[0], A.e[0] + SH), ldexp(A.m[1], A.e[1] + SH));
		s.Am23[lane] = make_double2(ldexp(A.m[2], A.e[2] + SH), ldexp(A.m[3], A.e[3] + SH));
		s.Bm01[lane] = make_double2(ldexp(B.m[0], B.e[0] + SH), ldexp(B.m[1], B.e[1] + SH));
		s.Bm23[lane] = make_double2(ldexp(B.m[2], B.e[2] + SH), ldexp(B.m[3], B.e[3] + SH));
		__syncthreads();
		double acc[4] = {0.0, 0.0, 0.0, 0.0};
		// addresses by hand: the broadcast operand advances by immediate offsets inside a trip of four chunks, the gathered
		// operand's chunk is one XOR of the lane's byte offset with a scalar (hipcc spends 2.5 vector instructions per chunk on
		// (lane ^ g) << 4 and on the broadcast address otherwise: 640 of the kernel's 9.7 k per check)
		const char *const Ab = (const char *)s.Am01, *const Bb = (const char *)s.Bm01;
		int lane16 = lane << 4;
		asm("" : "+v"(lane16)); // (opaque: hipcc would re-associate the XOR below into (lane ^ g) << 4, two instructions per chunk)
#pragma unroll 4
		for (int g = 0; g < 64; g++) {
			const int bo = lane16 ^ (g << 4);
			const double2 a01 = *(const double2 *)(Ab + (g << 4)), a23 = *(const double2 *)(Ab + 1024 + (g << 4));
			const double2 b01 = *(const double2 *)(Bb + bo), b23 = *(const double2 *)(Bb + 1024 + bo);
			const double a[4] = {a01.x, a01.y, a23.x, a23.y}, b[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
			for (int j = 0; j < 4; j++)
#pragma unroll
				for (int i = 0; i < 4; i++) acc[i] = __fma_rn(a[j], b[i ^ j], acc[i]);
		}
#pragma unroll
		for (int i = 0; i < 4; i++) { c.acc[i] = acc[i]; c.ex[i] = -2 * SH; }
	} else {
		s.Am01[lane] = make_double2(A.m[0], A.m[1]);
		s.Am23[lane] = make_double2(A.m[2], A.m[3]);
		s.Bm01[lane] = make_double2(B.m[0], B.m[1]);
		s.Bm23[lane] = make_double2(B.m[2], B.m[3]);
		s.Ae[lane] = make_int4(A.e[0], A.e[1], A.e[2], A.e[3]);
		s.Be[lane] = make_int4(B.e[0], B.e[1], B.e[2], B.e[3]);
		__syncthreads();
		// Reference exponent of every output.  The scaling below is exact for any reference that is within ~2^1000 of the
		// output's largest term, so an estimate is enough: the largest term among the pairs that contain one of the TOPK
		// entries of either operand (a lower bound of the true top).  If some term still overflows against it the exact top
		// exponent is taken (integer max-plus pass) and the sum is redone.
		int ex[4] = {INT_MIN, INT_MIN, INT_MIN, INT_MIN};
		const int *AeI = (const int *)s.Ae, *BeI = (const int *)s.Be;
		const TopK At = top_entries(A, lane), Bt = top_entries(B, lane);
#pragma unroll
		for (int k = 0; k < TOPK; k++) {
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const int z = 4 * lane + i;
				ex[i] = max(ex[i], max(At.te[k] + BeI[z ^ At.tsym[k]], Bt.te[k] + AeI[z ^ Bt.tsym[k]]));
			}
		}
		double acc[4];
		for (int attempt = 0;; attempt++) {
			int nex[4];
#pragma unroll
			for (int i = 0; i < 4; i++) {
				nex[i] = -ex[i];
				asm("" : "+v"(nex[i])); // opaque: keeps the compiler from folding the negation back into a sub + add pair
				acc[i] = 0.0;
			}
#pragma unroll 2
			for (int g = 0; g < 64; g++) {
				const double2 a01 = s.Am01[g], a23 = s.Am23[g];
				const double2 b01 = s.Bm01[lane ^ g], b23 = s.Bm23[lane ^ g];
				const int4 av = s.Ae[g], bv = s.Be[lane ^ g];
				const double a[4] = {a01.x, a01.y, a23.x, a23.y}, b[4] = {b01.x, b01.y, b23.x, b23.y};
				const int ae[4] = {av.x, av.y, av.z, av.w}, be[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
				for (int j = 0; j < 4; j++)
#pragma unroll
					for (int i = 0; i < 4; i++) {
						const int d = ae[j] + be[i ^ j] + nex[i]; // one v_add3_u32; exact power-of-two scaling, far below the range gives 0
						acc[i] = __fma_rn(a[j], ldexp(b[i ^ j], d), acc[i]);
					}
			}
			const bool fin = acc[0] < __builtin_huge_val() && acc[1] < __builtin_huge_val() && acc[2] < __builtin_huge_val() && acc[3] < __builtin_huge_val();
			if (attempt || __all(fin)) break;
			// exact top exponent of every output
#pragma unroll 2
			for (int g = 0; g < 64; g++) {
				const int4 av = s.Ae[g], bv = s.Be[lane ^ g];
				const int ae[4] = {av.x, av.y, av.z, av.w}, be[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
				for (int j = 0; j < 4; j++)
#pragma unroll
					for (int i = 0; i < 4; i++) ex[i] = max(ex[i], ae[j] + be[i ^ j]);
			}
		}
#pragma unroll
		for (int i = 0; i < 4; i++) { c.acc[i] = acc[i]; c.ex[i] = ex[i]; }
	}
	__syncthreads(); // operands are rewritten by the next convolution
	return c;
}
// ... as a message: out[z] = LSE_x(A[x] + B[z^x]) - LSE_x(A[x] + B[x]) (log domain, out[0] = 0)
__device__ __forceinline__ void lse_conv(const XVec &A, const XVec &B, double (&out)[4], const Lds &s, int lane, unsigned long long *stamps)
{
	const ConvAcc c = conv_core(A, B, s, lane, stamps);
	double lse[4];
#pragma unroll
	for (int i = 0; i < 4; i++) lse[i] = ((nbl_log_pos(c.acc[i]) + (double)c.ex[i] * LN2) + c.base_a) + c.base_b;
	const double norm = read_lane_f64(lse[0], 0); // z = 0
#pragma unroll
	for (int i = 0; i < 4; i++) out[i] = lse[i] - norm;
	if (lane == 0) out[0] = 0.0;
}
// ... as the operand of the next convolution: the sums already are probabilities, so mantissa and exponent come from the value
// itself (no logarithm per symbol, no exponential); reference = the power of two of the largest entry, whose log-domain level
// relative to symbol 0 (= LLR 0, like every vector here) is E_max ln 2 - ln(sum of symbol 0)
__device__ __forceinline__ XVec conv_xvec(const ConvAcc &c)
{
	XVec r;
	int e[4], emax = INT_MIN, nemin = INT_MIN;
#pragma unroll
	for (int i = 0; i < 4; i++) {
		int fe;
		r.m[i] = 2.0 * frexp(c.acc[i], &fe); // acc = (m / 2) 2^fe
		e[i] = fe - 1 + c.ex[i];
		emax = max(emax, e[i]);
		nemin = max(nemin, -e[i]);
	}
	emax = wave_imax_id(emax);
	nemin = wave_imax_id(nemin);
#pragma unroll
	for (int i = 0; i < 4; i++) r.e[i] = e[i] - emax;
	const double acc0 = read_lane_f64(c.acc[0], 0);
	const int ex0 = __builtin_amdgcn_readlane(c.ex[0], 0);
	r.mx = (double)(emax - ex0) * LN2 - nbl_log_pos(acc0);
	r.rng = (double)(emax + nemin + 1) * LN2;
	return r;
}

// Two convolutions that share their first operand, A [+] B1 and A [+] B2, when both are narrow: one loop, the broadcast reads of
// A serve both (the LDS port is what the plain-double path saturates: 24 instead of 32 port cycles per 32 FMAs).  Same terms and
// the same accumulation order as two lse_conv calls.  Returns false (nothing done) when either convolution is wide.
__device__ __forceinline__ bool lse_conv_pair(const XVec &A, const XVec &B1, const XVec &B2, double (&out1)[4], double (&out2)[4], const Lds &s,
                                              int lane, unsigned long long *stamps)
{
	constexpr int SH = 500;
	if (!(fmin(A.rng, B1.rng) < 1000.0 && fmin(A.rng, B2.rng) < 1000.0)) return false; // wave-uniform
#ifdef NBL_EMS_STAMPS
	if (stamps && lane == 0) atomicAdd(&stamps[0], 2ull);
#endif
	double2 *const Cm01 = (double2 *)s.Ae, *const Cm23 = Cm01 + 64; // the exponent arrays are idle on the narrow path
	s.Am01[lane] = make_double2(ldexp(A.m[0], A.e[0] + SH), ldexp(A.m[1], A.e[1] + SH));
	s.Am23[lane] = make_double2(ldexp(A.m[2], A.e[2] + SH), ldexp(A.m[3], A.e[3] + SH));
	s.Bm01[lane] = make_double2(ldexp(B1.m[0], B1.e[0] + SH), ldexp(B1.m[1], B1.e[1] + SH));
	s.Bm23[lane] = make_double2(ldexp(B1.m[2], B1.e[2] + SH), ldexp(B1.m[3], B1.e[3] + SH));
	Cm01[lane] = make_double2(ldexp(B2.m[0], B2.e[0] + SH), ldexp(B2.m[1], B2.e[1] + SH));
	Cm23[lane] = make_double2(ldexp(B2.m[2], B2.e[2] + SH), ldexp(B2.m[3], B2.e[3] + SH));
	__syncthreads();
	double acc1[4] = {0.0, 0.0, 0.0, 0.0}, acc2[4] = {0.0, 0.0, 0.0, 0.0};
	const char *const Ab = (const char *)s.Am01, *const Bb = (const char *)s.Bm01, *const Cb = (const char *)Cm01;
	int lane16 = lane << 4;
	asm("" : "+v"(lane16));
#pragma unroll 4
	for (int g = 0; g < 64; g++) {
		const int bo = lane16 ^ (g << 4); // (addresses by hand, as in conv_core)
		const double2 a01 = *(const double2 *)(Ab + (g << 4)), a23 = *(const double2 *)(Ab + 1024 + (g << 4));
		const double2 b01 = *(const double2 *)(Bb + bo), b23 = *(const double2 *)(Bb + 1024 + bo);
		const double2 c01 = *(const double2 *)(Cb + bo), c23 = *(const double2 *)(Cb + 1024 + bo);
		const double a[4] = {a01.x, a01.y, a23.x, a23.y}, b[4] = {b01.x, b01.y, b23.x, b23.y}, c[4] = {c01.x, c01.y, c23.x, c23.y};
#pragma unroll
		for (int j = 0; j < 4; j++)
#pragma unroll
			for (int i = 0; i < 4; i++) {
				acc1[i] = __fma_rn(a[j], b[i ^ j], acc1[i]);
				acc2[i] = __fma_rn(a[j], c[i ^ j], acc2[i]);
			}
	}
	double l1[4], l2[4];
#pragma unroll
	for (int i = 0; i < 4; i++) {
		l1[i] = ((nbl_log_pos(acc1[i]) - (2 * SH) * LN2) + A.mx) + B1.mx;
		l2[i] = ((nbl_log_pos(acc2[i]) - (2 * SH) * LN2) + A.mx) + B2.mx;
	}
	const double n1 = read_lane_f64(l1[0], 0), n2 = read_lane_f64(l2[0], 0);
#pragma unroll
	for (int i = 0; i < 4; i++) { out1[i] = l1[i] - n1; out2[i] = l2[i] - n2; }
	if (lane == 0) { out1[0] = 0.0; out2[0] = 0.0; }
	__syncthreads();
	return true;
}

template <bool FUSED>
__global__ __launch_bounds__(64, 3) void cn_bp_q256_dc4_kernel(NblGraphDev g, NblWork w, NblRun r)
{
	__shared__ __attribute__((aligned(16))) double smem[4 * Q]; // 8 KB: four permutation buffers, then the operands
	const int lane = lane_id();
	const NblXcdSlot xs = nbl_xcd_slot(g.M); // all checks of a codeword on one XCD (nbl_device.h)
	const int b = nbl_codeword(w, r, xs.slot), m = xs.unit;
	if (b < 0) return;
	if (!r.fixed_iters && w.done[b]) return;
	const int c0 = g.coff[m];

	double *V = w.v2c + (size_t)b * g.E * Q;
	double *C = w.c2v + ((size_t)b * g.E + c0) * Q;

	// inputs into the check domain: p_d[h_d a] = v2c_d[a], symbol 0 = LLR 0 (the A1 == 0 branch of LLR_BoxPlus, :1623-1632)
	int hsym[4][4]; // h_d * a for the lane's four symbols a = lane + 64 i (from the multiplication table, L1 / L2 resident)
#pragma unroll
	for (int d = 0; d < 4; d++) {
		double *Vd = V + (size_t)g.c_epos[c0 + d] * Q;
		{
			const uint8_t *mrow = g.mul + (size_t)g.c_h[c0 + d] * Q;
#pragma unroll
			for (int i = 0; i < 4; i++) hsym[d][i] = mrow[lane + 64 * i];
		}
		if (!FUSED) {
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const int a = lane + 64 * i;
				smem[d * Q + hsym[d][i]] = (a == 0) ? 0.0 : Vd[a];
			}
		} else {
			// FUSED: the variable-node pass of this iteration for this edge (NBLDPC.cpp:676-691, :718-744): post = (L_ch + c2v_0)
			// + c2v_1 of the edge's variable (dv = 2), hard decision by the check that holds the variable's first edge,
			// v2c = post - c2v of this edge, damped 1/2 : 1/2 against the previous v2c when its hard decision moves.
			const double *Cp = w.c2v_prev + (w.c2v_prev_shared ? (size_t)0 : (size_t)b * g.E * Q); // (iteration 1: one shared block of zeros)
			const int n = g.c_var[c0 + d], e = g.c_epos[c0 + d], e0 = g.voff[n];
			const double *pl = w.Lch + ((size_t)b * g.N + n) * Q;
			const double *pa = Cp + (size_t)g.v_cpos[e0] * Q, *pb = Cp + (size_t)g.v_cpos[e0 + 1] * Q;
			const bool ownA = (e == e0);
			double post[4], nv[4];
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const int a = lane + 64 * i;
				const double ca = pa[a], cb = pb[a];
				post[i] = (pl[a] + ca) + cb;
				nv[i] = post[i] - (ownA ? ca : cb);
			}
			if (ownA) {
				const int dec = wave_decide_keyed<4>(post, lane, Q);
				if (lane == 0) w.dec[(size_t)b * g.N + n] = dec;
				if (w.post) {
#pragma unroll
					for (int i = 0; i < 4; i++) w.post[((size_t)b * g.N + n) * Q + lane + 64 * i] = post[i];
				}
			}
			// Damping (:730-741) compares the hard decision of the previous v2c with the new one.  The previous decision is what this
			// stage recorded for that vector one iteration ago (w.edge_dec), so the previous v2c itself is loaded -- and reduced -- only
			// in iteration 1 (where it is L_ch and nothing has been recorded) or when the decision moved and the blend needs it.
			const bool first = (r.iter == 1);
			int before;
			if (first) {
				double ov[4];
#pragma unroll
				for (int i = 0; i < 4; i++) ov[i] = Vd[lane + 64 * i];
				before = wave_decide_keyed<4>(ov, lane, Q);
			} else before = w.edge_dec[(size_t)b * g.E + e];
			int after = wave_decide_keyed<4>(nv, lane, Q);
			if (before != after) {
#pragma unroll
				for (int i = 0; i < 4; i++) nv[i] = __dadd_rn(__dmul_rn(r.damp_old, Vd[lane + 64 * i]), __dmul_rn(r.damp_new, nv[i]));
				double t[4];
#pragma unroll
				for (int i = 0; i < 4; i++) t[i] = (lane + 64 * i == 0) ? 0.0 : nv[i];
				after = wave_decide_keyed<4>(t, lane, Q);
			}
			if (lane == 0) w.edge_dec[(size_t)b * g.E + e] = after;
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const int a = lane + 64 * i;
				const double x = (a == 0) ? 0.0 : nv[i];
				V[(size_t)e * Q + a] = x;
				smem[d * Q + hsym[d][i]] = x;
			}
		}
	}
	__syncthreads();
	BPMARK(0);
	XVec p[4];
#pragma unroll
	for (int d = 0; d < 4; d++) {
		double L[4];
#pragma unroll
		for (int i = 0; i < 4; i++) L[i] = smem[d * Q + 4 * lane + i];
		p[d] = to_xvec(L, lane);
	}
	__syncthreads();
	BPMARK(1);

	Lds s;
	s.Am01 = (double2 *)smem;
	s.Am23 = s.Am01 + 64;
	s.Bm01 = s.Am23 + 64;
	s.Bm23 = s.Bm01 + 64;
	s.Ae = (int4 *)(s.Bm23 + 64);
	s.Be = s.Ae + 64;
	double *T = (double *)(s.Be + 64); // [Q] output staging (bytes 6144..8191)

	auto emit = [&](const double (&o)[4], int d) {
#pragma unroll
		for (int i = 0; i < 4; i++) T[4 * lane + i] = o[i];
		__syncthreads();
		double *Cd = C + (size_t)d * Q;
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const int a = lane + 64 * i;
			Cd[a] = (a == 0) ? 0.0 : T[hsym[d][i]];
		}
		__syncthreads();
	};

	double o[4];
	// forward: F2 = p0 [+] p1, F3 = F2 [+] p2 = output 3 (A2 == 0 case, :761-764); output 2 = F2 [+] p3
	{
		const ConvAcc cF2 = conv_core(p[0], p[1], s, lane, w.stamps);
		BPMARK(2);
		const XVec F2 = conv_xvec(cF2);
		BPMARK(3);
		double o2[4];
		if (lse_conv_pair(F2, p[2], p[3], o, o2, s, lane, w.stamps)) {
			BPMARK(4);
			emit(o, 3);
			emit(o2, 2);
		} else {
			lse_conv(F2, p[2], o, s, lane, w.stamps);
			emit(o, 3);
			lse_conv(F2, p[3], o, s, lane, w.stamps);
			emit(o, 2);
		}
	}
	BPMARK(5);
	// backward: R1 = p3 [+] p2, R0 = R1 [+] p1 = output 0 (A1 == 0 case, :757-760); output 1 = p0 [+] R1
	{
		const XVec R1 = conv_xvec(conv_core(p[3], p[2], s, lane, w.stamps));
		double o2[4];
		// output 1 = p0 [+] R1 is taken as R1 [+] p0 here: the same products, summed in the mirrored order
		if (lse_conv_pair(R1, p[1], p[0], o, o2, s, lane, w.stamps)) {
			emit(o, 0);
			emit(o2, 1);
		} else {
			lse_conv(R1, p[1], o, s, lane, w.stamps);
			emit(o, 0);
			lse_conv(p[0], R1, o, s, lane, w.stamps);
			emit(o, 1);
		}
	}
}

} // namespace

bool nbl_bp256_applicable(const NblGraphDev &g, bool all_dc4) { return g.q == 256 && all_dc4; }

hipError_t nbl_launch_cn_bp256(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool fused, hipStream_t st)
{
	dim3 grid(nbl_xcd_grid(r.B, g.M)), block(64);
	if (fused) cn_bp_q256_dc4_kernel<true><<<grid, block, 0, st>>>(g, w, r);
	else cn_bp_q256_dc4_kernel<false><<<grid, block, 0, st>>>(g, w, r);
	return hipGetLastError();
}
