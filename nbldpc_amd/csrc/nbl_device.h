// nbldpc_amd/csrc/nbl_device.h -- wave-level device helpers (gfx950, wave64) shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "nbl_common.h"

#define NBL_NEG_INF (-__builtin_huge_val())
#define NBL_DBL_MAX 1.7976931348623157e308

// lane l of a wave owns symbols a = l + 64*i, i < NS.  For q < 64 only lanes < q own a symbol.
template <int Q> struct Fld {
	static constexpr int NS = (Q + 63) / 64;
	static constexpr int P = (Q == 4) ? 2 : (Q == 8) ? 3 : (Q == 16) ? 4 : (Q == 32) ? 5 : (Q == 64) ? 6 : (Q == 128) ? 7 : 8;
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// codeword of grid slot `slot` (wave-uniform), -1 = nothing to do.  With the active list (early exit on large batches) the grid
// only covers the codewords that were still iterating after the previous window.
// XCD-aware grids of the fused check-node kernels.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8), each with
// its own L2.  A fused iteration reads every L_ch / c2v vector of a codeword twice or three times (once per check the variable
// joins): all `units` workgroups of a codeword slot are therefore given to ONE XCD, so that the later reads can hit that L2
// instead of HBM.  Speed only: nothing depends on the placement.  The grid is rounded up to whole groups of 8 slots; slots
// beyond the batch find no codeword (nbl_codeword returns -1).
struct NblXcdSlot { int slot, unit; };
__device__ __forceinline__ NblXcdSlot nbl_xcd_slot(int units)
{
	const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
	NblXcdSlot s;
	s.slot = (idx / units) * 8 + xcd;
	s.unit = idx % units;
	return s;
}
inline unsigned nbl_xcd_grid(int B, int units) { return (unsigned)((long long)((B + 7) / 8) * 8 * units); }

__device__ __forceinline__ int nbl_codeword(const NblWork &w, const NblRun &r, int slot)
{
	if (slot >= r.B) return -1;
	if (!w.active) return slot;
	return slot < *w.n_act ? w.active[slot] : -1;
}

// v_max_f64 without the canonicalising pre-max hipcc adds in front of fmax() on loaded values
__device__ __forceinline__ double dmax(double a, double b)
{
	double r;
	asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
__device__ __forceinline__ double dmin(double a, double b)
{
	double r;
	asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}

__device__ __forceinline__ double read_lane_f64(double v, int lane)
{
	int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
	int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
	return __hiloint2double(hi, lo);
}

// wave-wide reductions on the DPP network: row_shr 1, 2, 4, 8 leave the row result in lane 15 of every row, row_bcast15 /
// row_bcast31 carry it across the rows; lane 63 holds the wave result.  Lanes without a source keep their own value, which is
// neutral for max / min.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int dpp_i32(int v)
{
	return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_f64(double v)
{
	return __hiloint2double(dpp_i32<CTRL, ROW_MASK>(__double2hiint(v)), dpp_i32<CTRL, ROW_MASK>(__double2loint(v)));
}
__device__ __forceinline__ double wave_fmax(double v)
{
	v = dmax(v, dpp_f64<0x111, 0xF>(v));
	v = dmax(v, dpp_f64<0x112, 0xF>(v));
	v = dmax(v, dpp_f64<0x114, 0xF>(v));
	v = dmax(v, dpp_f64<0x118, 0xF>(v));
	v = dmax(v, dpp_f64<0x142, 0xA>(v));
	v = dmax(v, dpp_f64<0x143, 0xC>(v));
	return read_lane_f64(v, 63);
}
__device__ __forceinline__ double wave_fmin(double v)
{
	v = dmin(v, dpp_f64<0x111, 0xF>(v));
	v = dmin(v, dpp_f64<0x112, 0xF>(v));
	v = dmin(v, dpp_f64<0x114, 0xF>(v));
	v = dmin(v, dpp_f64<0x118, 0xF>(v));
	v = dmin(v, dpp_f64<0x142, 0xA>(v));
	v = dmin(v, dpp_f64<0x143, 0xC>(v));
	return read_lane_f64(v, 63);
}
__device__ __forceinline__ int wave_imax(int v)
{
	v = max(v, dpp_i32<0x111, 0xF>(v));
	v = max(v, dpp_i32<0x112, 0xF>(v));
	v = max(v, dpp_i32<0x114, 0xF>(v));
	v = max(v, dpp_i32<0x118, 0xF>(v));
	v = max(v, dpp_i32<0x142, 0xA>(v));
	v = max(v, dpp_i32<0x143, 0xC>(v));
	return __builtin_amdgcn_readlane(v, 63);
}

// Order-preserving 32-bit key of a double (rounded to float, sign-magnitude folded to two's complement) and a wave maximum of
// such keys: with the operation's identity as the DPP `old` value hipcc folds the move into v_max_i32_dpp (six instructions).
__device__ __forceinline__ int nbl_key32(double x)
{
	const int b = __float_as_int((float)x);
	return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float nbl_unkey32(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }
__device__ __forceinline__ int wave_imax_id(int x)
{
	constexpr int ID = (int)0x80000000;
	x = max(x, __builtin_amdgcn_update_dpp(ID, x, 0x111, 0xF, 0xF, false));
	x = max(x, __builtin_amdgcn_update_dpp(ID, x, 0x112, 0xF, 0xF, false));
	x = max(x, __builtin_amdgcn_update_dpp(ID, x, 0x114, 0xF, 0xF, false));
	x = max(x, __builtin_amdgcn_update_dpp(ID, x, 0x118, 0xF, 0xF, false));
	x = max(x, __builtin_amdgcn_update_dpp(ID, x, 0x142, 0xA, 0xF, false));
	x = max(x, __builtin_amdgcn_update_dpp(ID, x, 0x143, 0xC, 0xF, false));
	return __builtin_amdgcn_readlane(x, 63);
}

__device__ __forceinline__ unsigned wave_umax_id(unsigned x)
{
	x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false));
	x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false));
	x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false));
	x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false));
	x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false));
	x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false));
	return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}
// Exact wave maximum of max(x, 0) -- what every hard decision and every "most reliable symbol" search needs: the running maximum
// of DecideLLRVector starts at 0 (NBLDPC.cpp:1545), and symbol 0 carries the value 0.  Non-negative doubles order like their bit
// patterns, so the maximum is two unsigned 32-bit reductions: the high words, then the low words of the lanes that hold the top
// high word -- 17 vector instructions, each DPP step folded into its v_max_u32, against 32 for the FP64 reduction (an FP64
// instruction cannot take a DPP operand: two moves in front of every maximum).  Same value as wave_fmax(max(x, 0)), bit for bit.
__device__ __forceinline__ double wave_fmax_nonneg(double x)
{
	x = dmax(x, 0.0);
	const unsigned hi = (unsigned)__double2hiint(x), mh = wave_umax_id(hi);
	const unsigned lo = (hi == mh) ? (unsigned)__double2loint(x) : 0u, ml = wave_umax_id(lo);
	return __hiloint2double((int)mh, (int)ml);
}

__device__ __forceinline__ unsigned wave_umin_id(unsigned x)
{
	constexpr int ID = -1;
	x = min(x, (unsigned)__builtin_amdgcn_update_dpp(ID, (int)x, 0x111, 0xF, 0xF, false));
	x = min(x, (unsigned)__builtin_amdgcn_update_dpp(ID, (int)x, 0x112, 0xF, 0xF, false));
	x = min(x, (unsigned)__builtin_amdgcn_update_dpp(ID, (int)x, 0x114, 0xF, 0xF, false));
	x = min(x, (unsigned)__builtin_amdgcn_update_dpp(ID, (int)x, 0x118, 0xF, 0xF, false));
	x = min(x, (unsigned)__builtin_amdgcn_update_dpp(ID, (int)x, 0x142, 0xA, 0xF, false));
	x = min(x, (unsigned)__builtin_amdgcn_update_dpp(ID, (int)x, 0x143, 0xC, 0xF, false));
	return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}
// Exact wave minimum of non-negative doubles (+inf allowed, no NaN), the same way: high words, then the low words of the lanes that
// hold the smallest high word.
__device__ __forceinline__ double wave_fmin_nonneg(double x)
{
	const unsigned hi = (unsigned)__double2hiint(x), mh = wave_umin_id(hi);
	const unsigned lo = (hi == mh) ? (unsigned)__double2loint(x) : 0xffffffffu, ml = wave_umin_id(lo);
	return __hiloint2double((int)mh, (int)ml);
}

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ double uniform_f64(double v)
{
	return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
	                        __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// number of set bits of `mask` below this lane
__device__ __forceinline__ int prefix_count(uint64_t mask)
{
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

// Multiplication by a wave-uniform GF(2^p) element h, done as a GF(2)-linear map instead of a table:
// h*a = XOR_{bit i of a} h*x^i.  `lane_part` folds the lane's bits 0..5 once; slot bits 6,7 are constants.
template <int Q> struct GfMul {
	int basis[8];
	int lane_part;
	__device__ __forceinline__ void init(int h, int poly, int lane)
	{
		int x = h;
#pragma unroll
		for (int i = 0; i < 8; i++) {
			basis[i] = x;
			x <<= 1;
			if (x & Q) x ^= poly;
		}
		lane_part = 0;
#pragma unroll
		for (int i = 0; i < 6; i++)
			if (i < Fld<Q>::P) lane_part ^= (-((lane >> i) & 1)) & basis[i];
	}
	// product with a = lane + 64*slot
	__device__ __forceinline__ int at_slot(int slot) const
	{
		int t = lane_part;
		if (slot & 1) t ^= basis[6];
		if (slot & 2) t ^= basis[7];
		return t;
	}
};

// Hard decision of DecideLLRVector (NBLDPC.cpp:1542-1562): running maximum starts at 0, strict '>', so the
// lowest symbol wins ties and a vector without a positive entry decides 0.  v[i] belongs to a = lane+64 i;
// slot a = 0 must hold a value <= 0 (it holds 0.0).
template <int NS> __device__ __forceinline__ int wave_decide(const double (&v)[NS], int lane, int q)
{
	double best = 0.0;
#pragma unroll
	for (int i = 0; i < NS; i++)
		if (lane + 64 * i < q) best = dmax(best, v[i]);
	if (!__ballot(best > 0.0)) return 0; // nothing positive: every lane's running maximum is still the initial 0 (one compare, no reduction)
	const double mx = wave_fmax_nonneg(best);
	int arg = 0;
#pragma unroll
	for (int i = NS - 1; i >= 0; i--) { // the lowest symbol that holds the maximum: lowest slot first, then lowest lane
		const uint64_t hit = __ballot(lane + 64 * i < q && v[i] == mx);
		arg = hit ? 64 * i + __builtin_ctzll(hit) : arg;
	}
	return arg; // (mx > 0 here)
}

// The same decision with the reduction on 32-bit keys: a vector without a positive entry decides 0 after one compare (every
// lane's running maximum is the initial 0.0 then: no reduction at all); otherwise the wave maximum of the order-preserving
// float keys (one DPP instruction per step instead of five for an FP64 step) names the lane that holds the maximum -- rounding
// to float is monotone -- and the exact value is read from it; several lanes with the maximal key (equal or nearly equal
// maxima) fall back to the FP64 reduction.  Same result as wave_decide for every input.
template <int NS> __device__ __forceinline__ int wave_decide_keyed(const double (&v)[NS], int lane, int q)
{
	double best = 0.0;
#pragma unroll
	for (int i = 0; i < NS; i++)
		if (lane + 64 * i < q) best = dmax(best, v[i]);
	if (!__ballot(best > 0.0)) return 0;
	const int k = nbl_key32(best), kmax = wave_imax_id(k);
	const uint64_t c = __ballot(k == kmax);
	const double mx = (__builtin_popcountll(c) == 1) ? read_lane_f64(best, __builtin_ctzll(c)) : wave_fmax(best);
	int arg = 0;
#pragma unroll
	for (int i = NS - 1; i >= 0; i--) {
		const uint64_t hit = __ballot(lane + 64 * i < q && v[i] == mx);
		arg = hit ? 64 * i + __builtin_ctzll(hit) : arg;
	}
	return arg; // (mx > 0 here)
}

// check-to-variable shaping, NBLDPC.cpp:903-916 / 1113-1126
__device__ __forceinline__ double shape_llr(double y, double factor, double offset)
{
	// unshaped (factor 1, offset 0 -- the values of the reference's sample profile): the dead zone is the single point 0, and
	// NaN maps to 0 as below
	if (factor == 1.0 && offset == 0.0) return (y < 0.0 || y > 0.0) ? y : 0.0;
	if (factor != 1.0) y = y / factor;
	if (y < -1 * offset) return y + offset;
	if (y > offset) return y - offset;
	return 0.0;
}
