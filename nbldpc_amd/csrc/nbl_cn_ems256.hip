// nbldpc_amd/csrc/nbl_cn_ems256.hip -- EMS check node specialised for the headline shape:
//   GF(256), every check of degree 4, any nc >= 1, any nm <= 64 (8, 16, 32, 64 as compile-time constants).
// (BASELINE configs 2 and 3: divsalar.UNBLDPC.{128.64,512.256}.GF.256, EMS nm=16/32.)
//
// Same arithmetic as cn_ems_kernel<256> (nbl_kernels.hip) -- the generic kernel is the readable statement of the
// algorithm and serves every other shape; tests check both against the oracle.  What differs is the mapping:
//   * lane l owns symbols {2l, 2l+1, 128+2l, 129+2l}: a q-vector is two 16-byte loads/stores per lane, and a
//     max-plus XOR-gather "P[s ^ t]" is two conflict-free ds_read_b128 (natural layout, lane XOR only permutes the
//     16-byte slots inside one 256-byte LDS row).
//   * the incoming vectors arrive in the variable domain (symbol a) and are moved ONCE into the check domain (t = h a)
//     through LDS -- the staging conf(q,1) needs anyway; everything after that (top-nm selection, list building, the
//     convolutions) works on check-domain registers, where the lane's symbols are the same for all four edges: slot
//     parity is a compile-time property, list entries need no per-element symbol arithmetic.  The byte offsets 8 h a
//     of the permutation come from a per-edge table built by nbl_create (g.ems_toff), not from GF arithmetic.
//   * top-nm selection without sorting: a 256-bucket histogram (LDS atomics + DPP prefix sum) finds the bucket that
//     holds the nm-th best value; if that bucket ends exactly at the nm-th entry the members are known, otherwise a ballot
//     quickselect inside the bucket finds the exact cut under SortLLRVector's order (value desc, higher symbol first
//     among equals).
//   * wave maxima / minima are reduced on order-preserving 32-bit keys (one DPP instruction per step); the lane that holds
//     the maximum key delivers the exact FP64 value.
//   * conf(nm, nc >= 3): the three pair convolutions e0(+)e1, e0(+)e2, e1(+)e2 are scattered in one pass into three LDS
//     buffers; the outputs 2, 1, 0 all gather over the list of edge 3 and share one loop (one broadcast read of the list
//     entry and one address per entry serve three convolutions).
//   * 8 KB LDS per wave at nm = 32 (three q-vectors + the lists), four waves per SIMD (VGPR-limited).
#include <hip/hip_runtime.h>
#ifndef NBL_EMS_PRUNE
#define NBL_EMS_PRUNE 1 // the exact bound on the gather lists (0: every entry, for A/B runs)
#endif
#include <type_traits>
#include "nbl_device.h"
#include "nbl_kernels.h"

#ifndef NBL_WHATIF
#define NBL_WHATIF 0 // 1..7: diagnostic builds that leave one kind of work out (tools/whatif.sh); the results are wrong on purpose
#endif

namespace {

constexpr int Q = 256;

// list entry of a check-domain symbol t: value + {t8 = 8 t: byte offset of the symbol, tt = (t & 0xFE) << 3: byte-offset XOR of the
// gather}; values and offset pairs live in separate arrays (two 8-byte stores from registers that already exist)
struct ListEnt { double v; int t8; int tt; };

// One wave per workgroup: LDS operations of one wave execute in issue order, so phases that hand data over through LDS only
// need the COMPILER to keep the order -- no s_barrier, no drain of the LDS queue.
#define WSYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")

// ---- 64-bit wave max through DPP moves (no LDS); used only where the 32-bit key reduction is ambiguous ---------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_mov_f64(double x)
{
	int lo = __builtin_amdgcn_update_dpp(__double2loint(x), __double2loint(x), CTRL, ROW_MASK, 0xF, false);
	int hi = __builtin_amdgcn_update_dpp(__double2hiint(x), __double2hiint(x), CTRL, ROW_MASK, 0xF, false);
	return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_max_f64(double x)
{
	x = dmax(x, dpp_mov_f64<0xB1>(x));        // quad_perm [1,0,3,2]
	x = dmax(x, dpp_mov_f64<0x4E>(x));        // quad_perm [2,3,0,1]
	x = dmax(x, dpp_mov_f64<0x141>(x));       // row_half_mirror
	x = dmax(x, dpp_mov_f64<0x140>(x));       // row_mirror
	x = dmax(x, dpp_mov_f64<0x142, 0xA>(x));  // row_bcast15 into rows 1,3
	x = dmax(x, dpp_mov_f64<0x143, 0xC>(x));  // row_bcast31 into rows 2,3
	return read_lane_f64(x, 63);
}

// inclusive prefix sum over the 64 lanes (the DPP sequence LLVM's atomic optimiser emits on gfx9)
__device__ __forceinline__ int wave_scan_add(int x)
{
	x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false); // row_shr:1
	x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false); // row_shr:2
	x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false); // row_shr:4
	x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false); // row_shr:8
	x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false); // row_bcast15 -> rows 1,3
	x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false); // row_bcast31 -> rows 2,3
	return x;
}

// wave max / min of 32-bit integers: with the operation's identity as the DPP `old` value hipcc folds the move into the
// v_max_i32_dpp / v_min_i32_dpp itself (six instructions per reduction)
__device__ __forceinline__ int wave_max_i32(int x)
{
	constexpr int ID = (int)0x80000000;
	auto mx = [](int a, int b) { return a > b ? a : b; };
	x = mx(x, __builtin_amdgcn_update_dpp(ID, x, 0xB1, 0xF, 0xF, false));
	x = mx(x, __builtin_amdgcn_update_dpp(ID, x, 0x4E, 0xF, 0xF, false));
	x = mx(x, __builtin_amdgcn_update_dpp(ID, x, 0x141, 0xF, 0xF, false));
	x = mx(x, __builtin_amdgcn_update_dpp(ID, x, 0x140, 0xF, 0xF, false));
	x = mx(x, __builtin_amdgcn_update_dpp(ID, x, 0x142, 0xA, 0xF, false));
	x = mx(x, __builtin_amdgcn_update_dpp(ID, x, 0x143, 0xC, 0xF, false));
	return __builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ int wave_min_i32(int x)
{
	constexpr int ID = 0x7fffffff;
	auto mn = [](int a, int b) { return a < b ? a : b; };
	x = mn(x, __builtin_amdgcn_update_dpp(ID, x, 0xB1, 0xF, 0xF, false));
	x = mn(x, __builtin_amdgcn_update_dpp(ID, x, 0x4E, 0xF, 0xF, false));
	x = mn(x, __builtin_amdgcn_update_dpp(ID, x, 0x141, 0xF, 0xF, false));
	x = mn(x, __builtin_amdgcn_update_dpp(ID, x, 0x140, 0xF, 0xF, false));
	x = mn(x, __builtin_amdgcn_update_dpp(ID, x, 0x142, 0xA, 0xF, false));
	x = mn(x, __builtin_amdgcn_update_dpp(ID, x, 0x143, 0xC, 0xF, false));
	return __builtin_amdgcn_readlane(x, 63);
}

// order-preserving 32-bit key of a double: round to float (monotone), then sign-magnitude -> two's complement
__device__ __forceinline__ int key32(double x)
{
	const int b = __float_as_int((float)x);
	return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float unkey32(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }

// exact wave maximum of `loc` given the wave maximum `kmax` of its keys: rounding is monotone, so the maximum sits in a
// lane whose key equals kmax; one such lane (the rule) -> read it, several -> full FP64 reduction
__device__ __forceinline__ double wave_max_exact(double loc, int key, int kmax)
{
	const uint64_t c = __ballot(key == kmax);
	if (__builtin_popcountll(c) == 1) return read_lane_f64(loc, __builtin_ctzll(c));
	return wave_max_f64(loc);
}

// symbol of slot i of lane l in this kernel's layout
__device__ __forceinline__ int sym_of(int lane, int i) { return 2 * lane + (i & 1) + 128 * (i >> 1); }

// highest symbol among the set bits of four slot masks (-1 if empty)
__device__ __forceinline__ int highest_sym(const uint64_t (&m)[4])
{
	int best = -1;
#pragma unroll
	for (int i = 0; i < 4; i++) {
		if (m[i]) {
			int l = 63 - __builtin_clzll(m[i]);
			int a = 2 * l + (i & 1) + 128 * (i >> 1);
			best = a > best ? a : best;
		}
	}
	return best;
}
// lowest symbol among the set bits of four slot masks (Q if empty)
__device__ __forceinline__ int lowest_sym(const uint64_t (&m)[4])
{
	int best = Q;
#pragma unroll
	for (int i = 0; i < 4; i++) {
		if (m[i]) {
			int l = __builtin_ctzll(m[i]);
			int a = 2 * l + (i & 1) + 128 * (i >> 1);
			best = a < best ? a : best;
		}
	}
	return best;
}

// h * a for this layout: lane bits 0..5 are symbol bits 1..6, slot bit 0 is symbol bit 0, slot bit 1 is symbol bit 7
// (only the rare tie path of the selection uses it; the permutation offsets of the main path come from g.ems_toff)
struct GfMulL {
	int b0, b7, lane_part;
	__device__ __forceinline__ void init(int h, int poly, int lane)
	{
		int x = h;
		b0 = x;
		lane_part = 0;
#pragma unroll
		for (int i = 1; i < 8; i++) {
			x <<= 1;
			if (x & Q) x ^= poly;
			if (i < 7) lane_part ^= (-((lane >> (i - 1)) & 1)) & x;
			else b7 = x;
		}
	}
	__device__ __forceinline__ int at_slot(int i) const { return lane_part ^ ((i & 1) ? b0 : 0) ^ ((i & 2) ? b7 : 0); }
};

struct SelState {
	uint64_t cand[4];
	uint64_t gt[4], eq[4];
	int done;
};

// one quickselect step for one edge: pivot = first remaining candidate, counts by ballot
__device__ __forceinline__ void select_step(const double (&v)[4], int nm, SelState &s)
{
	if (s.done) return;
	const uint64_t any = s.cand[0] | s.cand[1] | s.cand[2] | s.cand[3];
	if (!any) { s.done = 1; return; } // only reachable with NaN inputs
	double offer = __builtin_amdgcn_inverse_ballot_w64(s.cand[0]) ? v[0] : __builtin_amdgcn_inverse_ballot_w64(s.cand[1]) ? v[1]
	             : __builtin_amdgcn_inverse_ballot_w64(s.cand[2]) ? v[2] : v[3];
	const double pv = read_lane_f64(offer, __builtin_ctzll(any));
	uint64_t gt[4], ge[4];
	int cgt = 0, cge = 0;
#pragma unroll
	for (int i = 0; i < 4; i++) {
		gt[i] = __ballot(v[i] > pv);
		ge[i] = __ballot(v[i] >= pv);
		cgt += __popcll(gt[i]);
		cge += __popcll(ge[i]);
	}
	if (cgt >= nm) {
#pragma unroll
		for (int i = 0; i < 4; i++) s.cand[i] &= gt[i];
	} else if (cge >= nm) {
#pragma unroll
		for (int i = 0; i < 4; i++) { s.gt[i] = gt[i]; s.eq[i] = ge[i] & ~gt[i]; }
		s.done = 1;
	} else {
#pragma unroll
		for (int i = 0; i < 4; i++) s.cand[i] &= ~ge[i];
	}
}

// members of the nm best under SortLLRVector's order (value desc, higher symbol first among equals, NBLDPC.cpp:1731).
// The masks are over CHECK-domain slots; the tie rule is about the variable-domain symbol a = hinv * t.
__device__ __forceinline__ void finish_ties(uint64_t (&member)[4], const uint64_t (&eqin)[4], int need, int hinv, int poly, int lane)
{
	GfMulL mh;
	mh.init(hinv, poly, lane);
	uint64_t eq[4] = {eqin[0], eqin[1], eqin[2], eqin[3]};
	for (; need > 0; need--) {
		int best = -1;
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const int a = __builtin_amdgcn_inverse_ballot_w64(eq[i]) ? mh.at_slot(i) : -1;
			best = a > best ? a : best;
		}
		const int amax = wave_max_i32(best);
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const uint64_t hit = __ballot(__builtin_amdgcn_inverse_ballot_w64(eq[i]) && mh.at_slot(i) == amax);
			member[i] |= hit;
			eq[i] &= ~hit;
		}
	}
}

__device__ __forceinline__ void finish_members(SelState &s, int nm, uint64_t (&member)[4], int hinv, int poly, int lane)
{
	int cgt = 0, ceq = 0;
#pragma unroll
	for (int i = 0; i < 4; i++) { cgt += __popcll(s.gt[i]); ceq += __popcll(s.eq[i]); }
	const int need = nm - cgt;
#pragma unroll
	for (int i = 0; i < 4; i++) member[i] = s.gt[i];
	if (need == ceq) {
#pragma unroll
		for (int i = 0; i < 4; i++) member[i] |= s.eq[i];
	} else {
		finish_ties(member, s.eq, need, hinv, poly, lane); // ties straddle the cut: the highest symbols of the tie group (rare)
	}
}

} // namespace

// FUSED = true: one launch is a whole flooding iteration for dv = 2 codes.  The variable-node work of the reference
// (a-posteriori sum L_ch + c2v_0 + c2v_1 in that order, hard decision, v2c = L_post - c2v, NBLDPC.cpp:808-823, 848-857) is
// recomputed per edge from the previous iteration's c2v, so v2c never touches HBM; the wave that holds a variable's FIRST
// edge writes its hard decision.  c2v is double-buffered (w.c2v_prev -> w.c2v) because the schedule is flooding.
// NC = min(nc, 3): 3 = no deviation counting needed (every other edge may deviate); 2 = at least one of the three other
// edges stays at rank 0; 1 = conf(nm,1) is a subset of conf(q,1), nothing beyond the closed form is needed.
// NMT > 0: nm = NMT, a compile-time constant (8, 16, 32, 64).  NMT < 0: any other nm, given at run time, on the layout of the next
// power of two -NMT -- the unused tail of every list is pre-filled with entries of value -inf, which no maximum can pick up:
// the pair scatters run over them (cost of the larger layout), the gathers stop at nm.
template <int NMT, bool FUSED, int NC>
__global__ __launch_bounds__(64) void cn_ems_q256_dc4_kernel(NblGraphDev g, NblWork w, NblRun r)
{
	constexpr int NM = NMT < 0 ? -NMT : NMT;
	constexpr bool VARNM = NMT < 0;
	const int nmr = VARNM ? r.nm : NM; // the nm of the selection (the list layout and the loops over it use NM)
	constexpr int LOGNM = (NM == 8) ? 3 : (NM == 16) ? 4 : (NM == 32) ? 5 : 6;
	// static LDS: its base is a compile-time constant, so computed addresses fold into the ds_* offset fields (with `extern
	// __shared__` every computed address pays a v_add with the link-time base)
#if NBL_WHATIF == 5   // diagnostic: three waves per SIMD (13 KB of LDS per wave)
	__shared__ __attribute__((aligned(16))) char smem[13 * 1024];
#elif NBL_WHATIF == 6 // diagnostic: two waves per SIMD (19 KB)
	__shared__ __attribute__((aligned(16))) char smem[19 * 1024];
#else
	__shared__ __attribute__((aligned(16))) char smem[3 * Q * 8 + 4 * NM * 8 + 4 * NM * 8];
#endif
	const int lane = lane_id();
	// XCD-aware mapping (nbl_device.h): all checks of a codeword run on ONE XCD, so the second read of every L_ch / c2v vector
	// (each is used by two checks) can hit that XCD's L2.  Speed only: nothing depends on the placement.
	const NblXcdSlot xs = nbl_xcd_slot(g.M);
	const int m = xs.unit, b = nbl_codeword(w, r, xs.slot);
	if (b < 0) return;
	if (!r.fixed_iters && w.done[b]) return;
#if NBL_EMS_PRUNE
	// A codeword whose syndrome has been zero keeps iterating in fixed-iteration runs; its vectors have one dominant symbol each,
	// which is where the short lists below pay (they are exact for any input -- the flag only decides whether the bounds are worth
	// forming: on a codeword that is still searching they leave out next to nothing).
	const bool tryf = NC >= 2 && r.fixed_iters && w.done[b];
#else
	const bool tryf = false;
#endif
	const int c0 = g.coff[m];

	// three q-vectors: staging of the incoming vectors | histogram (B0..B1) | the three pair convolutions | staging of the outputs
	double *const B0 = (double *)smem, *const B1 = B0 + Q, *const B2 = B1 + Q;
	// the nm best of every edge, [even-symbol group | odd-symbol group]: values [4][NM], offsets {t8, tt} [4][NM]
	double *const lstv = B2 + Q;
	int2 *const lstt = (int2 *)(lstv + 4 * NM);
	auto list_at = [&](int j, int k) {
		ListEnt e;
		e.v = lstv[j * NM + k];
		const int2 o = lstt[j * NM + k];
		e.t8 = o.x;
		e.tt = o.y;
		return e;
	};

	// per-section cycle stamps (tools/stamps.py): compiled in only with -DNBL_EMS_STAMPS -- the accumulators
	// live in SGPRs for the whole kernel and push the selection code into SGPR spills
#ifdef NBL_EMS_STAMPS
	unsigned long long st_t0 = 0, st_acc[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	const bool st_on = (w.stamps != nullptr) && ((blockIdx.x & 63) == 0);
#define STAMP(i) do { if (st_on) { unsigned long long t1_ = clock64(); st_acc[i] += t1_ - st_t0; st_t0 = t1_; } } while (0)
#define STAMP_COUNT(i) do { if (st_on) st_acc[i]++; } while (0)
	if (st_on) st_t0 = clock64();
#elif defined(NBL_EMS_MARKS)
	// static instruction budget (tools/isa_budget.py): a comment line in the ISA at every section boundary
#define STAMP(i) asm volatile("; NBLMARK " #i)
#define STAMP_COUNT(i) do { } while (0)
#else
#define STAMP(i) do { } while (0)
#define STAMP_COUNT(i) do { } while (0)
#endif

	const double *V = w.v2c + (size_t)b * g.E * Q;
	double *C = w.c2v + ((size_t)b * g.E + c0) * Q;

	// byte offsets 8 * (h_j * a) of the lane's four variable-domain symbols in a check-domain vector, 16 bits each
	uint2 toff[4];
#define TOFF(j, i) ((int)((((i) & 2) ? toff[j].y : toff[j].x) >> (16 * ((i) & 1))) & 0xffff)
#pragma unroll
	for (int j = 0; j < 4; j++) toff[j] = ((const uint2 *)g.ems_toff)[(size_t)(c0 + j) * 64 + lane];

	// ---- load the four incoming vectors (two 16-byte loads each) ----------------------------------------------------
	double v[4][4];
	if (!FUSED) {
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const double2 *src = (const double2 *)(V + (size_t)g.c_epos[c0 + j] * Q);
			double2 d0 = src[lane], d1 = src[64 + lane];
			v[j][0] = (lane == 0) ? 0.0 : d0.x;
			v[j][1] = d0.y;
			v[j][2] = d1.x;
			v[j][3] = d1.y;
		}
	} else {
		const double *Cp = w.c2v_prev + (w.c2v_prev_shared ? (size_t)0 : (size_t)b * g.E * Q); // (iteration 1: one shared block of zeros)
		// issue every load of the four edges before anything consumes one (24 x 16 B per lane in flight)
		double2 l0[4], l1[4], a0[4], a1[4], b0[4], b1[4];
		int nvar[4], eidx[4];
		bool ownA[4];
		// (one 64-byte row per check holds what the chain c_var / c_epos -> voff -> v_cpos would deliver after three dependent
		// scalar loads: the first vector load of the wave is issued that much earlier)
		const int *row = g.dv2_row + (size_t)m * 16;
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const int n = row[j], cpA = row[4 + j], cpB = row[8 + j], e = row[12 + j] & 0x7fffffff;
#if NBL_WHATIF == 11 // diagnostic (wrong results): every wave reads the channel vectors of codewords 0..7 (L2 hits instead of HBM)
			const double2 *pl = (const double2 *)(w.Lch + ((size_t)(b & 7) * g.N + n) * Q);
#else
			const double2 *pl = (const double2 *)(w.Lch + ((size_t)b * g.N + n) * Q);
#endif
			const double2 *pa = (const double2 *)(Cp + (size_t)cpA * Q);
			const double2 *pb = (const double2 *)(Cp + (size_t)cpB * Q);
#if NBL_WHATIF == 4 // diagnostic (wrong results): no HBM reads, inputs made up from the addresses
			l0[j].x = (double)(long long)(pl + lane) * -1e-9; l0[j].y = l0[j].x * 1.5; l1[j].x = l0[j].x * 0.7; l1[j].y = l0[j].x * 0.3;
			a0[j].x = (double)(long long)(pa + lane) * -1e-9; a0[j].y = a0[j].x * 1.25; a1[j].x = a0[j].x * 0.6; a1[j].y = a0[j].x * 0.2;
			b0[j].x = (double)(long long)(pb + lane) * -1e-9; b0[j].y = b0[j].x * 1.125; b1[j].x = b0[j].x * 0.9; b1[j].y = b0[j].x * 0.4;
#else
			l0[j] = pl[lane]; l1[j] = pl[64 + lane];
			a0[j] = pa[lane]; a1[j] = pa[64 + lane];
			b0[j] = pb[lane]; b1[j] = pb[64 + lane];
#endif
			nvar[j] = n; eidx[j] = e;
			ownA[j] = row[12 + j] < 0; // this check is the variable's first edge
		}
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const int n = nvar[j], e = eidx[j];
			double post[4] = {(l0[j].x + a0[j].x) + b0[j].x, (l0[j].y + a0[j].y) + b0[j].y, (l1[j].x + a1[j].x) + b1[j].x,
			                  (l1[j].y + a1[j].y) + b1[j].y};
			// (ownA is wave-uniform: a branch, not eight per-lane selects)
			if (ownA[j]) {
				v[j][0] = post[0] - a0[j].x; v[j][1] = post[1] - a0[j].y; v[j][2] = post[2] - a1[j].x; v[j][3] = post[3] - a1[j].y;
			} else {
				asm volatile(""); // (keeps the arms apart: hipcc folds this diamond back into selects otherwise)
				v[j][0] = post[0] - b0[j].x; v[j][1] = post[1] - b0[j].y; v[j][2] = post[2] - b1[j].x; v[j][3] = post[3] - b1[j].y;
			}
			if (lane == 0) v[j][0] = 0.0;
			if (ownA[j]) {
				// hard decision (DecideLLRVector :1542-1562): lowest symbol among the maxima, 0 unless the maximum is positive
				if (lane == 0) post[0] = 0.0;
				const double locp = dmax(dmax(post[0], post[1]), dmax(post[2], post[3]));
				const int kp = key32(locp);
				const double pm = wave_max_exact(locp, kp, wave_max_i32(kp));
				uint64_t hit[4];
#pragma unroll
				for (int i = 0; i < 4; i++) hit[i] = __ballot(post[i] == pm);
				const int arg = lowest_sym(hit);
				if (lane == 0) w.dec[(size_t)b * g.N + n] = (pm > 0.0) ? arg : 0;
				if (w.post) {
					double2 *pp = (double2 *)(w.post + ((size_t)b * g.N + n) * Q);
					double2 p01, p23;
					p01.x = post[0]; p01.y = post[1]; p23.x = post[2]; p23.y = post[3];
					pp[lane] = p01;
					pp[64 + lane] = p23;
				}
			}
			if (w.store_v2c) {
				double2 *pv = (double2 *)(w.v2c + ((size_t)b * g.E + e) * Q);
				double2 v01, v23;
				v01.x = v[j][0]; v01.y = v[j][1]; v23.x = v[j][2]; v23.y = v[j][3];
				pv[lane] = v01;
				pv[64 + lane] = v23;
			}
		}
	}
	STAMP(0);

	// ---- rank 0 of every edge: value m_j, check-domain symbol z_j (highest symbol among equal maxima, :1731) -------
	double mtop[4], lmin[4];
	double sec[4] = {0, 0, 0, 0}; // an upper bound of the best value beside rank 0 (for the bounds in front of the gathers)
	int ztop[4];
#pragma unroll
	for (int j = 0; j < 4; j++) {
		const double loc = dmax(dmax(v[j][0], v[j][1]), dmax(v[j][2], v[j][3]));
		const int k = key32(loc);
		mtop[j] = wave_max_exact(loc, k, wave_max_i32(k));
		// 64 distinct entries are >= the smallest lane maximum, so the nm-th best (nm <= 64) is too; the key of that minimum is
		// taken one float step down (rounding may have gone up) -- any lower bound will do
		// (a float infinity must not become the bound: the key is clamped on the scalar unit, below the key of FLT_MAX)
		constexpr int KEY_3E38 = 0x7f61b1e6; // key32(3.0e38)
		const int kmin = wave_min_i32(k);
		const double lf = (double)unkey32(kmin < KEY_3E38 ? kmin : KEY_3E38);
		lmin[j] = ((lf - (mtop[j] - lf) * 0x1p-20) - __builtin_fabs(lf) * 0x1p-22) - 0x1p-120; // (margins: float rounding, rcp in the bucket scale)
		uint64_t eqm[4];
#pragma unroll
		for (int i = 0; i < 4; i++) eqm[i] = __ballot(v[j][i] == mtop[j]);
		const int topa = highest_sym(eqm); // variable-domain symbol of rank 0
		const int tl = (topa & 127) >> 1;
		const unsigned twl = (unsigned)__builtin_amdgcn_readlane((int)toff[j].x, tl), twh = (unsigned)__builtin_amdgcn_readlane((int)toff[j].y, tl);
		const unsigned tw = (topa & 128) ? twh : twl;
		ztop[j] = (int)((tw >> (16 * (topa & 1))) & 0xffffu) >> 3;
#if NBL_EMS_PRUNE
		if (tryf) {
			// the best value beside rank 0: every lane's maximum, the lane that holds rank 0 with that slot left out; on float keys,
			// rounded up by more than the rounding (an upper bound is all the pruning needs)
			const int ts = (topa & 1) + 2 * (topa >> 7); // slot of rank 0 (wave-uniform)
			// (selects, not v[j][ts]: a run-time index would put the vector into scratch memory)
			const double mate = (ts == 0) ? v[j][1] : (ts == 1) ? v[j][0] : (ts == 2) ? v[j][3] : v[j][2]; // the other value of rank 0's 16-byte pair
			const double far = (ts & 2) ? dmax(v[j][0], v[j][1]) : dmax(v[j][2], v[j][3]);                 // the other pair
			const double loc2 = (lane == tl) ? dmax(mate, far) : loc;
			const double sf = (double)unkey32(wave_max_i32(key32(loc2)));
			sec[j] = uniform_f64((sf + __builtin_fabs(sf) * 0x1p-22) + 0x1p-120);
		}
#endif
	}
	STAMP(1);

	// ---- conf(q,1): one edge deviates to any symbol, the others stay at rank 0 (:894).  The staging of edge j in the check
	// domain also brings the vector back as check-domain registers: from here on slot i of every edge is symbol sym_of(lane, i)
	double S[4][4];
#pragma unroll
	for (int x = 0; x < 4; x++)
#pragma unroll
		for (int i = 0; i < 4; i++) S[x][i] = -NBL_DBL_MAX;
	const int ztot = ztop[0] ^ ztop[1] ^ ztop[2] ^ ztop[3];
	const int lane16 = lane << 4;
#pragma unroll
	for (int j = 0; j < 4; j++) {
		char *const Uj = (char *)(j == 1 ? B1 : j == 2 ? B2 : B0); // B0, B1, B2, B0: no buffer is rewritten while reads of it are in flight
		WSYNC();
#pragma unroll
		for (int i = 0; i < 4; i++) *(double *)(Uj + TOFF(j, i)) = v[j][i];
		WSYNC();
#pragma unroll
		for (int x = 0; x < 4; x++) {
			if (x == j) continue;
			const int shift = ztot ^ ztop[x] ^ ztop[j];
			// U[s ^ shift] for the lane's four symbols: two 16-byte reads (the XOR permutes 16-byte slots; bit 0 swaps inside a
			// slot -- a wave-uniform property, resolved by a branch rather than by eight selects)
			const int ad = lane16 ^ ((shift & 0xFE) << 3);
			const double2 ra = *(const double2 *)(Uj + ad), rb = *(const double2 *)(Uj + (ad ^ 1024));
			auto upd = [&](double u0, double u1, double u2, double u3) {
				const double uu[4] = {u0, u1, u2, u3};
#pragma unroll
				for (int i = 0; i < 4; i++) {
					double acc = 0.0;
#pragma unroll
					for (int o = 0; o < 4; o++) {
						if (o == x) continue;
						acc = acc + ((o == j) ? uu[i] : mtop[o]);
					}
					S[x][i] = dmax(S[x][i], acc);
				}
			};
			if (shift & 1) upd(ra.y, ra.x, rb.y, rb.x);
			else upd(ra.x, ra.y, rb.x, rb.y);
		}
		if (NC >= 2) {
			const double2 d0 = *(const double2 *)(Uj + lane16), d1 = *(const double2 *)(Uj + lane16 + 1024);
			v[j][0] = d0.x; v[j][1] = d0.y; v[j][2] = d1.x; v[j][3] = d1.y;
		}
	}
	STAMP(5);

	// ---- top-nm selection (check-domain values) ----------------------------------------------------------------------
	// (a) a 256-bucket histogram of [lmin, mtop] (LDS atomics + DPP prefix sum) finds the bucket that holds the nm-th best
	// value -- bucketing only has to be monotone, so its arithmetic needs no care; (b) a ballot quickselect, restricted to
	// that bucket, finds the cut exactly; ties at the cut are resolved by symbol, higher first (:1731).
	// list layout per edge: the members of slot 0, of slot 2 (even symbols, bit 7 clear / set), of slot 1, of slot 3 (odd symbols);
	// nA = end of the slot-0 run, n0 = end of the even symbols, nC = end of the slot-1 run
	int n0[4] = {0, 0, 0, 0}, nA[4] = {0, 0, 0, 0}, nC[4] = {0, 0, 0, 0};
	int cntl[4] = {nmr, nmr, nmr, nmr}; // entries per list
	// Short lists.  Output x already holds conf(q,1): S[x][s] >= flo[x] for every s.  A configuration for output x that
	// conf(q,1) does not already contain has two or more deviating edges, so beside any one of its entries (value v, edge a) the other
	// two edges b, c contribute at most p_bc = max(m_b + sec_c, sec_b + m_c) (sec = best value beside rank 0): its sum is at most
	// v + p_bc, up to rounding.  An entry with v + p_bc < flo[x] for every output x != a therefore cannot change anything, in any
	// role (pair operand or gather entry): thr[a] = min_x (flo[x] - p_bc), lowered by far more than the rounding of the sums.  When
	// the entries at or above their thresholds number <= nm on all four edges, they ARE the useful part of the top-nm lists: the lists
	// are built from one compare per symbol, without histogram and cut search, and every later loop runs over the short lists.  On a
	// converged codeword (one dominant symbol per edge) that leaves a handful of entries.  Otherwise the full selection runs as before.
	bool fast = false;
	double thr[4] = {0, 0, 0, 0};
#if NBL_EMS_PRUNE
	if (tryf) {
		double flo[4]; // lower bounds of min_s S[x][s]
#pragma unroll
		for (int x = 0; x < 4; x++) {
			const double lf = (double)unkey32(wave_min_i32(key32(dmin(dmin(S[x][0], S[x][1]), dmin(S[x][2], S[x][3])))));
			flo[x] = uniform_f64((lf - __builtin_fabs(lf) * 0x1p-22) - 0x1p-120);
		}
		auto pb = [&](int b, int c) { return dmax(mtop[b] + sec[c], sec[b] + mtop[c]); };
		const double p01 = pb(0, 1), p02 = pb(0, 2), p03 = pb(0, 3), p12 = pb(1, 2), p13 = pb(1, 3), p23 = pb(2, 3);
		auto low = [&](double f, double p) { // f - p, lowered
			const double t = f - p;
			return (t - (__builtin_fabs(f) + __builtin_fabs(p)) * 0x1p-40) - 0x1p-100;
		};
		thr[0] = uniform_f64(dmin(dmin(low(flo[1], p23), low(flo[2], p13)), low(flo[3], p12)));
		thr[1] = uniform_f64(dmin(dmin(low(flo[0], p23), low(flo[2], p03)), low(flo[3], p02)));
		thr[2] = uniform_f64(dmin(dmin(low(flo[0], p13), low(flo[1], p03)), low(flo[3], p01)));
		thr[3] = uniform_f64(dmin(dmin(low(flo[0], p12), low(flo[1], p02)), low(flo[2], p01)));
		fast = true;
#pragma unroll
		for (int j = 0; j < 4; j++) {
			int c = 0;
#pragma unroll
			for (int i = 0; i < 4; i++) c += __popcll(__ballot(!(v[j][i] < thr[j]))); // (the masks themselves are formed again at the compaction: 64 scalar registers otherwise)
			cntl[j] = c;
			fast = fast && c <= nmr;
		}
		if (!fast) cntl[0] = cntl[1] = cntl[2] = cntl[3] = nmr;
#ifdef NBL_EMS_STAMPS
		if (st_on) { st_acc[13] += fast ? 1 : 0; st_acc[14] += cntl[0] + cntl[1] + cntl[2] + cntl[3]; } // short-list checks, list entries
#endif
	}
#endif
	if (NC >= 2) { // conf(nm,1) needs no lists: it is contained in conf(q,1)
		// list members (wave-uniform masks per slot) -> the list image: slots 0, 2 hold even symbols, slots 1, 3 odd ones
		auto compact = [&](int j, const uint64_t (&member)[4]) {
			const int c0n = __popcll(member[0]), c2n = __popcll(member[2]), c1n = __popcll(member[1]);
			const int ne = c0n + c2n;
			n0[j] = ne;
			nA[j] = c0n;
			nC[j] = ne + c1n;
			const int base[4] = {j * NM, j * NM + ne, j * NM + c0n, j * NM + ne + c1n};
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const int pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(member[i] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)member[i], base[i]));
				if (__builtin_amdgcn_inverse_ballot_w64(member[i])) {
					// list entry fields of the lane's slot: symbol and gather offset (formed here: two more registers per slot would be
					// live across the whole selection otherwise)
					lstv[pos] = v[j][i];
					lstt[pos] = make_int2(sym_of(lane, i) << 3, (sym_of(lane, i) & 0x7E) << 3);
				}
			}
		};
		if (VARNM || fast) { // list entries beyond the nm selected ones (run-time nm, short lists): value -inf, pairwise different symbols (a scatter of several lanes
			// to ONE address is serialised by the LDS)
			WSYNC();
#pragma unroll
			for (int j = 0; j < 4; j++) {
				lstv[j * NM + (lane & (NM - 1))] = NBL_NEG_INF;
				lstt[j * NM + (lane & (NM - 1))] = make_int2((lane & (NM - 1)) << 3, 0);
			}
			WSYNC();
		}
		if (fast) {
#pragma unroll
			for (int j = 0; j < 4; j++) {
				uint64_t member[4];
#pragma unroll
				for (int i = 0; i < 4; i++) member[i] = __ballot(!(v[j][i] < thr[j]));
				compact(j, member);
			}
		} else {
			int bk[4][4];
			int *H = (int *)B0; // [4 edges][256 buckets] (spans B0 and B1); lane l reads buckets 4l .. 4l+3 of every edge
			int one;            // the increment of the sixteen histogram atomics, held in ONE register (hipcc re-materialises a literal 1 per atomic)
			asm volatile("v_mov_b32 %0, 1" : "=v"(one));
			WSYNC();
			{
				int4 z4 = {0, 0, 0, 0};
#pragma unroll
				for (int i = 0; i < 4; i++) ((int4 *)H)[i * 64 + lane] = z4;
			}
			WSYNC();
#pragma unroll
			for (int j = 0; j < 4; j++) {
				// bucketing only has to be monotone and the same in every lane: the hardware reciprocal will do (an IEEE division is
				// ~14 instructions).  lmin lies strictly below every lane maximum, so range > 0, entries >= the smallest lane maximum land
				// in buckets 0..255 and everything below lmin is clamped to 256 = "not counted"
				const double scale = 256.0 * __builtin_amdgcn_rcp(mtop[j] - lmin[j]);
#pragma unroll
				for (int i = 0; i < 4; i++) {
					// v_cvt_i32_f64 saturates (and maps NaN to 0): no FP64 minimum against a constant that would have to be materialised in
					// a register pair.  A bucket >= 256 lies below the lower bound and is not counted; every later use compares bk against
					// a bucket <= 255, so it needs no clamp either
					int bi;
					asm("v_cvt_i32_f64 %0, %1" : "=v"(bi) : "v"((mtop[j] - v[j][i]) * scale));
					bk[j][i] = bi;
					if (bi < 256) __hip_atomic_fetch_add(&H[j * 256 + bi], one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
			}
			WSYNC();
			int hc[4][4]; // [bucket slot of this lane][edge]
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const int4 h4 = ((int4 *)H)[j * 64 + lane];
				hc[0][j] = h4.x; hc[1][j] = h4.y; hc[2][j] = h4.z; hc[3][j] = h4.w;
			}
			STAMP(2);
			// per edge: locate the cut bucket, settle the members, compact them into the list image
			// [even-symbol group | odd-symbol group]
#pragma unroll
			for (int j = 0; j < 4; j++) {
				uint64_t member[4];
				const int tot = hc[0][j] + hc[1][j] + hc[2][j] + hc[3][j];
				const int cum = wave_scan_add(tot);
				const uint64_t reach = __ballot(cum >= nmr);
				const int lstar = reach ? __builtin_ctzll(reach) : 63;
				// the cut lane's four bucket counts and its running total: which of its buckets holds the nm-th entry is scalar work
				const int cl = __builtin_amdgcn_readlane(cum, lstar);
				const int q1 = __builtin_amdgcn_readlane(hc[1][j], lstar),
				          q2 = __builtin_amdgcn_readlane(hc[2][j], lstar), q3 = __builtin_amdgcn_readlane(hc[3][j], lstar);
				const int s0 = cl - q3 - q2 - q1, s1 = cl - q3 - q2, s2 = cl - q3;
				const int bsel = (s0 >= nmr) ? 0 : (s1 >= nmr) ? 1 : (s2 >= nmr) ? 2 : 3;
				const int upto = (s0 >= nmr) ? s0 : (s1 >= nmr) ? s1 : (s2 >= nmr) ? s2 : cl; // entries up to and including the cut bucket
				const int bstar = 4 * lstar + bsel;
				const bool exact = reach && upto == nmr; // the cut bucket ends exactly at the nm-th entry
				if (exact) {
#pragma unroll
					for (int i = 0; i < 4; i++) member[i] = __ballot(bk[j][i] <= bstar);
				} else {
					// (no bucket reaches nm -- cannot happen with finite inputs: every entry is a candidate)
					STAMP_COUNT(10);
					SelState ss;
#pragma unroll
					for (int i = 0; i < 4; i++) { ss.cand[i] = reach ? __ballot(bk[j][i] == bstar) : ~0ull; ss.gt[i] = ss.eq[i] = 0; }
					ss.done = 0;
					for (int guard = 0; guard < 300 && !ss.done; guard++) {
						select_step(v[j], nmr, ss);
						STAMP_COUNT(9);
					}
					finish_members(ss, nmr, member, g.c_hinv[c0 + j], g.poly, lane);
				}
				compact(j, member);
			}
		}
	}
	WSYNC();
	STAMP(4);

	// ---- conf(nm,nc): truncated max-plus convolutions --------------------------------------------------------------
	// dst[t_a ^ t_b ^ sxor] = max over the nm x nm entry pairs of (v_a + bias) + v_b   (bias = 0: plain pair convolution)
	constexpr int PER = 64 >> LOGNM, ROUNDS = (NM << LOGNM) >> 6;
	const int rounds = VARNM ? (nmr + PER - 1) >> (6 - LOGNM) : ROUNDS; // (run-time nm: the moving operand stops at the nm-th entry)
	// list position of the lane's fixed operand: neighbouring lanes alternate between the front (even symbols) and the back (odd
	// symbols) of the list -- a run of 16 even symbols would only reach half of the LDS banks of the scatter
	const int apos = (lane & 1) ? (NM - 1) - ((lane & (NM - 1)) >> 1) : ((lane & (NM - 1)) >> 1);
	auto pair_scatter = [&](double *dst, int ja, int jb, double bias, int sxor) __attribute__((always_inline)) {
		WSYNC();
		double2 ninf;
		ninf.x = NBL_NEG_INF;
		ninf.y = NBL_NEG_INF;
		((double2 *)dst)[lane] = ninf;
		((double2 *)dst)[64 + lane] = ninf;
		ListEnt ea = list_at(ja, apos);
		ea.v = ea.v + bias;
		ea.t8 ^= sxor << 3;
		WSYNC();
		if (fast) { // short lists: the moving operand stops at its last entry
			for (int it = 0; it < ((cntl[jb] + PER - 1) >> (6 - LOGNM)); it++) {
				const ListEnt eb = list_at(jb, it * PER + (lane >> LOGNM));
				__hip_atomic_fetch_max((double *)((char *)dst + (ea.t8 ^ eb.t8)), ea.v + eb.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		} else
#pragma unroll
		for (int it = 0; it < rounds; it++) {
			const ListEnt eb = list_at(jb, it * PER + (lane >> LOGNM));
			__hip_atomic_fetch_max((double *)((char *)dst + (ea.t8 ^ eb.t8)), ea.v + eb.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		WSYNC();
	};
	// same, accumulating into dst without clearing it
	auto pair_scatter_more = [&](double *dst, int ja, int jb, double bias, int sxor) __attribute__((always_inline)) {
		ListEnt ea = list_at(ja, apos);
		ea.v = ea.v + bias;
		ea.t8 ^= sxor << 3;
		if (fast) { // short lists: the moving operand stops at its last entry
			for (int it = 0; it < ((cntl[jb] + PER - 1) >> (6 - LOGNM)); it++) {
				const ListEnt eb = list_at(jb, it * PER + (lane >> LOGNM));
				__hip_atomic_fetch_max((double *)((char *)dst + (ea.t8 ^ eb.t8)), ea.v + eb.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		} else
#pragma unroll
		for (int it = 0; it < rounds; it++) {
			const ListEnt eb = list_at(jb, it * PER + (lane >> LOGNM));
			__hip_atomic_fetch_max((double *)((char *)dst + (ea.t8 ^ eb.t8)), ea.v + eb.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		WSYNC();
	};
	// Sout[s] = max(Sout[s], src[s ^ sx] + add) for the lane's four symbols
	auto fold = [&](const double *src, int sx, double add, double (&Sout)[4]) {
		const char *Sb = (const char *)src;
		const int ad = lane16 ^ ((sx & 0xFE) << 3);
		const double2 ra = *(const double2 *)(Sb + ad), rb = *(const double2 *)(Sb + (ad ^ 1024));
		const bool sw = sx & 1;
		Sout[0] = dmax(Sout[0], (sw ? ra.y : ra.x) + add);
		Sout[1] = dmax(Sout[1], (sw ? ra.x : ra.y) + add);
		Sout[2] = dmax(Sout[2], (sw ? rb.y : rb.x) + add);
		Sout[3] = dmax(Sout[3], (sw ? rb.x : rb.y) + add);
	};
	// conf(nm,2) for output x with others o1 < o2 < o3 (B0 already holds o1 (+) o2): one of the three stays at rank 0 (:1769)
	auto conf_nc2 = [&](int x, int o1, int o2, int o3) {
		fold(B0, ztop[o3], mtop[o3], S[x]);                                   // (v1 + v2) + m3
		pair_scatter(B1, o2, o3, mtop[o1], ztop[o1]);                         // (m1 + v2) + v3
		pair_scatter_more(B1, o1, o3, mtop[o2], ztop[o2]);                    // (v1 + m2) + v3
		{
			const double2 ra = ((const double2 *)B1)[lane], rb = ((const double2 *)B1)[64 + lane];
			S[x][0] = dmax(S[x][0], ra.x); S[x][1] = dmax(S[x][1], ra.y);
			S[x][2] = dmax(S[x][2], rb.x); S[x][3] = dmax(S[x][3], rb.y);
		}
	};
	// NP max-plus gather convolutions over ONE list: acc[p][s] = max_k  P_p[s ^ t_k] + v_k, P_p = B0 + p * Q
	auto gather_conv = [&](int jc, auto np_tag, double (&acc)[decltype(np_tag)::value][4]) {
		constexpr int NP = decltype(np_tag)::value;
		constexpr int UN = 4; // entries per trip (twelve 16-byte gathers in flight in the three-way loop)
		const char *Pb = (const char *)B0;
		// four runs of entries, by bit 0 and bit 7 of the entry's symbol t (both fixed by the slot the entry came from): bit 0 swaps
		// the values inside a 16-byte slot (SW), bit 7 swaps the lower and the upper half of the vector (HI) -- as compile-time
		// properties of the run neither costs an instruction, and the second gather address is the first plus 1024 (an offset
		// field, no address arithmetic).  UN entries per trip, then the remainder one by one (no padding entries)
		auto body = [&](const ListEnt &en, auto swapped, auto upper) {
			constexpr bool SW = decltype(swapped)::value, HI = decltype(upper)::value;
#if NBL_WHATIF == 10 // diagnostic (wrong results): every lane gathers from ONE address (same instructions, one LDS pass per read)
			const int ad = en.tt & 1008;
#else
			const int ad = lane16 ^ en.tt; // (tt holds bits 1..6 of t only)
#endif
			double2 ra[NP], rb[NP];
#pragma unroll
			for (int p = 0; p < NP; p++) {
#if NBL_WHATIF == 1 // diagnostic (wrong results): the gathers without their LDS reads
				ra[p].x = acc[p][1]; ra[p].y = acc[p][0]; rb[p].x = acc[p][3]; rb[p].y = acc[p][2];
				asm volatile("" : "+v"(ra[p].x), "+v"(ra[p].y), "+v"(rb[p].x), "+v"(rb[p].y) : "v"(ad));
#else
				ra[p] = *(const double2 *)(Pb + p * (Q * 8) + ad + (HI ? 1024 : 0));
				rb[p] = *(const double2 *)(Pb + p * (Q * 8) + ad + (HI ? 0 : 1024));
#endif
			}
#pragma unroll
			for (int p = 0; p < NP; p++) {
#if NBL_WHATIF == 2 // diagnostic (wrong results): the gathers without their additions
				acc[p][0] = dmax(acc[p][0], (SW ? ra[p].y : ra[p].x));
				acc[p][1] = dmax(acc[p][1], (SW ? ra[p].x : ra[p].y));
				acc[p][2] = dmax(acc[p][2], (SW ? rb[p].y : rb[p].x));
				acc[p][3] = dmax(acc[p][3], (SW ? rb[p].x : rb[p].y));
#else
				acc[p][0] = dmax(acc[p][0], (SW ? ra[p].y : ra[p].x) + en.v);
				acc[p][1] = dmax(acc[p][1], (SW ? ra[p].x : ra[p].y) + en.v);
				acc[p][2] = dmax(acc[p][2], (SW ? rb[p].y : rb[p].x) + en.v);
				acc[p][3] = dmax(acc[p][3], (SW ? rb[p].x : rb[p].y) + en.v);
#endif
			}
		};
		// the list lives in the registers of lanes 0..NM-1 (one entry each); an entry reaches the whole wave through v_readlane
		// (SGPR operands of the adds) -- the LDS port is the busier resource of this kernel, a broadcast read per entry costs more
		const double myv = lstv[jc * NM + (lane & (NM - 1))];
		const int mytt = lstt[jc * NM + (lane & (NM - 1))].y;
		auto entry = [&](int k) {
			ListEnt e;
			e.v = read_lane_f64(myv, k);
			e.tt = __builtin_amdgcn_readlane(mytt, k);
			e.t8 = 0;
			return e;
		};
		auto run = [&](int k0, int k1, auto swapped, auto upper) {
			int k = k0;
			for (; k + UN <= k1; k += UN) {
				STAMP_COUNT(11);
				ListEnt en[UN];
#pragma unroll
				for (int u = 0; u < UN; u++) en[u] = entry(k + u);
#pragma unroll
				for (int u = 0; u < UN; u++) body(en[u], swapped, upper);
			}
			for (; k < k1; k++) { STAMP_COUNT(12); body(entry(k), swapped, upper); }
		};
		run(0, nA[jc], std::false_type{}, std::false_type{});
		run(nA[jc], n0[jc], std::false_type{}, std::true_type{});
		run(n0[jc], nC[jc], std::true_type{}, std::false_type{});
		run(nC[jc], cntl[jc], std::true_type{}, std::true_type{});
	};
	// c2v[a] = shape(S[h_x a] - S[0]) (:899-916): the output goes back to the variable domain through LDS
	auto emit_stage = [&](int x, double *Sx) {
		double2 s01, s23;
		s01.x = S[x][0]; s01.y = S[x][1]; s23.x = S[x][2]; s23.y = S[x][3];
		((double2 *)Sx)[lane] = s01;
		((double2 *)Sx)[64 + lane] = s23;
	};
	auto emit_store = [&](int x, const double *Sx) {
		const double s0 = Sx[0];
		double y[4];
#pragma unroll
		for (int i = 0; i < 4; i++) y[i] = shape_llr(*(const double *)((const char *)Sx + TOFF(x, i)) - s0, r.factor, r.offset);
		if (lane == 0) y[0] = 0.0;
		double2 o01, o23;
		o01.x = y[0]; o01.y = y[1]; o23.x = y[2]; o23.y = y[3];
		double2 *dst = (double2 *)(C + (size_t)x * Q);
		dst[lane] = o01;
		dst[64 + lane] = o23;
	};

	if (NC >= 3) {
		// the three pair convolutions in one pass: B0 = e0 (+) e1, B1 = e0 (+) e2, B2 = e1 (+) e2
		{
			double2 ninf;
			ninf.x = NBL_NEG_INF;
			ninf.y = NBL_NEG_INF;
#pragma unroll
			for (int i = 0; i < 6; i++) ((double2 *)B0)[i * 64 + lane] = ninf;
			const ListEnt ea0 = list_at(0, apos), ea1 = list_at(1, apos);
			WSYNC();
			if (fast) { // short lists: the moving operands stop at the last entry of the longer one (beyond a list's end: -inf entries)
				const int mc = cntl[1] > cntl[2] ? cntl[1] : cntl[2];
				for (int it = 0; it < ((mc + PER - 1) >> (6 - LOGNM)); it++) {
					const ListEnt eb1 = list_at(1, it * PER + (lane >> LOGNM)), eb2 = list_at(2, it * PER + (lane >> LOGNM));
					__hip_atomic_fetch_max((double *)((char *)B0 + (ea0.t8 ^ eb1.t8)), ea0.v + eb1.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					__hip_atomic_fetch_max((double *)((char *)B1 + (ea0.t8 ^ eb2.t8)), ea0.v + eb2.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					__hip_atomic_fetch_max((double *)((char *)B2 + (ea1.t8 ^ eb2.t8)), ea1.v + eb2.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
			} else
#pragma unroll
			for (int it = 0; it < rounds; it++) {
				const ListEnt eb1 = list_at(1, it * PER + (lane >> LOGNM)), eb2 = list_at(2, it * PER + (lane >> LOGNM));
#if NBL_WHATIF == 3 // diagnostic (wrong results): the pair convolutions without their LDS atomics
				double keep = (ea0.v + eb1.v) + ((ea0.v + eb2.v) + (ea1.v + eb2.v));
				int keepa = (ea0.t8 ^ eb1.t8) + (ea0.t8 ^ eb2.t8) + (ea1.t8 ^ eb2.t8);
				asm volatile("" : : "v"(keep), "v"(keepa));
#elif NBL_WHATIF == 7 // diagnostic (wrong results): plain stores to the same addresses instead of the atomic maxima
				*(volatile double *)((char *)B0 + (ea0.t8 ^ eb1.t8)) = ea0.v + eb1.v;
				*(volatile double *)((char *)B1 + (ea0.t8 ^ eb2.t8)) = ea0.v + eb2.v;
				*(volatile double *)((char *)B2 + (ea1.t8 ^ eb2.t8)) = ea1.v + eb2.v;
#else
				__hip_atomic_fetch_max((double *)((char *)B0 + (ea0.t8 ^ eb1.t8)), ea0.v + eb1.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				__hip_atomic_fetch_max((double *)((char *)B1 + (ea0.t8 ^ eb2.t8)), ea0.v + eb2.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				__hip_atomic_fetch_max((double *)((char *)B2 + (ea1.t8 ^ eb2.t8)), ea1.v + eb2.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
			}
			WSYNC();
		}
		STAMP(6);
		// output 3 = (e0 (+) e1) (+) e2; outputs 2, 1, 0 = {e0 (+) e1, e0 (+) e2, e1 (+) e2} (+) e3 share the loop over list 3
		{
			double a3[1][4], a210[3][4];
#pragma unroll
			for (int i = 0; i < 4; i++) { a3[0][i] = S[3][i]; a210[0][i] = S[2][i]; a210[1][i] = S[1][i]; a210[2][i] = S[0][i]; }
			gather_conv(2, std::integral_constant<int, 1>{}, a3);
			gather_conv(3, std::integral_constant<int, 3>{}, a210);
#pragma unroll
			for (int i = 0; i < 4; i++) { S[3][i] = a3[0][i]; S[2][i] = a210[0][i]; S[1][i] = a210[1][i]; S[0][i] = a210[2][i]; }
		}
		STAMP(7);
	} else if (NC == 2) {
		pair_scatter(B0, 0, 1, 0.0, 0);
		conf_nc2(3, 0, 1, 2);
		conf_nc2(2, 0, 1, 3);
		pair_scatter(B0, 0, 2, 0.0, 0);
		conf_nc2(1, 0, 2, 3);
		pair_scatter(B0, 1, 2, 0.0, 0);
		conf_nc2(0, 1, 2, 3);
		STAMP(6);
	}
	WSYNC();
	emit_stage(3, B0);
	emit_stage(2, B1);
	emit_stage(1, B2);
	WSYNC();
	emit_store(3, B0);
	emit_store(2, B1);
	emit_store(1, B2);
	WSYNC();
	emit_stage(0, B0);
	WSYNC();
	emit_store(0, B0);
	STAMP(8);

#ifdef NBL_EMS_STAMPS
	if (st_on && lane == 0) {
		for (int i = 0; i < 15; i++) atomicAdd(&w.stamps[i], st_acc[i]);
		atomicAdd(&w.stamps[15], 1ull);
	}
#endif
#undef STAMP
#undef STAMP_COUNT
#undef TOFF
}

bool nbl_ems256_applicable(const NblGraphDev &g, bool all_dc4, int nm, int nc)
{
	return g.q == 256 && all_dc4 && nc >= 1 && nm >= 1 && nm <= 64 && g.ems_toff != nullptr; // (fused: g.dv2_row too, see fused_shape)
}

size_t nbl_ems256_lds_bytes(int nm) { return 3 * Q * 8 + (size_t)4 * (nm <= 8 ? 8 : nm <= 16 ? 16 : nm <= 32 ? 32 : 64) * 16; }

template <int NM, bool FUSED>
static void launch_nc(int nc, dim3 grid, dim3 block, hipStream_t st, const NblGraphDev &g, const NblWork &w, const NblRun &r)
{
	if (nc >= 3) cn_ems_q256_dc4_kernel<NM, FUSED, 3><<<grid, block, 0, st>>>(g, w, r);
	else if (nc == 2) cn_ems_q256_dc4_kernel<NM, FUSED, 2><<<grid, block, 0, st>>>(g, w, r);
	else cn_ems_q256_dc4_kernel<NM, FUSED, 1><<<grid, block, 0, st>>>(g, w, r);
}

hipError_t nbl_launch_cn_ems256(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool fused, hipStream_t st)
{
	dim3 grid(nbl_xcd_grid(r.B, g.M)), block(64);
	if (fused) {
		switch (r.nm) {
		case 8: launch_nc<8, true>(r.nc, grid, block, st, g, w, r); break;
		case 16: launch_nc<16, true>(r.nc, grid, block, st, g, w, r); break;
		case 32: launch_nc<32, true>(r.nc, grid, block, st, g, w, r); break;
		case 64: launch_nc<64, true>(r.nc, grid, block, st, g, w, r); break;
		default: // any other nm <= 64 (nbl_ems256_applicable): the layout of the next power of two
			if (r.nm < 8) launch_nc<-8, true>(r.nc, grid, block, st, g, w, r);
			else if (r.nm < 16) launch_nc<-16, true>(r.nc, grid, block, st, g, w, r);
			else if (r.nm < 32) launch_nc<-32, true>(r.nc, grid, block, st, g, w, r);
			else launch_nc<-64, true>(r.nc, grid, block, st, g, w, r);
			break;
		}
	} else {
		switch (r.nm) {
		case 8: launch_nc<8, false>(r.nc, grid, block, st, g, w, r); break;
		case 16: launch_nc<16, false>(r.nc, grid, block, st, g, w, r); break;
		case 32: launch_nc<32, false>(r.nc, grid, block, st, g, w, r); break;
		case 64: launch_nc<64, false>(r.nc, grid, block, st, g, w, r); break;
		default:
			if (r.nm < 8) launch_nc<-8, false>(r.nc, grid, block, st, g, w, r);
			else if (r.nm < 16) launch_nc<-16, false>(r.nc, grid, block, st, g, w, r);
			else if (r.nm < 32) launch_nc<-32, false>(r.nc, grid, block, st, g, w, r);
			else launch_nc<-64, false>(r.nc, grid, block, st, g, w, r);
			break;
		}
	}
	return hipGetLastError();
}
