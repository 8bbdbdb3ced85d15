// nbldpc_amd/csrc/nbl_cn_ems256.hip -- EMS check node specialised for the headline shape:
//   GF(256), every check of degree 4, any nc >= 1, nm in {8,16,32}.
// (BASELINE configs 2 and 3: divsalar.UNBLDPC.{128.64,512.256}.GF.256, EMS nm=16/32.)
//
// Same arithmetic as cn_ems_kernel<256> (nbl_kernels.hip) -- the generic kernel is the readable statement of the
// algorithm and serves every other shape; tests check both against the oracle.  What differs is the mapping:
//   * lane l owns symbols {2l, 2l+1, 128+2l, 129+2l}: a q-vector is two 16-byte loads/stores per lane, and a
//     max-plus XOR-gather "P[s ^ t]" is two conflict-free ds_read_b128 (natural layout, lane XOR only permutes the
//     16-byte slots inside one 256-byte LDS row).
//   * the four conf(q,1) result vectors stay in registers; 2 KB LDS buffers U (also histogram / S staging) and P plus
//     the packed lists -> 7 KB per wave at nm = 32, four waves per SIMD (VGPR-limited).
//   * the nm-best lists are packed {value, symbol} 16-byte entries, split by bit 0 of the symbol so the pair swap
//     of the gather is resolved by loop structure instead of per-element selects.
//   * top-nm selection without sorting: a 256-bucket histogram (LDS atomics + DPP prefix sum) finds the bucket that
//     holds the nm-th best value; if that bucket ends exactly at the nm-th entry the members are known, otherwise a ballot
//     quickselect inside the bucket finds the exact cut under SortLLRVector's order (value desc, higher symbol first
//     among equals).
#include <hip/hip_runtime.h>
#include <type_traits>
#include "nbl_device.h"
#include "nbl_kernels.h"

namespace {

constexpr int Q = 256;

struct __attribute__((aligned(16))) ListEnt { double v; int t; int tt; };   // tt = (t & 0xFE) << 3: byte offset XOR of the gather

// ---- 64-bit wave max through DPP moves (no LDS) ------------------------------------------------------------------
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

__device__ __forceinline__ double wave_min_f64(double x)
{
	x = dmin(x, dpp_mov_f64<0xB1>(x));
	x = dmin(x, dpp_mov_f64<0x4E>(x));
	x = dmin(x, dpp_mov_f64<0x141>(x));
	x = dmin(x, dpp_mov_f64<0x140>(x));
	x = dmin(x, dpp_mov_f64<0x142, 0xA>(x));
	x = dmin(x, dpp_mov_f64<0x143, 0xC>(x));
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

__device__ __forceinline__ int wave_max_i32(int x)
{
	auto mx = [](int a, int b) { return a > b ? a : b; };
	x = mx(x, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false));
	x = mx(x, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false));
	x = mx(x, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false));
	x = mx(x, __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false));
	x = mx(x, __builtin_amdgcn_update_dpp(x, x, 0x142, 0xA, 0xF, false));
	x = mx(x, __builtin_amdgcn_update_dpp(x, x, 0x143, 0xC, 0xF, false));
	return __builtin_amdgcn_readlane(x, 63);
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

// h * a for this layout: lane bits 0..5 are symbol bits 1..6, slot bit 0 is symbol bit 0, slot bit 1 is symbol bit 7
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
	__device__ __forceinline__ int scalar(int a, int h, int poly) const
	{
		int x = h, acc = 0;
#pragma unroll
		for (int i = 0; i < 8; i++) {
			if ((a >> i) & 1) acc ^= x;
			x <<= 1;
			if (x & Q) x ^= poly;
		}
		return acc;
	}
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

// members of the nm best under SortLLRVector's order (value desc, higher symbol first among equals, NBLDPC.cpp:1731)
__device__ __forceinline__ void finish_members(SelState &s, int nm, uint64_t (&member)[4])
{
	int cgt = 0, ceq = 0;
#pragma unroll
	for (int i = 0; i < 4; i++) { cgt += __popcll(s.gt[i]); ceq += __popcll(s.eq[i]); }
	int need = nm - cgt;
#pragma unroll
	for (int i = 0; i < 4; i++) member[i] = s.gt[i];
	if (need == ceq) {
#pragma unroll
		for (int i = 0; i < 4; i++) member[i] |= s.eq[i];
	} else {
		// ties straddle the cut: take the highest symbols of the tie group (rare)
		uint64_t eq[4] = {s.eq[0], s.eq[1], s.eq[2], s.eq[3]};
		for (; need > 0; need--) {
			int a = highest_sym(eq);
			int i = ((a >> 7) << 1) | (a & 1), l = (a & 127) >> 1;
#pragma unroll
			for (int k = 0; k < 4; k++)
				if (k == i) { member[k] |= 1ull << l; eq[k] &= ~(1ull << l); }
		}
	}
}

} // namespace

// FUSED = true: one launch is a whole flooding iteration for dv = 2 codes.  The variable-node work of the reference
// (a-posteriori sum L_ch + c2v_0 + c2v_1 in that order, hard decision, v2c = L_post - c2v, NBLDPC.cpp:808-823, 848-857) is
// recomputed per edge from the previous iteration's c2v, so v2c never touches HBM; the wave that holds a variable's FIRST
// edge writes its hard decision.  c2v is double-buffered (w.c2v_prev -> w.c2v) because the schedule is flooding.
// NC = min(nc, 3): 3 = no deviation counting needed (every other edge may deviate); 2 = at least one of the three other
// edges stays at rank 0; 1 = conf(nm,1) is a subset of conf(q,1), nothing beyond the closed form is needed.
template <int NM, bool FUSED, int NC>
__global__ __launch_bounds__(64) void cn_ems_q256_dc4_kernel(NblGraphDev g, NblWork w, NblRun r)
{
	constexpr int LOGNM = (NM == 8) ? 3 : (NM == 16) ? 4 : 5;
	constexpr int NMP = NM; // list stride per edge: [even-symbol group | odd-symbol group], back to back
	// static LDS: its base is a compile-time constant, so computed addresses fold into the ds_* offset fields (with `extern
	// __shared__` every computed address pays a v_add with the link-time base)
	__shared__ __attribute__((aligned(16))) char smem[2 * Q * 8 + Q / 2 * 8 + 4 * NM * 16];
	const int lane = lane_id();
	// XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8), each with its own L2.  All checks of
	// a codeword are given to ONE XCD, so the second read of every L_ch / c2v vector (each is used by two checks) can hit
	// that L2.  Speed only: nothing depends on the placement.
	const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
	const int m = idx % g.M, b = (idx / g.M) * 8 + xcd;
	if (b >= r.B) return;
	if (!r.fixed_iters && w.done[b]) return;
	const int c0 = g.coff[m];

	double *U = (double *)smem;              // [256] check-domain copy of one input vector | histogram | S of one output
	double *P = U + Q;                       // [256] pair convolution (U..P together: candidate buffer of the selection)
	// P[256..383] mirrors P[0..127] while a gather runs: the partner half of chunk `ad` is then always at ad + 1024
	ListEnt *lstp = (ListEnt *)(P + Q + Q / 2); // [4][NM]  the nm best of every edge, grouped by symbol bit 0

	// per-section cycle stamps (tools/stamps.py): compiled in only with -DNBL_EMS_STAMPS -- the thirteen 64-bit accumulators
	// live in SGPRs for the whole kernel and push the selection code into SGPR spills
#ifdef NBL_EMS_STAMPS
	unsigned long long st_t0 = 0, st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	const bool st_on = (w.stamps != nullptr) && ((blockIdx.x & 63) == 0);
#define STAMP(i) do { if (st_on) { unsigned long long t1_ = clock64(); st_acc[i] += t1_ - st_t0; st_t0 = t1_; } } while (0)
#define STAMP_COUNT(i) do { if (st_on) st_acc[i]++; } while (0)
	if (st_on) st_t0 = clock64();
#else
#define STAMP(i) do { } while (0)
#define STAMP_COUNT(i) do { } while (0)
#endif

	const double *V = w.v2c + (size_t)b * g.E * Q;
	double *C = w.c2v + ((size_t)b * g.E + c0) * Q;

	// ---- load the four incoming vectors (two 16-byte loads each) and their check-domain symbols ------------------
	double v[4][4];
	unsigned tp[4]; // check-domain symbols h_j * a of the lane's four slots, packed one byte each (keeps 12 VGPRs free:
	                // with 16 separate registers hipcc rematerialises the GF multiplication at every use)
#define TSYM(j, i) ((int)((tp[j] >> (8 * (i))) & 255u))
	int hcoef[4];
	if (!FUSED) {
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const double2 *src = (const double2 *)(V + (size_t)g.c_epos[c0 + j] * Q);
			double2 d0 = src[lane], d1 = src[64 + lane];
			v[j][0] = (lane == 0) ? 0.0 : d0.x;
			v[j][1] = d0.y;
			v[j][2] = d1.x;
			v[j][3] = d1.y;
			hcoef[j] = g.c_h[c0 + j];
		}
	} else {
		const double *Cp = w.c2v_prev + (size_t)b * g.E * Q;
		// issue every load of the four edges before anything consumes one (24 x 16 B per lane in flight)
		double2 l0[4], l1[4], a0[4], a1[4], b0[4], b1[4];
		int nvar[4], eidx[4];
		bool ownA[4];
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const int n = g.c_var[c0 + j], e = g.c_epos[c0 + j], e0 = g.voff[n];
			const int cpA = g.v_cpos[e0], cpB = g.v_cpos[e0 + 1];
			const double2 *pl = (const double2 *)(w.Lch + ((size_t)b * g.N + n) * Q);
			const double2 *pa = (const double2 *)(Cp + (size_t)cpA * Q);
			const double2 *pb = (const double2 *)(Cp + (size_t)cpB * Q);
			l0[j] = pl[lane]; l1[j] = pl[64 + lane];
			a0[j] = pa[lane]; a1[j] = pa[64 + lane];
			b0[j] = pb[lane]; b1[j] = pb[64 + lane];
			nvar[j] = n; eidx[j] = e;
			ownA[j] = (e == e0); // this check is the variable's first edge
			hcoef[j] = g.c_h[c0 + j];
		}
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const int n = nvar[j], e = eidx[j];
			double post[4] = {(l0[j].x + a0[j].x) + b0[j].x, (l0[j].y + a0[j].y) + b0[j].y, (l1[j].x + a1[j].x) + b1[j].x,
			                  (l1[j].y + a1[j].y) + b1[j].y};
			v[j][0] = post[0] - (ownA[j] ? a0[j].x : b0[j].x);
			v[j][1] = post[1] - (ownA[j] ? a0[j].y : b0[j].y);
			v[j][2] = post[2] - (ownA[j] ? a1[j].x : b1[j].x);
			v[j][3] = post[3] - (ownA[j] ? a1[j].y : b1[j].y);
			if (lane == 0) v[j][0] = 0.0;
			if (ownA[j]) {
				// hard decision (DecideLLRVector :1542-1562): lowest symbol among the maxima, 0 unless the maximum is positive
				if (lane == 0) post[0] = 0.0;
				const double pm = wave_max_f64(dmax(dmax(post[0], post[1]), dmax(post[2], post[3])));
				int cand = 0x7fffffff;
#pragma unroll
				for (int i = 3; i >= 0; i--) cand = (post[i] == pm) ? sym_of(lane, i) : cand;
				const int arg = -wave_max_i32(-cand);
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
#pragma unroll
	for (int j = 0; j < 4; j++) {
		GfMulL mh;
		mh.init(hcoef[j], g.poly, lane);
		tp[j] = (unsigned)mh.at_slot(0) | ((unsigned)mh.at_slot(1) << 8) | ((unsigned)mh.at_slot(2) << 16) | ((unsigned)mh.at_slot(3) << 24);
	}
	STAMP(0);

	// ---- rank 0 of every edge: value m_j, check-domain symbol z_j (highest symbol among equal maxima, :1731) -------
	double mtop[4], lmin[4];
	int ztop[4];
#pragma unroll
	for (int j = 0; j < 4; j++) {
		double loc = dmax(dmax(v[j][0], v[j][1]), dmax(v[j][2], v[j][3]));
		mtop[j] = wave_max_f64(loc);
		lmin[j] = wave_min_f64(loc); // 64 distinct entries are >= lmin, so the nm-th best (nm <= 64) is too
		uint64_t eqm[4];
#pragma unroll
		for (int i = 0; i < 4; i++) eqm[i] = __ballot(v[j][i] == mtop[j]);
		const int topa = highest_sym(eqm);
		ztop[j] = g.mul[hcoef[j] * Q + topa];
	}
	STAMP(1);

	// ---- top-nm selection -------------------------------------------------------------------------------------------
	// (a) a 256-bucket histogram of [lmin, mtop] (LDS atomics + DPP prefix sum) finds the bucket that holds the nm-th best
	// value -- bucketing only has to be monotone, so its arithmetic needs no care; (b) a ballot quickselect, restricted to
	// that bucket, finds the cut exactly; ties at the cut are resolved by symbol, higher first (:1731).
	int n0[4] = {0, 0, 0, 0}; // entries with an even check-domain symbol per edge (they come first in the list)
	if (NC >= 2) { // conf(nm,1) needs no lists: it is contained in conf(q,1)
		int bk[4][4];
		int *H = (int *)U; // [256 buckets][4 edges] (spans U and P); lane l reads buckets 4l .. 4l+3
		{
			int4 z4 = {0, 0, 0, 0};
#pragma unroll
			for (int i = 0; i < 4; i++) ((int4 *)H)[lane * 4 + i] = z4;
		}
		__syncthreads();
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const double range = mtop[j] - lmin[j];
			// bucketing only has to be monotone and the same in every lane: the hardware reciprocal will do (an IEEE division is ~14 instructions)
			const double scale = range > 0.0 ? 256.0 * __builtin_amdgcn_rcp(range) : 0.0;
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const double d = (mtop[j] - v[j][i]) * scale;
				int bi = (int)dmin(d, 255.0);
				bk[j][i] = (v[j][i] >= lmin[j]) ? bi : 256;
				if (bk[j][i] < 256) atomicAdd(&H[bk[j][i] * 4 + j], 1);
			}
		}
		__syncthreads();
		int hc[4][4]; // [bucket slot of this lane][edge]
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const int4 h4 = ((int4 *)H)[lane * 4 + i];
			hc[i][0] = h4.x; hc[i][1] = h4.y; hc[i][2] = h4.z; hc[i][3] = h4.w;
		}
		STAMP(2);
		// per edge: locate the cut bucket, settle the members, compact them into the list image
		// [even-symbol group | odd-symbol group]
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const int tot = hc[0][j] + hc[1][j] + hc[2][j] + hc[3][j];
			const int cum = wave_scan_add(tot);
			const uint64_t reach = __ballot(cum >= NM);
			const int lstar = reach ? __builtin_ctzll(reach) : 63;
			// every lane works out which of its four buckets would be the cut; the cut lane's answer is read back
			const int pre = cum - tot, s0 = pre + hc[0][j], s1 = s0 + hc[1][j], s2 = s1 + hc[2][j];
			const int bsel = (s0 >= NM) ? 0 : (s1 >= NM) ? 1 : (s2 >= NM) ? 2 : 3;
			const int upto = (s0 >= NM) ? s0 : (s1 >= NM) ? s1 : (s2 >= NM) ? s2 : cum; // entries up to and including the cut bucket
			const int bstar = 4 * lstar + __builtin_amdgcn_readlane(bsel, lstar);
			const bool exact = __builtin_amdgcn_readlane(upto, lstar) == NM; // the cut bucket ends exactly at the nm-th entry
			uint64_t member[4];
			if (exact) {
#pragma unroll
				for (int i = 0; i < 4; i++) member[i] = __ballot(bk[j][i] <= bstar);
			} else {
				SelState ss;
#pragma unroll
				for (int i = 0; i < 4; i++) { ss.cand[i] = __ballot(bk[j][i] == bstar); ss.gt[i] = ss.eq[i] = 0; }
				ss.done = 0;
				for (int guard = 0; guard < 300 && !ss.done; guard++) {
					select_step(v[j], NM, ss);
					STAMP_COUNT(9);
				}
				finish_members(ss, NM, member);
			}
			// compaction straight from the (wave-uniform) member masks
			uint64_t g0[4], g1[4];
			int c0n = 0;
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const uint64_t even = __ballot(((tp[j] >> (8 * i)) & 1u) == 0u);
				g0[i] = member[i] & even;
				g1[i] = member[i] & ~even;
				c0n += __popcll(g0[i]);
			}
			n0[j] = c0n;
			ListEnt *Lj = lstp + j * NMP;
			int base0 = 0, base1 = c0n; // [even-symbol group | odd-symbol group], back to back
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const int sym = TSYM(j, i);
				const int pe_ = base0 + prefix_count(g0[i]), po_ = base1 + prefix_count(g1[i]);
				const int pos = (sym & 1) ? po_ : pe_;
				if (__builtin_amdgcn_inverse_ballot_w64(member[i])) {
					ListEnt e;
					e.v = v[j][i];
					e.t = sym;
					e.tt = (sym & 0xFE) << 3;
					Lj[pos] = e;
				}
				base0 += __popcll(g0[i]);
				base1 += __popcll(g1[i]);
			}
		}
	}
	__syncthreads();
	STAMP(4);

	// ---- conf(q,1): one edge deviates to any symbol, the others stay at rank 0 (:894) ---------------------------
	double S[4][4];
#pragma unroll
	for (int x = 0; x < 4; x++)
#pragma unroll
		for (int i = 0; i < 4; i++) S[x][i] = -NBL_DBL_MAX;
	const int ztot = ztop[0] ^ ztop[1] ^ ztop[2] ^ ztop[3];
#pragma unroll
	for (int j = 0; j < 4; j++) {
		__syncthreads();
#pragma unroll
		for (int i = 0; i < 4; i++) U[TSYM(j, i)] = v[j][i];
		__syncthreads();
#pragma unroll
		for (int x = 0; x < 4; x++) {
			if (x == j) continue;
			const int shift = ztot ^ ztop[x] ^ ztop[j];
			// U[s ^ shift] for the lane's four symbols: two 16-byte reads (the XOR permutes 16-byte slots; bit 0 swaps inside a slot)
			const int ad = (lane << 4) ^ ((shift & 0xFE) << 3);
			const double2 ra = *(const double2 *)((const char *)U + ad), rb = *(const double2 *)((const char *)U + (ad ^ 1024));
			const bool sw = shift & 1;
			const double uu[4] = {sw ? ra.y : ra.x, sw ? ra.x : ra.y, sw ? rb.y : rb.x, sw ? rb.x : rb.y};
#pragma unroll
			for (int i = 0; i < 4; i++) {
				double acc = 0.0;
				const double u = uu[i];
#pragma unroll
				for (int o = 0; o < 4; o++) {
					if (o == x) continue;
					acc = acc + ((o == j) ? u : mtop[o]);
				}
				S[x][i] = dmax(S[x][i], acc);
			}
		}
	}
	STAMP(5);

	// ---- conf(nm,nc >= 3): truncated max-plus convolutions; outputs 3,2 share P = e0 (+) e1 ----------------------
	const int lane16 = lane << 4;
	// dst[t_a ^ t_b ^ sxor] = max over the nm x nm entry pairs of (v_a + bias) + v_b   (bias = 0: plain pair convolution)
	auto pair_scatter = [&](double *dst, int ja, int jb, double bias, int sxor) {
		__syncthreads();
		double2 ninf;
		ninf.x = NBL_NEG_INF;
		ninf.y = NBL_NEG_INF;
		((double2 *)dst)[lane] = ninf;
		((double2 *)dst)[64 + lane] = ninf;
		ListEnt ea = lstp[ja * NMP + (lane & (NM - 1))];
		ea.v = ea.v + bias;
		ea.t ^= sxor;
		__syncthreads();
		constexpr int PER = 64 >> LOGNM, ROUNDS = (NM << LOGNM) >> 6;
#pragma unroll
		for (int it = 0; it < ROUNDS; it++) {
			const ListEnt eb = lstp[jb * NMP + it * PER + (lane >> LOGNM)];
			__hip_atomic_fetch_max(&dst[ea.t ^ eb.t], ea.v + eb.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		__syncthreads();
	};
	// same, accumulating into dst without clearing it
	auto pair_scatter_more = [&](double *dst, int ja, int jb, double bias, int sxor) {
		ListEnt ea = lstp[ja * NMP + (lane & (NM - 1))];
		ea.v = ea.v + bias;
		ea.t ^= sxor;
		constexpr int PER = 64 >> LOGNM, ROUNDS = (NM << LOGNM) >> 6;
#pragma unroll
		for (int it = 0; it < ROUNDS; it++) {
			const ListEnt eb = lstp[jb * NMP + it * PER + (lane >> LOGNM)];
			__hip_atomic_fetch_max(&dst[ea.t ^ eb.t], ea.v + eb.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		__syncthreads();
	};
	auto pair_conv = [&](int ja, int jb) {
		pair_scatter(P, ja, jb, 0.0, 0);
		((double2 *)P)[128 + lane] = ((const double2 *)P)[lane]; // mirror of the low half for the gathers
		__syncthreads();
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
	// conf(nm,2) for output x with others o1 < o2 < o3 (P already holds o1 (+) o2): one of the three stays at rank 0 (:1769)
	auto conf_nc2 = [&](int x, int o1, int o2, int o3) {
		fold(P, ztop[o3], mtop[o3], S[x]);                                   // (v1 + v2) + m3
		pair_scatter(U, o2, o3, mtop[o1], ztop[o1]);                         // (m1 + v2) + v3
		pair_scatter_more(U, o1, o3, mtop[o2], ztop[o2]);                    // (v1 + m2) + v3
		{
			const double2 ra = ((const double2 *)U)[lane], rb = ((const double2 *)U)[64 + lane];
			S[x][0] = dmax(S[x][0], ra.x); S[x][1] = dmax(S[x][1], ra.y);
			S[x][2] = dmax(S[x][2], rb.x); S[x][3] = dmax(S[x][3], rb.y);
		}
	};
	auto gather_conv = [&](int jc, double (&Sout)[4]) {
		double a0 = NBL_NEG_INF, a1 = NBL_NEG_INF, a2 = NBL_NEG_INF, a3 = NBL_NEG_INF;
		const char *Pb = (const char *)P;
		const ListEnt *L = lstp + jc * NMP;
		// entries [0, n0): even symbols, the lane's slot pairs line up; [n0, NM): odd symbols, the pairs are swapped.  Four entries
		// per trip, then the remainder one by one (no padding entries: every trip is real work)
		auto run = [&](int k0, int k1, auto swapped) {
			constexpr bool SW = decltype(swapped)::value;
			int k = k0;
			for (; k + 4 <= k1; k += 4) {
				ListEnt en[4];
				double2 ra[4], rb[4];
#pragma unroll
				for (int u = 0; u < 4; u++) en[u] = L[k + u]; // same address in every lane: LDS broadcast
#pragma unroll
				for (int u = 0; u < 4; u++) {
					const int ad = lane16 ^ en[u].tt;
					ra[u] = *(const double2 *)(Pb + ad);
					rb[u] = *(const double2 *)(Pb + ad + 1024); // the other half (P[256..383] mirrors P[0..127])
				}
#pragma unroll
				for (int u = 0; u < 4; u++) {
					a0 = dmax(a0, (SW ? ra[u].y : ra[u].x) + en[u].v);
					a1 = dmax(a1, (SW ? ra[u].x : ra[u].y) + en[u].v);
					a2 = dmax(a2, (SW ? rb[u].y : rb[u].x) + en[u].v);
					a3 = dmax(a3, (SW ? rb[u].x : rb[u].y) + en[u].v);
				}
			}
			for (; k < k1; k++) {
				const ListEnt en = L[k];
				const int ad = lane16 ^ en.tt;
				const double2 ra = *(const double2 *)(Pb + ad), rb = *(const double2 *)(Pb + ad + 1024);
				a0 = dmax(a0, (SW ? ra.y : ra.x) + en.v);
				a1 = dmax(a1, (SW ? ra.x : ra.y) + en.v);
				a2 = dmax(a2, (SW ? rb.y : rb.x) + en.v);
				a3 = dmax(a3, (SW ? rb.x : rb.y) + en.v);
			}
		};
		run(0, n0[jc], std::false_type{});
		run(n0[jc], NM, std::true_type{});
		Sout[0] = dmax(Sout[0], a0);
		Sout[1] = dmax(Sout[1], a1);
		Sout[2] = dmax(Sout[2], a2);
		Sout[3] = dmax(Sout[3], a3);
	};
	// c2v[a] = shape(S[h_x a] - S[0]) (:899-916)
	auto emit = [&](int x) {
		double *Sx = U;
		__syncthreads();
		double2 s01, s23;
		s01.x = S[x][0]; s01.y = S[x][1]; s23.x = S[x][2]; s23.y = S[x][3];
		((double2 *)Sx)[lane] = s01;
		((double2 *)Sx)[64 + lane] = s23;
		__syncthreads();
		const double s0 = Sx[0];
		double y[4];
#pragma unroll
		for (int i = 0; i < 4; i++) y[i] = shape_llr(Sx[TSYM(x, i)] - s0, r.factor, r.offset);
		if (lane == 0) y[0] = 0.0;
		double2 o01, o23;
		o01.x = y[0]; o01.y = y[1]; o23.x = y[2]; o23.y = y[3];
		double2 *dst = (double2 *)(C + (size_t)x * Q);
		dst[lane] = o01;
		dst[64 + lane] = o23;
	};

	if (NC >= 3) {
		pair_conv(0, 1);
		STAMP(6);
		gather_conv(2, S[3]);
		gather_conv(3, S[2]);
		STAMP(7);
		emit(3);
		emit(2);
		STAMP(8);
		pair_conv(0, 2);
		STAMP(6);
		gather_conv(3, S[1]);
		STAMP(7);
		emit(1);
		STAMP(8);
		pair_conv(1, 2);
		STAMP(6);
		gather_conv(3, S[0]);
		STAMP(7);
		emit(0);
		STAMP(8);
	} else if (NC == 2) {
		pair_conv(0, 1);
		conf_nc2(3, 0, 1, 2);
		STAMP(6);
		emit(3);
		STAMP(8);
		conf_nc2(2, 0, 1, 3);
		STAMP(6);
		emit(2);
		STAMP(8);
		pair_conv(0, 2);
		conf_nc2(1, 0, 2, 3);
		STAMP(6);
		emit(1);
		STAMP(8);
		pair_conv(1, 2);
		conf_nc2(0, 1, 2, 3);
		STAMP(6);
		emit(0);
		STAMP(8);
	} else {
		emit(3);
		emit(2);
		emit(1);
		emit(0);
		STAMP(8);
	}

#ifdef NBL_EMS_STAMPS
	if (st_on && lane == 0) {
		for (int i = 0; i < 12; i++) atomicAdd(&w.stamps[i], st_acc[i]);
		atomicAdd(&w.stamps[15], 1ull);
	}
#endif
#undef STAMP
#undef STAMP_COUNT
#undef TSYM
}

bool nbl_ems256_applicable(const NblGraphDev &g, bool all_dc4, int nm, int nc)
{
	return g.q == 256 && all_dc4 && nc >= 1 && (nm == 8 || nm == 16 || nm == 32);
}

size_t nbl_ems256_lds_bytes(int nm) { return 2 * Q * 8 + Q / 2 * 8 + (size_t)4 * nm * 16; }

template <int NM, bool FUSED>
static void launch_nc(int nc, dim3 grid, dim3 block, size_t lds, hipStream_t st, const NblGraphDev &g, const NblWork &w, const NblRun &r)
{
	if (nc >= 3) cn_ems_q256_dc4_kernel<NM, FUSED, 3><<<grid, block, 0, st>>>(g, w, r);
	else if (nc == 2) cn_ems_q256_dc4_kernel<NM, FUSED, 2><<<grid, block, 0, st>>>(g, w, r);
	else cn_ems_q256_dc4_kernel<NM, FUSED, 1><<<grid, block, 0, st>>>(g, w, r);
}

hipError_t nbl_launch_cn_ems256(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool fused, hipStream_t st)
{
	dim3 grid((unsigned)((long long)((r.B + 7) / 8) * 8 * g.M)), block(64);
	const size_t lds = nbl_ems256_lds_bytes(r.nm);
	if (fused) {
		switch (r.nm) {
		case 8: launch_nc<8, true>(r.nc, grid, block, lds, st, g, w, r); break;
		case 16: launch_nc<16, true>(r.nc, grid, block, lds, st, g, w, r); break;
		case 32: launch_nc<32, true>(r.nc, grid, block, lds, st, g, w, r); break;
		default: return hipErrorInvalidValue;
		}
	} else {
		switch (r.nm) {
		case 8: launch_nc<8, false>(r.nc, grid, block, lds, st, g, w, r); break;
		case 16: launch_nc<16, false>(r.nc, grid, block, lds, st, g, w, r); break;
		case 32: launch_nc<32, false>(r.nc, grid, block, lds, st, g, w, r); break;
		default: return hipErrorInvalidValue;
		}
	}
	return hipGetLastError();
}
