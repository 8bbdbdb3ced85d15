// tools/repro/tems256_dp_soa.hip -- self-checking reproducer of the wrong-path-code fault of the GF(256) T-EMS kernel
// (nbl_cn_tems256.hip, DESIGN.md section 4 "toolchain note"), reduced to the dynamic programme over the trellis columns.
//
// The SAME programme is compiled twice into one binary: with the per-symbol DP state as six parallel [4] arrays (SOA, the
// layout the kernel had first) and as an array of structs (AOS, the layout it has now).  The host evaluates the programme's
// definition directly (lexicographic minimum of (cost, path code) over the candidate set -- independent of evaluation order)
// and compares.  Build and run on a gfx950 box:
//     hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 tools/repro/tems256_dp_soa.hip -o /tmp/repro && /tmp/repro
// Exit code 0 = both layouts agree with the host; 1 = a layout differs (printed).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int Q = 256, P = 8, DC = 4, NS = 4;

struct __attribute__((aligned(16))) Cand { double u; int q16; unsigned dig; };

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int prefix_count(uint64_t mask)
{
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

__device__ __forceinline__ void relax(double &bv, unsigned &bc, double val, unsigned code)
{
	const bool lt = val < bv;
	bv = lt ? val : bv;
	bc = lt ? code : bc;
	const unsigned cm = code < bc ? code : bc;
	bc = (val == bv) ? cm : bc;
}

template <bool SOA, int NC>
__global__ __launch_bounds__(64) void dp_kernel(const double *U, int nr, double *dW_out, unsigned *eta_out)
{
	__shared__ __attribute__((aligned(16))) char lds[2 * Q * 16 + (Q + 4) * 16];
	char *Sraw = lds;
	double2 *Sv = (double2 *)Sraw;
	uint4 *Sc = (uint4 *)(Sraw + Q * 16);
	Cand *cl = (Cand *)(lds + 2 * Q * 16);
	const int lane = lane_id();
	constexpr int nc = NC;
	const double *Ub = U + (size_t)blockIdx.x * DC * Q;
	double u[NS][DC];
#pragma unroll
	for (int d = 0; d < DC; d++)
#pragma unroll
		for (int i = 0; i < NS; i++) u[i][d] = Ub[d * Q + lane + 64 * i];

	int mask[NS];
#pragma unroll
	for (int i = 0; i < NS; i++) {
		mask[i] = 0;
#pragma unroll
		for (int d = 0; d < DC; d++) {
			int rank = 0;
#pragma unroll
			for (int e = 0; e < DC; e++)
				if (e != d) rank += (u[i][e] < u[i][d] || (u[i][e] == u[i][d] && e < d)) ? 1 : 0;
			mask[i] |= (rank < nr) ? (1 << d) : 0;
		}
	}

	const double INF = __builtin_huge_val();
	// ---- the two layouts of the DP state ----
	double v1[NS], v2[NS], v3[NS];
	unsigned c1[NS], c2[NS], c3[NS];
	struct DpState { double v1; unsigned c1; double v2; unsigned c2; double v3; unsigned c3; } st[NS];
#define V1(i) (SOA ? v1[i] : st[i].v1)
#define V2(i) (SOA ? v2[i] : st[i].v2)
#define V3(i) (SOA ? v3[i] : st[i].v3)
#define C1(i) (SOA ? c1[i] : st[i].c1)
#define C2(i) (SOA ? c2[i] : st[i].c2)
#define C3(i) (SOA ? c3[i] : st[i].c3)
#pragma unroll
	for (int i = 0; i < NS; i++) {
		v1[i] = v2[i] = v3[i] = INF;
		c1[i] = c2[i] = c3[i] = 0;
		st[i].v1 = st[i].v2 = st[i].v3 = INF;
		st[i].c1 = st[i].c2 = st[i].c3 = 0;
		const int s = lane + 64 * i;
		if ((mask[i] & 1) && s > 0) {
			if (SOA) { v1[i] = u[i][0]; c1[i] = (unsigned)s << (P * (DC - 1)); }
			else { st[i].v1 = u[i][0]; st[i].c1 = (unsigned)s << (P * (DC - 1)); }
		}
	}
	const int lane16 = lane << 4;
#pragma unroll
	for (int d = 1; d < DC; d++) {
		const int sh = P * (DC - 1 - d);
		int base = 0;
#pragma unroll
		for (int i = 0; i < NS; i++) {
			const int s = lane + 64 * i;
			Sv[s] = make_double2(V1(i), V2(i));
			Sc[s] = make_uint4(C1(i), C2(i), 0u, 0u);
			const bool c = ((mask[i] >> d) & 1) && s > 0;
			const uint64_t bal = __ballot(c);
			if (c) {
				Cand e;
				e.u = u[i][d];
				e.q16 = s << 4;
				e.dig = (unsigned)s << sh;
				cl[base + prefix_count(bal)] = e;
			}
			base += uniform(__builtin_popcountll(bal));
		}
		const int n = base, n4 = (n + 3) & ~3;
		if (lane < 4 && n + lane < n4) {
			Cand e;
			e.u = INF;
			e.q16 = 0;
			e.dig = 0;
			cl[n + lane] = e;
		}
		__syncthreads();
		if (nc >= 2) {
			for (int k = 0; k < n4; k += 2) {
				Cand e[2];
#pragma unroll
				for (int t = 0; t < 2; t++) e[t] = cl[k + t];
#pragma unroll
				for (int t = 0; t < 2; t++) {
					const int A = lane16 ^ e[t].q16;
#pragma unroll
					for (int i = 0; i < NS; i++) {
						const int off = A ^ (i << 10);
						const double2 sv = *(const double2 *)(Sraw + off);
						const uint2 sc = *(const uint2 *)(Sraw + Q * 16 + off);
						if (SOA) {
							relax(v2[i], c2[i], sv.x + e[t].u, sc.x + e[t].dig);
							if (d >= 2 && nc >= 3) relax(v3[i], c3[i], sv.y + e[t].u, sc.y + e[t].dig);
						} else {
							relax(st[i].v2, st[i].c2, sv.x + e[t].u, sc.x + e[t].dig);
							if (d >= 2 && nc >= 3) relax(st[i].v3, st[i].c3, sv.y + e[t].u, sc.y + e[t].dig);
						}
					}
				}
			}
		}
#pragma unroll
		for (int i = 0; i < NS; i++) {
			const int s = lane + 64 * i;
			if (((mask[i] >> d) & 1) && s > 0) {
				if (SOA) relax(v1[i], c1[i], u[i][d], (unsigned)s << sh);
				else relax(st[i].v1, st[i].c1, u[i][d], (unsigned)s << sh);
			}
		}
		__syncthreads();
	}
#pragma unroll
	for (int i = 0; i < NS; i++) {
		double dW = INF;
		unsigned eta = 0xffffffffu;
		const double v0 = (lane == 0 && i == 0) ? 0.0 : INF;
		if (v0 < dW || (v0 == dW && 0u < eta)) { dW = v0; eta = 0u; }
		if (nc >= 1 && (V1(i) < dW || (V1(i) == dW && C1(i) < eta))) { dW = V1(i); eta = C1(i); }
		if (nc >= 2 && (V2(i) < dW || (V2(i) == dW && C2(i) < eta))) { dW = V2(i); eta = C2(i); }
		if (nc >= 3 && (V3(i) < dW || (V3(i) == dW && C3(i) < eta))) { dW = V3(i); eta = C3(i); }
		dW_out[(size_t)blockIdx.x * Q + lane + 64 * i] = dW;
		eta_out[(size_t)blockIdx.x * Q + lane + 64 * i] = eta;
	}
}

// the programme's definition on the host: per layer the lexicographic minimum of (cost, code)
static void host_dp(const double *U, int nr, int nc, double *dW, unsigned *eta)
{
	const double INF = __builtin_huge_val();
	std::vector<int> mask(Q, 0);
	for (int s = 0; s < Q; s++)
		for (int d = 0; d < DC; d++) {
			int rank = 0;
			for (int e = 0; e < DC; e++)
				if (e != d) rank += (U[e * Q + s] < U[d * Q + s] || (U[e * Q + s] == U[d * Q + s] && e < d)) ? 1 : 0;
			if (rank < nr) mask[s] |= 1 << d;
		}
	auto better = [](double v, unsigned c, double bv, unsigned bc) { return v < bv || (v == bv && c < bc); };
	std::vector<double> v1(Q, INF), v2(Q, INF), v3(Q, INF);
	std::vector<unsigned> c1(Q, 0), c2(Q, 0), c3(Q, 0);
	for (int s = 1; s < Q; s++)
		if (mask[s] & 1) { v1[s] = U[s]; c1[s] = (unsigned)s << (P * (DC - 1)); }
	for (int d = 1; d < DC; d++) {
		const int sh = P * (DC - 1 - d);
		std::vector<double> p1 = v1, p2 = v2;
		std::vector<unsigned> q1 = c1, q2 = c2;
		for (int s = 0; s < Q; s++) {
			for (int q = 1; q < Q; q++) {
				if (!((mask[q] >> d) & 1)) continue;
				const double uq = U[d * Q + q];
				const unsigned dig = (unsigned)q << sh;
				if (nc >= 2) {
					// (the kernel's relax() also moves the code to the smaller one on equal INFINITE costs: same rule here)
					const double val = p1[s ^ q] + uq;
					const unsigned code = q1[s ^ q] + dig;
					if (better(val, code, v2[s], c2[s])) { v2[s] = val; c2[s] = code; }
				}
				if (d >= 2 && nc >= 3) {
					const double val = p2[s ^ q] + uq;
					const unsigned code = q2[s ^ q] + dig;
					if (better(val, code, v3[s], c3[s])) { v3[s] = val; c3[s] = code; }
				}
			}
			if (s > 0 && ((mask[s] >> d) & 1)) {
				const double val = U[d * Q + s];
				const unsigned code = (unsigned)s << sh;
				if (better(val, code, v1[s], c1[s])) { v1[s] = val; c1[s] = code; }
			}
		}
	}
	for (int s = 0; s < Q; s++) {
		double w = INF;
		unsigned e = 0xffffffffu;
		const double v0 = s == 0 ? 0.0 : INF;
		if (better(v0, 0u, w, e)) { w = v0; e = 0u; }
		if (nc >= 1 && better(v1[s], c1[s], w, e)) { w = v1[s]; e = c1[s]; }
		if (nc >= 2 && better(v2[s], c2[s], w, e)) { w = v2[s]; e = c2[s]; }
		if (nc >= 3 && better(v3[s], c3[s], w, e)) { w = v3[s]; e = c3[s]; }
		dW[s] = w;
		eta[s] = e;
	}
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

template <bool SOA, int NC> static int run(const char *name, const std::vector<double> &U, int blocks, int nr)
{
	double *dU, *ddW;
	unsigned *deta;
	CK(hipMalloc(&dU, U.size() * 8));
	CK(hipMalloc(&ddW, (size_t)blocks * Q * 8));
	CK(hipMalloc(&deta, (size_t)blocks * Q * 4));
	CK(hipMemcpy(dU, U.data(), U.size() * 8, hipMemcpyHostToDevice));
	dp_kernel<SOA, NC><<<blocks, 64>>>(dU, nr, ddW, deta);
	CK(hipDeviceSynchronize());
	std::vector<double> dW((size_t)blocks * Q), rW(Q);
	std::vector<unsigned> eta((size_t)blocks * Q), re(Q);
	CK(hipMemcpy(dW.data(), ddW, dW.size() * 8, hipMemcpyDeviceToHost));
	CK(hipMemcpy(eta.data(), deta, eta.size() * 4, hipMemcpyDeviceToHost));
	long bad_w = 0, bad_e = 0;
	for (int b = 0; b < blocks; b++) {
		host_dp(&U[(size_t)b * DC * Q], nr, NC, rW.data(), re.data());
		for (int s = 0; s < Q; s++) {
			const bool infs = rW[s] == __builtin_huge_val() && dW[(size_t)b * Q + s] == rW[s]; // unreachable check sum: code is not used
			if (dW[(size_t)b * Q + s] != rW[s]) bad_w++;
			else if (!infs && eta[(size_t)b * Q + s] != re[s]) {
				if (bad_e < 4) printf("  %s nc=%d block %d s=%d: cost %.1f path code %08x, host %08x\n", name, NC, b, s, rW[s], eta[(size_t)b * Q + s], re[s]);
				bad_e++;
			}
		}
	}
	printf("%s nr=%d nc=%d: %ld wrong costs, %ld wrong path codes of %d\n", name, nr, NC, bad_w, bad_e, blocks * Q);
	(void)hipFree(dU); (void)hipFree(ddW); (void)hipFree(deta);
	return (bad_w || bad_e) ? 1 : 0;
}

int main()
{
	const int blocks = 64;
	std::vector<double> U((size_t)blocks * DC * Q);
	unsigned x = 12345u;
	for (size_t i = 0; i < U.size(); i++) { // small integers: ties everywhere (the failing shape has integer LLRs)
		x = x * 1664525u + 1013904223u;
		U[i] = (double)((x >> 24) % 7);
	}
	for (int b = 0; b < blocks; b++)
		for (int d = 0; d < DC; d++) U[((size_t)b * DC + d) * Q] = 0.0; // dU[d][0] = 0: the most reliable symbol
	int rc = 0;
	rc |= run<true, 2>("SOA", U, blocks, 3);
	rc |= run<false, 2>("AOS", U, blocks, 3);
	rc |= run<true, 3>("SOA", U, blocks, 3);
	rc |= run<false, 3>("AOS", U, blocks, 3);
	rc |= run<true, 2>("SOA", U, blocks, 2);
	rc |= run<false, 2>("AOS", U, blocks, 2);
	return rc;
}
