// nbldpc_amd/csrc/nbl_api.cpp -- C ABI (include/nbldpc.h) on top of the HIP kernels.
//
// Replaces CNBLDPC::Initial's decoder set-up (NBLDPC.cpp:140-377: graph cross indices, message buffers) and
// CNBLDPC::Decoding's iteration loop (NBLDPC.cpp:607-641 -> Decoding_BP/EMS/TEMS) for a batch of codewords.
// No CPU decode path exists in this library: without a HIP device nbl_create() fails.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <cmath>
#include <string>
#include <thread>
#include <vector>
#include "../../include/nbldpc.h"
#include "nbl_kernels.h"

static thread_local std::string g_create_error;

struct nbl_decoder {
	int device = -1;
	nbl_params prm{};
	NblGraphDev g{};
	NblWork w{};
	std::vector<void *> graph_allocs;
	int *d_e2c_map = nullptr;   // variable-major edge -> check-major slot (same as v_cpos), for c2v read-back
	int cap = 0;                // codewords the workspace holds
	size_t ws_bytes = 0;
	double *d_Lin = nullptr;    // staging of the host-layout input [cap][N][q-1]
	uint8_t *d_conv8 = nullptr;
	hipStream_t stream = nullptr;
	bool record_state = false;
	bool all_dc4 = false;       // every check has degree 4
	int min_dc = 0;             // smallest check degree
	bool all_dv2 = false;       // every variable has degree 2
	double *c2v_alt = nullptr;  // second c2v buffer of the fused EMS iteration (flooding schedule -> double buffer)
	double *c2v_zero = nullptr; // fused iterations: the all-zero c2v of iteration 0, ONE [E][q] block shared by all codewords, written once
	                            // when the workspace is made and only ever read (iteration 1 reads it instead of a buffer that would
	                            // have to be cleared on every call)
	const double *last_c2v = nullptr;
	bool last_fused = false;    // the last decode ran fused iterations (nbl_read_state picks the c2v buffer per codeword)
	// device-side demodulator (nbl_set_demodulator)
	int dm_order = 0, dm_L = 0;
	double *d_cons = nullptr;
	int *d_src = nullptr;
	double *d_rx = nullptr;
	size_t d_rx_cap = 0;
	// device-side AWGN channel (nbl_decode_batch_noise)
	std::vector<double> h_cons;  // constellation as given to nbl_set_demodulator
	uint32_t *d_jump = nullptr;  // [3][L] A^(4 s) mod m of the three generators
	uint32_t *d_state = nullptr; // [cap][3]
	uint8_t *d_txi = nullptr;    // [cap][L]
	double *d_fn = nullptr;      // [cap][L][2][2]: log(1 - u1), cos(2 pi u2) of every normal draw
	uint32_t *d_fidx = nullptr;  // uncertain values: index into d_fn, argument, host-evaluated value
	double *d_farg = nullptr, *d_fval = nullptr;
	unsigned *d_fcount = nullptr;
	uint32_t *h_fidx = nullptr;  // pinned host mirrors
	double *h_farg = nullptr, *h_fval = nullptr;
	size_t noise_cap = 0, flag_cap = 0;
	double last_flag_frac = 0.0;
	hipStream_t stream2 = nullptr; // the channel of batch k+1 runs here while batch k is decoded on `stream`
	double *d_rxs[2] = {nullptr, nullptr}; // resident received samples of nbl_channel_batch, one per slot
	size_t d_rxs_cap[2] = {0, 0};
	int rxs_B[2] = {0, 0};
	std::string err2;              // error text of the channel thread (nbl_channel_batch); nbl_last_error reports both
	// hipGraph replay of the iteration loop: one executable graph per window of iterations (fixed iterations: the whole loop;
	// early exit: the `poll_every` iterations between two polls), captured on the decoder's own stream the first time a window is
	// run with a given set of buffers, replayed on the caller's stream afterwards.  NBL_GRAPH=0 switches it off.
	struct GraphKey { const void *lin, *lch, *v2c, *c2v, *alt, *post; int B, record, generic, fused; };
	GraphKey gkey{};
	std::vector<hipGraphExec_t> gexec; // index = window number
	int *d_active = nullptr;           // [cap] active list (early exit, batches of NBL_COMPACT_MIN codewords or more)
	bool use_compact = true;           // NBL_COMPACT=0 switches it off
	bool use_graph = false;            // opt-in (NBL_GRAPH=1): measured gain is nil, see DESIGN.md section 7
	int force_generic = 0;      // debug: 1 = always the generic kernels, 2 = specialised kernels but no VN/CN fusion
	bool profiling = false;
	hipEvent_t ev[2] = {nullptr, nullptr};
	int *h_ndone = nullptr;      // pinned [2]: converged-codeword counts read back after each window of iterations
	std::vector<hipEvent_t> pev; // per-launch events (profiling only)
	double ms[4] = {0, 0, 0, 0};
	long long launches[3] = {0, 0, 0};
	int last_B = 0;
	std::string err;
};

// HIP_TRY_E: the failing call's text goes into `errstr` -- dec->err for everything the decode thread does, dec->err2 for the channel
// thread (nbl_channel_batch may run beside a decode on the same handle, so the two never share a string)
#define HIP_TRY_E(errstr, call)                                                                    \
	do {                                                                                           \
		hipError_t e_ = (call);                                                                    \
		if (e_ != hipSuccess) {                                                                    \
			(errstr) = std::string(#call) + ": " + hipGetErrorString(e_);                          \
			return NBL_ERR_HIP;                                                                    \
		}                                                                                          \
	} while (0)
#define HIP_TRY(dec, call) HIP_TRY_E((dec)->err, call)

static int ilog2(int q)
{
	int p = 0;
	while ((1 << p) < q) p++;
	return p;
}

template <typename T> static nbl_status upload(nbl_decoder *d, const std::vector<T> &h, const T **dst)
{
	void *p = nullptr;
	HIP_TRY(d, hipMalloc(&p, h.size() * sizeof(T) + 16));
	d->graph_allocs.push_back(p);
	HIP_TRY(d, hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
	*dst = (const T *)p;
	return NBL_OK;
}

static void drop_graphs(nbl_decoder *d)
{
	for (auto &ge : d->gexec)
		if (ge) (void)hipGraphExecDestroy(ge);
	d->gexec.clear();
}

static void free_workspace(nbl_decoder *d)
{
	drop_graphs(d);
	void *ptrs[] = {d->w.Lch, d->w.v2c, d->w.c2v, d->w.post, d->w.dec, d->w.out, d->w.iters, d->w.done, d->d_Lin, d->d_conv8, d->c2v_alt, d->w.edge_dec, d->d_active, d->c2v_zero};
	d->c2v_zero = nullptr;
	d->d_active = nullptr;
	d->w.active = nullptr;
	d->c2v_alt = nullptr;
	for (void *p : ptrs)
		if (p) (void)hipFree(p);
	d->w.Lch = d->w.v2c = d->w.c2v = d->w.post = nullptr;
	d->w.dec = d->w.out = d->w.iters = d->w.edge_dec = nullptr;
	d->w.done = nullptr;
	d->d_Lin = nullptr;
	d->d_conv8 = nullptr;
	d->cap = 0;
	d->ws_bytes = 0;
}

// Shapes whose whole iteration is ONE launch (variable-node pass recomputed inside the check-node kernel, c2v double-buffered):
// (2,4)-regular codes, EMS over GF(256) with nm <= 64, T-EMS over GF(64) and GF(256), log-QSPA over GF(256).
static bool small_enabled()
{
	static const bool on = !getenv("NBL_NO_SMALL"); // A/B measurements: the one-check-per-wave kernels on small fields
	return on;
}

// small fields (q <= 64): 64 / q checks per wave, any variable degree (nbl_cn_small.hip)
static bool small_shape(const nbl_decoder *d)
{
	if (!small_enabled()) return false;
	const nbl_params &p = d->prm;
	// (GF(64), check degree 4, T-EMS has its own kernel: config 4)
	if (p.method == NBL_METHOD_TEMS && d->all_dv2 && nbl_tems64_applicable(d->g, d->all_dc4, p.tems_nr, p.tems_nc)) return false;
	if (p.method == NBL_METHOD_BP && nbl_bp64_applicable(d->g, d->all_dc4)) return false; // (and log-QSPA: nbl_cn_bp64.hip)
	if (p.method == NBL_METHOD_EMS) return nbl_small_applicable(d->g, p.method, d->min_dc, p.ems_nm, p.ems_nc);
	if (p.method == NBL_METHOD_TEMS) return nbl_small_applicable(d->g, p.method, d->min_dc, 0, p.tems_nc);
	if (p.method == NBL_METHOD_BP) return nbl_small_applicable(d->g, p.method, d->min_dc, 0, 0);
	return false;
}

static bool ems64_shape(const nbl_decoder *d)
{
	return small_enabled() && d->prm.method == NBL_METHOD_EMS && nbl_ems64_applicable(d->g, d->min_dc, d->prm.ems_nm, d->prm.ems_nc);
}

static bool fused_shape(const nbl_decoder *d)
{
	// (variable degrees above 3: these kernels behind the separate VN pass)
	if (small_shape(d) || ems64_shape(d)) return d->g.c_nbr != nullptr;
	if (!d->all_dv2) return false;
	if (d->prm.method == NBL_METHOD_EMS) return nbl_ems256_applicable(d->g, d->all_dc4, d->prm.ems_nm, d->prm.ems_nc);
	if (d->prm.method == NBL_METHOD_TEMS)
		return nbl_tems64_applicable(d->g, d->all_dc4, d->prm.tems_nr, d->prm.tems_nc) || nbl_tems256_applicable(d->g, d->all_dc4, d->prm.tems_nr, d->prm.tems_nc);
	if (d->prm.method == NBL_METHOD_BP) return nbl_bp256_applicable(d->g, d->all_dc4) || nbl_bp64_applicable(d->g, d->all_dc4);
	return false;
}

static nbl_status ensure_workspace(nbl_decoder *d, int B)
{
	// v2c only exists in HBM when something reads it: the unfused path, or state read-back
	// (damped methods always keep it: the damping reads the previous iteration's v2c)
	const bool want_v2c = d->prm.method != NBL_METHOD_EMS || !fused_shape(d) || d->record_state || d->force_generic != 0;
	if (B <= d->cap && (!d->record_state || d->w.post) && (!want_v2c || d->w.v2c)) return NBL_OK;
	int cap = B > d->cap ? B : d->cap;
	free_workspace(d);
	const size_t q = d->g.q, N = d->g.N, E = d->g.E;
	size_t bytes = 0;
	auto alloc = [&](void **p, size_t n) -> hipError_t { bytes += n; return hipMalloc(p, n); };
	HIP_TRY(d, alloc((void **)&d->w.Lch, (size_t)cap * N * q * 8));
	if (want_v2c) HIP_TRY(d, alloc((void **)&d->w.v2c, (size_t)cap * E * q * 8));
	HIP_TRY(d, alloc((void **)&d->w.c2v, (size_t)cap * E * q * 8));
	if (fused_shape(d)) {
		HIP_TRY(d, alloc((void **)&d->c2v_alt, (size_t)cap * E * q * 8));
		HIP_TRY(d, alloc((void **)&d->c2v_zero, E * q * 8)); // one block for every codeword (NblWork::c2v_prev_shared)
		HIP_TRY(d, hipMemset(d->c2v_zero, 0, E * q * 8));
	}
	if (d->record_state) HIP_TRY(d, alloc((void **)&d->w.post, (size_t)cap * N * q * 8));
	HIP_TRY(d, alloc((void **)&d->w.dec, (size_t)cap * N * 4));
	if (d->prm.method != NBL_METHOD_EMS) HIP_TRY(d, alloc((void **)&d->w.edge_dec, (size_t)cap * E * 4));
	HIP_TRY(d, alloc((void **)&d->w.out, (size_t)cap * N * 4));
	HIP_TRY(d, alloc((void **)&d->w.iters, (size_t)cap * 4));
	HIP_TRY(d, alloc((void **)&d->w.done, (size_t)cap));
	HIP_TRY(d, alloc((void **)&d->d_active, (size_t)cap * 4));
	d->cap = cap;
	d->ws_bytes = bytes;
	return NBL_OK;
}

extern "C" int32_t nbl_abi_version(void) { return NBL_ABI_VERSION; }

extern "C" const char *nbl_last_error(const nbl_decoder *dec)
{
	if (!dec) return g_create_error.c_str();
	if (dec->err2.empty()) return dec->err.c_str();
	if (dec->err.empty()) return dec->err2.c_str();
	static thread_local std::string both;
	both = dec->err + " | channel: " + dec->err2;
	return both.c_str();
}

extern "C" size_t nbl_workspace_bytes(const nbl_decoder *dec) { return dec ? dec->ws_bytes : 0; }

static nbl_status fail_create(nbl_decoder *d, nbl_status st, const std::string &msg)
{
	g_create_error = msg.empty() && d ? d->err : msg;
	if (d) nbl_destroy(d);
	return st;
}

extern "C" nbl_status nbl_create(const nbl_code_desc *code, const uint16_t *gf_mul, const uint16_t *gf_inv,
                                 const nbl_params *params, int device, nbl_decoder **out)
{
	if (!out) return NBL_ERR_ARG;
	*out = nullptr;
	if (!code || !gf_mul || !gf_inv || !params) return fail_create(nullptr, NBL_ERR_ARG, "null argument");
	const int N = code->N, M = code->M, q = code->q;
	if (N <= 0 || M <= 0 || q < 4 || (q & (q - 1))) return fail_create(nullptr, NBL_ERR_ARG, "N, M must be positive and q a power of two, at least 4");
	// (the reference ships arithmetic tables up to GF(512) but no code above GF(256); a valid request this library cannot serve)
	if (q > 256) return fail_create(nullptr, NBL_ERR_UNSUPPORTED, "fields above GF(256) are not supported (one wave holds at most 4 symbols per lane)");
	switch (params->method) {
	case NBL_METHOD_EMS: case NBL_METHOD_BP: case NBL_METHOD_TEMS: break;
	default: return fail_create(nullptr, NBL_ERR_UNSUPPORTED, "decode method not supported (reference: 'has not been developed' / OSD / BS-TEMS)");
	}
	if (params->max_iter < 0) return fail_create(nullptr, NBL_ERR_ARG, "max_iter < 0");
	if (params->method == NBL_METHOD_EMS) {
		// reference: "EMS configuration error! EMS_Nm is too large!" + exit(-1), NBLDPC.cpp:282-286
		if (params->ems_nm > q) return fail_create(nullptr, NBL_ERR_ARG, "EMS configuration error! EMS_Nm is too large!");
		if (params->ems_nm < 1 || params->ems_nc < 0) return fail_create(nullptr, NBL_ERR_ARG, "ems_nm < 1 or ems_nc < 0");
	}
	if (params->method == NBL_METHOD_TEMS && (params->tems_nr < 1 || params->tems_nc < 0))
		return fail_create(nullptr, NBL_ERR_ARG, "tems_nr < 1 or tems_nc < 0");

	// ---- host-side graph indices (NBLDPC.cpp:236-263) -------------------------------------------------------
	std::vector<int> voff(N + 1, 0), coff(M + 1, 0);
	int maxdv = 0, maxdc = 0;
	for (int n = 0; n < N; n++) {
		if (code->var_deg[n] < 1) return fail_create(nullptr, NBL_ERR_ARG, "variable of degree < 1");
		voff[n + 1] = voff[n] + code->var_deg[n];
		if (code->var_deg[n] > maxdv) maxdv = code->var_deg[n];
	}
	for (int m = 0; m < M; m++) {
		if (code->chk_deg[m] < 2) return fail_create(nullptr, NBL_ERR_ARG, "check of degree < 2");
		coff[m + 1] = coff[m] + code->chk_deg[m];
		if (code->chk_deg[m] > maxdc) maxdc = code->chk_deg[m];
	}
	const int E = voff[N];
	if (coff[M] != E) return fail_create(nullptr, NBL_ERR_ARG, "variable-side and check-side edge counts differ");
	if (maxdc > NBL_MAXDC || maxdv > NBL_MAXDV) return fail_create(nullptr, NBL_ERR_ARG, "node degree above the supported maximum (8)");
	std::vector<int> v_cpos(E, -1), c_epos(E, -1), c_var(E), c_h(E), c_hinv(E);
	for (int ce = 0; ce < E; ce++) {
		int n = code->chk_var[ce], h = code->chk_h[ce];
		if (n < 0 || n >= N || h <= 0 || h >= q) return fail_create(nullptr, NBL_ERR_ARG, "check-side edge out of range / zero coefficient");
		c_var[ce] = n;
		c_h[ce] = h;
		c_hinv[ce] = gf_inv[h];
		if (gf_mul[(size_t)h * q + gf_inv[h]] != 1) return fail_create(nullptr, NBL_ERR_ARG, "gf_inv inconsistent with gf_mul");
	}
	for (int n = 0; n < N; n++)
		for (int e = voff[n]; e < voff[n + 1]; e++) {
			int m = code->var_chk[e];
			if (m < 0 || m >= M) return fail_create(nullptr, NBL_ERR_ARG, "variable-side edge out of range");
			for (int ce = coff[m]; ce < coff[m + 1]; ce++)
				if (c_var[ce] == n) v_cpos[e] = ce; // last match wins, like VarLinkDc
			if (v_cpos[e] < 0 || c_h[v_cpos[e]] != code->var_h[e]) return fail_create(nullptr, NBL_ERR_ARG, "variable-side and check-side edge lists disagree");
		}
	for (int m = 0; m < M; m++)
		for (int ce = coff[m]; ce < coff[m + 1]; ce++) {
			int n = c_var[ce];
			for (int e = voff[n]; e < voff[n + 1]; e++)
				if (code->var_chk[e] == m) c_epos[ce] = e; // like ChkLinkDv
			if (c_epos[ce] < 0) return fail_create(nullptr, NBL_ERR_ARG, "check-side edge without variable-side partner");
		}
	const int p = ilog2(q);
	// shape limits of the kernels, refused here rather than at the first decode
	if (params->method == NBL_METHOD_TEMS && p * maxdc > 32)
		return fail_create(nullptr, NBL_ERR_UNSUPPORTED, "T-EMS: log2(q) * (largest check degree) must not exceed 32 (the trellis path code is one 32-bit word)");
	if (params->method == NBL_METHOD_EMS) {
		const int layers = (params->ems_nc >= maxdc - 1) ? 1 : params->ems_nc + 1;
		const size_t lds = ((size_t)maxdc * q + (2 * (size_t)layers + 1) * q + (size_t)maxdc * params->ems_nm) * 8 + (size_t)maxdc * params->ems_nm * 4 + 16;
		const bool special = q == 256 && maxdc == 4 && params->ems_nc >= 1 && params->ems_nm <= 64;
		if (!special && lds > 160 * 1024)
			return fail_create(nullptr, NBL_ERR_UNSUPPORTED, "EMS: this (q, check degree, nm, nc) needs more than the 160 KB of LDS one wave can have");
	}
	// primitive polynomial recovered from the table: x * x^(p-1) = x^p = poly - q
	const int poly = q | gf_mul[(size_t)2 * q + (q >> 1)];
	for (int a = 0; a < q; a++) {
		int expect = (a << 1);
		if (expect & q) expect ^= poly;
		if (gf_mul[(size_t)a * q + 2] != expect) return fail_create(nullptr, NBL_ERR_ARG, "gf_mul is not a polynomial-basis GF(2^p) table");
	}

	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail_create(nullptr, NBL_ERR_NO_DEVICE, "no HIP device (this library has no CPU decode path)");
	if (device < 0 || device >= ndev) return fail_create(nullptr, NBL_ERR_ARG, "device index out of range");

	nbl_decoder *d = new nbl_decoder();
	d->device = device;
	d->prm = *params;
	if (hipSetDevice(device) != hipSuccess) return fail_create(d, NBL_ERR_HIP, "hipSetDevice failed");
	if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) return fail_create(d, NBL_ERR_HIP, "hipStreamCreate failed");
	if (const char *e = getenv("NBL_GRAPH")) d->use_graph = atoi(e) != 0;
	if (const char *e = getenv("NBL_COMPACT")) d->use_compact = atoi(e) != 0;
	d->g.N = N; d->g.M = M; d->g.E = E; d->g.q = q; d->g.p = p; d->g.poly = poly; d->g.maxdc = maxdc; d->g.maxdv = maxdv;
	std::vector<uint8_t> mul8((size_t)q * q);
	for (size_t i = 0; i < mul8.size(); i++) mul8[i] = (uint8_t)gf_mul[i];
	nbl_status st;
	if ((st = upload(d, voff, &d->g.voff)) || (st = upload(d, coff, &d->g.coff)) || (st = upload(d, v_cpos, &d->g.v_cpos)) ||
	    (st = upload(d, c_epos, &d->g.c_epos)) || (st = upload(d, c_var, &d->g.c_var)) || (st = upload(d, c_h, &d->g.c_h)) ||
	    (st = upload(d, c_hinv, &d->g.c_hinv)) || (st = upload(d, mul8, &d->g.mul)))
		return fail_create(d, st, "");
	d->d_e2c_map = (int *)d->g.v_cpos;
	d->all_dc4 = true;
	for (int m = 0; m < M; m++) d->all_dc4 = d->all_dc4 && (code->chk_deg[m] == 4);
	d->min_dc = code->chk_deg[0];
	for (int m = 0; m < M; m++) d->min_dc = code->chk_deg[m] < d->min_dc ? code->chk_deg[m] : d->min_dc;
	d->all_dv2 = true;
	for (int n = 0; n < N; n++) d->all_dv2 = d->all_dv2 && (code->var_deg[n] == 2);
	if (q == 256 && d->all_dc4) {
		// permutation offsets of the specialised EMS kernel: variable-domain symbol a of lane l -> byte offset of h*a in a q-vector
		std::vector<unsigned long long> toff((size_t)E * 64);
		for (int ce = 0; ce < E; ce++)
			for (int l = 0; l < 64; l++) {
				unsigned long long pk = 0;
				for (int i = 0; i < 4; i++) {
					const int a = 2 * l + (i & 1) + 128 * (i >> 1);
					pk |= (unsigned long long)(8u * gf_mul[(size_t)c_h[ce] * q + a]) << (16 * i);
				}
				toff[(size_t)ce * 64 + l] = pk;
			}
		if ((st = upload(d, toff, &d->g.ems_toff))) return fail_create(d, st, "");
	}
	if (d->all_dc4 && d->all_dv2) {
		std::vector<int> row((size_t)M * 16);
		for (int m = 0; m < M; m++)
			for (int j = 0; j < 4; j++) {
				const int ce = coff[m] + j, n = c_var[ce], e = c_epos[ce], e0 = voff[n];
				row[(size_t)m * 16 + j] = n;
				row[(size_t)m * 16 + 4 + j] = v_cpos[e0];
				row[(size_t)m * 16 + 8 + j] = v_cpos[e0 + 1];
				row[(size_t)m * 16 + 12 + j] = e | ((e == e0) ? (int)0x80000000 : 0);
			}
		if ((st = upload(d, row, &d->g.dv2_row))) return fail_create(d, st, "");
	}
	int mindv = maxdv;
	for (int n = 0; n < N; n++) mindv = code->var_deg[n] < mindv ? code->var_deg[n] : mindv;
	if (q <= 64 && maxdv <= 3 && mindv >= 2) {
		// fused small-field iteration: everything the variable-node stage of a check-major edge needs, in one 16-byte row
		// (variable degrees 2 and 3 only: the fused loaders add the second c2v vector unconditionally; a code with a degree-1
		// variable takes the separate variable-node launch, which handles any degree)
		std::vector<int> nbr((size_t)E * 4);
		for (int ce = 0; ce < E; ce++) {
			const int n = c_var[ce], e0 = voff[n], dv = voff[n + 1] - e0;
			for (int k = 0; k < 3; k++) nbr[(size_t)ce * 4 + k] = k < dv ? v_cpos[e0 + k] : -1;
			nbr[(size_t)ce * 4 + 3] = (c_epos[ce] == e0) ? 1 : 0;
		}
		if ((st = upload(d, nbr, &d->g.c_nbr))) return fail_create(d, st, "");
	}
	void *cnt = nullptr;
	if (hipMalloc(&cnt, 16) != hipSuccess) return fail_create(d, NBL_ERR_NOMEM, "hipMalloc failed");
	d->graph_allocs.push_back(cnt);
	d->w.n_done = (int *)cnt;
	d->w.n_act = (int *)cnt + 1;
	if (hipEventCreate(&d->ev[0]) != hipSuccess || hipEventCreate(&d->ev[1]) != hipSuccess) return fail_create(d, NBL_ERR_HIP, "hipEventCreate failed");
	if (params->max_batch > 0 && (st = ensure_workspace(d, params->max_batch))) return fail_create(d, st, "");
	*out = d;
	return NBL_OK;
}

extern "C" void nbl_destroy(nbl_decoder *d)
{
	if (!d) return;
	if (d->device >= 0) (void)hipSetDevice(d->device);
	if (d->stream) { (void)hipStreamSynchronize(d->stream); }
	(void)hipDeviceSynchronize(); // graphs may still be running on a caller's stream
	free_workspace(d);
	for (void *p : d->graph_allocs) (void)hipFree(p);
	if (d->d_src) (void)hipFree(d->d_src);
	if (d->d_cons) (void)hipFree(d->d_cons);
	if (d->d_rx) (void)hipFree(d->d_rx);
	for (double *p : d->d_rxs)
		if (p) (void)hipFree(p);
	if (d->stream2) { (void)hipStreamSynchronize(d->stream2); (void)hipStreamDestroy(d->stream2); }
	for (void *p : {(void *)d->d_jump, (void *)d->d_state, (void *)d->d_txi, (void *)d->d_fn, (void *)d->d_fidx, (void *)d->d_farg, (void *)d->d_fval, (void *)d->d_fcount})
		if (p) (void)hipFree(p);
	for (void *p : {(void *)d->h_fidx, (void *)d->h_farg, (void *)d->h_fval})
		if (p) (void)hipHostFree(p);
	for (auto &e : d->ev)
		if (e) (void)hipEventDestroy(e);
	if (d->h_ndone) (void)hipHostFree(d->h_ndone);
	for (auto &e : d->pev) (void)hipEventDestroy(e);
	if (d->stream) (void)hipStreamDestroy(d->stream);
	delete d;
}

extern "C" nbl_status nbl_set_profiling(nbl_decoder *d, int32_t on)
{
	if (!d) return NBL_ERR_ARG;
	d->profiling = on != 0;
	return NBL_OK;
}

// Diagnostic only (not part of include/nbldpc.h): route every shape through the generic kernels.
extern "C" nbl_status nbl_debug_force_generic(nbl_decoder *d, int32_t on)
{
	if (!d) return NBL_ERR_ARG;
	d->force_generic = on;
	return NBL_OK;
}

// Diagnostic only (not part of include/nbldpc.h): number of iteration windows currently held as executable hipGraphs
extern "C" int32_t nbl_debug_graph_windows(nbl_decoder *d)
{
	int n = 0;
	if (d)
		for (auto &ge : d->gexec) n += ge != nullptr;
	return n;
}

// Diagnostic only (not part of include/nbldpc.h): in-kernel cycle stamps of the check-node kernel.
extern "C" nbl_status nbl_debug_stamps(nbl_decoder *d, int32_t on, unsigned long long out[16])
{
	if (!d) return NBL_ERR_ARG;
	d->err.clear();
	HIP_TRY(d, hipSetDevice(d->device));
	if (out && d->w.stamps) {
		HIP_TRY(d, hipStreamSynchronize(d->stream));
		HIP_TRY(d, hipDeviceSynchronize());
		HIP_TRY(d, hipMemcpy(out, d->w.stamps, 16 * 8, hipMemcpyDeviceToHost));
	}
	if (on && !d->w.stamps) {
		void *p = nullptr;
		HIP_TRY(d, hipMalloc(&p, 16 * 8));
		d->graph_allocs.push_back(p);
		d->w.stamps = (unsigned long long *)p;
	}
	if (d->w.stamps && on) HIP_TRY(d, hipMemset(d->w.stamps, 0, 16 * 8));
	if (!on) d->w.stamps = nullptr;
	return NBL_OK;
}

extern "C" nbl_status nbl_set_record_state(nbl_decoder *d, int32_t on)
{
	if (!d) return NBL_ERR_ARG;
	d->record_state = on != 0;
	return NBL_OK;
}

extern "C" nbl_status nbl_last_timing(nbl_decoder *d, double ms[4], int64_t launches[3])
{
	if (!d) return NBL_ERR_ARG;
	if (ms) memcpy(ms, d->ms, sizeof d->ms);
	if (launches) for (int i = 0; i < 3; i++) launches[i] = d->launches[i];
	return NBL_OK;
}

static nbl_status launch_cn(nbl_decoder *d, const NblRun &r, hipStream_t st)
{
	const bool small_on = small_shape(d);
	switch (d->prm.method) {
	case NBL_METHOD_EMS:
		if (d->force_generic != 1 && nbl_ems256_applicable(d->g, d->all_dc4, r.nm, r.nc)) HIP_TRY(d, nbl_launch_cn_ems256(d->g, d->w, r, false, st));
		else if (d->force_generic != 1 && small_on) HIP_TRY(d, nbl_launch_cn_ems_small(d->g, d->w, r, false, st));
		else if (d->force_generic != 1 && ems64_shape(d)) HIP_TRY(d, nbl_launch_cn_ems64(d->g, d->w, r, false, st));
		else HIP_TRY(d, nbl_launch_cn_ems(d->g, d->w, r, st));
		break;
	case NBL_METHOD_TEMS:
		if (d->force_generic != 1 && nbl_tems64_applicable(d->g, d->all_dc4, r.nr, r.nc)) HIP_TRY(d, nbl_launch_cn_tems64(d->g, d->w, r, false, st));
		else if (d->force_generic != 1 && nbl_tems256_applicable(d->g, d->all_dc4, r.nr, r.nc)) HIP_TRY(d, nbl_launch_cn_tems256(d->g, d->w, r, false, st));
		else if (d->force_generic != 1 && small_on) HIP_TRY(d, nbl_launch_cn_tems_small(d->g, d->w, r, false, st));
		else HIP_TRY(d, nbl_launch_cn_tems(d->g, d->w, r, st));
		break;
	case NBL_METHOD_BP:
		if (d->force_generic != 1 && nbl_bp256_applicable(d->g, d->all_dc4)) HIP_TRY(d, nbl_launch_cn_bp256(d->g, d->w, r, false, st));
		else if (d->force_generic != 1 && nbl_bp64_applicable(d->g, d->all_dc4)) HIP_TRY(d, nbl_launch_cn_bp64(d->g, d->w, r, false, st));
		else if (d->force_generic != 1 && small_on) HIP_TRY(d, nbl_launch_cn_bp_small(d->g, d->w, r, false, st));
		else HIP_TRY(d, nbl_launch_cn_bp(d->g, d->w, r, st));
		break;
	default: d->err = "check-node kernel for this method is not built yet"; return NBL_ERR_UNSUPPORTED;
	}
	return NBL_OK;
}

// The iteration loop of Decoding_BP / _EMS / _TEMS (NBLDPC.cpp:673 / 805 / 973): per iteration one fused launch + syndrome
// (specialised (2,4)-regular shapes) or the launch triple VN, syndrome, CN.  The launches of a window of iterations are captured
// into a hipGraph once and replayed (a decode is 100+ back-to-back launches: at small batches the gaps between them are most
// of the time).
struct IterCtx {
	nbl_decoder *d;
	NblRun r;
	bool damp, fused;
	double *bufA, *bufB;
	const double *zeros = nullptr; // stands in for bufA in iteration 1 (then bufA needs no clearing)
	// profiling: one event after every launch on the launch stream; phase time = sum of the gaps it closes
	size_t nev = 0;
	std::vector<int> tag; // 0 vn, 1 syn, 2 cn, 3 other
};

static hipError_t mark(IterCtx &c, int t, hipStream_t st)
{
	nbl_decoder *d = c.d;
	if (!d->profiling) return hipSuccess;
	if (c.nev == d->pev.size()) { hipEvent_t e; hipError_t rc = hipEventCreate(&e); if (rc != hipSuccess) return rc; d->pev.push_back(e); }
	c.tag.push_back(t);
	return hipEventRecord(d->pev[c.nev++], st);
}

// launches of iterations it_lo .. it_hi on `st`
static nbl_status enqueue_window(IterCtx &c, int it_lo, int it_hi, hipStream_t st, bool count)
{
	nbl_decoder *d = c.d;
	const nbl_params &p = d->prm;
	for (int it = it_lo; it <= it_hi; it++) {
		c.r.iter = it;
		if (c.fused) {
			// one launch = variable-node pass + check-node pass; c2v ping-pongs between the two buffers
			NblWork wf = d->w;
			wf.c2v_prev = (it == 1 && c.zeros) ? c.zeros : (it & 1) ? c.bufA : c.bufB;
			wf.c2v_prev_shared = (it == 1 && c.zeros) ? 1 : 0;
			wf.c2v = (it & 1) ? c.bufB : c.bufA;
			wf.store_v2c = d->record_state ? 1 : 0;
			if (small_shape(d)) {
				if (p.method == NBL_METHOD_EMS) HIP_TRY(d, nbl_launch_cn_ems_small(d->g, wf, c.r, true, st));
				else if (p.method == NBL_METHOD_TEMS) HIP_TRY(d, nbl_launch_cn_tems_small(d->g, wf, c.r, true, st));
				else HIP_TRY(d, nbl_launch_cn_bp_small(d->g, wf, c.r, true, st));
			} else if (p.method == NBL_METHOD_EMS && d->g.q == 64) HIP_TRY(d, nbl_launch_cn_ems64(d->g, wf, c.r, true, st));
			else if (p.method == NBL_METHOD_EMS) HIP_TRY(d, nbl_launch_cn_ems256(d->g, wf, c.r, true, st));
			else if (p.method == NBL_METHOD_TEMS && d->g.q == 64) HIP_TRY(d, nbl_launch_cn_tems64(d->g, wf, c.r, true, st));
			else if (p.method == NBL_METHOD_TEMS) HIP_TRY(d, nbl_launch_cn_tems256(d->g, wf, c.r, true, st));
			else if (d->g.q == 64) HIP_TRY(d, nbl_launch_cn_bp64(d->g, wf, c.r, true, st));
			else HIP_TRY(d, nbl_launch_cn_bp256(d->g, wf, c.r, true, st));
			HIP_TRY(d, mark(c, 2, st));
			HIP_TRY(d, nbl_launch_syn(d->g, d->w, c.r, st));
			HIP_TRY(d, mark(c, 1, st));
			if (count) { d->launches[2]++; d->launches[1]++; }
			continue;
		}
		HIP_TRY(d, nbl_launch_vn(d->g, d->w, c.r, c.damp, st));
		HIP_TRY(d, mark(c, 0, st));
		HIP_TRY(d, nbl_launch_syn(d->g, d->w, c.r, st));
		HIP_TRY(d, mark(c, 1, st));
		if (count) { d->launches[0]++; d->launches[1]++; }
		// (the reference leaves the loop after the syndrome check of the last iteration it runs; the check-node pass of a
		// window's last iteration is only needed if another window follows -- it is cheap to keep the windows uniform)
		nbl_status s = launch_cn(d, c.r, st);
		if (s) return s;
		HIP_TRY(d, mark(c, 2, st));
		if (count) d->launches[2]++;
	}
	return NBL_OK;
}

// Run window number `widx` (iterations it_lo..it_hi): replay its graph, capturing it first if needed; plain launches when
// graphs are off, while profiling (events between the launches) or if the capture fails.
static nbl_status run_window(IterCtx &c, int widx, int it_lo, int it_hi, hipStream_t st)
{
	nbl_decoder *d = c.d;
	const bool graph_ok = d->use_graph && !d->profiling && !d->w.stamps;
	if (graph_ok) {
		if ((int)d->gexec.size() <= widx) d->gexec.resize(widx + 1, nullptr);
		if (!d->gexec[widx]) {
			hipGraph_t graph = nullptr;
			if (hipStreamBeginCapture(d->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
				const nbl_status s = enqueue_window(c, it_lo, it_hi, d->stream, false);
				const hipError_t e = hipStreamEndCapture(d->stream, &graph);
				if (s == NBL_OK && e == hipSuccess && graph) {
					if (hipGraphInstantiate(&d->gexec[widx], graph, nullptr, nullptr, 0) != hipSuccess) d->gexec[widx] = nullptr;
				}
				if (graph) (void)hipGraphDestroy(graph);
				(void)hipGetLastError();
			}
			if (!d->gexec[widx]) d->use_graph = false; // capture is not available here: plain launches from now on
		}
		if (d->gexec[widx]) {
			HIP_TRY(d, hipGraphLaunch(d->gexec[widx], st));
			const int n = it_hi - it_lo + 1;
			d->launches[1] += n; d->launches[2] += n;
			if (!c.fused) d->launches[0] += n;
			return NBL_OK;
		}
	}
	return enqueue_window(c, it_lo, it_hi, st, true);
}

static nbl_status run_iterations(nbl_decoder *d, const double *d_Lin, int B, hipStream_t st)
{
	const nbl_params &p = d->prm;
	IterCtx c;
	c.d = d;
	c.damp = p.method != NBL_METHOD_EMS;
	NblRun &r = c.r;
	r = NblRun{};
	r.B = B;
	r.fixed_iters = p.fixed_iters;
	if (p.method == NBL_METHOD_EMS) { r.nm = p.ems_nm; r.nc = p.ems_nc; r.factor = p.ems_factor; r.offset = p.ems_offset; }
	else { r.nr = p.tems_nr; r.nc = p.tems_nc; r.factor = p.tems_factor; r.offset = p.tems_offset; }
	r.damp_old = (p.method == NBL_METHOD_BP) ? 0.5 : 0.25;  // NBLDPC.cpp:739 / :1046
	r.damp_new = (p.method == NBL_METHOD_BP) ? 0.5 : 0.75;
	d->launches[0] = d->launches[1] = d->launches[2] = 0;
	c.fused = fused_shape(d) && d->force_generic == 0 && d->c2v_alt;
	c.bufA = d->w.c2v;
	c.bufB = d->c2v_alt;
	c.zeros = c.fused ? d->c2v_zero : nullptr;
	// the captured graphs hold buffer addresses and the batch size: any change drops them
	const nbl_decoder::GraphKey key = {d_Lin, d->w.Lch, d->w.v2c, d->w.c2v, d->c2v_alt, d->w.post, B, d->record_state ? 1 : 0, d->force_generic, c.fused ? 1 : 0};
	if (memcmp(&key, &d->gkey, sizeof key) != 0) { drop_graphs(d); d->gkey = key; }
	HIP_TRY(d, mark(c, 3, st));
	HIP_TRY(d, nbl_launch_init(d_Lin, d->g, d->w, B, (c.damp ? 1 : 0) | (c.zeros ? 2 : 0), st)); // bit 1: c2v is not cleared
	HIP_TRY(d, mark(c, 3, st));
	d->last_c2v = c.zeros ? c.zeros : c.bufA;
	d->last_fused = c.fused;
	// windows: fixed iterations or no polling -> one window; early exit -> `poll_every` iterations, then ask the device
	const bool polling = !p.fixed_iters && p.poll_every > 0;
	const int wlen = polling ? p.poll_every : (p.max_iter > 0 ? p.max_iter : 1);
	int last_it = 0;
	if (polling && !d->h_ndone) HIP_TRY(d, hipHostMalloc((void **)&d->h_ndone, 2 * sizeof(int), hipHostMallocDefault));
	// Early exit without idling the GPU: window w+1 is queued BEFORE the host looks at the count of converged codewords that
	// window w left behind (read back through pinned memory behind an event).  If everything had converged, the extra window
	// finds every codeword frozen and its kernels return at once; outputs, flags and iteration counts are unaffected.
	// Large batches: after every window the device rebuilds the list of codewords still iterating, and the grids of the next
	// windows cover that list -- sized by the newest converged count the host has seen, an upper bound of the list's length --
	// instead of the whole batch (most codewords of a waterfall batch are done long before the stragglers).
	const bool compact = polling && d->use_compact && !d->use_graph && B >= 1024;
	d->w.active = nullptr;
	int pending = -1; // parity of the read-back that has not been looked at yet
	for (int it_lo = 1, widx = 0; it_lo <= p.max_iter; it_lo += wlen, widx++) {
		const int it_hi = (it_lo + wlen - 1 < p.max_iter) ? it_lo + wlen - 1 : p.max_iter;
		nbl_status s = run_window(c, widx, it_lo, it_hi, st);
		if (s) { d->w.active = nullptr; return s; }
		last_it = it_hi;
		if (polling) {
			const int par = widx & 1;
			HIP_TRY(d, hipMemcpyAsync(&d->h_ndone[par], d->w.n_done, sizeof(int), hipMemcpyDeviceToHost, st));
			HIP_TRY(d, hipEventRecord(d->ev[par], st));
			if (compact) {
				HIP_TRY(d, nbl_launch_compact(d->w.done, B, d->d_active, (int *)d->w.n_act, st));
				d->w.active = d->d_active;
			}
			if (pending >= 0) {
				HIP_TRY(d, hipEventSynchronize(d->ev[pending]));
				if (d->h_ndone[pending] >= B) break;
				if (compact) c.r.B = B - d->h_ndone[pending];
			}
			pending = par;
		}
	}
	d->w.active = nullptr;
	if (polling) HIP_TRY(d, hipStreamSynchronize(st));
	if (c.fused && last_it > 0) d->last_c2v = (last_it & 1) ? c.bufB : c.bufA;
	if (d->profiling && c.nev > 0) {
		HIP_TRY(d, hipEventSynchronize(d->pev[c.nev - 1]));
		d->ms[0] = d->ms[1] = d->ms[2] = d->ms[3] = 0;
		for (size_t i = 1; i < c.nev; i++) {
			float ms = 0;
			HIP_TRY(d, hipEventElapsedTime(&ms, d->pev[i - 1], d->pev[i]));
			if (c.tag[i] < 3) d->ms[c.tag[i]] += ms;
			d->ms[3] += ms;
		}
	}
	d->last_B = B;
	return NBL_OK;
}

extern "C" nbl_status nbl_decode_batch_device(nbl_decoder *d, const double *d_L_ch, int32_t B, int32_t *d_out_sym,
                                              uint8_t *d_converged, int32_t *d_iters, void *stream)
{
	if (!d || !d_L_ch || !d_out_sym || B < 0) return NBL_ERR_ARG;
	if (B == 0) return NBL_OK;
	d->err.clear();
	HIP_TRY(d, hipSetDevice(d->device));
	hipStream_t st = stream ? (hipStream_t)stream : d->stream;
	nbl_status s = ensure_workspace(d, B);
	if (s) return s;
	if ((s = run_iterations(d, d_L_ch, B, st))) return s;
	HIP_TRY(d, hipMemcpyAsync(d_out_sym, d->w.out, (size_t)B * d->g.N * 4, hipMemcpyDeviceToDevice, st));
	if (d_converged) HIP_TRY(d, hipMemcpyAsync(d_converged, d->w.done, (size_t)B, hipMemcpyDeviceToDevice, st));
	if (d_iters) HIP_TRY(d, hipMemcpyAsync(d_iters, d->w.iters, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
	return NBL_OK;
}

extern "C" nbl_status nbl_decode_batch(nbl_decoder *d, const double *L_ch, int32_t B, int32_t *out_sym, uint8_t *converged,
                                       int32_t *iters)
{
	if (!d || !L_ch || !out_sym || B < 0) return NBL_ERR_ARG;
	if (B == 0) return NBL_OK;
	d->err.clear();
	HIP_TRY(d, hipSetDevice(d->device));
	nbl_status s = ensure_workspace(d, B);
	if (s) return s;
	const size_t in_bytes = (size_t)B * d->g.N * (d->g.q - 1) * 8;
	if (!d->d_Lin) HIP_TRY(d, hipMalloc((void **)&d->d_Lin, (size_t)d->cap * d->g.N * (d->g.q - 1) * 8));
	HIP_TRY(d, hipMemcpyAsync(d->d_Lin, L_ch, in_bytes, hipMemcpyHostToDevice, d->stream));
	if ((s = run_iterations(d, d->d_Lin, B, d->stream))) return s;
	HIP_TRY(d, hipMemcpyAsync(out_sym, d->w.out, (size_t)B * d->g.N * 4, hipMemcpyDeviceToHost, d->stream));
	if (converged) HIP_TRY(d, hipMemcpyAsync(converged, d->w.done, (size_t)B, hipMemcpyDeviceToHost, d->stream));
	if (iters) HIP_TRY(d, hipMemcpyAsync(iters, d->w.iters, (size_t)B * 4, hipMemcpyDeviceToHost, d->stream));
	HIP_TRY(d, hipStreamSynchronize(d->stream));
	return NBL_OK;
}

extern "C" nbl_status nbl_set_demodulator(nbl_decoder *d, const nbl_demod_desc *dm)
{
	if (!d || !dm || !dm->src) return NBL_ERR_ARG;
	const int q = d->g.q, N = d->g.N, p = d->g.p;
	if (dm->mod_order != 2 && dm->mod_order != q) {
		d->err = "This module ( code order ~= modulation order ) haven't been developed!"; // Comm.cpp:400-404
		return NBL_ERR_UNSUPPORTED;
	}
	if (dm->n_mod_sym <= 0 || (dm->mod_order == q && !dm->constellation)) return NBL_ERR_ARG;
	d->err.clear();
	HIP_TRY(d, hipSetDevice(d->device));
	const size_t nsrc = dm->mod_order == 2 ? (size_t)N * p : (size_t)N;
	for (size_t i = 0; i < nsrc; i++)
		if (dm->src[i] >= dm->n_mod_sym) { d->err = "demodulator source index out of range"; return NBL_ERR_ARG; }
	if (d->d_src) (void)hipFree(d->d_src);
	if (d->d_cons) (void)hipFree(d->d_cons);
	d->d_src = nullptr; d->d_cons = nullptr;
	HIP_TRY(d, hipMalloc((void **)&d->d_src, nsrc * 4));
	HIP_TRY(d, hipMemcpy(d->d_src, dm->src, nsrc * 4, hipMemcpyHostToDevice));
	if (dm->mod_order == q) {
		HIP_TRY(d, hipMalloc((void **)&d->d_cons, (size_t)q * 16));
		HIP_TRY(d, hipMemcpy(d->d_cons, dm->constellation, (size_t)q * 16, hipMemcpyHostToDevice));
	}
	d->h_cons.clear();
	if (dm->constellation) d->h_cons.assign(dm->constellation, dm->constellation + (size_t)2 * dm->mod_order);
	if (!d->d_cons && dm->constellation) { // BPSK: the demodulator does not need the points, the channel does
		HIP_TRY(d, hipMalloc((void **)&d->d_cons, (size_t)dm->mod_order * 16));
		HIP_TRY(d, hipMemcpy(d->d_cons, dm->constellation, (size_t)dm->mod_order * 16, hipMemcpyHostToDevice));
	}
	// the channel's buffers are sized per lane of dm_L symbols and the jump table is per symbol position: both are rebuilt for
	// the new L by the next channel call
	if (d->d_jump) { (void)hipFree(d->d_jump); d->d_jump = nullptr; }
	d->noise_cap = 0;
	d->dm_order = dm->mod_order;
	d->dm_L = dm->n_mod_sym;
	return NBL_OK;
}

static nbl_status run_iterations(nbl_decoder *d, const double *d_Lin, int B, hipStream_t st);

extern "C" nbl_status nbl_decode_batch_samples(nbl_decoder *d, const double *rx, double sigma, int32_t B, int32_t *out_sym,
                                               uint8_t *converged, int32_t *iters)
{
	if (!d || !rx || !out_sym || B < 0 || !(sigma > 0)) return NBL_ERR_ARG;
	if (!d->dm_order) { d->err = "nbl_set_demodulator has not been called"; return NBL_ERR_ARG; }
	if (B == 0) return NBL_OK;
	d->err.clear();
	HIP_TRY(d, hipSetDevice(d->device));
	nbl_status s = ensure_workspace(d, B);
	if (s) return s;
	const size_t bytes = (size_t)B * d->dm_L * 16;
	if (bytes > d->d_rx_cap) {
		if (d->d_rx) (void)hipFree(d->d_rx);
		d->d_rx = nullptr;
		HIP_TRY(d, hipMalloc((void **)&d->d_rx, bytes));
		d->d_rx_cap = bytes;
	}
	HIP_TRY(d, hipMemcpyAsync(d->d_rx, rx, bytes, hipMemcpyHostToDevice, d->stream));
	HIP_TRY(d, nbl_launch_demod(d->d_rx, d->dm_L, sigma, d->dm_order, d->d_cons, d->d_src, d->g, d->w, B, d->stream));
	if ((s = run_iterations(d, nullptr, B, d->stream))) return s;
	HIP_TRY(d, hipMemcpyAsync(out_sym, d->w.out, (size_t)B * d->g.N * 4, hipMemcpyDeviceToHost, d->stream));
	if (converged) HIP_TRY(d, hipMemcpyAsync(converged, d->w.done, (size_t)B, hipMemcpyDeviceToHost, d->stream));
	if (iters) HIP_TRY(d, hipMemcpyAsync(iters, d->w.iters, (size_t)B * 4, hipMemcpyDeviceToHost, d->stream));
	HIP_TRY(d, hipStreamSynchronize(d->stream));
	return NBL_OK;
}


// ---- AWGN channel + CRand on the device ------------------------------------------------------------------

static uint32_t mod_pow(uint32_t a, uint64_t k, uint32_t m)
{
	uint64_t r = 1 % m, x = a % m;
	for (; k; k >>= 1) {
		if (k & 1) r = r * x % m;
		x = x * x % m;
	}
	return (uint32_t)r;
}

extern "C" void nbl_rand_advance(uint32_t state[3], uint64_t draws)
{
	state[0] = (uint32_t)((uint64_t)(state[0] % 61967u) * mod_pow(249, draws, 61967) % 61967u);
	state[1] = (uint32_t)((uint64_t)(state[1] % 63443u) * mod_pow(251, draws, 63443) % 63443u);
	state[2] = (uint32_t)((uint64_t)(state[2] % 63599u) * mod_pow(252, draws, 63599) % 63599u);
}

static nbl_status ensure_noise(nbl_decoder *d, int B, std::string &err)
{
	const size_t L = d->dm_L;
	if (!d->d_jump) {
		std::vector<uint32_t> jump(3 * L);
		const uint32_t A[3] = {249, 251, 252}, M[3] = {61967, 63443, 63599};
		for (int g = 0; g < 3; g++) {
			uint64_t x = 1;
			const uint32_t a4 = mod_pow(A[g], 4, M[g]);
			for (size_t s = 0; s < L; s++) { jump[g * L + s] = (uint32_t)x; x = x * a4 % M[g]; }
		}
		HIP_TRY_E(err, hipMalloc((void **)&d->d_jump, jump.size() * 4));
		HIP_TRY_E(err, hipMemcpy(d->d_jump, jump.data(), jump.size() * 4, hipMemcpyHostToDevice));
	}
	if ((size_t)B <= d->noise_cap) return NBL_OK; // (capacity in lanes of dm_L symbols; nbl_set_demodulator resets it)
	for (void *p : {(void *)d->d_state, (void *)d->d_txi, (void *)d->d_fn, (void *)d->d_fidx, (void *)d->d_farg, (void *)d->d_fval})
		if (p) (void)hipFree(p);
	for (void *p : {(void *)d->h_fidx, (void *)d->h_farg, (void *)d->h_fval})
		if (p) (void)hipHostFree(p);
	d->d_state = nullptr; d->d_txi = nullptr; d->d_fn = nullptr; d->d_fidx = nullptr; d->d_farg = d->d_fval = nullptr;
	d->h_fidx = nullptr; d->h_farg = d->h_fval = nullptr;
	d->noise_cap = 0;
	const size_t nval = (size_t)B * L * 4; // two functions per normal draw, two draws per symbol
	if (nval > 0xffffffffull) { err = "nbl_decode_batch_noise: batch * symbols too large for 32-bit value indices"; return NBL_ERR_ARG; }
	// about 16 % of the values are uncertain (5 % of the logarithms, 11 % of the cosines); room for 30 %
	const size_t cap = nval * 3 / 10 + 4096;
	HIP_TRY_E(err, hipMalloc((void **)&d->d_state, (size_t)B * 12));
	HIP_TRY_E(err, hipMalloc((void **)&d->d_txi, (size_t)B * L));
	HIP_TRY_E(err, hipMalloc((void **)&d->d_fn, nval * 8));
	HIP_TRY_E(err, hipMalloc((void **)&d->d_fidx, cap * 4));
	HIP_TRY_E(err, hipMalloc((void **)&d->d_farg, cap * 8));
	HIP_TRY_E(err, hipMalloc((void **)&d->d_fval, cap * 8));
	if (!d->d_fcount) HIP_TRY_E(err, hipMalloc((void **)&d->d_fcount, 16));
	HIP_TRY_E(err, hipHostMalloc((void **)&d->h_farg, cap * 8, hipHostMallocDefault));
	HIP_TRY_E(err, hipHostMalloc((void **)&d->h_fval, cap * 8, hipHostMallocDefault));
	d->noise_cap = B;
	d->flag_cap = cap;
	return NBL_OK;
}

// Forms RX = TX + noise for B lanes in *rx_buf (grown on demand) on stream `st`: the three kernels of nbl_noise.hip with the host's
// libm in between.  Returns with the samples complete in HBM.
static nbl_status run_channel(nbl_decoder *d, const uint8_t *tx_index, const uint32_t *lane_state, double sigma, int B, hipStream_t st,
                              double **rx_buf, size_t *rx_cap, std::string &err)
{
	if (!d->dm_order) { err = "nbl_set_demodulator has not been called"; return NBL_ERR_ARG; }
	if (d->h_cons.empty() || !d->d_cons) { err = "the channel needs the constellation points (nbl_demod_desc.constellation), also for BPSK"; return NBL_ERR_ARG; }
	nbl_status s = ensure_noise(d, B, err);
	if (s) return s;
	const size_t L = d->dm_L;
	if (d->dm_order < 256) { // an index beyond the constellation would read past d_cons in the finish kernel (8 bytes per step:
		// the modulation orders are powers of two, so "some byte >= order" is "some bit above the order's bits is set")
		const size_t n = (size_t)B * L;
		const uint8_t hi = (uint8_t)~(d->dm_order - 1);
		uint64_t m8 = 0, any = 0;
		for (int k = 0; k < 8; k++) m8 = (m8 << 8) | hi;
		size_t i = 0;
		for (; i + 8 <= n; i += 8) { uint64_t w8; memcpy(&w8, tx_index + i, 8); any |= w8 & m8; }
		for (; i < n; i++) any |= (uint64_t)(tx_index[i] & hi);
		if (any || (d->dm_order & (d->dm_order - 1))) {
			for (size_t k = 0; k < n; k++)
				if ((int)tx_index[k] >= d->dm_order) { err = "tx_index holds a value >= mod_order"; return NBL_ERR_ARG; }
		}
	}
	const size_t bytes = (size_t)B * L * 16;
	if (bytes > *rx_cap) {
		if (*rx_buf) (void)hipFree(*rx_buf);
		*rx_buf = nullptr;
		HIP_TRY_E(err, hipMalloc((void **)rx_buf, bytes));
		*rx_cap = bytes;
	}
	const bool timing = getenv("NBL_CHANNEL_TIMING") != nullptr;
	auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t0 = now();
	double t1 = t0, t2 = t0, t3 = t0;
	HIP_TRY_E(err, hipMemcpyAsync(d->d_state, lane_state, (size_t)B * 12, hipMemcpyHostToDevice, st));
	HIP_TRY_E(err, hipMemcpyAsync(d->d_txi, tx_index, (size_t)B * L, hipMemcpyHostToDevice, st));
	HIP_TRY_E(err, hipMemsetAsync(d->d_fcount, 0, 4, st));
	HIP_TRY_E(err, nbl_launch_noise_gen(d->d_state, d->d_jump, (int)L, B, d->d_fn, d->d_fidx, d->d_farg, d->d_fcount, (unsigned)d->flag_cap, st));
	unsigned nflag = 0;
	HIP_TRY_E(err, hipMemcpyAsync(&nflag, d->d_fcount, 4, hipMemcpyDeviceToHost, st));
	HIP_TRY_E(err, hipStreamSynchronize(st));
	if (nflag > d->flag_cap) { err = "nbl_decode_batch_noise: more uncertain values than the list holds (30 % of all)"; return NBL_ERR_NOMEM; }
	d->last_flag_frac = (double)nflag / ((double)B * L * 4);
	t1 = now();
	if (nflag) {
		HIP_TRY_E(err, hipMemcpyAsync(d->h_farg, d->d_farg, (size_t)nflag * 8, hipMemcpyDeviceToHost, st));
		HIP_TRY_E(err, hipStreamSynchronize(st));
		t2 = now();
		// the host's own libm decides the uncertain values: log(1 - u1) or cos(2 pi u2), Rand.cpp:35
		int T = (int)std::thread::hardware_concurrency();
		if (const char *e = getenv("NBL_HOST_THREADS")) T = atoi(e);
		if (T > 16) T = 16;
		if (T < 1 || nflag < 4096) T = 1;
		auto work = [&](unsigned lo, unsigned hi) {
			// arguments of the logarithm are 1 - u1 in (0, 1]; arguments of the cosine are 2 pi u2 in [0, 2 pi): the list marks a
			// cosine argument by its sign bit (the kernel stores -x, and -0.0 for x = 0), so only the arguments travel
			for (unsigned k = lo; k < hi; k++) {
				const double a = d->h_farg[k];
				d->h_fval[k] = std::signbit(a) ? std::cos(-a) : std::log(a);
			}
		};
		if (T == 1) work(0, nflag);
		else {
			std::vector<std::thread> th;
			for (int t = 0; t < T; t++) th.emplace_back(work, (unsigned)((uint64_t)nflag * t / T), (unsigned)((uint64_t)nflag * (t + 1) / T));
			for (auto &x : th) x.join();
		}
		t3 = now();
		HIP_TRY_E(err, hipMemcpyAsync(d->d_fval, d->h_fval, (size_t)nflag * 8, hipMemcpyHostToDevice, st));
		HIP_TRY_E(err, nbl_launch_noise_patch(d->d_fn, d->d_fidx, d->d_fval, nflag, st));
	}
	HIP_TRY_E(err, nbl_launch_noise_finish(d->d_fn, d->d_txi, d->d_cons, sigma, (int)L, B, *rx_buf, st));
	HIP_TRY_E(err, hipStreamSynchronize(st));
	if (timing)
		fprintf(stderr, "[channel] B=%d: generate %.2f ms, list to host %.2f ms, host libm (%u values) %.2f ms, patch + finish %.2f ms\n", B,
		        (t1 - t0) * 1e3, (t2 - t1) * 1e3, nflag, (t3 - t2) * 1e3, (now() - t3) * 1e3);
	return NBL_OK;
}

extern "C" nbl_status nbl_decode_batch_noise(nbl_decoder *d, const uint8_t *tx_index, const uint32_t *lane_state, double sigma, int32_t B,
                                             int32_t *out_sym, uint8_t *converged, int32_t *iters)
{
	if (!d || !tx_index || !lane_state || !out_sym || B < 0 || !(sigma > 0)) return NBL_ERR_ARG;
	if (B == 0) return NBL_OK;
	d->err.clear();
	HIP_TRY(d, hipSetDevice(d->device));
	nbl_status s = ensure_workspace(d, B);
	if (s) return s;
	if ((s = run_channel(d, tx_index, lane_state, sigma, B, d->stream, &d->d_rx, &d->d_rx_cap, d->err))) return s;
	HIP_TRY(d, nbl_launch_demod(d->d_rx, d->dm_L, sigma, d->dm_order, d->d_cons, d->d_src, d->g, d->w, B, d->stream));
	if ((s = run_iterations(d, nullptr, B, d->stream))) return s;
	HIP_TRY(d, hipMemcpyAsync(out_sym, d->w.out, (size_t)B * d->g.N * 4, hipMemcpyDeviceToHost, d->stream));
	if (converged) HIP_TRY(d, hipMemcpyAsync(converged, d->w.done, (size_t)B, hipMemcpyDeviceToHost, d->stream));
	if (iters) HIP_TRY(d, hipMemcpyAsync(iters, d->w.iters, (size_t)B * 4, hipMemcpyDeviceToHost, d->stream));
	HIP_TRY(d, hipStreamSynchronize(d->stream));
	return NBL_OK;
}

// Two-phase form: the channel of batch k+1 (second stream, host libm) may run on another host thread while batch k is decoded.
extern "C" nbl_status nbl_channel_batch(nbl_decoder *d, int32_t slot, const uint8_t *tx_index, const uint32_t *lane_state, double sigma, int32_t B)
{
	if (!d || !tx_index || !lane_state || B <= 0 || slot < 0 || slot > 1 || !(sigma > 0)) return NBL_ERR_ARG;
	d->err2.clear();
	HIP_TRY_E(d->err2, hipSetDevice(d->device));
	if (!d->stream2) HIP_TRY_E(d->err2, hipStreamCreateWithFlags(&d->stream2, hipStreamNonBlocking));
	d->rxs_B[slot] = 0;
	const nbl_status s = run_channel(d, tx_index, lane_state, sigma, B, d->stream2, &d->d_rxs[slot], &d->d_rxs_cap[slot], d->err2);
	if (s == NBL_OK) d->rxs_B[slot] = B;
	return s;
}

extern "C" nbl_status nbl_decode_batch_resident(nbl_decoder *d, int32_t slot, double sigma, int32_t B, int32_t *out_sym, uint8_t *converged, int32_t *iters)
{
	if (!d || !out_sym || B <= 0 || slot < 0 || slot > 1 || !(sigma > 0)) return NBL_ERR_ARG;
	if (d->rxs_B[slot] != B) { d->err = "nbl_decode_batch_resident: slot does not hold the samples of a batch of this size (nbl_channel_batch first)"; return NBL_ERR_ARG; }
	d->err.clear();
	HIP_TRY(d, hipSetDevice(d->device));
	nbl_status s = ensure_workspace(d, B);
	if (s) return s;
	HIP_TRY(d, nbl_launch_demod(d->d_rxs[slot], d->dm_L, sigma, d->dm_order, d->d_cons, d->d_src, d->g, d->w, B, d->stream));
	if ((s = run_iterations(d, nullptr, B, d->stream))) return s;
	HIP_TRY(d, hipMemcpyAsync(out_sym, d->w.out, (size_t)B * d->g.N * 4, hipMemcpyDeviceToHost, d->stream));
	if (converged) HIP_TRY(d, hipMemcpyAsync(converged, d->w.done, (size_t)B, hipMemcpyDeviceToHost, d->stream));
	if (iters) HIP_TRY(d, hipMemcpyAsync(iters, d->w.iters, (size_t)B * 4, hipMemcpyDeviceToHost, d->stream));
	HIP_TRY(d, hipStreamSynchronize(d->stream));
	return NBL_OK;
}

// Diagnostic only (not part of include/nbldpc.h): run the channel alone and return the received samples [B][L][2] and the
// fraction of log / cos values that went to the host's libm.
extern "C" nbl_status nbl_debug_channel(nbl_decoder *d, const uint8_t *tx_index, const uint32_t *lane_state, double sigma, int32_t B,
                                        double *rx_out, double *flag_frac)
{
	if (!d || !tx_index || !lane_state || !rx_out || B <= 0) return NBL_ERR_ARG;
	d->err.clear();
	HIP_TRY(d, hipSetDevice(d->device));
	nbl_status s = run_channel(d, tx_index, lane_state, sigma, B, d->stream, &d->d_rx, &d->d_rx_cap, d->err);
	if (s) return s;
	HIP_TRY(d, hipMemcpyAsync(rx_out, d->d_rx, (size_t)B * d->dm_L * 16, hipMemcpyDeviceToHost, d->stream));
	HIP_TRY(d, hipStreamSynchronize(d->stream));
	if (flag_frac) *flag_frac = d->last_flag_frac;
	return NBL_OK;
}

// Diagnostic only: channel LLRs of codeword b as the decoder holds them, [N][q-1]
extern "C" nbl_status nbl_debug_read_lch(nbl_decoder *d, int32_t b, double *out)
{
	if (!d || !out || b < 0 || b >= d->last_B) return NBL_ERR_ARG;
	d->err.clear();
	HIP_TRY(d, hipSetDevice(d->device));
	const int q = d->g.q, N = d->g.N;
	double *tmp = nullptr;
	HIP_TRY(d, hipMalloc((void **)&tmp, (size_t)N * (q - 1) * 8));
	HIP_TRY(d, nbl_launch_unpad(d->w.Lch + (size_t)b * N * q, tmp, nullptr, N, q, d->stream));
	HIP_TRY(d, hipMemcpyAsync(out, tmp, (size_t)N * (q - 1) * 8, hipMemcpyDeviceToHost, d->stream));
	HIP_TRY(d, hipStreamSynchronize(d->stream));
	(void)hipFree(tmp);
	return NBL_OK;
}

extern "C" nbl_status nbl_read_state(nbl_decoder *d, int32_t b, double *post, double *v2c, double *c2v)
{
	if (!d || b < 0 || b >= d->last_B) return NBL_ERR_ARG;
	d->err.clear();
	HIP_TRY(d, hipSetDevice(d->device));
	const int q = d->g.q, N = d->g.N, E = d->g.E;
	double *tmp = nullptr;
	HIP_TRY(d, hipMalloc((void **)&tmp, (size_t)E * (q - 1) * 8));
	nbl_status rc = NBL_OK;
	auto grab = [&](const double *src, const int *map, int rows, double *dst) -> nbl_status {
		HIP_TRY(d, nbl_launch_unpad(src, tmp, map, rows, q, d->stream));
		HIP_TRY(d, hipMemcpyAsync(dst, tmp, (size_t)rows * (q - 1) * 8, hipMemcpyDeviceToHost, d->stream));
		HIP_TRY(d, hipStreamSynchronize(d->stream));
		return NBL_OK;
	};
	if (post) {
		if (!d->w.post) { d->err = "state recording was off during the last decode (nbl_set_record_state)"; rc = NBL_ERR_ARG; }
		else rc = grab(d->w.post + (size_t)b * N * q, nullptr, N, post);
	}
	if (!rc && v2c) {
		if (!d->w.v2c) { d->err = "v2c is not kept in HBM on the fused path unless state recording is on (nbl_set_record_state)"; rc = NBL_ERR_ARG; }
		else rc = grab(d->w.v2c + (size_t)b * E * q, nullptr, E, v2c);
	}
	if (!rc && c2v) {
		const double *src = d->last_c2v ? d->last_c2v : d->w.c2v;
		if (d->last_fused && d->c2v_alt && !d->prm.fixed_iters) {
			// fused iterations: a codeword that converged at iteration k has c2v(k) in one buffer (computed before its syndrome was
			// known) and c2v(k-1), what the reference returns with, intact in the other; iteration i writes bufB when i is odd
			int it = 0;
			uint8_t done = 0;
			HIP_TRY(d, hipMemcpy(&it, d->w.iters + b, sizeof it, hipMemcpyDeviceToHost));
			HIP_TRY(d, hipMemcpy(&done, d->w.done + b, 1, hipMemcpyDeviceToHost));
			if (done) src = (it == 1 && d->c2v_zero) ? d->c2v_zero : ((it - 1) & 1) ? d->c2v_alt : d->w.c2v; // buffer written by iteration it-1 (it = 1: the zeros)
		}
		rc = grab(src == d->c2v_zero ? src : src + (size_t)b * E * q, d->d_e2c_map, E, c2v); // (the zeros are one shared block)
	}
	(void)hipFree(tmp);
	return rc;
}
