// nbldpc_amd/host/nbldpc_host.cpp -- see nbldpc_host.h.
#include "nbldpc_host.h"
#include <cstring>
#include <fstream>
#include <iostream>

CNBLDPC::~CNBLDPC()
{
	if (dec) nbl_destroy(dec);
}

bool CNBLDPC::Initial(CSimulation &sim, int device, int fixed_iters)
{
	GFq = sim.GFq;
	maxIter = sim.maxIter;
	DecodeMethod = sim.decodeMethod;
	if (!GF.Initial(GFq)) { error = GF.error; return false; }

	// code file: "N M q" / "maxdv maxdc" / dv[N] / dc[M] / N rows of (check, h) / M rows of (var, h), 1-based (NBLDPC.cpp:147-205)
	std::ifstream f(sim.NonBinaryFileName);
	if (!f.is_open()) { error = "Cannot open " + sim.NonBinaryFileName; std::cerr << error << std::endl; return false; }
	f >> CodeLen >> ChkLen >> GFq >> maxVarDegree >> maxChkDegree;
	VarDegree.assign(CodeLen, 0);
	ChkDegree.assign(ChkLen, 0);
	for (auto &d : VarDegree) f >> d;
	for (auto &d : ChkDegree) f >> d;
	VarLink.assign(CodeLen, {}); VarLinkGFe.assign(CodeLen, {});
	ChkLink.assign(ChkLen, {}); ChkLinkGFe.assign(ChkLen, {});
	for (int n = 0; n < CodeLen; n++)
		for (int d = 0; d < VarDegree[n]; d++) { int c, h; f >> c >> h; VarLink[n].push_back(c - 1); VarLinkGFe[n].push_back(h); }
	for (int m = 0; m < ChkLen; m++)
		for (int d = 0; d < ChkDegree[m]; d++) { int v, h; f >> v >> h; ChkLink[m].push_back(v - 1); ChkLinkGFe[m].push_back(h); }
	if (!f) { error = "Malformed code file " + sim.NonBinaryFileName; std::cerr << error << std::endl; return false; }

	// every variable whose degree equals PuntureVarDegree is punctured (NBLDPC.cpp:163-176)
	PuncturePositionV.clear();
	for (int n = 0; n < CodeLen; n++)
		if (VarDegree[n] == sim.PuntureVarDegree) PuncturePositionV.push_back(n);
	PunctureLen = (int)PuncturePositionV.size();
	PuncturePosition = PuncturePositionV.data();

	if (sim.OSD_order >= 0) { error = "OSD post-processing is outside this decode path: set OSD_order to -1"; std::cerr << error << std::endl; return false; }
	if (sim.randomMsg && !InitialEncode()) return false;

	if (device < 0) return true; // host-only use (link-chain front-end, encoder): no decoder handle is created

	// hand the graph and the parameters to the device library
	std::vector<int32_t> vchk, vh, cvar, ch;
	for (int n = 0; n < CodeLen; n++) for (int d = 0; d < VarDegree[n]; d++) { vchk.push_back(VarLink[n][d]); vh.push_back(VarLinkGFe[n][d]); }
	for (int m = 0; m < ChkLen; m++) for (int d = 0; d < ChkDegree[m]; d++) { cvar.push_back(ChkLink[m][d]); ch.push_back(ChkLinkGFe[m][d]); }
	nbl_code_desc code = {CodeLen, ChkLen, GFq, VarDegree.data(), ChkDegree.data(), vchk.data(), vh.data(), cvar.data(), ch.data()};
	std::vector<uint16_t> mul((size_t)GFq * GFq), inv(GFq, 0);
	for (int i = 0; i < GFq * GFq; i++) mul[i] = (uint16_t)GF.TableMultiply[i];
	for (int a = 1; a < GFq; a++) inv[a] = (uint16_t)GF.TableInverse[a];
	nbl_params p;
	memset(&p, 0, sizeof p);
	p.method = sim.decodeMethod;
	p.max_iter = sim.maxIter;
	p.ems_nm = sim.ems_nm; p.ems_nc = sim.ems_nc; p.ems_factor = sim.ems_factor; p.ems_offset = sim.ems_offset;
	p.tems_nr = sim.tems_nr; p.tems_nc = sim.tems_nc; p.tems_factor = sim.tems_factor; p.tems_offset = sim.tems_offset;
	p.fixed_iters = fixed_iters;
	p.poll_every = fixed_iters ? 0 : 2;
	p.max_batch = 0; // the workspace is sized at the first DecodingBatch call
	if (dec) { nbl_destroy(dec); dec = nullptr; }
	nbl_status st = nbl_create(&code, mul.data(), inv.data(), &p, device, &dec);
	if (st != NBL_OK) {
		error = nbl_last_error(nullptr);
		std::cerr << error << std::endl; // the reference prints and exits for its own configuration errors (NBLDPC.cpp:284-285)
		return false;
	}
	return true;
}

// Systematic form by Gauss elimination from the last row up, pivot in column (row + N - M); a missing pivot is fetched from
// a row above, else from a column to the left (the column swap is recorded and undone after encoding).  NBLDPC.cpp:1474-1538.
bool CNBLDPC::InitialEncode()
{
	const int N = CodeLen, M = ChkLen;
	std::vector<std::vector<int>> H(M, std::vector<int>(N, 0));
	for (int m = 0; m < M; m++)
		for (size_t d = 0; d < ChkLink[m].size(); d++) H[m][ChkLink[m][d]] = ChkLinkGFe[m][d];
	swap_src.clear(); swap_dst.clear();
	for (int row = M - 1; row >= 0; row--) {
		const int col = row + N - M;
		if (H[row][col] == 0) {
			bool found = false;
			for (int up = row - 1; up >= 0 && !found; up--)
				if (H[up][col] != 0) { std::swap(H[row], H[up]); found = true; }
			for (int left = col - 1; left >= 0 && !found; left--)
				if (H[row][left] != 0) {
					for (int m = 0; m < M; m++) std::swap(H[m][col], H[m][left]);
					swap_src.push_back(col); swap_dst.push_back(left);
					found = true;
				}
			if (!found) { error = "NB matrix is not full rank"; std::cerr << error << std::endl; return false; }
		}
		const int hinv = GF.GFInverse(H[row][col]);
		for (int up = row - 1; up >= 0; up--)
			if (H[up][col] != 0) {
				const int x = GF.GFMultiply(hinv, H[up][col]);
				for (int c = 0; c < N; c++) H[up][c] = GF.GFAdd(H[up][c], GF.GFMultiply(x, H[row][c]));
			}
		for (int c = 0; c <= col; c++) H[row][c] = GF.GFMultiply(H[row][c], hinv);
	}
	enc_link.assign(M, {}); enc_coef.assign(M, {});
	for (int p = 0; p < M; p++)
		for (int c = 0; c < N - M + p; c++)
			if (H[p][c] != 0) { enc_link[p].push_back(c); enc_coef[p].push_back(H[p][c]); }
	return true;
}

int CNBLDPC::Encode(int *msg_sym, int *code_sym) // NBLDPC.cpp:562-604
{
	const int K = CodeLen - ChkLen;
	for (int c = 0; c < K; c++) code_sym[c] = msg_sym[c];
	for (int c = K; c < CodeLen; c++) code_sym[c] = 0;
	for (int p = 0; p < ChkLen; p++) {
		int acc = 0;
		for (size_t d = 0; d < enc_link[p].size(); d++) acc = GF.GFAdd(acc, GF.GFMultiply(enc_coef[p][d], code_sym[enc_link[p][d]]));
		code_sym[K + p] = acc;
	}
	for (int k = (int)swap_src.size() - 1; k >= 0; k--) std::swap(code_sym[swap_src[k]], code_sym[swap_dst[k]]);
	for (int c = 0; c < K; c++) msg_sym[c] = code_sym[c];
	return 0;
}

int CNBLDPC::DecodingBatch(const double *L_ch, int B, int *out, uint8_t *converged, int *iters)
{
	if (!dec) { error = "decoder not initialised"; return -1; }
	nbl_status st = nbl_decode_batch(dec, L_ch, B, out, converged, iters);
	if (st != NBL_OK) { error = nbl_last_error(dec); std::cerr << error << std::endl; return (int)st; }
	return 0;
}

int CNBLDPC::SetDemodulator(int mod_order, int n_mod_sym, const double *constellation, const int *src)
{
	if (!dec) { error = "decoder not initialised"; return -1; }
	nbl_demod_desc dm = {mod_order, n_mod_sym, constellation, src};
	nbl_status st = nbl_set_demodulator(dec, &dm);
	if (st != NBL_OK) { error = nbl_last_error(dec); std::cerr << error << std::endl; return (int)st; }
	return 0;
}

int CNBLDPC::DecodingBatchSamples(const double *rx, double sigma, int B, int *out, uint8_t *converged, int *iters)
{
	if (!dec) { error = "decoder not initialised"; return -1; }
	nbl_status st = nbl_decode_batch_samples(dec, rx, sigma, B, out, converged, iters);
	if (st != NBL_OK) { error = nbl_last_error(dec); std::cerr << error << std::endl; return (int)st; }
	return 0;
}

int CNBLDPC::DecodingBatchNoise(const unsigned char *tx_index, const unsigned int *lane_state, double sigma, int B, int *out, uint8_t *converged, int *iters)
{
	if (!dec) { error = "decoder not initialised"; return -1; }
	nbl_status st = nbl_decode_batch_noise(dec, tx_index, lane_state, sigma, B, out, converged, iters);
	if (st != NBL_OK) { error = nbl_last_error(dec); std::cerr << error << std::endl; return (int)st; }
	return 0;
}

int CNBLDPC::ChannelBatch(int slot, const unsigned char *tx_index, const unsigned int *lane_state, double sigma, int B)
{
	if (!dec) { error = "decoder not initialised"; return -1; }
	nbl_status st = nbl_channel_batch(dec, slot, tx_index, lane_state, sigma, B);
	if (st != NBL_OK) { error = nbl_last_error(dec); std::cerr << error << std::endl; return (int)st; }
	return 0;
}

int CNBLDPC::DecodingBatchResident(int slot, double sigma, int B, int *out, uint8_t *converged, int *iters)
{
	if (!dec) { error = "decoder not initialised"; return -1; }
	nbl_status st = nbl_decode_batch_resident(dec, slot, sigma, B, out, converged, iters);
	if (st != NBL_OK) { error = nbl_last_error(dec); std::cerr << error << std::endl; return (int)st; }
	return 0;
}

int CNBLDPC::Decoding(double **L_ch, int *DecodeOutput, int *, int *) // NBLDPC.cpp:607
{
	std::vector<double> flat((size_t)CodeLen * (GFq - 1));
	for (int n = 0; n < CodeLen; n++) memcpy(&flat[(size_t)n * (GFq - 1)], L_ch[n], sizeof(double) * (GFq - 1));
	uint8_t ok = 0;
	if (DecodingBatch(flat.data(), 1, DecodeOutput, &ok, nullptr) != 0) return 0;
	return ok;
}
