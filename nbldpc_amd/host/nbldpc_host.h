// nbldpc_amd/host/nbldpc_host.h -- CNBLDPC with the reference's public interface (NBLDPC.h:13-71) on top of the C ABI.
//
//   bool Initial(CSimulation&)                          NBLDPC.h:42   code-file parse, puncture list, encoder, nbl_create
//   int  Encode(int* msg_sym, int* code_sym)            NBLDPC.h:61   systematic encode (also rewrites msg_sym, :598-601)
//   int  Decoding(double** L_ch, int* out, int*, int*)  NBLDPC.h:71   one codeword (batch of one)
//   int  DecodingBatch(...)                             new            B codewords, ONE device call
// All decoding happens in libnbldpc_hip.so; this class holds no CPU decoder.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/nbldpc.h"
#include "gf.h"
#include "simulation.h"

class CNBLDPC {
public:
	CNBLDPC() = default;
	~CNBLDPC();
	CNBLDPC(const CNBLDPC &) = delete;
	CNBLDPC &operator=(const CNBLDPC &) = delete;

	int GFq = 0;
	CGF GF;
	int maxIter = 0;
	int CodeLen = 0, ChkLen = 0, PunctureLen = 0;
	std::vector<int> PuncturePositionV;
	int *PuncturePosition = nullptr; // = PuncturePositionV.data()
	int maxVarDegree = 0, maxChkDegree = 0;
	std::vector<int> VarDegree, ChkDegree;
	std::vector<std::vector<int>> VarLink, ChkLink, VarLinkGFe, ChkLinkGFe;
	int DecodeMethod = 0;

	bool Initial(CSimulation &sim, int device = 0, int fixed_iters = 0); // device < 0: graph + encoder only, no GPU touched
	int Encode(int *msg_sym, int *code_sym);
	int Decoding(double **L_ch, int *DecodeOutput, int *RelySeri_symbol, int *RelySeri_bit);
	// L_ch [B][CodeLen][GFq-1]; out [B][CodeLen]; converged [B] (may be null); iters [B] (may be null). 0 on success.
	int DecodingBatch(const double *L_ch, int B, int *out, uint8_t *converged, int *iters);
	// device-side demodulation (replaces CComm::Demodulate, Comm.cpp:340-407): rx [B][L][2] received samples
	int SetDemodulator(int mod_order, int n_mod_sym, const double *constellation, const int *src);
	int DecodingBatchSamples(const double *rx, double sigma, int B, int *out, uint8_t *converged, int *iters);
	// device-side channel (replaces CComm::Channel_AWGN + CRand, Comm.cpp:328-337 / Rand.cpp:17-37): tx_index [B][L] constellation
	// indices, lane_state [B][3] generator states in front of the frame
	int DecodingBatchNoise(const unsigned char *tx_index, const unsigned int *lane_state, double sigma, int B, int *out, uint8_t *converged, int *iters);
	// the same in two phases (slot 0 / 1): the channel of cycle k+1 may run while cycle k is decoded
	int ChannelBatch(int slot, const unsigned char *tx_index, const unsigned int *lane_state, double sigma, int B);
	int DecodingBatchResident(int slot, double sigma, int B, int *out, uint8_t *converged, int *iters);
	const std::string &LastError() const { return error; }
	nbl_decoder *Handle() const { return dec; }

private:
	// encoder state (InitialEncode NBLDPC.cpp:477-560)
	std::vector<int> swap_src, swap_dst;
	std::vector<std::vector<int>> enc_link, enc_coef;
	bool InitialEncode();
	nbl_decoder *dec = nullptr;
	std::string error;
};
