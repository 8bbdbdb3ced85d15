// nbldpc_amd/host/comm.h -- CComm: one lane of the reference's link chain (Comm.h / Comm.cpp), minus the decoder it used to
// own: all lanes share one CNBLDPC (code parameters + encoder) and their frames are decoded together by the caller.
//   FrontEnd()    = GenerateMessage, Encode, Puncture, Modulate, Channel_AWGN, Demodulate      (Comm.cpp:181-189)
//   TakeDecoded() = the symbol/bit unpacking of CComm::Decode after NBLDPC.Decoding             (Comm.cpp:421-443)
//   Err()         = error counting                                                              (Comm.cpp:446-503)
#pragma once
#include <string>
#include <vector>
#include "nbldpc_host.h"
#include "rand.h"
#include "simulation.h"

struct CComplex { double Real = 0, Image = 0; };

class CComm {
public:
	int GFq = 0, parallel_num = 1, Bit_Len_PerSYM = 0;
	CNBLDPC *NBLDPC = nullptr;
	CRand Rand;
	int pn = 0;                        // the 11-stage PN register, bit i = regPN[i] of the reference (Comm.h)
	const unsigned short *pn_jump = nullptr; // state after `parallel - 1` clocks, per state (one look-up per message bit)
	int randomMsg = 0, crcLen = 0, crcLen_correct = 0;
	int MSG_SYM_LEN = 0, MSG_BIT_LEN = 0, CODE_SYM_LEN = 0, CODE_BIT_LEN = 0, PUN_SYM_LEN = 0, PUN_BIT_LEN = 0;
	int modOrder = 0, MOD_BIT_PER_SYM = 0, MOD_SYM_LEN = 0, MOD_BIT_LEN = 0;
	double CodeRate = 0, sigma_n = 0;
	std::vector<int> TX_MSG_BIT_beforeCRC, TX_MSG_BIT, TX_MSG_SYM, TX_CODE_SYM, TX_CODE_BIT, PUN_SYM, PUN_BIT, TX_MOD_BIT;
	std::vector<int> RX_DECODE_SYM, RX_DECODE_BIT, RX_MSG_SYM, RX_MSG_BIT;
	std::vector<CComplex> CONSTELLATION, TX_MOD_SYM, RX_MOD_SYM;
	std::vector<unsigned char> TX_MOD_IDX; // constellation index of every transmitted symbol (what Modulate looked up)
	std::vector<double> RX_LLR_BIT;
	std::vector<double> RX_LLR_SYM; // [CODE_SYM_LEN][GFq-1], row-major (the reference's double**)
	bool DecodeCorrect = false;

	bool Initial(CSimulation &sim, int parallel_order, CNBLDPC *shared);
	double SetEbN0(CSimulation &sim, int parallel_order);
	int FrontEnd();
	int FrontEndToChannel(); // everything up to Channel_AWGN: the demodulator runs on the device
	// everything up to Modulate: channel and demodulator run on the device.  The lane's generator state in front of the frame is
	// returned and the generator is moved past the 4 * MOD_SYM_LEN uniform draws Channel_AWGN would have made (Comm.cpp:328-337).
	int FrontEndToModulate(unsigned int state_out[3]);
	void DemodSource(std::vector<int> &src) const; // which received sample carries each code bit (BPSK) / code symbol (q-ary)
	int GenerateMessage();
	int GenPN();
	void CRCEncode(int *seqOut, const int *seqIn, int seqInLen, int crcLen, int crc24Type);
	int CrcCheck(const int *seqIn, int seqInLen, int crcLen, int crc24Type);
	int Encode();
	int Puncture();
	int Modulate();
	int Channel_AWGN();
	int Demodulate();
	int TakeDecoded(const int *decoded_sym, bool converged);
	int Err(CSimulation &sim);
	// Err() in two halves for the pipelined driver: HoldTx keeps the transmitted message of this cycle while the next cycle's
	// front-end already runs; ErrCount is lane-local (thread-safe); ErrAccumulate adds into the shared counters in lane order.
	void HoldTx(int slot);
	void ErrCount(int slot, double &errSym, double &errBit, int &crc_ok);
	static void ErrAccumulate(CSimulation &sim, double errSym, double errBit, int crc_ok);
	void ErrRates(CSimulation &sim) const;
	std::vector<int> HOLD_MSG_SYM[2], HOLD_MSG_BIT[2];
	std::string error;

private:
	void ResetSources(CSimulation &sim, int parallel_order);
	void SkipPN(int n); // clock the PN register n times (table walk)
	static const unsigned short *PnJumpTable(int n); // state -> state after n clocks (cached per n)
};
