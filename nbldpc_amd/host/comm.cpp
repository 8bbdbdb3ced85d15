// nbldpc_amd/host/comm.cpp -- see comm.h.  Every arithmetic expression that feeds the channel LLRs keeps the reference's
// operation order, because L_ch has to be bit-identical for FER parity (tests/test_host_frontend.py).
#include "comm.h"
#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <mutex>
#include <vector>

// seed + lane for the three generators, PN register pre-advanced lane steps (Comm.cpp:58-74, :160-173)
void CComm::ResetSources(CSimulation &sim, int lane)
{
	Rand.IX = Rand.IY = Rand.IZ = sim.randomseed;
	if (lane) { Rand.IX += lane; Rand.IY += lane; Rand.IZ += lane; }
	static const int init[11] = {1, 0, 1, 0, 0, 0, 1, 1, 0, 0, 1};
	pn = 0;
	for (int i = 0; i < 11; i++) pn |= init[i] << i;
	SkipPN(lane);
	pn_jump = PnJumpTable(sim.parallel - 1);
}

bool CComm::Initial(CSimulation &sim, int lane, CNBLDPC *shared)
{
	NBLDPC = shared;
	GFq = sim.GFq;
	parallel_num = sim.parallel;
	Bit_Len_PerSYM = log(GFq) / log(2);
	ResetSources(sim, lane);
	randomMsg = sim.randomMsg;
	crcLen = sim.crcLen;
	crcLen_correct = sim.crc_correctLen;

	MSG_SYM_LEN = NBLDPC->CodeLen - NBLDPC->ChkLen;
	MSG_BIT_LEN = MSG_SYM_LEN * Bit_Len_PerSYM;
	CODE_SYM_LEN = NBLDPC->CodeLen;
	CODE_BIT_LEN = CODE_SYM_LEN * Bit_Len_PerSYM;
	PUN_SYM_LEN = NBLDPC->PunctureLen;
	PUN_BIT_LEN = PUN_SYM_LEN * Bit_Len_PerSYM;
	if (MSG_BIT_LEN - crcLen < 0) { error = "crcLen exceeds the message length"; return false; }
	TX_MSG_BIT.assign(MSG_BIT_LEN, 0);
	TX_MSG_SYM.assign(MSG_SYM_LEN, 0);
	TX_MSG_BIT_beforeCRC.assign(MSG_BIT_LEN - crcLen, 0);
	TX_CODE_SYM.assign(CODE_SYM_LEN, 0);
	TX_CODE_BIT.assign(CODE_BIT_LEN, 0);
	PUN_SYM.assign(PUN_SYM_LEN, 0);
	PUN_BIT.assign(PUN_BIT_LEN, 0);
	CodeRate = double(MSG_SYM_LEN) / double(CODE_SYM_LEN - PUN_SYM_LEN);
	for (int s = 0; s < PUN_SYM_LEN; s++) {
		PUN_SYM[s] = NBLDPC->PuncturePosition[s];
		for (int k = 0; k < Bit_Len_PerSYM; k++) PUN_BIT[s * Bit_Len_PerSYM + k] = Bit_Len_PerSYM * NBLDPC->PuncturePosition[s] + k;
	}
	modOrder = sim.nQAM;
	MOD_BIT_PER_SYM = log(sim.nQAM * 1.0) / log(2.0);
	MOD_SYM_LEN = (CODE_SYM_LEN - PUN_SYM_LEN) * Bit_Len_PerSYM / MOD_BIT_PER_SYM;
	MOD_BIT_LEN = MOD_SYM_LEN * MOD_BIT_PER_SYM;
	if (modOrder != 2 && modOrder != GFq) {
		error = "This module ( code order ~= modulation order ) haven't been developed!"; // Comm.cpp:400-404
		std::cerr << error << std::endl;
		return false;
	}

	// constellation file: "Point: i Real: x Imag: y" per line (Comm.cpp:113-126)
	CONSTELLATION.assign(modOrder, CComplex());
	std::ifstream fc(sim.ConstellationFileName);
	if (!fc.is_open()) { error = "Cannot open " + sim.ConstellationFileName; std::cerr << error << std::endl; return false; }
	for (int k = 0; k < modOrder; k++) {
		std::string w;
		int idx = 0;
		double re = 0, im = 0;
		fc >> w >> idx >> w >> re >> w >> im;
		if (!fc || idx < 0 || idx >= modOrder) { error = "Malformed constellation file"; return false; }
		CONSTELLATION[idx].Real = re;
		CONSTELLATION[idx].Image = im;
	}
	TX_MOD_BIT.assign(MOD_BIT_LEN, 0);
	TX_MOD_SYM.assign(MOD_SYM_LEN, CComplex());
	RX_MOD_SYM.assign(MOD_SYM_LEN, CComplex());
	TX_MOD_IDX.assign(MOD_SYM_LEN, 0);
	RX_LLR_BIT.assign(CODE_BIT_LEN, 0.0);
	RX_LLR_SYM.assign((size_t)CODE_SYM_LEN * (GFq - 1), 0.0);
	RX_DECODE_SYM.assign(CODE_SYM_LEN, 0);
	RX_DECODE_BIT.assign(CODE_BIT_LEN, 0);
	RX_MSG_SYM.assign(MSG_SYM_LEN, 0);
	RX_MSG_BIT.assign(MSG_BIT_LEN, 0);
	return true;
}

double CComm::SetEbN0(CSimulation &sim, int lane)
{
	ResetSources(sim, lane);
	double EbN0 = pow(10.0, sim.EbN0 / 10.0);
	sigma_n = 1.0 / sqrt(2 * MOD_BIT_PER_SYM * CodeRate * EbN0); // Comm.cpp:176-177
	return sigma_n;
}

int CComm::FrontEnd()
{
	GenerateMessage();
	Encode();
	Puncture();
	Modulate();
	Channel_AWGN();
	return Demodulate();
}

int CComm::FrontEndToChannel()
{
	GenerateMessage();
	Encode();
	Puncture();
	Modulate();
	return Channel_AWGN();
}

int CComm::FrontEndToModulate(unsigned int state_out[3])
{
	// Nothing after this call reads the chain's intermediates (message bits in front of the CRC, code symbols and bits, modulator
	// bits, constellation points): only TX_MSG_BIT / TX_MSG_SYM (error count) and TX_MOD_IDX (shipped to the device) are kept.  With
	// thousands of lanes per thread the intermediates of a lane are cold every cycle -- 14 KB of cache misses per frame, five times
	// the chain's arithmetic -- so they are swapped (pointer swaps) for one set of per-thread buffers that stays in the cache.
	struct Scratch { std::vector<int> before_crc, code_sym, code_bit, mod_bit; std::vector<CComplex> mod_sym; };
	static thread_local Scratch tl;
	if (tl.before_crc.size() != TX_MSG_BIT_beforeCRC.size()) tl.before_crc.assign(TX_MSG_BIT_beforeCRC.size(), 0);
	if (tl.code_sym.size() != TX_CODE_SYM.size()) tl.code_sym.assign(TX_CODE_SYM.size(), 0);
	if (tl.code_bit.size() != TX_CODE_BIT.size()) tl.code_bit.assign(TX_CODE_BIT.size(), 0);
	if (tl.mod_bit.size() != TX_MOD_BIT.size()) tl.mod_bit.assign(TX_MOD_BIT.size(), 0);
	if (tl.mod_sym.size() != TX_MOD_SYM.size()) tl.mod_sym.assign(TX_MOD_SYM.size(), CComplex());
	auto exchange = [&] {
		TX_MSG_BIT_beforeCRC.swap(tl.before_crc);
		TX_CODE_SYM.swap(tl.code_sym);
		TX_CODE_BIT.swap(tl.code_bit);
		TX_MOD_BIT.swap(tl.mod_bit);
		TX_MOD_SYM.swap(tl.mod_sym);
	};
	exchange();
	GenerateMessage();
	Encode();
	Puncture();
	Modulate();
	exchange();
	state_out[0] = (unsigned int)(Rand.IX % 61967ul);
	state_out[1] = (unsigned int)(Rand.IY % 63443ul);
	state_out[2] = (unsigned int)(Rand.IZ % 63599ul);
	Rand.Skip(4ul * (unsigned long)MOD_SYM_LEN);
	return 0;
}

// the index bookkeeping of Demodulate (Comm.cpp:348-357 / :384-396) as a table: -1 marks a punctured position
void CComm::DemodSource(std::vector<int> &src) const
{
	src.clear();
	int pi = 0;
	if (modOrder == 2) {
		for (int b = 0; b < CODE_BIT_LEN; b++) {
			if (PUN_BIT_LEN != 0 && pi < PUN_BIT_LEN && PUN_BIT[pi] == b) { pi++; src.push_back(-1); }
			else src.push_back(b - pi);
		}
	} else {
		for (int s = 0; s < CODE_SYM_LEN; s++) {
			if (PUN_SYM_LEN != 0 && pi < PUN_SYM_LEN && PUN_SYM[pi] == s) { pi++; src.push_back(-1); }
			else src.push_back(s - pi);
		}
	}
}

// 11-stage shift register, output r10, feedback r10 ^ r3 after the shift (Comm.cpp:241-252)
int CComm::GenPN()
{
	const int out = (pn >> 9) & 1, fb = ((pn >> 9) ^ (pn >> 2)) & 1; // shift, then regPN[0] = regPN[10] ^ regPN[3]; output regPN[10]
	pn = ((pn << 1) & 2047) | fb;
	return out;
}

// The register update is a linear map of an 11-bit state, so "clock it n times" is a table walk: the map is squared
// repeatedly (2048-entry tables) and the tables of the set bits of n are applied; for the one n the message generator needs
// (parallel - 1, Comm.cpp:201) the walk is folded into a single table once.  Same register contents as n calls of GenPN();
// the reference clocks one by one, which is O(parallel) per message bit.
namespace {
struct PnTables {
	std::vector<std::vector<unsigned short>> pow2; // pow2[k][state] = state after 2^k clocks
	std::map<int, std::vector<unsigned short>> jumps;
	std::mutex mu;
	PnTables()
	{
		std::vector<unsigned short> t(2048);
		for (int s = 0; s < 2048; s++) {
			const int fb = ((s >> 9) ^ (s >> 2)) & 1; // the feedback is taken after the shift (Comm.cpp:246-250)
			t[s] = (unsigned short)(((s << 1) & 2047) | fb);
		}
		pow2.push_back(t);
		for (int k = 1; k < 31; k++) {
			std::vector<unsigned short> n(2048);
			for (int s = 0; s < 2048; s++) n[s] = pow2[k - 1][pow2[k - 1][s]];
			pow2.push_back(n);
		}
	}
	int walk(int s, int n) const
	{
		for (int k = 0; k < 31; k++)
			if ((n >> k) & 1) s = pow2[k][s];
		return s;
	}
};
PnTables &pn_tables()
{
	static PnTables t;
	return t;
}
} // namespace

void CComm::SkipPN(int n)
{
	if (n > 0) pn = pn_tables().walk(pn, n);
}

const unsigned short *CComm::PnJumpTable(int n)
{
	PnTables &t = pn_tables();
	std::lock_guard<std::mutex> lock(t.mu);
	auto it = t.jumps.find(n);
	if (it == t.jumps.end()) {
		std::vector<unsigned short> j(2048);
		for (int s = 0; s < 2048; s++) j[s] = (unsigned short)(n > 0 ? t.walk(s, n) : s);
		it = t.jumps.emplace(n, std::move(j)).first;
	}
	return it->second.data();
}

// every lane draws from the SAME PN sequence, interleaved: lane i uses outputs i, i+P, i+2P, .. (Comm.cpp:199-202)
int CComm::GenerateMessage()
{
	const int nb = MSG_BIT_LEN - crcLen;
	for (int b = 0; b < nb; b++) {
		if (randomMsg) {
			pn = pn_jump[pn]; // parallel - 1 clocks
			TX_MSG_BIT_beforeCRC[b] = GenPN();
		} else {
			TX_MSG_BIT_beforeCRC[b] = 0;
		}
	}
	CRCEncode(TX_MSG_BIT.data(), TX_MSG_BIT_beforeCRC.data(), nb, crcLen, 0);
	for (int s = 0; s < MSG_SYM_LEN; s++) {
		int sym = 0;
		for (int k = Bit_Len_PerSYM - 1; k >= 0; k--) sym = 2 * sym + TX_MSG_BIT[Bit_Len_PerSYM * s + k]; // bit k of the symbol = bit s*p+k
		TX_MSG_SYM[s] = sym;
	}
	return 0;
}

// generator polynomials of Comm.cpp:513-533 as tap lists (G[0] and G[L] are always set)
static void crc_taps(int len, int type24, int G[25])
{
	for (int i = 0; i < 25; i++) G[i] = 0;
	if (len == 8) { const int t[] = {0, 1, 3, 4, 7, 8}; for (int x : t) G[x] = 1; }
	else if (len == 16) { const int t[] = {0, 5, 12, 16}; for (int x : t) G[x] = 1; }
	else if (len == 24 && !type24) { const int t[] = {0, 1, 3, 4, 5, 6, 7, 10, 11, 14, 17, 18, 23, 24}; for (int x : t) G[x] = 1; }
	else if (len == 24) { const int t[] = {0, 1, 5, 6, 23, 24}; for (int x : t) G[x] = 1; }
}

// The shift registers of Comm.cpp:506-636 as one integer: bit j = reg[j]; a clock is a shift and, when the feedback bit is set, an
// XOR with the tap mask (bit j set for G[j], j = 0 .. len-1) -- the same register contents as the reference's inner loop over j.
static unsigned crc_mask(int len, int type24)
{
	int G[25];
	crc_taps(len, type24, G);
	unsigned m = 0;
	for (int j = 0; j < len; j++) m |= (unsigned)(G[j] != 0) << j;
	return m;
}

void CComm::CRCEncode(int *out, const int *in, int n, int len, int type24) // Comm.cpp:506-561
{
	if (in != out) memcpy(out, in, sizeof(int) * n);
	if (len == 0) return;
	const unsigned taps = crc_mask(len, type24), keep = (len == 32) ? 0xffffffffu : ((1u << len) - 1u);
	unsigned reg = 0;
	for (int i = 0; i < n; i++) {
		const unsigned fb = ((reg >> (len - 1)) ^ (unsigned)in[i]) & 1u; // reg[len-1] ^ in[i]; reg[0] = fb (G[0] = 1)
		reg = ((reg << 1) & keep) ^ (fb ? taps : 0u);
	}
	for (int i = 0; i < len; i++) out[n + i] = (int)((reg >> (len - 1 - i)) & 1u);
}

int CComm::CrcCheck(const int *in, int n, int len, int type24) // Comm.cpp:564-636
{
	if (len == 0) return 1;
	const unsigned taps = crc_mask(len, type24), keep = (len == 32) ? 0xffffffffu : ((1u << len) - 1u);
	unsigned reg = 0;
	int ones = 0;
	for (int i = 0; i < n; i++) {
		const unsigned fb = (reg >> (len - 1)) & 1u;                      // reg[len-1]; reg[0] = fb ^ in[i]
		reg = (((reg << 1) & keep) ^ (fb ? taps : 0u)) ^ ((unsigned)in[i] & 1u);
		ones += in[i];
	}
	return (reg == 0 && ones != 0) ? 1 : 0; // an all-zero word does not count as a CRC pass
}

int CComm::Encode() // Comm.cpp:255-289
{
	if (randomMsg) NBLDPC->Encode(TX_MSG_SYM.data(), TX_CODE_SYM.data());
	else std::fill(TX_CODE_SYM.begin(), TX_CODE_SYM.end(), 0);
	for (int s = 0; s < CODE_SYM_LEN; s++)
		for (int k = 0; k < Bit_Len_PerSYM; k++) TX_CODE_BIT[s * Bit_Len_PerSYM + k] = (TX_CODE_SYM[s] >> k) & 1; // LSB first
	for (int b = 0; b < MSG_BIT_LEN; b++) TX_MSG_BIT[b] = TX_CODE_BIT[b];
	return 0;
}

int CComm::Puncture() // Comm.cpp:290-308
{
	int pi = 0, mb = 0;
	for (int b = 0; b < CODE_BIT_LEN; b++) {
		if (PUN_BIT_LEN != 0 && pi < PUN_BIT_LEN && PUN_BIT[pi] == b) pi++;
		else TX_MOD_BIT[mb++] = TX_CODE_BIT[b];
	}
	return 0;
}

int CComm::Modulate() // Comm.cpp:310-325: MSB first (the reverse of Encode's unpacking -- reproduced as is)
{
	for (int s = 0; s < MOD_SYM_LEN; s++) {
		int idx = 0;
		for (int k = 0; k < MOD_BIT_PER_SYM; k++) idx += TX_MOD_BIT[s * MOD_BIT_PER_SYM + k] << (MOD_BIT_PER_SYM - 1 - k);
		TX_MOD_SYM[s] = CONSTELLATION[idx];
		TX_MOD_IDX[s] = (unsigned char)idx;
	}
	return 0;
}

int CComm::Channel_AWGN() // Comm.cpp:328-337: real AND imaginary noise are drawn, also for BPSK
{
	for (int s = 0; s < MOD_SYM_LEN; s++) {
		RX_MOD_SYM[s].Real = TX_MOD_SYM[s].Real + Rand.Rand_Norm(0, sigma_n);
		RX_MOD_SYM[s].Image = TX_MOD_SYM[s].Image + Rand.Rand_Norm(0, sigma_n);
	}
	return 0;
}

int CComm::Demodulate() // Comm.cpp:340-407
{
	const int w = GFq - 1;
	if (modOrder == 2) {
		int pi = 0;
		for (int b = 0; b < CODE_BIT_LEN; b++) {
			if (PUN_BIT_LEN != 0 && pi < PUN_BIT_LEN && PUN_BIT[pi] == b) { pi++; RX_LLR_BIT[b] = 0; }
			else RX_LLR_BIT[b] = -2 * RX_MOD_SYM[b - pi].Real / (sigma_n * sigma_n);
		}
		for (int s = 0; s < CODE_SYM_LEN; s++)
			for (int q = 1; q < GFq; q++) {
				double acc = 0;
				for (int k = 0; k < Bit_Len_PerSYM; k++)
					if ((q & (1 << k)) != 0) acc += RX_LLR_BIT[s * Bit_Len_PerSYM + k];
				RX_LLR_SYM[(size_t)s * w + q - 1] = acc;
			}
	} else { // modOrder == GFq: one constellation point per code symbol
		int pi = 0;
		const CComplex &c0 = CONSTELLATION[0];
		for (int s = 0; s < CODE_SYM_LEN; s++) {
			if (PUN_SYM_LEN != 0 && pi < PUN_SYM_LEN && PUN_SYM[pi] == s) {
				pi++;
				for (int q = 1; q < GFq; q++) RX_LLR_SYM[(size_t)s * w + q - 1] = 0;
			} else {
				const CComplex &r = RX_MOD_SYM[s - pi];
				for (int q = 1; q < GFq; q++) {
					const CComplex &cq = CONSTELLATION[q];
					RX_LLR_SYM[(size_t)s * w + q - 1] =
					    ((2 * r.Real - c0.Real - cq.Real) * (cq.Real - c0.Real) + (2 * r.Image - c0.Image - cq.Image) * (cq.Image - c0.Image)) /
					    (2 * sigma_n * sigma_n);
				}
			}
		}
	}
	return 0;
}

int CComm::TakeDecoded(const int *decoded, bool converged) // Comm.cpp:421-443
{
	DecodeCorrect = converged;
	for (int s = 0; s < CODE_SYM_LEN; s++) {
		RX_DECODE_SYM[s] = decoded[s];
		for (int k = 0; k < Bit_Len_PerSYM; k++) RX_DECODE_BIT[s * Bit_Len_PerSYM + k] = (decoded[s] >> k) & 1;
	}
	for (int s = 0; s < MSG_SYM_LEN; s++) RX_MSG_SYM[s] = RX_DECODE_SYM[s];
	for (int b = 0; b < MSG_BIT_LEN; b++) RX_MSG_BIT[b] = RX_DECODE_BIT[b];
	return 0;
}

int CComm::Err(CSimulation &sim) // Comm.cpp:446-503
{
	double errSym = 0, errBit = 0;
	int crc_ok = 0;
	ErrCount(-1, errSym, errBit, crc_ok);
	ErrAccumulate(sim, errSym, errBit, crc_ok);
	ErrRates(sim);
	return 0;
}

void CComm::HoldTx(int slot)
{
	HOLD_MSG_SYM[slot] = TX_MSG_SYM;
	HOLD_MSG_BIT[slot] = TX_MSG_BIT;
}

// slot < 0: compare with the live TX buffers (serial driver), else with the message held for that cycle
void CComm::ErrCount(int slot, double &errSym, double &errBit, int &crc_ok)
{
	const std::vector<int> &msg_sym = slot < 0 ? TX_MSG_SYM : HOLD_MSG_SYM[slot];
	const std::vector<int> &msg_bit = slot < 0 ? TX_MSG_BIT : HOLD_MSG_BIT[slot];
	errSym = errBit = 0;
	for (int s = 0; s < MSG_SYM_LEN; s++) errSym += (msg_sym[s] != RX_MSG_SYM[s]);
	for (int b = 0; b < MSG_BIT_LEN; b++) errBit += (msg_bit[b] != RX_MSG_BIT[b]);
	crc_ok = CrcCheck(RX_MSG_BIT.data(), MSG_BIT_LEN, crcLen, 1);
}

void CComm::ErrAccumulate(CSimulation &sim, double errSym, double errBit, int crc_ok)
{
	sim.errSym += errSym;
	sim.errBit += errBit;
	sim.errFrame += (errSym != 0) ? 1 : 0;
	if (crc_ok && errSym != 0) { sim.U_errSym += errSym; sim.U_errBit += errBit; sim.U_errFrame += 1; }
}

void CComm::ErrRates(CSimulation &sim) const
{
	sim.SER = sim.errSym / (sim.simCycle * MSG_SYM_LEN * sim.parallel);
	sim.BER = sim.errBit / (sim.simCycle * MSG_BIT_LEN * sim.parallel);
	sim.FER = sim.errFrame / (sim.simCycle * sim.parallel);
	sim.U_SER = sim.U_errSym / (sim.simCycle * MSG_SYM_LEN * sim.parallel);
	sim.U_BER = sim.U_errBit / (sim.simCycle * MSG_BIT_LEN * sim.parallel);
	sim.U_FER = sim.U_errFrame / (sim.simCycle * sim.parallel);
}
