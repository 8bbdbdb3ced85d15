// nbldpc_amd/host/capi.cpp -- small C entry points over the host layer for the Python tests (ctypes).
#include <cstdlib>
#include <cstring>
#include <thread>
#include "link.h"

extern "C" {

// Run the link chain of every lane for `frames` cycles WITHOUT decoding and return what the reference's chain produced:
// L_ch [frames*P][N][q-1] (frame-major, b = f*P + lane), tx_code [frames*P][N], tx_msg [frames*P][K].  No GPU needed.
int nblh_frontend(const char *profile, double ebn0, int frames, double *L_ch, int *tx_code, int *tx_msg, double *sigma_out)
{
	CSimulation sim;
	if (sim.Initial(profile) != 0) return -1;
	sim.EbN0 = ebn0;
	// host-only CNBLDPC use: parse + encoder, no device (Initial would need a GPU) -> replicate the needed part
	CLink link;
	link.sim = sim;
	CNBLDPC &code = link.code;
	// the front-end needs the graph and the encoder only: no device decoder is created (device = -1)
	if (!code.Initial(link.sim, -1)) return -2;
	const int P = sim.parallel, N = code.CodeLen, K = code.CodeLen - code.ChkLen, w = code.GFq - 1;
	std::vector<std::unique_ptr<CComm>> lanes;
	for (int i = 0; i < P; i++) {
		lanes.emplace_back(new CComm());
		if (!lanes.back()->Initial(link.sim, i, &code)) return -3;
		lanes.back()->SetEbN0(link.sim, i);
	}
	if (sigma_out) *sigma_out = lanes[0]->sigma_n;
	// lanes are independent (own RNG, PN register, buffers): run them on host threads, frame after frame per lane
	auto work = [&](int lo, int hi) {
		for (int i = lo; i < hi; i++) {
			CComm &c = *lanes[i];
			for (int f = 0; f < frames; f++) {
				c.FrontEnd();
				const size_t b = (size_t)f * P + i;
				memcpy(L_ch + b * N * w, c.RX_LLR_SYM.data(), sizeof(double) * N * w);
				if (tx_code) for (int n = 0; n < N; n++) tx_code[b * N + n] = c.TX_CODE_SYM[n];
				if (tx_msg) for (int n = 0; n < K; n++) tx_msg[b * K + n] = c.TX_MSG_SYM[n];
			}
		}
	};
	int T = (int)std::thread::hardware_concurrency();
	if (const char *e = getenv("NBL_HOST_THREADS")) T = atoi(e);
	if (T > 16) T = 16;
	if (T > P) T = P;
	if (T <= 1) work(0, P);
	else {
		std::vector<std::thread> th;
		for (int t = 0; t < T; t++) th.emplace_back(work, (int)((long long)P * t / T), (int)((long long)P * (t + 1) / T));
		for (auto &x : th) x.join();
	}
	return 0;
}

// Host link chain up to and including Channel_AWGN for `frames` cycles (no GPU): received samples rx [frames*P][L][2], the
// constellation index of every transmitted symbol tx_index [frames*P][L] and each lane's generator state in front of each
// frame state [frames*P][3] -- what the device-side channel is given, and what it must reproduce.
int nblh_channel(const char *profile, double ebn0, int frames, double *rx, unsigned char *tx_index, unsigned int *state, double *sigma_out)
{
	CSimulation sim;
	if (sim.Initial(profile) != 0) return -1;
	sim.EbN0 = ebn0;
	CLink link;
	link.sim = sim;
	CNBLDPC &code = link.code;
	if (!code.Initial(link.sim, -1)) return -2;
	const int P = sim.parallel;
	std::vector<std::unique_ptr<CComm>> lanes;
	for (int i = 0; i < P; i++) {
		lanes.emplace_back(new CComm());
		if (!lanes.back()->Initial(link.sim, i, &code)) return -3;
		lanes.back()->SetEbN0(link.sim, i);
	}
	if (sigma_out) *sigma_out = lanes[0]->sigma_n;
	const int L = lanes[0]->MOD_SYM_LEN;
	auto work = [&](int lo, int hi) {
		for (int i = lo; i < hi; i++) {
			CComm &c = *lanes[i];
			for (int f = 0; f < frames; f++) {
				const size_t b = (size_t)f * P + i;
				state[b * 3 + 0] = (unsigned int)(c.Rand.IX % 61967ul);
				state[b * 3 + 1] = (unsigned int)(c.Rand.IY % 63443ul);
				state[b * 3 + 2] = (unsigned int)(c.Rand.IZ % 63599ul);
				c.FrontEndToChannel();
				for (int s = 0; s < L; s++) {
					rx[(b * L + s) * 2] = c.RX_MOD_SYM[s].Real;
					rx[(b * L + s) * 2 + 1] = c.RX_MOD_SYM[s].Image;
					tx_index[b * L + s] = c.TX_MOD_IDX[s];
				}
			}
		}
	};
	int T = (int)std::thread::hardware_concurrency();
	if (const char *e = getenv("NBL_HOST_THREADS")) T = atoi(e);
	if (T > 16) T = 16;
	if (T > P) T = P;
	if (T <= 1) work(0, P);
	else {
		std::vector<std::thread> th;
		for (int t = 0; t < T; t++) th.emplace_back(work, (int)((long long)P * t / T), (int)((long long)P * (t + 1) / T));
		for (auto &x : th) x.join();
	}
	return L;
}

// Full simulation of one profile on the GPU; per Eb/N0 point: EbN0, errFrame, errSym, errBit, U_errFrame, frames, BER, SER, FER.
int nblh_simulate(const char *profile, int device, double *rows, int max_rows)
{
	CLink link;
	// device = -2: rehearsal of the multi-GPU split on a one-GPU box (two decoders, both on device 0)
	const std::vector<int> devs = device == -2 ? std::vector<int>{0, 0} : std::vector<int>{device};
	if (!link.Initial(profile, devs)) return -1;
	int n = 0;
	while (link.sim.NextSNR()) {
		if (!link.RunPoint(false)) return -2;
		if (n < max_rows) {
			double *r = rows + 9 * n;
			r[0] = link.sim.EbN0; r[1] = link.sim.errFrame; r[2] = link.sim.errSym; r[3] = link.sim.errBit; r[4] = link.sim.U_errFrame;
			r[5] = (link.sim.simCycle - 1) * link.sim.parallel; r[6] = link.sim.BER; r[7] = link.sim.SER; r[8] = link.sim.FER;
		}
		n++;
	}
	return n;
}

// encoder check: encode `count` random messages, return 0 if every codeword satisfies every parity check
int nblh_encode(const char *profile, const int *msg, int count, int *code_out)
{
	CSimulation sim;
	if (sim.Initial(profile) != 0) return -1;
	CNBLDPC code;
	if (!code.Initial(sim, -1)) return -2;
	const int N = code.CodeLen, K = N - code.ChkLen;
	std::vector<int> m(K);
	for (int c = 0; c < count; c++) {
		memcpy(m.data(), msg + (size_t)c * K, sizeof(int) * K);
		code.Encode(m.data(), code_out + (size_t)c * N);
	}
	return 0;
}
}
