// nbldpc_amd/host/link.h -- CLink: the body of the reference's main() (main.cpp:13-65) with the parallel_for over lanes
// (main.cpp:46) replaced by ONE batched decode per simulation cycle.
#pragma once
#include <memory>
#include <vector>
#include "comm.h"

class CLink {
public:
	CSimulation sim;
	CNBLDPC code;                      // code parameters + encoder + the decoder of devices[0]
	std::vector<std::unique_ptr<CNBLDPC>> extra; // one more decoder per additional GPU (lanes sharded, no collective)
	std::vector<int> devices;
	std::vector<std::unique_ptr<CComm>> lanes;
	std::vector<double> L_batch[2];    // [parallel][N][q-1], one buffer per cycle in flight
	bool device_demod = true;          // ship received samples, demodulate on the GPU (bit-identical L_ch);
	                                   // NBL_DEVICE_DEMOD=0: build L_ch on the host as the reference's Demodulate does
	std::vector<double> rx_batch[2];   // [parallel][MOD_SYM_LEN][2]
	bool device_noise = true;          // AWGN channel + CRand on the GPU as well (bit-identical samples);
	                                   // NBL_DEVICE_NOISE=0 draws the noise on the host threads.  Needs device_demod.
	std::vector<unsigned char> txi_batch[2]; // [parallel][MOD_SYM_LEN] constellation indices
	std::vector<unsigned int> state_batch[2]; // [parallel][3] generator states in front of the frame
	bool channel_ok[2] = {true, true};
	bool Channel(int slot);            // device-side channel of the slot's frames (runs under the previous cycle's decode)
	bool pipeline = true;              // NBL_PIPELINE=0: strictly serial cycles
	int host_threads = 1;
	std::vector<int> out_batch, iters;
	std::vector<uint8_t> conv;
	std::string error;
	double t_init = 0, t_front = 0, t_decode = 0, t_err = 0; // wall seconds per phase, summed over all cycles
	long long n_frames = 0;

	bool Initial(const std::string &profile, int device = 0);
	bool Initial(const std::string &profile, const std::vector<int> &device_list);
	void BeginSNR();                   // ClearSimuCount + SetEbN0 on every lane
	bool Cycle();                      // front-ends, one batched decode, Err per lane in lane order
	void FrontEnds(int slot);          // every lane's link chain for one cycle, into buffer `slot` (threaded over lanes)
	bool Decode(int slot);             // one batched decode per GPU of buffer `slot`
	void CountErrors(int slot);        // unpack + compare per lane (threaded), counters added in lane order
	bool RunPoint(bool verbose);       // one Eb/N0 point: BeginSNR, then cycles until the stop rule ends it (pipelined by default)
	void RunAll(bool verbose);
};
