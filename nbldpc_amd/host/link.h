// nbldpc_amd/host/link.h -- CLink: the body of the reference's main() (main.cpp:13-65) with the parallel_for over lanes
// (main.cpp:46) replaced by ONE batched decode per simulation cycle.
#pragma once
#include <memory>
#include <vector>
#include "comm.h"

class CLink {
public:
	CSimulation sim;
	CNBLDPC code;
	std::vector<std::unique_ptr<CComm>> lanes;
	std::vector<double> L_batch;       // [parallel][N][q-1]
	std::vector<int> out_batch, iters;
	std::vector<uint8_t> conv;
	std::string error;

	bool Initial(const std::string &profile, int device = 0);
	void BeginSNR();                   // ClearSimuCount + SetEbN0 on every lane
	bool Cycle();                      // front-ends, one batched decode, Err per lane in lane order
	void RunAll(bool verbose);
};
