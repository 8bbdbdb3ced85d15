// nbldpc_amd/host/link.cpp -- see link.h.
#include "link.h"
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <thread>

bool CLink::Initial(const std::string &profile, int device)
{
	if (sim.Initial(profile) != 0) { error = "cannot read profile " + profile; return false; }
	if (!code.Initial(sim, device)) { error = code.LastError(); return false; }
	lanes.clear();
	for (int i = 0; i < sim.parallel; i++) {
		lanes.emplace_back(new CComm());
		if (!lanes.back()->Initial(sim, i, &code)) { error = lanes.back()->error; return false; }
	}
	const size_t per = (size_t)code.CodeLen * (code.GFq - 1);
	L_batch.assign(per * sim.parallel, 0.0);
	out_batch.assign((size_t)code.CodeLen * sim.parallel, 0);
	iters.assign(sim.parallel, 0);
	conv.assign(sim.parallel, 0);
	return true;
}

void CLink::BeginSNR()
{
	sim.ClearSimuCount();
	for (int i = 0; i < sim.parallel; i++) lanes[i]->SetEbN0(sim, i);
}

bool CLink::Cycle()
{
	const int P = sim.parallel;
	const size_t per = (size_t)code.CodeLen * (code.GFq - 1);
	// lanes own their RNG / PN / buffers, so their front-ends run in parallel exactly like the reference's parallel_for
	// (main.cpp:46); results do not depend on the thread count
	auto work = [&](int lo, int hi) {
		for (int i = lo; i < hi; i++) {
			lanes[i]->FrontEnd();
			memcpy(&L_batch[per * i], lanes[i]->RX_LLR_SYM.data(), sizeof(double) * per);
		}
	};
	int T = 1;
	if (const char *e = getenv("NBL_HOST_THREADS")) T = atoi(e);
	else { T = (int)std::thread::hardware_concurrency(); if (T > 16) T = 16; }
	if (T > P) T = P;
	if (T <= 1) work(0, P);
	else {
		std::vector<std::thread> th;
		for (int t = 0; t < T; t++) th.emplace_back(work, (int)((long long)P * t / T), (int)((long long)P * (t + 1) / T));
		for (auto &x : th) x.join();
	}
	if (code.DecodingBatch(L_batch.data(), P, out_batch.data(), conv.data(), iters.data()) != 0) { error = code.LastError(); return false; }
	for (int i = 0; i < P; i++) {
		lanes[i]->TakeDecoded(&out_batch[(size_t)code.CodeLen * i], conv[i] != 0);
		lanes[i]->Err(sim); // serial, lane order: same accumulation order as main.cpp:48-51
	}
	sim.decoded_frames += P;
	return true;
}

void CLink::RunAll(bool verbose)
{
	if (verbose) { sim.Show(Screen_Logo); sim.Show(Screen_Conf); sim.Show(Screen_Head); }
	while (sim.NextSNR()) {
		BeginSNR();
		while (sim.SimulateThisSNR()) {
			if (!Cycle()) { std::cerr << error << std::endl; return; }
			if (verbose) sim.Show(Screen_Sim_Data);
		}
		if (verbose) sim.Show(Screen_Sim_End_Data);
	}
}
