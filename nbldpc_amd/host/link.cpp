// nbldpc_amd/host/link.cpp -- see link.h.
#include "link.h"
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <thread>

bool CLink::Initial(const std::string &profile, int device) { return Initial(profile, std::vector<int>{device}); }

// One decoder per listed GPU; lanes [P*g/G, P*(g+1)/G) go to the g-th one.  Frames are independent (main.cpp:46), so the
// multi-GPU path is a static split of the batch with no exchange between devices; Err() still runs in lane order.
bool CLink::Initial(const std::string &profile, const std::vector<int> &device_list)
{
	if (sim.Initial(profile) != 0) { error = "cannot read profile " + profile; return false; }
	devices = device_list.empty() ? std::vector<int>{0} : device_list;
	if (!code.Initial(sim, devices[0])) { error = code.LastError(); return false; }
	extra.clear();
	for (size_t g = 1; g < devices.size(); g++) {
		extra.emplace_back(new CNBLDPC());
		if (!extra.back()->Initial(sim, devices[g])) { error = extra.back()->LastError(); return false; }
	}
	lanes.clear();
	for (int i = 0; i < sim.parallel; i++) {
		lanes.emplace_back(new CComm());
		if (!lanes.back()->Initial(sim, i, &code)) { error = lanes.back()->error; return false; }
	}
	const size_t per = (size_t)code.CodeLen * (code.GFq - 1);
	if (const char *e = getenv("NBL_DEVICE_DEMOD")) device_demod = atoi(e) != 0;
	if (device_demod) {
		std::vector<int> src;
		lanes[0]->DemodSource(src);
		std::vector<double> cons;
		for (const CComplex &c : lanes[0]->CONSTELLATION) { cons.push_back(c.Real); cons.push_back(c.Image); }
		if (code.SetDemodulator(lanes[0]->modOrder, lanes[0]->MOD_SYM_LEN, cons.data(), src.data()) != 0) { error = code.LastError(); return false; }
		for (auto &x : extra)
			if (x->SetDemodulator(lanes[0]->modOrder, lanes[0]->MOD_SYM_LEN, cons.data(), src.data()) != 0) { error = x->LastError(); return false; }
		rx_batch.assign((size_t)2 * lanes[0]->MOD_SYM_LEN * sim.parallel, 0.0);
	}
	L_batch.assign(device_demod ? 0 : per * sim.parallel, 0.0);
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
			if (device_demod) {
				lanes[i]->FrontEndToChannel();
				const int L = lanes[i]->MOD_SYM_LEN;
				for (int s = 0; s < L; s++) {
					rx_batch[((size_t)i * L + s) * 2] = lanes[i]->RX_MOD_SYM[s].Real;
					rx_batch[((size_t)i * L + s) * 2 + 1] = lanes[i]->RX_MOD_SYM[s].Image;
				}
			} else {
				lanes[i]->FrontEnd();
				memcpy(&L_batch[per * i], lanes[i]->RX_LLR_SYM.data(), sizeof(double) * per);
			}
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
	const int G = (int)devices.size();
	const size_t rxper = device_demod ? (size_t)2 * lanes[0]->MOD_SYM_LEN : 0;
	const double sigma = lanes[0]->sigma_n;
	if (G == 1) {
		const int rc1 = device_demod ? code.DecodingBatchSamples(rx_batch.data(), sigma, P, out_batch.data(), conv.data(), iters.data())
		                             : code.DecodingBatch(L_batch.data(), P, out_batch.data(), conv.data(), iters.data());
		if (rc1 != 0) { error = code.LastError(); return false; }
	} else {
		std::vector<std::thread> th;
		std::vector<int> rc(G, 0);
		for (int gidx = 0; gidx < G; gidx++) {
			th.emplace_back([&, gidx]() {
				const int lo = (int)((long long)P * gidx / G), hi = (int)((long long)P * (gidx + 1) / G);
				CNBLDPC &dec = gidx == 0 ? code : *extra[gidx - 1];
				if (hi > lo)
					rc[gidx] = device_demod
					    ? dec.DecodingBatchSamples(&rx_batch[rxper * lo], sigma, hi - lo, &out_batch[(size_t)code.CodeLen * lo], &conv[lo], &iters[lo])
					    : dec.DecodingBatch(&L_batch[per * lo], hi - lo, &out_batch[(size_t)code.CodeLen * lo], &conv[lo], &iters[lo]);
			});
		}
		for (auto &x : th) x.join();
		for (int gidx = 0; gidx < G; gidx++)
			if (rc[gidx] != 0) { error = (gidx == 0 ? code : *extra[gidx - 1]).LastError(); return false; }
	}
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
