// nbldpc_amd/host/link.cpp -- see link.h.
#include "link.h"
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <chrono>
#include <thread>

static double now_s()
{
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

bool CLink::Initial(const std::string &profile, int device) { return Initial(profile, std::vector<int>{device}); }

// One decoder per listed GPU; lanes [P*g/G, P*(g+1)/G) go to the g-th one.  Frames are independent (main.cpp:46), so the
// multi-GPU path is a static split of the batch with no exchange between devices; Err() still runs in lane order.
bool CLink::Initial(const std::string &profile, const std::vector<int> &device_list)
{
	const double t0 = now_s();
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
	if (const char *e = getenv("NBL_DEVICE_NOISE")) device_noise = atoi(e) != 0;
	if (!device_demod || lanes[0]->CONSTELLATION.size() > 256) device_noise = false;
	if (const char *e = getenv("NBL_PIPELINE")) pipeline = atoi(e) != 0;
	if (const char *e = getenv("NBL_HOST_THREADS")) host_threads = atoi(e);
	else { // 16 threads keep one GPU fed; more GPUs decode more lanes per cycle
		host_threads = (int)std::thread::hardware_concurrency();
		const int cap = 16 * (int)devices.size();
		if (host_threads > cap) host_threads = cap;
	}
	if (host_threads > sim.parallel) host_threads = sim.parallel;
	if (host_threads < 1) host_threads = 1;
	if (device_demod) {
		std::vector<int> src;
		lanes[0]->DemodSource(src);
		std::vector<double> cons;
		for (const CComplex &c : lanes[0]->CONSTELLATION) { cons.push_back(c.Real); cons.push_back(c.Image); }
		if (code.SetDemodulator(lanes[0]->modOrder, lanes[0]->MOD_SYM_LEN, cons.data(), src.data()) != 0) { error = code.LastError(); return false; }
		for (auto &x : extra)
			if (x->SetDemodulator(lanes[0]->modOrder, lanes[0]->MOD_SYM_LEN, cons.data(), src.data()) != 0) { error = x->LastError(); return false; }
	}
	for (int slot = 0; slot < (pipeline ? 2 : 1); slot++) {
		rx_batch[slot].assign(device_demod && !device_noise ? (size_t)2 * lanes[0]->MOD_SYM_LEN * sim.parallel : 0, 0.0);
		txi_batch[slot].assign(device_noise ? (size_t)lanes[0]->MOD_SYM_LEN * sim.parallel : 0, 0);
		state_batch[slot].assign(device_noise ? (size_t)3 * sim.parallel : 0, 0);
		L_batch[slot].assign(device_demod ? 0 : per * sim.parallel, 0.0);
	}
	{ // algorithmic bytes of the decode: per frame 8(q-1)N + 4N + 4, per iteration 8(q-1)(N + 4E + D E), D = 1 with damping
		double E = 0;
		for (int n = 0; n < code.CodeLen; n++) E += code.VarDegree[n];
		const double w = 8.0 * (code.GFq - 1), N = code.CodeLen, D = (sim.decodeMethod == 2) ? 0 : 1;
		sim.bytes_per_frame = w * N + 4 * N + 4;
		sim.bytes_per_iter = w * (N + 4 * E + D * E);
	}
	out_batch.assign((size_t)code.CodeLen * sim.parallel, 0);
	iters.assign(sim.parallel, 0);
	conv.assign(sim.parallel, 0);
	t_init = now_s() - t0;
	return true;
}

void CLink::BeginSNR()
{
	sim.ClearSimuCount();
	for (int i = 0; i < sim.parallel; i++) lanes[i]->SetEbN0(sim, i);
}

template <class F> static void over_lanes(int P, int T, F work)
{
	if (T <= 1) { work(0, P); return; }
	std::vector<std::thread> th;
	for (int t = 0; t < T; t++) th.emplace_back(work, (int)((long long)P * t / T), (int)((long long)P * (t + 1) / T));
	for (auto &x : th) x.join();
}

// lanes own their RNG / PN / buffers, so their front-ends run in parallel exactly like the reference's parallel_for
// (main.cpp:46); results do not depend on the thread count
void CLink::FrontEnds(int slot)
{
	const size_t per = (size_t)code.CodeLen * (code.GFq - 1);
	over_lanes(sim.parallel, host_threads, [&](int lo, int hi) {
		for (int i = lo; i < hi; i++) {
			if (device_noise) {
				lanes[i]->FrontEndToModulate(&state_batch[slot][(size_t)3 * i]);
				memcpy(&txi_batch[slot][(size_t)i * lanes[i]->MOD_SYM_LEN], lanes[i]->TX_MOD_IDX.data(), lanes[i]->MOD_SYM_LEN);
			} else if (device_demod) {
				lanes[i]->FrontEndToChannel();
				const int L = lanes[i]->MOD_SYM_LEN;
				double *rx = &rx_batch[slot][(size_t)i * L * 2];
				for (int s = 0; s < L; s++) {
					rx[2 * s] = lanes[i]->RX_MOD_SYM[s].Real;
					rx[2 * s + 1] = lanes[i]->RX_MOD_SYM[s].Image;
				}
			} else {
				lanes[i]->FrontEnd();
				memcpy(&L_batch[slot][per * i], lanes[i]->RX_LLR_SYM.data(), sizeof(double) * per);
			}
			lanes[i]->HoldTx(slot);
		}
	});
	if (device_noise) channel_ok[slot] = Channel(slot);
}

// device-side AWGN channel of the frames the front-ends just produced (second stream of every decoder): in the pipelined
// driver this runs while the previous cycle is being decoded
bool CLink::Channel(int slot)
{
	const int P = sim.parallel, G = (int)devices.size(), L = lanes[0]->MOD_SYM_LEN;
	const double sigma = lanes[0]->sigma_n;
	std::vector<int> rc(G, 0);
	auto shard = [&](int gidx) {
		const int lo = (int)((long long)P * gidx / G), hi = (int)((long long)P * (gidx + 1) / G);
		CNBLDPC &dec = gidx == 0 ? code : *extra[gidx - 1];
		if (hi > lo) rc[gidx] = dec.ChannelBatch(slot, &txi_batch[slot][(size_t)L * lo], &state_batch[slot][(size_t)3 * lo], sigma, hi - lo);
	};
	if (G == 1) shard(0);
	else {
		std::vector<std::thread> th;
		for (int gidx = 0; gidx < G; gidx++) th.emplace_back(shard, gidx);
		for (auto &x : th) x.join();
	}
	for (int gidx = 0; gidx < G; gidx++)
		if (rc[gidx] != 0) { error = (gidx == 0 ? code : *extra[gidx - 1]).LastError(); return false; }
	return true;
}

bool CLink::Decode(int slot)
{
	if (device_noise && !channel_ok[slot]) return false;
	const int P = sim.parallel;
	const size_t per = (size_t)code.CodeLen * (code.GFq - 1);
	const int G = (int)devices.size();
	const size_t rxper = device_demod ? (size_t)2 * lanes[0]->MOD_SYM_LEN : 0;
	const double sigma = lanes[0]->sigma_n;
	std::vector<int> rc(G, 0);
	auto shard = [&](int gidx) {
		const int lo = (int)((long long)P * gidx / G), hi = (int)((long long)P * (gidx + 1) / G);
		CNBLDPC &dec = gidx == 0 ? code : *extra[gidx - 1];
		if (hi > lo && device_noise)
			rc[gidx] = dec.DecodingBatchResident(slot, sigma, hi - lo, &out_batch[(size_t)code.CodeLen * lo], &conv[lo], &iters[lo]);
		else if (hi > lo)
			rc[gidx] = device_demod
			    ? dec.DecodingBatchSamples(&rx_batch[slot][rxper * lo], sigma, hi - lo, &out_batch[(size_t)code.CodeLen * lo], &conv[lo], &iters[lo])
			    : dec.DecodingBatch(&L_batch[slot][per * lo], hi - lo, &out_batch[(size_t)code.CodeLen * lo], &conv[lo], &iters[lo]);
	};
	if (G == 1) shard(0);
	else {
		std::vector<std::thread> th;
		for (int gidx = 0; gidx < G; gidx++) th.emplace_back(shard, gidx);
		for (auto &x : th) x.join();
	}
	for (int gidx = 0; gidx < G; gidx++)
		if (rc[gidx] != 0) { error = (gidx == 0 ? code : *extra[gidx - 1]).LastError(); return false; }
	return true;
}

// The counters are integer-valued doubles, so adding the per-lane counts in lane order afterwards gives exactly what the
// reference's serial Err() loop gives (main.cpp:48-51); the rates are its last lane's expressions on the final totals.
void CLink::CountErrors(int slot)
{
	const int P = sim.parallel;
	std::vector<double> es(P), eb(P);
	std::vector<int> ok(P);
	over_lanes(P, host_threads, [&](int lo, int hi) {
		for (int i = lo; i < hi; i++) {
			lanes[i]->TakeDecoded(&out_batch[(size_t)code.CodeLen * i], conv[i] != 0);
			lanes[i]->ErrCount(slot, es[i], eb[i], ok[i]);
		}
	});
	for (int i = 0; i < P; i++) CComm::ErrAccumulate(sim, es[i], eb[i], ok[i]);
	lanes[P - 1]->ErrRates(sim);
	sim.decoded_frames += P;
	for (int i = 0; i < P; i++) sim.decoded_iters += iters[i];
	n_frames += P;
}

bool CLink::Cycle()
{
	const double t0 = now_s();
	FrontEnds(0);
	const double t1 = now_s();
	if (!Decode(0)) return false;
	const double t2 = now_s();
	CountErrors(0);
	t_front += t1 - t0;
	t_decode += t2 - t1;
	t_err += now_s() - t2;
	return true;
}

// Pipelined driver: while the GPUs decode cycle k the host threads already run the link chain of cycle k+1 into the other
// buffer.  Every cycle sees exactly the frames it would see in the serial order (lanes keep their own generators), and the
// stop rule is evaluated after each cycle's error count as before; the one speculative front-end at the end of an Eb/N0 point
// is discarded, and SetEbN0 re-seeds every lane for the next point (Comm.cpp:160-173), so the results are identical.
bool CLink::RunPoint(bool verbose)
{
	BeginSNR();
	if (!pipeline) {
		while (sim.SimulateThisSNR()) {
			if (!Cycle()) return false;
			if (verbose) sim.Show(Screen_Sim_Data);
		}
		return true;
	}
	int slot = 0;
	bool primed = false;
	while (sim.SimulateThisSNR()) {
		double t0 = now_s();
		if (!primed) { FrontEnds(slot); primed = true; t_front += now_s() - t0; t0 = now_s(); }
		bool ok = true;
		std::thread dec([&]() { ok = Decode(slot); });
		FrontEnds(slot ^ 1); // cycle k+1 (speculative if the stop rule ends this point)
		const double t1 = now_s();
		dec.join();
		const double t2 = now_s();
		if (!ok) return false;
		CountErrors(slot);
		t_front += t1 - t0;   // host time that ran under the decode
		t_decode += t2 - t1;  // decode time NOT hidden by the front-end
		t_err += now_s() - t2;
		slot ^= 1;
		if (verbose) sim.Show(Screen_Sim_Data);
	}
	return true;
}

void CLink::RunAll(bool verbose)
{
	if (verbose) { sim.Show(Screen_Logo); sim.Show(Screen_Conf); sim.Show(Screen_Head); }
	while (sim.NextSNR()) {
		if (!RunPoint(verbose)) { std::cerr << error << std::endl; return; }
		if (verbose) sim.Show(Screen_Sim_End_Data);
	}
}
