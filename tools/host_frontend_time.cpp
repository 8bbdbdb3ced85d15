// tools/host_frontend_time.cpp -- per-frame cost of the host link chain up to Modulate (what nbldpc_sim's front-end threads run per
// lane and cycle), by stage, for P lanes on T threads: is the front-end bound by its arithmetic or by the lanes' working set?
//   tools/host_frontend_time.sh [P] [T] [cycles] [stages]   (builds this file against nbldpc_amd/host and runs it in a scratch work dir;
//   stages = 1 times the stages one by one on the lane's own buffers, default: FrontEndToModulate as a whole, reported as message_crc)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <thread>
#include <vector>
#include "comm.h"
#include "link.h"
int main(int argc, char **argv)
{
	const int P = argc > 1 ? atoi(argv[1]) : 16384, T = argc > 2 ? atoi(argv[2]) : 16, F = argc > 3 ? atoi(argv[3]) : 4;
	const bool stages = argc > 4 && atoi(argv[4]) != 0; // 1: the stages one by one on the lane's own buffers; 0: FrontEndToModulate as the harness calls it
	CSimulation sim;
	if (sim.Initial("NBLDPC.Profile.txt") != 0) return 1;
	sim.EbN0 = 1.5;
	CLink link;
	link.sim = sim;
	CNBLDPC &code = link.code;
	if (!code.Initial(link.sim, -1)) return 2;
	std::vector<std::unique_ptr<CComm>> lanes;
	for (int i = 0; i < P; i++) {
		lanes.emplace_back(new CComm());
		if (!lanes.back()->Initial(link.sim, i, &code)) return 3;
		lanes.back()->SetEbN0(link.sim, i);
	}
	std::vector<std::vector<double>> t(T, std::vector<double>(5, 0.0));
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto sec = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
	const auto w0 = now();
	auto work = [&](int th) {
		for (int f = 0; f < F; f++)
			for (int i = (int)((long long)P * th / T); i < (int)((long long)P * (th + 1) / T); i++) {
				CComm &c = *lanes[i];
				if (!stages) {
					unsigned int st[3];
					const auto a = now();
					c.FrontEndToModulate(st);
					t[th][0] += sec(a, now());
					continue;
				}
				const auto a = now();
				c.GenerateMessage();
				const auto b = now();
				c.Encode();
				const auto d = now();
				c.Puncture();
				const auto e = now();
				c.Modulate();
				const auto g = now();
				c.Rand.Skip(4ul * (unsigned long)c.MOD_SYM_LEN);
				const auto h = now();
				t[th][0] += sec(a, b); t[th][1] += sec(b, d); t[th][2] += sec(d, e); t[th][3] += sec(e, g); t[th][4] += sec(g, h);
			}
	};
	std::vector<std::thread> th;
	for (int k = 0; k < T; k++) th.emplace_back(work, k);
	for (auto &x : th) x.join();
	const double wall = sec(w0, now()), n = (double)F * P;
	double s[5] = {0, 0, 0, 0, 0};
	for (int k = 0; k < T; k++)
		for (int j = 0; j < 5; j++) s[j] += t[k][j];
	printf("{\"lanes\": %d, \"threads\": %d, \"cycles\": %d, \"wall_ms_per_cycle\": %.2f, \"us_per_frame_thread\": {\"message_crc\": %.2f, \"encode\": %.2f, "
	       "\"puncture\": %.2f, \"modulate\": %.2f, \"rand_skip\": %.2f, \"total\": %.2f}}\n",
	       P, T, F, 1e3 * wall / F, 1e6 * s[0] / n, 1e6 * s[1] / n, 1e6 * s[2] / n, 1e6 * s[3] / n, 1e6 * s[4] / n, 1e6 * (s[0] + s[1] + s[2] + s[3] + s[4]) / n);
	return 0;
}
