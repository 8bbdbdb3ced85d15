// nbldpc_amd/host/main.cpp -- nbldpc_sim: the reference's driver (main.cpp) on the batched GPU decode path.
// usage: nbldpc_sim [profile = NBLDPC.Profile.txt] [devices = 0]     e.g. "0,1,2,3,4,5,6,7": lanes are sharded over the GPUs
// (run it where ./SRC/ and the profile's files are)
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>
#include "link.h"

int main(int argc, char **argv)
{
	CLink link;
	const std::string profile = argc > 1 ? argv[1] : "NBLDPC.Profile.txt";
	std::vector<int> devices;
	{
		std::string list = argc > 2 ? argv[2] : "0";
		size_t pos = 0;
		while (pos <= list.size()) {
			size_t c = list.find(',', pos);
			if (c == std::string::npos) c = list.size();
			if (c > pos) devices.push_back(atoi(list.substr(pos, c - pos).c_str()));
			pos = c + 1;
		}
	}
	if (!link.Initial(profile, devices)) {
		std::cerr << "initialisation failed: " << link.error << std::endl;
		return 1;
	}
	link.RunAll(true);
	std::cout << std::endl;
	const double tot = link.t_front + link.t_decode + link.t_err;
	std::cerr << "[nbldpc_sim] " << link.n_frames << " frames: set-up " << link.t_init << " s, front-end " << link.t_front << " s, decode "
	          << link.t_decode << " s, error count " << link.t_err << " s  -> " << (tot > 0 ? link.n_frames / tot : 0.0) << " frames/s" << std::endl;
	return 0;
}
