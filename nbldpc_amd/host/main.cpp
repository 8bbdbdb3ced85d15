// nbldpc_amd/host/main.cpp -- nbldpc_sim: the reference's driver (main.cpp) on the batched GPU decode path.
// usage: nbldpc_sim [profile = NBLDPC.Profile.txt] [device = 0]      (run it where ./SRC/ and the profile's files are)
#include <cstdlib>
#include <iostream>
#include "link.h"

int main(int argc, char **argv)
{
	CLink link;
	const std::string profile = argc > 1 ? argv[1] : "NBLDPC.Profile.txt";
	const int device = argc > 2 ? atoi(argv[2]) : 0;
	if (!link.Initial(profile, device)) {
		std::cerr << "initialisation failed: " << link.error << std::endl;
		return 1;
	}
	link.RunAll(true);
	std::cout << std::endl;
	return 0;
}
