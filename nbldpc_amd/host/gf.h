// nbldpc_amd/host/gf.h -- CGF: GF(q) arithmetic tables loaded from ./SRC/Arith.Table.GF.<q>.txt (GF.cpp:50-113).
#pragma once
#include <string>
#include <vector>

class CGF {
public:
	int q = 0, p = 0;
	std::vector<int> TableMultiply; // [q*q]
	std::vector<int> TableAdd;      // [q*q]
	std::vector<int> TableInverse;  // [q]

	bool Initial(int GFq, const std::string &src_dir = "./SRC");
	int GFAdd(int a, int b) const { return TableAdd[a * q + b]; }
	int GFMultiply(int a, int b) const { return TableMultiply[a * q + b]; }
	int GFInverse(int a) const; // GFInverse(0): "Div 0 Error!" (GF.cpp:41-46) -> returns -1 here instead of exit(-1)
	std::string error;
};
