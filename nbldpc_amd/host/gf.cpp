// nbldpc_amd/host/gf.cpp -- see gf.h.  File format (GF.cpp:81-113): one title line, then three blocks each announced by two
// label tokens: q*q products, q*q sums, q inverses.
#include "gf.h"
#include <cmath>
#include <fstream>
#include <iostream>
#include <sstream>

int CGF::GFInverse(int a) const
{
	if (a == 0) {
		std::cerr << "Div 0 Error!" << std::endl;
		return -1;
	}
	return TableInverse[a];
}

bool CGF::Initial(int GFq, const std::string &src_dir)
{
	q = GFq;
	p = int(std::log(double(GFq)) / std::log(2.0));
	std::ostringstream name;
	name << src_dir << "/Arith.Table.GF." << q << ".txt";
	std::ifstream fin(name.str());
	if (!fin.is_open()) {
		error = "Cannot open " + name.str();
		std::cerr << error << std::endl;
		return false;
	}
	std::string rub;
	std::getline(fin, rub);
	TableMultiply.assign((size_t)q * q, 0);
	TableAdd.assign((size_t)q * q, 0);
	TableInverse.assign(q, 0);
	fin >> rub >> rub;
	for (int i = 0; i < q * q; i++) fin >> TableMultiply[i];
	fin >> rub >> rub;
	for (int i = 0; i < q * q; i++) fin >> TableAdd[i];
	fin >> rub >> rub;
	for (int i = 0; i < q; i++) fin >> TableInverse[i];
	if (!fin) {
		error = "Truncated arithmetic table " + name.str();
		std::cerr << error << std::endl;
		return false;
	}
	return true;
}
