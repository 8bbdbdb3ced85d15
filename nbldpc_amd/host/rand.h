// nbldpc_amd/host/rand.h -- CRand: the reference's noise source, needed bit for bit for FER parity.
// Three multiplicative congruential generators summed modulo 1 (Rand.cpp:17-28) and the cosine branch of Box-Muller
// (Rand.cpp:31-37).  The expression order below is the reference's; do not "simplify" it.
#pragma once
#include <cmath>

class CRand {
public:
	unsigned long IX = 0, IY = 0, IZ = 0;

	double Rand_Uniform()
	{
		IX = (IX * 249) % 61967;
		IY = (IY * 251) % 63443;
		IZ = (IZ * 252) % 63599;
		double t = (((double)IX) / ((double)61967)) + (((double)IY) / ((double)63443)) + (((double)IZ) / ((double)63599));
		t -= (int)t;
		return t;
	}

	// the state after n calls of Rand_Uniform: X <- X * A^n mod m for each of the three generators
	void Skip(unsigned long n)
	{
		auto pw = [](unsigned long a, unsigned long k, unsigned long m) {
			unsigned long r = 1 % m, x = a % m;
			for (; k; k >>= 1) { if (k & 1) r = r * x % m; x = x * x % m; }
			return r;
		};
		IX = (IX % 61967) * pw(249, n, 61967) % 61967;
		IY = (IY % 63443) * pw(251, n, 63443) % 63443;
		IZ = (IZ % 63599) * pw(252, n, 63599) % 63599;
	}

	double Rand_Norm(double mu, double sigma)
	{
		const double u1 = Rand_Uniform();
		const double u2 = Rand_Uniform();
		return mu + sigma * std::cos(2 * std::acos(-1.0) * u2) * std::sqrt(-2.0 * std::log(1.0 - u1));
	}
};
