// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Our own driver translation unit that links the UNMODIFIED reference objects (everything except
// main.cpp, which needs MSVC <ppl.h>) and dumps golden vectors as .npy files.  It uses only the public
// members of the reference classes:
//   CSimulation::Initial            /root/reference/Simulation.cpp:53
//   CComm::Initial / SetEbN0        /root/reference/Comm.cpp:48 / :157
//   CComm::GenerateMessage..Demodulate  /root/reference/Comm.cpp:194-407
//   CNBLDPC::Decoding               /root/reference/NBLDPC.cpp:607
// It exists only in the build container (the reference does not travel to the GPU box); its outputs are
// packed into tests/golden/*.npz by tests/golden/make_golden.py.
//
// usage (cwd must contain ./SRC/ with the GF tables, see oracle/Makefile):
//   ref_driver dump <profile> <outdir> <EbN0> <frames> <iters,csv> <state_iters,csv>
//   ref_driver fer  <profile>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <sstream>
#include "Simulation.h"
#include "Comm.h"

static void write_npy(const std::string& path, const char* descr, const std::vector<size_t>& shape,
                      const void* data, size_t nbytes)
{
	FILE* f = fopen(path.c_str(), "wb");
	if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
	std::ostringstream hd;
	hd << "{'descr': '" << descr << "', 'fortran_order': False, 'shape': (";
	for (size_t i = 0; i < shape.size(); i++) { hd << shape[i] << ","; }
	hd << "), }";
	std::string h = hd.str();
	size_t total = 10 + h.size() + 1;
	size_t pad = (64 - total % 64) % 64;
	h.append(pad, ' ');
	h.push_back('\n');
	unsigned char magic[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, 0, 0};
	magic[8] = (unsigned char)(h.size() & 0xff);
	magic[9] = (unsigned char)(h.size() >> 8);
	fwrite(magic, 1, 10, f);
	fwrite(h.data(), 1, h.size(), f);
	fwrite(data, 1, nbytes, f);
	fclose(f);
}

static std::vector<int> parse_csv(const char* s)
{
	std::vector<int> v;
	std::stringstream ss(s);
	std::string tok;
	while (std::getline(ss, tok, ',')) { if (!tok.empty()) v.push_back(atoi(tok.c_str())); }
	return v;
}

static int syndrome_ok(CNBLDPC& c, const int* out)
{
	for (int row = 0; row < c.ChkLen; row++) {
		int s = 0;
		for (int d = 0; d < c.ChkDegree[row]; d++) {
			s = c.GF.GFAdd(s, c.GF.GFMultiply(c.ChkLinkGFe[row][d], out[c.ChkLink[row][d]]));
		}
		if (s) return 0;
	}
	return 1;
}

static int run_dump(int argc, char** argv)
{
	if (argc < 8) { fprintf(stderr, "dump: need profile outdir ebn0 frames iters state_iters\n"); return 2; }
	std::string profile = argv[2], outdir = argv[3];
	double ebn0 = atof(argv[4]);
	int frames = atoi(argv[5]);
	std::vector<int> iters = parse_csv(argv[6]);
	std::vector<int> state_iters = parse_csv(argv[7]);

	CSimulation sim;
	sim.Initial(profile);
	sim.EbN0 = ebn0;
	int P = sim.parallel;
	CComm* comm = new CComm[P];
	for (int i = 0; i < P; i++) comm[i].Initial(sim, i);
	for (int i = 0; i < P; i++) comm[i].SetEbN0(sim, i);

	CNBLDPC& c0 = comm[0].NBLDPC;
	const int N = c0.CodeLen, M = c0.ChkLen, q = c0.GFq;
	int E = 0;
	for (int n = 0; n < N; n++) E += c0.VarDegree[n];
	const size_t B = (size_t)frames * P;              // frame-major: b = f*P + lane
	std::vector<double> Lch(B * N * (q - 1));
	std::vector<int32_t> txcode(B * N), txmsg(B * (N - M));
	std::vector<int32_t> out(iters.size() * B * N), flag(iters.size() * B), synok(iters.size() * B);
	std::vector<double> sigma(P);
	// per-edge states (var-major edge order: e = sum_{n'<n} dv[n'] + d) for frame 0, every lane
	std::vector<double> st_post(state_iters.size() * P * N * (q - 1));
	std::vector<double> st_v2c(state_iters.size() * P * E * (q - 1));
	std::vector<double> st_c2v(state_iters.size() * P * E * (q - 1));   // c2v stored in the same var-major edge order

	for (int f = 0; f < frames; f++) {
		for (int i = 0; i < P; i++) {
			CComm& c = comm[i];
			size_t b = (size_t)f * P + i;
			sigma[i] = c.sigma_n;
			c.GenerateMessage(); c.Encode(); c.Puncture(); c.Modulate(); c.Channel_AWGN(); c.Demodulate();
			for (int n = 0; n < N; n++) {
				txcode[b * N + n] = c.TX_CODE_SYM[n];
				memcpy(&Lch[(b * N + n) * (q - 1)], c.RX_LLR_SYM[n], sizeof(double) * (q - 1));
			}
			for (int n = 0; n < N - M; n++) txmsg[b * (N - M) + n] = c.TX_MSG_SYM[n];
			for (size_t k = 0; k < iters.size(); k++) {
				c.NBLDPC.maxIter = iters[k];
				int r = c.NBLDPC.Decoding(c.RX_LLR_SYM, c.RX_DECODE_SYM, c.ReliableSeri_Symbol, c.ReliableSeri_BIT);
				for (int n = 0; n < N; n++) out[(k * B + b) * N + n] = c.RX_DECODE_SYM[n];
				flag[k * B + b] = r;
				synok[k * B + b] = syndrome_ok(c.NBLDPC, c.RX_DECODE_SYM);
			}
			if (f == 0) {
				for (size_t k = 0; k < state_iters.size(); k++) {
					c.NBLDPC.maxIter = state_iters[k];
					c.NBLDPC.Decoding(c.RX_LLR_SYM, c.RX_DECODE_SYM, c.ReliableSeri_Symbol, c.ReliableSeri_BIT);
					size_t e = 0;
					for (int n = 0; n < N; n++) {
						memcpy(&st_post[((k * P + i) * N + n) * (q - 1)], c.NBLDPC.L_post[n], sizeof(double) * (q - 1));
						for (int d = 0; d < c.NBLDPC.VarDegree[n]; d++, e++) {
							int row = c.NBLDPC.VarLink[n][d], dc = c.NBLDPC.VarLinkDc[n][d];
							memcpy(&st_v2c[((k * P + i) * E + e) * (q - 1)], c.NBLDPC.L_v2c[n][d], sizeof(double) * (q - 1));
							memcpy(&st_c2v[((k * P + i) * E + e) * (q - 1)], c.NBLDPC.L_c2v[row][dc], sizeof(double) * (q - 1));
						}
					}
				}
			}
		}
		fprintf(stderr, "frame %d/%d done\n", f + 1, frames);
	}
	std::vector<int32_t> it32(iters.begin(), iters.end()), st32(state_iters.begin(), state_iters.end());
	write_npy(outdir + "/L_ch.npy", "<f8", {B, (size_t)N, (size_t)(q - 1)}, Lch.data(), Lch.size() * 8);
	write_npy(outdir + "/tx_code.npy", "<i4", {B, (size_t)N}, txcode.data(), txcode.size() * 4);
	write_npy(outdir + "/tx_msg.npy", "<i4", {B, (size_t)(N - M)}, txmsg.data(), txmsg.size() * 4);
	write_npy(outdir + "/iters.npy", "<i4", {iters.size()}, it32.data(), it32.size() * 4);
	write_npy(outdir + "/out.npy", "<i4", {iters.size(), B, (size_t)N}, out.data(), out.size() * 4);
	write_npy(outdir + "/ret.npy", "<i4", {iters.size(), B}, flag.data(), flag.size() * 4);
	write_npy(outdir + "/syn_ok.npy", "<i4", {iters.size(), B}, synok.data(), synok.size() * 4);
	write_npy(outdir + "/sigma.npy", "<f8", {(size_t)P}, sigma.data(), sigma.size() * 8);
	write_npy(outdir + "/state_iters.npy", "<i4", {state_iters.size()}, st32.data(), st32.size() * 4);
	if (!state_iters.empty()) {
		write_npy(outdir + "/st_post.npy", "<f8", {state_iters.size(), (size_t)P, (size_t)N, (size_t)(q - 1)}, st_post.data(), st_post.size() * 8);
		write_npy(outdir + "/st_v2c.npy", "<f8", {state_iters.size(), (size_t)P, (size_t)E, (size_t)(q - 1)}, st_v2c.data(), st_v2c.size() * 8);
		write_npy(outdir + "/st_c2v.npy", "<f8", {state_iters.size(), (size_t)P, (size_t)E, (size_t)(q - 1)}, st_c2v.data(), st_c2v.size() * 8);
	}
	return 0;
}

// Same control flow as the reference's main() (main.cpp:13-65) with parallel_for replaced by a serial loop
// (lanes are independent, main.cpp:46).  Prints one machine-readable line per Eb/N0 point.
static int run_fer(int argc, char** argv)
{
	if (argc < 3) return 2;
	CSimulation sim;
	sim.Initial(argv[2]);
	int P = sim.parallel;
	CComm* comm = new CComm[P];
	for (int i = 0; i < P; i++) comm[i].Initial(sim, i);
	while (sim.NextSNR()) {
		sim.ClearSimuCount();
		for (int i = 0; i < P; i++) comm[i].SetEbN0(sim, i);
		while (sim.SimulateThisSNR()) {
			for (int i = 0; i < P; i++) comm[i].Transmission();
			for (int i = 0; i < P; i++) comm[i].Err(sim);
		}
		printf("{\"EbN0\": %.17g, \"errFrame\": %.17g, \"errSym\": %.17g, \"errBit\": %.17g, \"U_errFrame\": %.17g, "
		       "\"frames\": %.17g, \"BER\": %.17g, \"SER\": %.17g, \"FER\": %.17g, \"cpu_s\": %.3f}\n",
		       sim.EbN0, sim.errFrame, sim.errSym, sim.errBit, sim.U_errFrame, (sim.simCycle - 1) * P,
		       sim.BER, sim.SER, sim.FER, double(clock() - sim.start) / CLOCKS_PER_SEC);
		fflush(stdout);
	}
	return 0;
}

int main(int argc, char** argv)
{
	if (argc < 2) { fprintf(stderr, "usage: ref_driver dump|fer ...\n"); return 2; }
	if (!strcmp(argv[1], "dump")) return run_dump(argc, argv);
	if (!strcmp(argv[1], "fer")) return run_fer(argc, argv);
	return 2;
}
