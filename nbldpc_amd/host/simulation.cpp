// nbldpc_amd/host/simulation.cpp -- see simulation.h.
#include "simulation.h"
#include <chrono>
#include <iomanip>
#include <iostream>

using std::cout;
using std::endl;

static double wall_now()
{
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// The profile is positional: every value is preceded by a fixed number of label tokens and nothing else identifies it
// (Simulation.cpp:58-107).  `skip(n)` eats the labels.
int CSimulation::Initial(const std::string &profilename)
{
	ProfileFileName = profilename;
	std::ifstream in(profilename);
	if (!in.is_open()) return -1;
	auto skip = [&](int n) { std::string w; for (int i = 0; i < n; i++) in >> w; };
	skip(1); in >> GFq;
	skip(2); in >> NonBinaryFileName;
	skip(2); in >> PuntureVarDegree;
	skip(2); in >> decodeMethod;
	skip(2); in >> maxIter;
	skip(1); in >> parallel;
	skip(1); in >> crcLen;
	skip(1); in >> crc_correctLen;
	skip(1); in >> OSD_order;
	skip(1); in >> OSD_factor;
	skip(1); in >> OSD_flag;
	skip(2); in >> ems_nm;
	skip(2); in >> ems_nc;
	skip(2); in >> ems_factor;
	skip(2); in >> ems_offset;
	skip(2); in >> tems_nr;
	skip(2); in >> tems_nc;
	skip(2); in >> tems_factor;
	skip(2); in >> tems_offset;
	skip(2); in >> bs_tems_nm;
	skip(2); in >> bs_tems_nc;
	skip(2); in >> bs_tems_factor;
	skip(2); in >> bs_tems_offset;
	skip(2); in >> snrBegin;
	skip(2); in >> snrStep;
	skip(2); in >> snrStop;
	skip(1); in >> nQAM;
	skip(1); in >> ConstellationFileName;
	skip(2); in >> randomMsg;
	skip(3); in >> minErrFrame;
	skip(3); in >> U_minErrFrame;
	skip(3); in >> minSimCycle;
	skip(2); in >> randomseed;
	skip(3); in >> showSimFrameStep;
	if (!in) return -2;
	EbN0 = snrBegin - snrStep; // Simulation.cpp:119
	return 0;
}

int CSimulation::ClearSimuCount()
{
	simCycle = 0;
	errFrame = errBit = errSym = 0;
	U_errFrame = U_errBit = U_errSym = 0;
	decoded_frames = 0;
	decoded_iters = 0;
	start = clock();
	wall_start = wall_now();
	return 0;
}

bool CSimulation::NextSNR()
{
	EbN0 += snrStep;
	return !(EbN0 > snrStop);
}

// Simulation.cpp:373-375: note the post-increment and the '<=' comparisons (minErrFrame = -1 runs an exact frame count)
bool CSimulation::SimulateThisSNR()
{
	return ((simCycle++) * parallel <= minSimCycle) || (errFrame <= minErrFrame) || (U_errFrame <= U_minErrFrame);
}

int CSimulation::Show(int mode)
{
	switch (mode) {
	case Screen_Logo:
		cout << "******************************************************************************\n"
		     << "*******   NB-LDPC decoding simulation -- MI355X batched decode path   *******\n"
		     << "******************************************************************************\n" << endl;
		break;
	case Screen_Conf:
		cout << "Configuration lists as follows:\n"
		     << "Code: " << NonBinaryFileName << "\nGF: " << GFq << "\tPuncture Variable Degree: " << PuntureVarDegree
		     << "\tMaximum iterations: " << maxIter << "\tparallel: " << parallel << endl;
		cout << "crcLen: " << crcLen << "\tcrcusedforcorrect: " << crc_correctLen << "\tOSD_order: " << OSD_order
		     << "\tOSD_factor: " << OSD_factor << "\tOSD_flag: " << OSD_flag << endl;
		if (decodeMethod == BP_DECODE) cout << "Algorithm: BP" << decodeMethod << endl;
		else if (decodeMethod == EMS_DECODE)
			cout << "Algorithm: EMS\tNm: " << ems_nm << "\tNc: " << ems_nc << "\tFactor: " << ems_factor << "\tOffset: " << ems_offset << endl;
		else if (decodeMethod == T_EMS_DECODE)
			cout << "Algorithm: Trellis EMS\tNr: " << tems_nr << "\tNc: " << tems_nc << "\tFactor: " << tems_factor << "\tOffset: " << tems_offset << endl;
		else cout << "Algorithm: method " << decodeMethod << " (not available on this decode path)" << endl;
		cout << "Modulation: " << nQAM << "-QAM" << "\tConf: " << ConstellationFileName << endl;
		cout << "SNR: " << snrBegin << ":" << snrStep << ":" << snrStop << (randomMsg ? "\tRandom Sequence\n" : "\tALL 0 sequence\n")
		     << "Min Err Frame: " << minErrFrame << "\tMin U_Err Frame: " << U_minErrFrame << "\tMin Sim Frame: " << minSimCycle
		     << "\tRand Seed: " << randomseed << endl;
		break;
	case Screen_Head:
		cout << "EbN0\t" << "Error\t" << "CRCmiss\t" << "BER\t\t" << "SER\t\t" << "FER\t\t" << "U-FER\t\t" << "Time" << endl;
		break;
	case Screen_Sim_Data:
		if (long(simCycle) % showSimFrameStep == 0) {
			cout << std::defaultfloat << EbN0 << '\t' << errFrame << '\t' << U_errFrame << '\t';
			cout << std::scientific << BER << '\t' << SER << '\t' << FER << '\t' << U_FER << '\t' << simCycle << '\r';
		}
		break;
	case Screen_Sim_End_Data: {
		stop = clock();
		cout << std::defaultfloat << EbN0 << '\t' << errFrame << '\t' << U_errFrame << '\t';
		cout << std::scientific << BER << '\t' << SER << '\t' << FER << '\t' << U_FER << '\t'
		     << 1.0 * (stop - start) / CLOCKS_PER_SEC << endl;
		const double wall = wall_now() - wall_start;
		cout << std::defaultfloat << "      frames " << decoded_frames << "  wall " << wall << " s  " << (wall > 0 ? decoded_frames / wall : 0.0)
		     << " codewords/s (link chain + decode)";
		if (wall > 0 && bytes_per_iter > 0)
			cout << "  " << (decoded_frames * bytes_per_frame + decoded_iters * bytes_per_iter) / wall * 1e-9 << " algorithmic GB/s ("
			     << (decoded_frames > 0 ? decoded_iters / decoded_frames : 0.0) << " iterations per frame)";
		cout << endl;
		break;
	}
	default: break;
	}
	return 0;
}
