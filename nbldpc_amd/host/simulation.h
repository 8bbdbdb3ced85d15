// nbldpc_amd/host/simulation.h -- CSimulation: configuration bag, Eb/N0 loop, stop rule and report lines of the
// reference (Simulation.h, Simulation.cpp).  Same public field names, so code written against the reference compiles.
#pragma once
#include <ctime>
#include <fstream>
#include <string>

#define BP_DECODE 1
#define EMS_DECODE 2
#define MinMax_DECODE 3
#define T_EMS_DECODE 4
#define T_MinMax_DECODE 5
#define OSD_DECODE 6
#define BS_TEMS_DECODE 7

#define Screen_Logo 0
#define Screen_Conf 1
#define Screen_Head 3
#define Screen_Sim_Data 7
#define Screen_Sim_End_Data 8

class CSimulation {
public:
	std::string ProfileFileName, NonBinaryFileName, ConstellationFileName;
	int GFq = 0, decodeMethod = 0, parallel = 1, maxIter = 0, randomMsg = 0, PuntureVarDegree = 0, nQAM = 0, randomseed = 0;
	double snrBegin = 0, snrStep = 0, snrStop = 0, EbN0 = 0;
	int crcLen = 0, crc_correctLen = 0, OSD_order = -1, OSD_flag = 0;
	double OSD_factor = 0;
	int ems_nm = 0, ems_nc = 0, tems_nr = 0, tems_nc = 0, bs_tems_nm = 0, bs_tems_nc = 0;
	double ems_factor = 0, ems_offset = 0, tems_factor = 0, tems_offset = 0, bs_tems_factor = 0, bs_tems_offset = 0;
	double errFrame = 0, errBit = 0, errSym = 0, FER = 0, BER = 0, SER = 0;
	double U_errFrame = 0, U_errBit = 0, U_errSym = 0, U_FER = 0, U_BER = 0, U_SER = 0;
	double simCycle = 0;
	int minSimCycle = 0, minErrFrame = 0, U_minErrFrame = 0, showSimFrameStep = 1;
	clock_t start = 0, stop = 0;
	double wall_start = 0;       // extension: wall-clock seconds (the reference reports CPU time only)
	double decoded_frames = 0;   // extension: frames decoded at this Eb/N0 point
	double decoded_iters = 0;    // extension: iterations those frames ran (early exit: up to the first zero syndrome)
	double bytes_per_frame = 0, bytes_per_iter = 0; // extension: algorithmic bytes: 8(q-1)[N + I (N + 4E + D E)] per frame

	int Initial(const std::string &profilename);
	int Show(int mode);
	int ClearSimuCount();
	bool NextSNR();
	bool SimulateThisSNR();
};
