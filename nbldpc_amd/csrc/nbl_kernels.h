// nbldpc_amd/csrc/nbl_kernels.h -- host-callable launchers of nbl_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include "nbl_common.h"

hipError_t nbl_launch_init(const double *d_Lin, const NblGraphDev &g, const NblWork &w, int B, int write_v2c, hipStream_t st);
hipError_t nbl_launch_demod(const double *d_rx, int L, double sigma, int mod_order, const double *d_cons, const int *d_src,
                            const NblGraphDev &g, const NblWork &w, int B, hipStream_t st);
hipError_t nbl_launch_vn(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool damp, hipStream_t st);
hipError_t nbl_launch_syn(const NblGraphDev &g, const NblWork &w, const NblRun &r, hipStream_t st);
hipError_t nbl_launch_compact(const uint8_t *done, int B, int *active, int *n_act, hipStream_t st);
hipError_t nbl_launch_cn_ems(const NblGraphDev &g, const NblWork &w, const NblRun &r, hipStream_t st);
hipError_t nbl_launch_unpad(const double *src, double *dst, const int *map, int rows, int q, hipStream_t st);
size_t nbl_ems_lds_bytes(const NblGraphDev &g, int nm, int layers);
int nbl_ems_layers(const NblGraphDev &g, int nc);

// specialised EMS check node (nbl_cn_ems256.hip)
bool nbl_ems256_applicable(const NblGraphDev &g, bool all_dc4, int nm, int nc);
size_t nbl_ems256_lds_bytes(int nm);
hipError_t nbl_launch_cn_ems256(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool fused, hipStream_t st);

// T-EMS and log-QSPA check nodes (nbl_cn_tems.hip, nbl_cn_bp.hip)
hipError_t nbl_launch_cn_tems(const NblGraphDev &g, const NblWork &w, const NblRun &r, hipStream_t st);
hipError_t nbl_launch_cn_bp(const NblGraphDev &g, const NblWork &w, const NblRun &r, hipStream_t st);

// small fields (q <= 32), 64 / q checks per wave (nbl_cn_small.hip); method as in include/nbldpc.h (1 BP, 2 EMS, 4 T-EMS)
bool nbl_small_applicable(const NblGraphDev &g, int method, int min_dc, int nm, int nc);
hipError_t nbl_launch_cn_ems_small(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool fused, hipStream_t st);
hipError_t nbl_launch_cn_tems_small(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool fused, hipStream_t st);
hipError_t nbl_launch_cn_bp_small(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool fused, hipStream_t st);

// T-EMS check node for GF(64), check degree 4 (nbl_cn_tems64.hip)
bool nbl_tems64_applicable(const NblGraphDev &g, bool all_dc4, int nr, int nc);
hipError_t nbl_launch_cn_tems64(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool fused, hipStream_t st);

// T-EMS check node for GF(256), check degree 4 (nbl_cn_tems256.hip)
bool nbl_tems256_applicable(const NblGraphDev &g, bool all_dc4, int nr, int nc);
hipError_t nbl_launch_cn_tems256(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool fused, hipStream_t st);

// log-QSPA check node for GF(256), check degree 4 (nbl_cn_bp256.hip)
bool nbl_bp256_applicable(const NblGraphDev &g, bool all_dc4);
hipError_t nbl_launch_cn_bp256(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool fused, hipStream_t st);

// EMS check node for GF(64): four checks per wave, four symbols per lane (nbl_cn_ems64.hip)
bool nbl_ems64_applicable(const NblGraphDev &g, int min_dc, int nm, int nc);
hipError_t nbl_launch_cn_ems64(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool fused, hipStream_t st);

// log-QSPA check node for GF(64), check degree 4: four checks per wave, four symbols per lane (nbl_cn_bp64.hip)
bool nbl_bp64_applicable(const NblGraphDev &g, bool all_dc4);
hipError_t nbl_launch_cn_bp64(const NblGraphDev &g, const NblWork &w, const NblRun &r, bool fused, hipStream_t st);

// AWGN channel + CRand on the device (nbl_noise.hip)
hipError_t nbl_launch_noise_gen(const uint32_t *state, const uint32_t *jump, int L, int B, double *fn, uint32_t *flag_idx, double *flag_arg,
                                unsigned *flag_count, unsigned cap, hipStream_t st);
hipError_t nbl_launch_noise_patch(double *fn, const uint32_t *flag_idx, const double *val, unsigned n, hipStream_t st);
hipError_t nbl_launch_noise_finish(const double *fn, const uint8_t *tx_index, const double *cons, double sigma, int L, int B, double *rx, hipStream_t st);
