// nbldpc_amd/csrc/nbl_common.h -- shared host/device declarations of the MI355X decode path.
//
// HBM layout (all FP64, the reference's message precision):
//   Lch [B][N][q]    channel LLRs (slot a holds ln P(a)/P(0); slot 0 holds 0.0,
//                    which is the reference's implicit "symbol 0 carries LLR 0", NBLDPC.cpp:1718)
//   v2c [B][E][q]    variable-to-check messages, variable-major edge order  (L_v2c[col][d], NBLDPC.h:65); absent on the
//                    fused EMS path, where it is recomputed inside the check-node kernel
//   c2v [B][E][q]    check-to-variable messages, check-major edge order     (L_c2v[row][d], NBLDPC.h:66); two buffers on
//                    the fused path (flooding schedule: read iteration t-1, write iteration t)
//   dec [B][N]       tentative hard decisions of the current iteration
// A q-vector is one contiguous, 8*q-byte aligned run, so a wave streams it with fully coalesced loads.
#pragma once
#include <stdint.h>

#define NBL_WAVE 64
#define NBL_MAXDC 8   // largest check degree supported by the kernels (reference codes: 4 and 5)
#define NBL_MAXDV 8

struct NblGraphDev {
	int N, M, E, q, p, poly, maxdc, maxdv;
	const int *voff;    // [N+1] variable-major edge offsets
	const int *coff;    // [M+1] check-major edge offsets
	const int *v_cpos;  // [E] variable-major edge -> check-major position (its c2v slot)
	const int *c_epos;  // [E] check-major edge -> variable-major position (its v2c slot)
	const int *c_var;   // [E] check-major edge -> variable
	const int *c_h;     // [E] check-major edge coefficient
	const int *c_hinv;  // [E] inverse coefficient
	const uint8_t *mul; // [q*q] GF multiplication table (syndrome kernel)
	const int *c_nbr;   // [E][4] q <= 64, variable degrees 2 and 3 only (else NULL): for check-major edge e of variable n the c2v slots of
	                    // n's edges in order (third = -1 at degree 2) and, in [3], 1 if e is n's first edge (nbl_cn_small.hip, fused)
	const unsigned long long *ems_toff; // [E][64] GF(256), all checks of degree 4 only (else NULL): for check-major edge e and lane l the
	                    // byte offsets 8 * (h_e * a) of a = 2l, 2l+1, 128+2l, 129+2l, 16 bits each (nbl_cn_ems256.hip)
	const int *dv2_row; // [M][16] every check of degree 4 and every variable of degree 2 only (else NULL): all a fused iteration needs
	                    // to address the inputs of check m, in ONE 64-byte row: [0..3] variable of edge j, [4..7] c2v slot of that
	                    // variable's first edge, [8..11] of its second edge, [12..15] variable-major position of edge j, bit 31 set
	                    // when edge j IS the variable's first edge (one scalar load instead of a chain of four dependent ones)
};

struct NblWork {
	double *Lch, *v2c, *c2v, *post; // post only when state recording is on
	const double *c2v_prev;         // fused EMS iteration: c2v of the previous iteration (read), c2v = this iteration (written)
	int c2v_prev_shared;            // 1: c2v_prev is ONE [E][q] block that every codeword reads (iteration 1: the all-zero c2v of iteration 0)
	int store_v2c;                  // fused EMS iteration: also write v2c (state read-back only)
	int *dec, *out, *iters;
	int *edge_dec;                  // fused damped iterations (T-EMS, BP): hard decision of every v2c vector of the previous iteration
	uint8_t *done;
	int *n_done;                    // device counter of converged codewords
	const int *active;              // early exit, large batches: codewords still iterating, ascending, rebuilt after every window of
	                                // iterations (NULL = every codeword has its own slot); grids then cover r.B SLOTS, not codewords
	const int *n_act;               // number of valid entries of `active`
	unsigned long long *stamps;     // [16] debug: per-section cycle sums of the check-node kernel (NULL = off)
};

struct NblRun {
	int B;                          // codeword slots the grid covers (= batch size unless w.active is set, then an upper bound of *w.n_act)
	int iter, fixed_iters;
	int nm, nc, nr;
	double factor, offset;
	double damp_old, damp_new;
};
