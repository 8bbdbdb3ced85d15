/* include/nbldpc.h -- C ABI of the MI355X non-binary LDPC decode path (libnbldpc_hip.so).
 *
 * Drop-in boundary for the reference's message-passing hot path:
 *
 *   reference interface (YongonY/NBLDPC)                      replaced by
 *   --------------------------------------------------------  -----------------------------------------
 *   bool CNBLDPC::Initial(CSimulation&)   NBLDPC.h:42,         nbl_create()  (graph + GF tables + decoder
 *        NBLDPC.cpp:140-377 (graph parse, cross indices,                      parameters; device buffers)
 *        per-algorithm scratch)
 *   int  CNBLDPC::Decoding(double** L_ch, int* DecodeOutput,   nbl_decode_batch()         host buffers
 *        int*, int*)        NBLDPC.h:71, NBLDPC.cpp:607-641    nbl_decode_batch_device()  HBM-resident
 *        -> Decoding_BP :643, Decoding_EMS :778,
 *           Decoding_TEMS :929
 *   public members L_post / L_v2c / L_c2v  NBLDPC.h:65-68      nbl_read_state()  (parity tests only)
 *   ~CNBLDPC                                                   nbl_destroy()
 *
 * One reference call decodes ONE codeword on the calling thread; one call here decodes a BATCH of B
 * independent codewords (the reference's `parallel` lanes, main.cpp:46) on one GPU.
 * Plain C types only: no exceptions, no exit(); every failure is a negative nbl_status and a message
 * retrievable with nbl_last_error().  There is NO CPU fallback: without a HIP device nbl_create fails.
 */
#ifndef NBLDPC_H
#define NBLDPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NBL_ABI_VERSION 1

/* decode methods: the reference's numbering, Simulation.h:3-9 */
#define NBL_METHOD_BP   1 /* exact log-domain QSPA, forward/backward */
#define NBL_METHOD_EMS  2 /* configuration-set EMS, nm/nc truncated   */
#define NBL_METHOD_TEMS 4 /* trellis EMS                              */

typedef enum nbl_status {
	NBL_OK = 0,
	NBL_ERR_ARG = -1,        /* bad argument / inconsistent graph (reference: undefined behaviour or exit(-1)) */
	NBL_ERR_UNSUPPORTED = -2,/* method 3/5/6/7 (reference prints "not developed" and exits, NBLDPC.cpp:618-638) */
	NBL_ERR_NO_DEVICE = -3,  /* no HIP device: there is deliberately no CPU path                               */
	NBL_ERR_HIP = -4,        /* a HIP runtime call failed                                                      */
	NBL_ERR_NOMEM = -5
} nbl_status;

/* Tanner graph exactly as the reference's code file lists it (NBLDPC.cpp:147-205), 0-based indices.
 * Both directions are given because their ORDER is semantically relevant: the order of a check's edges
 * fixes the floating-point association order and the tie-breaks of every check-node algorithm. */
typedef struct nbl_code_desc {
	int32_t N, M, q;            /* CodeLen, ChkLen, GFq                                        */
	const int32_t *var_deg;     /* [N]  VarDegree                                              */
	const int32_t *chk_deg;     /* [M]  ChkDegree                                              */
	const int32_t *var_chk;     /* [E]  VarLink, variable-major                                */
	const int32_t *var_h;       /* [E]  VarLinkGFe                                             */
	const int32_t *chk_var;     /* [E]  ChkLink, check-major                                   */
	const int32_t *chk_h;       /* [E]  ChkLinkGFe                                             */
} nbl_code_desc;

/* Decoder parameters: the fields CNBLDPC::Initial copies out of CSimulation (NBLDPC.cpp:142-143, 267-304). */
typedef struct nbl_params {
	int32_t method;             /* NBL_METHOD_*                                                */
	int32_t max_iter;           /* sim.maxIter                                                 */
	int32_t ems_nm, ems_nc;     /* sim.ems_nm, sim.ems_nc                                      */
	double  ems_factor, ems_offset;
	int32_t tems_nr, tems_nc;   /* sim.tems_nr, sim.tems_nc                                    */
	double  tems_factor, tems_offset;
	int32_t fixed_iters;        /* 0: a codeword stops at its first zero syndrome (reference).   */
	                            /* 1: every codeword runs max_iter iterations (throughput runs); */
	                            /*    outputs are frozen at the first zero syndrome either way.  */
	int32_t poll_every;         /* fixed_iters==0: ask the device every k iterations whether all */
	                            /* codewords are done (0 = never, run max_iter launches)         */
	int32_t max_batch;          /* workspace is sized for this many codewords (grows on demand)  */
} nbl_params;

typedef struct nbl_decoder nbl_decoder;

/* gf_mul: q*q multiplication table, gf_inv: q inverses (gf_inv[0] ignored) -- the tables CGF::Initial loads
 * from ./SRC/Arith.Table.GF.<q>.txt (GF.cpp:81-113).  Addition must be XOR (checked against gf_mul's
 * distributivity is not attempted; the reference's tables are polynomial-basis for every q it ships).
 * Shapes: q = 4 .. 256 (a power of two), check and variable degrees up to 8, ems_nm <= q, T-EMS with p * (largest check
 * degree) <= 32 (q = 2^p; the path code of TEMS_ConstructConf in 32 bits); anything else is NBL_ERR_UNSUPPORTED / NBL_ERR_ARG
 * with a message, at creation, never at the first decode. */
nbl_status nbl_create(const nbl_code_desc *code, const uint16_t *gf_mul, const uint16_t *gf_inv,
                      const nbl_params *params, int device, nbl_decoder **out);
void nbl_destroy(nbl_decoder *dec);

/* L_ch: [B][N][q-1] doubles, L_ch[b][n][a-1] = ln P(x_n=a)/P(x_n=0)  (RX_LLR_SYM, Comm.cpp:340-407).
 * out_sym: [B][N] decided symbols (DecodeOutput).  converged: [B] 1 = zero syndrome reached (the
 * reference's return value).  iters: [B] iteration of the first zero syndrome, or max_iter.
 * converged / iters may be NULL. */
nbl_status nbl_decode_batch(nbl_decoder *dec, const double *L_ch, int32_t B, int32_t *out_sym,
                            uint8_t *converged, int32_t *iters);

/* Same, every pointer is DEVICE memory on the decoder's device; work is enqueued on `stream`
 * (a hipStream_t, NULL = the decoder's own stream) and NOT synchronised -- except when poll_every > 0. */
nbl_status nbl_decode_batch_device(nbl_decoder *dec, const double *d_L_ch, int32_t B, int32_t *d_out_sym,
                                   uint8_t *d_converged, int32_t *d_iters, void *stream);

/* ---- device-side soft demodulator: L_ch is built in HBM from the received samples ------------------
 * Replaces CComm::Demodulate (Comm.cpp:340-407) for the two cases the reference implements: BPSK (modOrder == 2, :342-380)
 * and one constellation point per code symbol (modOrder == GFq, :382-398), with the reference's expression order, so
 * the LLRs are bit-identical to the host computation.  Host->device traffic drops from N(q-1) doubles to 2 L doubles per
 * codeword (16x for BPSK GF(256), 128x for 256-QAM). */
typedef struct nbl_demod_desc {
	int32_t mod_order;           /* 2, or q                                                              */
	int32_t n_mod_sym;           /* L = received samples per codeword (MOD_SYM_LEN)                      */
	const double *constellation; /* [mod_order][2] (Real, Image) = CONSTELLATION[]; used when mod_order == q */
	const int32_t *src;          /* mod_order == 2: [N*p] sample index carrying code bit b, -1 = punctured (LLR 0, :350-354)
	                                mod_order == q: [N]   sample index of code symbol n,    -1 = punctured (:386-393)   */
} nbl_demod_desc;
nbl_status nbl_set_demodulator(nbl_decoder *dec, const nbl_demod_desc *demod);
/* rx: HOST buffer [B][L][2] (Real, Image) = RX_MOD_SYM after the channel; sigma = sigma_n (Comm.cpp:176-177) */
nbl_status nbl_decode_batch_samples(nbl_decoder *dec, const double *rx, double sigma, int32_t B, int32_t *out_sym,
                                    uint8_t *converged, int32_t *iters);

/* ---- AWGN channel on the device ------------------------------------------------------------------------
 * Replaces CComm::Channel_AWGN (Comm.cpp:328-337) and its noise source CRand (Rand.cpp:17-37) for a batch of lanes: the received
 * samples RX = TX + Rand_Norm(0, sigma) are formed in HBM with the reference's generator and expression order, bit for bit (the
 * log / cos values whose rounding a GPU cannot settle are evaluated by the host's libm inside the call), then demodulated
 * (nbl_set_demodulator must have been called, WITH the constellation points, also for BPSK) and decoded.
 *   tx_index   HOST [B][L] uint8   index into the constellation of every transmitted symbol (CComm::Modulate, Comm.cpp:310-325)
 *   lane_state HOST [B][3] uint32  IX, IY, IZ of each lane's CRand before the frame's first draw
 * The frame consumes 4 L uniform draws per lane (real and imaginary part of every symbol, two draws each); the caller moves
 * its copy of each lane's generator on with nbl_rand_advance(state, 4 L). */
nbl_status nbl_decode_batch_noise(nbl_decoder *dec, const uint8_t *tx_index, const uint32_t *lane_state, double sigma, int32_t B,
                                  int32_t *out_sym, uint8_t *converged, int32_t *iters);
/* The same in two phases, for callers that overlap the channel of batch k+1 with the decode of batch k (two host threads):
 * nbl_channel_batch forms the samples of a batch into the decoder's resident buffer `slot` (0 or 1) on a second stream and returns
 * when they are complete; nbl_decode_batch_resident demodulates and decodes what a slot holds.  A channel call and a decode call
 * on DIFFERENT slots may run concurrently; otherwise the handle is single-threaded like every other call.  The two threads keep
 * separate error texts (nbl_last_error reports the decode side's, then the channel side's after " | channel: ").
 * tx_index values must be below mod_order (NBL_ERR_ARG otherwise). */
nbl_status nbl_channel_batch(nbl_decoder *dec, int32_t slot, const uint8_t *tx_index, const uint32_t *lane_state, double sigma, int32_t B);
nbl_status nbl_decode_batch_resident(nbl_decoder *dec, int32_t slot, double sigma, int32_t B, int32_t *out_sym, uint8_t *converged,
                                     int32_t *iters);
/* state <- state after `draws` calls of CRand::Rand_Uniform (Rand.cpp:17-28); pure host arithmetic */
void nbl_rand_advance(uint32_t state[3], uint64_t draws);

/* Message state of codeword b after the last decode call (host buffers, any may be NULL):
 * post [N][q-1], v2c [E][q-1], c2v [E][q-1], edges in variable-major order.  For parity tests.
 * post and c2v are the reference's members at return.  v2c differs for a codeword that CONVERGED at iteration k >= 2: the
 * variable-node pass writes the messages of iteration k before the syndrome is known, the reference returns with those of
 * iteration k-1 (NBLDPC.cpp:693-715 precedes :718-744); they are never used afterwards.  In fixed-iteration mode (timing)
 * the message buffers of a converged codeword keep being updated; only its outputs are frozen. */
nbl_status nbl_read_state(nbl_decoder *dec, int32_t b, double *post, double *v2c, double *c2v);

/* Keep L_post of the last variable-node pass so nbl_read_state can return it (costs N q-vectors of
 * extra HBM writes per iteration; off by default). */
nbl_status nbl_set_record_state(nbl_decoder *dec, int32_t on);

/* Per-phase device time of the last decode call in milliseconds (HIP events on the launch stream):
 * ms[0] variable-node kernels, ms[1] syndrome kernels, ms[2] check-node kernels, ms[3] whole call.
 * Only filled when profiling was switched on with nbl_set_profiling(dec, 1). */
nbl_status nbl_set_profiling(nbl_decoder *dec, int32_t on);
nbl_status nbl_last_timing(nbl_decoder *dec, double ms[4], int64_t launches[3]);

int32_t nbl_abi_version(void);
const char *nbl_last_error(const nbl_decoder *dec); /* dec may be NULL: error of the last failed nbl_create */
size_t nbl_workspace_bytes(const nbl_decoder *dec);

#ifdef __cplusplus
}
#endif
#endif
