/* oracle/nbl_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the YongonY/NBLDPC decode hot path, used as the parity checker by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  It is NEVER linked into, loaded by, or
 * called from the product library (nbldpc_amd/csrc/libnbldpc_hip.so).
 *
 * Parity pin: `literal` modes reproduce the compiled reference (oracle/_ref, built from the unmodified
 * sources by oracle/Makefile) bit for bit on LLR state, hard decisions and return flags -- checked by
 * tests/test_oracle_golden.py against the tests/golden npz files generated with tests/golden/make_golden.py.
 * `canonical` modes define the residue-free value the HIP kernels compute.
 */
#ifndef NBL_ORACLE_H
#define NBL_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

/* decode methods: same codes as the reference's Simulation.h:3-9 */
#define NBLO_BP   1
#define NBLO_EMS  2
#define NBLO_TEMS 4

/* arithmetic modes */
#define NBLO_LITERAL   0 /* reference operation order incl. DFS add/subtract residue; BP in long double */
#define NBLO_CANONICAL 1 /* same configuration sets / same association order, no residue; BP in double   */
#define NBLO_CANONICAL_DFS 2 /* canonical values by plain enumeration (cross-check of the EMS dynamic program) */

typedef struct nblo_gf {
	int q, p, poly;
	int *mul; /* q*q */
	int *inv; /* q, inv[0] = 0 (the reference aborts on GFInverse(0), GF.cpp:41-46) */
} nblo_gf;

typedef struct nblo_code {
	int N, M, q, E, maxdv, maxdc;
	int *dv, *dc;            /* degrees */
	int *voff, *coff;        /* prefix sums: var-major edge id e = voff[n]+d, check-major id ce = coff[m]+k */
	int *v_chk, *v_h, *v_k;  /* per var-major edge: check index, coefficient, position inside that check  */
	int *c_var, *c_h, *c_d;  /* per check-major edge: var index, coefficient, position inside that var    */
	int *c2e;                /* check-major edge -> var-major edge */
} nblo_code;

typedef struct nblo_params {
	int method;      /* NBLO_BP / NBLO_EMS / NBLO_TEMS */
	int max_iter;
	int mode;        /* NBLO_LITERAL / NBLO_CANONICAL */
	int ems_nm, ems_nc;
	double ems_factor, ems_offset;
	int tems_nr, tems_nc;
	double tems_factor, tems_offset;
	int fixed_iters; /* 0: return at first zero syndrome (reference behaviour). 1: keep iterating to max_iter
	                    (timing only); outputs are frozen at the first zero syndrome either way. */
} nblo_params;

typedef struct nblo_decoder nblo_decoder;

int  nblo_gf_build(nblo_gf *gf, int q);                       /* from the primitive polynomial */
int  nblo_gf_load(nblo_gf *gf, int q, const char *arith_path);/* reference text format, GF.cpp:81-113 */
void nblo_gf_free(nblo_gf *gf);
int  nblo_primitive_poly(int q);

nblo_code *nblo_code_load(const char *path);                  /* reference code-file format, NBLDPC.cpp:147-205 */
nblo_code *nblo_code_from_edges(int N, int M, int q, int E, const int *edge_var, const int *edge_chk,
                                const int *edge_h);           /* edges in var-major order */
void nblo_code_free(nblo_code *c);

nblo_decoder *nblo_decoder_create(const nblo_code *code, const nblo_gf *gf, const nblo_params *prm);
void nblo_decoder_free(nblo_decoder *d);

/* One codeword.  L_ch: [N][q-1] doubles (L[a-1] = ln P(a)/P(0)).  out: [N] symbols.
 * returns 1 if a zero syndrome was reached, else 0; *iters = iteration at which it was reached (1-based) or
 * max_iter.  Mirrors CNBLDPC::Decoding (NBLDPC.cpp:607). */
int nblo_decode(nblo_decoder *d, const double *L_ch, int *out, int *iters);

/* Batch helper: B codewords back to back ([B][N][q-1]), optional threads over codewords. */
int nblo_decode_batch(nblo_decoder *const *decs, int nthreads, const double *L_ch, int B, int *out,
                      unsigned char *converged, int *iters);

/* state after the last nblo_decode call, var-major edge order, [.][q-1] doubles */
const double *nblo_state_post(const nblo_decoder *d);
const double *nblo_state_v2c(const nblo_decoder *d);
const double *nblo_state_c2v(const nblo_decoder *d); /* c2v re-ordered to var-major edge order */

/* single check-node update entry points (unit tests): v2c_in/c2v_out are [dc][q-1] for check m */
void nblo_check_ems(nblo_decoder *d, int m, const double *v2c_in, double *c2v_out);
void nblo_check_tems(nblo_decoder *d, int m, const double *v2c_in, double *c2v_out);
void nblo_check_bp(nblo_decoder *d, int m, const double *v2c_in, double *c2v_out);

#ifdef __cplusplus
}
#endif
#endif
