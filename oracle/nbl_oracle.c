/* oracle/nbl_oracle.c -- TEST INFRASTRUCTURE ONLY (see nbl_oracle.h).
 *
 * CPU restatement of the YongonY/NBLDPC message-passing decoders.  Written from the behaviour of the
 * reference (file:line cited per function), on flat arrays instead of the reference's ragged pointers.
 * Floating-point operation ORDER follows the reference exactly in NBLO_LITERAL mode, because the parity
 * claim is bit-level: every add/subtract below that looks redundant is there on purpose.
 */
#define _GNU_SOURCE
#include "nbl_oracle.h"
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ */
/* GF(2^p)                                                                                           */
/* ------------------------------------------------------------------------------------------------ */

/* Primitive polynomials named in the first line of the reference's Arith.Table.GF.<q>.txt files. */
int nblo_primitive_poly(int q)
{
	switch (q) {
	case 4: return 7;
	case 8: return 11;
	case 16: return 19;
	case 32: return 37;
	case 64: return 67;
	case 128: return 137;
	case 256: return 285;
	case 512: return 529;
	default: return 0;
	}
}

static int ilog2(int q)
{
	int p = 0;
	while ((1 << p) < q) p++;
	return p;
}

static int gf_alloc(nblo_gf *gf, int q)
{
	gf->q = q;
	gf->p = ilog2(q);
	gf->poly = nblo_primitive_poly(q);
	gf->mul = (int *)malloc(sizeof(int) * q * q);
	gf->inv = (int *)calloc(q, sizeof(int));
	return (gf->mul && gf->inv) ? 0 : -1;
}

/* Tables the reference loads from text (GF.cpp:81-113) are polynomial-basis arithmetic: add = XOR,
 * mul = carry-less product reduced by the primitive polynomial.  tests/ check this against the files. */
int nblo_gf_build(nblo_gf *gf, int q)
{
	if ((1 << ilog2(q)) != q || !nblo_primitive_poly(q)) return -1;
	if (gf_alloc(gf, q)) return -1;
	for (int a = 0; a < q; a++) {
		for (int b = 0; b < q; b++) {
			int acc = 0, x = a;
			for (int i = 0; i < gf->p; i++) {
				if ((b >> i) & 1) acc ^= x;
				x <<= 1;
				if (x & q) x ^= gf->poly;
			}
			gf->mul[a * q + b] = acc;
			if (acc == 1) gf->inv[a] = b;
		}
	}
	return 0;
}

int nblo_gf_load(nblo_gf *gf, int q, const char *path)
{
	FILE *f = fopen(path, "r");
	char line[512], w1[64], w2[64];
	int v;
	if (!f) return -1;
	if (gf_alloc(gf, q)) { fclose(f); return -1; }
	if (!fgets(line, sizeof line, f)) goto bad;
	if (fscanf(f, "%63s %63s", w1, w2) != 2) goto bad;
	for (int i = 0; i < q * q; i++) if (fscanf(f, "%d", &gf->mul[i]) != 1) goto bad;
	if (fscanf(f, "%63s %63s", w1, w2) != 2) goto bad;
	for (int i = 0; i < q * q; i++) {
		if (fscanf(f, "%d", &v) != 1) goto bad;
		if (v != ((i / q) ^ (i % q))) goto bad; /* addition must be XOR */
	}
	if (fscanf(f, "%63s %63s", w1, w2) != 2) goto bad;
	for (int i = 0; i < q; i++) if (fscanf(f, "%d", &gf->inv[i]) != 1) goto bad;
	fclose(f);
	return 0;
bad:
	fclose(f);
	return -2;
}

void nblo_gf_free(nblo_gf *gf)
{
	free(gf->mul);
	free(gf->inv);
	gf->mul = gf->inv = NULL;
}

/* ------------------------------------------------------------------------------------------------ */
/* Tanner graph                                                                                      */
/* ------------------------------------------------------------------------------------------------ */

static int code_finish(nblo_code *c, const int *chk_var, const int *chk_h)
{
	/* cross indices, as NBLDPC.cpp:236-263 builds VarLinkDc / ChkLinkDv (last match wins) */
	for (int n = 0; n < c->N; n++) {
		for (int d = 0; d < c->dv[n]; d++) {
			int e = c->voff[n] + d, m = c->v_chk[e];
			c->v_k[e] = -1;
			for (int k = 0; k < c->dc[m]; k++)
				if (chk_var[c->coff[m] + k] == n) c->v_k[e] = k;
			if (c->v_k[e] < 0) return -1;
		}
	}
	for (int m = 0; m < c->M; m++) {
		for (int k = 0; k < c->dc[m]; k++) {
			int ce = c->coff[m] + k, n = chk_var[ce];
			c->c_var[ce] = n;
			c->c_h[ce] = chk_h[ce];
			c->c_d[ce] = -1;
			for (int d = 0; d < c->dv[n]; d++)
				if (c->v_chk[c->voff[n] + d] == m) c->c_d[ce] = d;
			if (c->c_d[ce] < 0) return -1;
			c->c2e[ce] = c->voff[n] + c->c_d[ce];
		}
	}
	return 0;
}

static nblo_code *code_alloc(int N, int M, int q)
{
	nblo_code *c = (nblo_code *)calloc(1, sizeof *c);
	c->N = N; c->M = M; c->q = q;
	c->dv = (int *)calloc(N, sizeof(int));
	c->dc = (int *)calloc(M, sizeof(int));
	c->voff = (int *)calloc(N + 1, sizeof(int));
	c->coff = (int *)calloc(M + 1, sizeof(int));
	return c;
}

static void code_alloc_edges(nblo_code *c)
{
	int E = c->E;
	c->v_chk = (int *)calloc(E, sizeof(int)); c->v_h = (int *)calloc(E, sizeof(int));
	c->v_k = (int *)calloc(E, sizeof(int));   c->c_var = (int *)calloc(E, sizeof(int));
	c->c_h = (int *)calloc(E, sizeof(int));   c->c_d = (int *)calloc(E, sizeof(int));
	c->c2e = (int *)calloc(E, sizeof(int));
}

/* File layout (NBLDPC.cpp:147-205): "N M q" / "maxdv maxdc" / dv[N] / dc[M] / per var: (check 1-based, h)
 * pairs / per check: (var 1-based, h) pairs. */
nblo_code *nblo_code_load(const char *path)
{
	FILE *f = fopen(path, "r");
	int N, M, q, *chk_var = NULL, *chk_h = NULL;
	nblo_code *c;
	if (!f) return NULL;
	if (fscanf(f, "%d %d %d", &N, &M, &q) != 3) { fclose(f); return NULL; }
	c = code_alloc(N, M, q);
	if (fscanf(f, "%d %d", &c->maxdv, &c->maxdc) != 2) goto bad;
	for (int n = 0; n < N; n++) { if (fscanf(f, "%d", &c->dv[n]) != 1) goto bad; c->voff[n + 1] = c->voff[n] + c->dv[n]; }
	for (int m = 0; m < M; m++) { if (fscanf(f, "%d", &c->dc[m]) != 1) goto bad; c->coff[m + 1] = c->coff[m] + c->dc[m]; }
	c->E = c->voff[N];
	if (c->coff[M] != c->E) goto bad;
	code_alloc_edges(c);
	for (int e = 0; e < c->E; e++) {
		if (fscanf(f, "%d %d", &c->v_chk[e], &c->v_h[e]) != 2) goto bad;
		c->v_chk[e]--;
	}
	chk_var = (int *)calloc(c->E, sizeof(int));
	chk_h = (int *)calloc(c->E, sizeof(int));
	for (int e = 0; e < c->E; e++) {
		if (fscanf(f, "%d %d", &chk_var[e], &chk_h[e]) != 2) goto bad;
		chk_var[e]--;
	}
	fclose(f); f = NULL;
	if (code_finish(c, chk_var, chk_h)) goto bad;
	free(chk_var); free(chk_h);
	return c;
bad:
	if (f) fclose(f);
	free(chk_var); free(chk_h);
	nblo_code_free(c);
	return NULL;
}

/* Build from a var-major edge list; the check-side order is increasing variable index, which is how every
 * code file shipped with the reference lists its check rows. */
nblo_code *nblo_code_from_edges(int N, int M, int q, int E, const int *edge_var, const int *edge_chk, const int *edge_h)
{
	nblo_code *c = code_alloc(N, M, q);
	int *chk_var, *chk_h, *fill;
	for (int e = 0; e < E; e++) { c->dv[edge_var[e]]++; c->dc[edge_chk[e]]++; }
	for (int n = 0; n < N; n++) { c->voff[n + 1] = c->voff[n] + c->dv[n]; if (c->dv[n] > c->maxdv) c->maxdv = c->dv[n]; }
	for (int m = 0; m < M; m++) { c->coff[m + 1] = c->coff[m] + c->dc[m]; if (c->dc[m] > c->maxdc) c->maxdc = c->dc[m]; }
	c->E = E;
	code_alloc_edges(c);
	chk_var = (int *)calloc(E, sizeof(int)); chk_h = (int *)calloc(E, sizeof(int)); fill = (int *)calloc(M, sizeof(int));
	for (int e = 0; e < E; e++) {
		int m = edge_chk[e];
		c->v_chk[e] = m; c->v_h[e] = edge_h[e];
		chk_var[c->coff[m] + fill[m]] = edge_var[e];
		chk_h[c->coff[m] + fill[m]] = edge_h[e];
		fill[m]++;
	}
	if (code_finish(c, chk_var, chk_h)) { nblo_code_free(c); c = NULL; }
	free(chk_var); free(chk_h); free(fill);
	return c;
}

void nblo_code_free(nblo_code *c)
{
	if (!c) return;
	free(c->dv); free(c->dc); free(c->voff); free(c->coff);
	free(c->v_chk); free(c->v_h); free(c->v_k); free(c->c_var); free(c->c_h); free(c->c_d); free(c->c2e);
	free(c);
}

/* ------------------------------------------------------------------------------------------------ */
/* decoder object                                                                                    */
/* ------------------------------------------------------------------------------------------------ */

struct nblo_decoder {
	const nblo_code *code;
	const nblo_gf *gf;
	nblo_params prm;
	int q, w; /* w = q-1 */
	double *post, *v2c, *c2v; /* [N][w], [E][w] var-major, [E][w] check-major */
	double *c2v_vm;           /* c2v copied to var-major order on request */
	double *old;              /* [w] damping scratch */
	int *dec;
	/* EMS scratch */
	double *srt_val;  /* [maxdc][q] values by rank */
	int *srt_sym;     /* [maxdc][q] check-domain symbol h*a by rank */
	double *S;        /* [q] */
	double *dpA, *dpB;/* [(nc+1)][q] */
	/* T-EMS scratch */
	double *dU;       /* [maxdc][q] */
	int *beta;        /* [maxdc] */
	int *tmin;        /* [q][maxdc] */
	unsigned char *inconf; /* [q][maxdc] */
	double *dW;       /* [q] */
	int *eta;         /* [q][maxdc] */
	int *eta_cand;    /* [maxdc] */
	unsigned char *selected; /* [q] */
	double *lc;       /* [q] */
	unsigned char *upd; /* [q] */
	/* BP scratch */
	double *sig, *rho, *tmpv; /* [w] */
	double *fw, *bw;          /* canonical: [maxdc][w] shared partials */
};

#define MUL(d, a, b) ((d)->gf->mul[(a) * (d)->q + (b)])
#define INV(d, a) ((d)->gf->inv[(a)])

nblo_decoder *nblo_decoder_create(const nblo_code *code, const nblo_gf *gf, const nblo_params *prm)
{
	nblo_decoder *d = (nblo_decoder *)calloc(1, sizeof *d);
	int q = code->q, w = q - 1, mdc = code->maxdc, nc;
	if (gf->q != q) { free(d); return NULL; }
	d->code = code; d->gf = gf; d->prm = *prm; d->q = q; d->w = w;
	if (prm->method == NBLO_EMS && prm->ems_nm > q) { free(d); return NULL; } /* NBLDPC.cpp:282-286 */
	d->post = (double *)calloc((size_t)code->N * w, sizeof(double));
	d->v2c = (double *)calloc((size_t)code->E * w, sizeof(double));
	d->c2v = (double *)calloc((size_t)code->E * w, sizeof(double));
	d->c2v_vm = (double *)calloc((size_t)code->E * w, sizeof(double));
	d->old = (double *)calloc(w, sizeof(double));
	d->dec = (int *)calloc(code->N, sizeof(int));
	d->srt_val = (double *)calloc((size_t)mdc * q, sizeof(double));
	d->srt_sym = (int *)calloc((size_t)mdc * q, sizeof(int));
	d->S = (double *)calloc(q, sizeof(double));
	nc = prm->ems_nc > 1 ? prm->ems_nc : 1; /* (conf(q,1) runs the same DP with one deviation) */
	d->dpA = (double *)calloc((size_t)(nc + 1) * q, sizeof(double));
	d->dpB = (double *)calloc((size_t)(nc + 1) * q, sizeof(double));
	d->dU = (double *)calloc((size_t)mdc * q, sizeof(double));
	d->beta = (int *)calloc(mdc, sizeof(int));
	d->tmin = (int *)calloc((size_t)q * mdc, sizeof(int));
	d->inconf = (unsigned char *)calloc((size_t)q * mdc, 1);
	d->dW = (double *)calloc(q, sizeof(double));
	d->eta = (int *)calloc((size_t)q * mdc, sizeof(int));
	d->eta_cand = (int *)calloc(mdc, sizeof(int));
	d->selected = (unsigned char *)calloc(q, 1);
	d->lc = (double *)calloc(q, sizeof(double));
	d->upd = (unsigned char *)calloc(q, 1);
	d->sig = (double *)calloc(w, sizeof(double));
	d->rho = (double *)calloc(w, sizeof(double));
	d->tmpv = (double *)calloc(w, sizeof(double));
	d->fw = (double *)calloc((size_t)mdc * w, sizeof(double));
	d->bw = (double *)calloc((size_t)mdc * w, sizeof(double));
	return d;
}

void nblo_decoder_free(nblo_decoder *d)
{
	if (!d) return;
	free(d->post); free(d->v2c); free(d->c2v); free(d->c2v_vm); free(d->old); free(d->dec);
	free(d->srt_val); free(d->srt_sym); free(d->S); free(d->dpA); free(d->dpB);
	free(d->dU); free(d->beta); free(d->tmin); free(d->inconf); free(d->dW); free(d->eta); free(d->eta_cand);
	free(d->selected); free(d->lc); free(d->upd); free(d->sig); free(d->rho); free(d->tmpv); free(d->fw); free(d->bw);
	free(d);
}

const double *nblo_state_post(const nblo_decoder *d) { return d->post; }
const double *nblo_state_v2c(const nblo_decoder *d) { return d->v2c; }
const double *nblo_state_c2v(const nblo_decoder *d)
{
	const nblo_code *c = d->code;
	for (int ce = 0; ce < c->E; ce++)
		memcpy(d->c2v_vm + (size_t)c->c2e[ce] * d->w, d->c2v + (size_t)ce * d->w, sizeof(double) * d->w);
	return d->c2v_vm;
}

/* Hard decision, DecideLLRVector NBLDPC.cpp:1542-1562: strict '>' against a running max that starts at 0,
 * so the lowest index wins ties and a vector with no positive entry decides symbol 0. */
static int decide(const double *L, int w)
{
	double best = 0;
	int arg = 0;
	for (int a = 0; a < w; a++)
		if (L[a] > best) { best = L[a]; arg = a + 1; }
	return arg;
}

/* dead-zone applied to every check-to-variable value, NBLDPC.cpp:903-916 / 1113-1126 */
static double shape(double y, double factor, double offset)
{
	y = y / factor;
	if (y < -1 * offset) return y + offset;
	if (y > offset) return y - offset;
	return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* EMS check node (NBLDPC.cpp:859-917, 1715-1786)                                                    */
/* ------------------------------------------------------------------------------------------------ */

typedef struct { double v; int a; } ems_item;

/* order produced by SortLLRVector (NBLDPC.cpp:1715-1746): descending value; among equal values the element
 * inserted later (higher symbol index) ends up in front because the insertion test is '>='. */
static int ems_cmp(const void *pa, const void *pb)
{
	const ems_item *x = (const ems_item *)pa, *y = (const ems_item *)pb;
	if (x->v > y->v) return -1;
	if (x->v < y->v) return 1;
	return (x->a > y->a) ? -1 : (x->a < y->a);
}

static void ems_sort_edge(nblo_decoder *d, int slot, const double *v2c, int h)
{
	int q = d->q;
	ems_item *it = (ems_item *)alloca(sizeof(ems_item) * q);
	it[0].v = 0; it[0].a = 0; /* symbol 0 carries LLR 0 (NBLDPC.cpp:1718-1719) */
	for (int a = 1; a < q; a++) { it[a].v = v2c[a - 1]; it[a].a = a; }
	qsort(it, q, sizeof(ems_item), ems_cmp);
	for (int r = 0; r < q; r++) {
		d->srt_val[slot * q + r] = it[r].v;
		d->srt_sym[slot * q + r] = MUL(d, it[r].a, h); /* contribution h*a to the check sum (:1766) */
	}
}

typedef struct {
	nblo_decoder *d;
	int nm, nc, skip, dc;
	double sum; /* the reference's running sumNonLLR: added to and subtracted from in place */
	int sym, diff;
} ems_dfs;

/* ConstructConf NBLDPC.cpp:1748-1786, literal: `sum` is one running variable, so fl(fl(s+x)-x) residue
 * survives into the following leaves exactly as in the reference. */
static void ems_dfs_literal(ems_dfs *st, int j)
{
	nblo_decoder *d = st->d;
	int q = d->q;
	if (j >= st->dc) {
		if (st->sum > d->S[st->sym]) d->S[st->sym] = st->sum;
		return;
	}
	if (j == st->skip) { ems_dfs_literal(st, j + 1); return; }
	for (int k = 0; k < st->nm; k++) {
		double x = d->srt_val[j * q + k];
		int s = d->srt_sym[j * q + k], dev = (k != 0);
		st->sym ^= s;
		st->sum = st->sum + x;
		st->diff += dev;
		if (st->diff <= st->nc) {
			ems_dfs_literal(st, j + 1);
			st->sym ^= s; st->sum = st->sum - x; st->diff -= dev;
		} else {
			st->sym ^= s; st->sum = st->sum - x; st->diff -= dev;
			break;
		}
	}
}

/* Same enumeration, but every leaf value is the fresh left-to-right sum ((x1+x2)+x3)...: no residue. */
static void ems_dfs_value(nblo_decoder *d, int nm, int nc, int skip, int dc, int j, double sum, int sym, int diff)
{
	int q = d->q;
	if (j >= dc) {
		if (sum > d->S[sym]) d->S[sym] = sum;
		return;
	}
	if (j == skip) { ems_dfs_value(d, nm, nc, skip, dc, j + 1, sum, sym, diff); return; }
	for (int k = 0; k < nm; k++) {
		int nd = diff + (k != 0);
		if (nd > nc) break;
		ems_dfs_value(d, nm, nc, skip, dc, j + 1, sum + d->srt_val[j * q + k], sym ^ d->srt_sym[j * q + k], nd);
	}
}

/* Residue-free value by dynamic programming over the other edges in index order.  Exact w.r.t.
 * ems_dfs_value because x -> fl(x + c) is monotone, so max and the rounded add commute. */
static void ems_conf_dp(nblo_decoder *d, int nm, int nc, int skip, int dc)
{
	int q = d->q, used = 0;
	double *A = d->dpA, *B = d->dpB, *T;
	for (int i = 0; i < (nc + 1) * q; i++) A[i] = -INFINITY;
	A[0] = 0.0;
	for (int j = 0; j < dc; j++) {
		const double *val = d->srt_val + j * q;
		const int *sym = d->srt_sym + j * q;
		if (j == skip) continue;
		for (int i = 0; i < (nc + 1) * q; i++) B[i] = -INFINITY;
		for (int dv = 0; dv <= used && dv <= nc; dv++) {
			for (int s = 0; s < q; s++) {
				double base = A[dv * q + s], t;
				if (base == -INFINITY) continue;
				t = base + val[0];
				if (t > B[dv * q + (s ^ sym[0])]) B[dv * q + (s ^ sym[0])] = t;
				if (dv + 1 > nc) continue;
				for (int k = 1; k < nm; k++) {
					t = base + val[k];
					if (t > B[(dv + 1) * q + (s ^ sym[k])]) B[(dv + 1) * q + (s ^ sym[k])] = t;
				}
			}
		}
		used++;
		T = A; A = B; B = T;
	}
	for (int dv = 0; dv <= nc; dv++)
		for (int s = 0; s < q; s++)
			if (A[dv * q + s] > d->S[s]) d->S[s] = A[dv * q + s];
}

static void ems_check_core(nblo_decoder *d, int dc, const int *h, const double *const *vin, double *const *cout)
{
	int q = d->q, mode = d->prm.mode;
	for (int k = 0; k < dc; k++) ems_sort_edge(d, k, vin[k], h[k]);
	for (int x = 0; x < dc; x++) {
		for (int s = 0; s < q; s++) d->S[s] = -DBL_MAX; /* :885-888 */
		if (mode == NBLO_LITERAL) {
			ems_dfs st = { d, q, 1, x, dc, 0.0, 0, 0 };
			ems_dfs_literal(&st, 0);                 /* conf(q,1)   :894 */
			st.nm = d->prm.ems_nm; st.nc = d->prm.ems_nc; st.sum = 0.0; st.sym = 0; st.diff = 0;
			ems_dfs_literal(&st, 0);                 /* conf(nm,nc) :897 */
		} else if (mode == NBLO_CANONICAL) {
			ems_conf_dp(d, q, 1, x, dc);
			ems_conf_dp(d, d->prm.ems_nm, d->prm.ems_nc, x, dc);
		} else {
			ems_dfs_value(d, q, 1, x, dc, 0, 0.0, 0, 0);
			ems_dfs_value(d, d->prm.ems_nm, d->prm.ems_nc, x, dc, 0, 0.0, 0, 0);
		}
		for (int a = 1; a < q; a++) {                /* :899-916 */
			int v = MUL(d, a, h[x]);
			cout[x][a - 1] = shape(d->S[v] - d->S[0], d->prm.ems_factor, d->prm.ems_offset);
		}
	}
}

/* ------------------------------------------------------------------------------------------------ */
/* T-EMS check node (NBLDPC.cpp:1055-1130, 1789-1944)                                                */
/* ------------------------------------------------------------------------------------------------ */

typedef struct {
	nblo_decoder *d;
	int nc, dc, literal;
	double sum;
	int sym, diff;
} tems_dfs;

/* TEMS_ConstructConf NBLDPC.cpp:1892-1944.  Columns in order, per column deviation symbols ascending; a
 * non-zero deviation symbol may be used by one column only; strict '<' keeps the first minimum met. */
static void tems_enumerate(tems_dfs *st, int col, double vsum)
{
	nblo_decoder *d = st->d;
	int q = d->q, mdc = d->code->maxdc;
	if (col >= st->dc) {
		double leaf = st->literal ? st->sum : vsum;
		if (leaf < d->dW[st->sym]) {
			d->dW[st->sym] = leaf;
			for (int c = 0; c < st->dc; c++) d->eta[st->sym * mdc + c] = d->eta_cand[c];
		}
		return;
	}
	for (int s = 0; s < q; s++) {
		int dev = (s != 0);
		double u;
		if (!d->inconf[s * mdc + col] || d->selected[s]) continue;
		st->diff += dev;
		if (st->diff > st->nc) { st->diff -= dev; break; }
		u = d->dU[col * q + s];
		d->selected[s] = (unsigned char)dev;
		st->sym ^= s;
		st->sum += u;
		d->eta_cand[col] = s;
		tems_enumerate(st, col + 1, vsum + u);
		st->sym ^= s;
		st->sum -= u;
		d->selected[s] = 0;
		st->diff -= dev;
	}
}

static void tems_check_core(nblo_decoder *d, int dc, const int *h, const double *const *vin, double *const *cout)
{
	int q = d->q, w = d->w, mdc = d->code->maxdc, syn = 0, nr = d->prm.tems_nr;
	tems_dfs st;
	/* most reliable symbol per edge in the check domain, TEMS_Get_Beta :1789-1812 */
	for (int k = 0; k < dc; k++) {
		double best = 0;
		int arg = 0;
		for (int a = 1; a < q; a++)
			if (vin[k][a - 1] > best) { best = vin[k][a - 1]; arg = MUL(d, a, h[k]); }
		d->beta[k] = arg;
		syn ^= arg;
	}
	/* delta domain, TEMS_Get_deltaU :1814-1834 */
	for (int k = 0; k < dc; k++) {
		int hi = INV(d, h[k]), bp = MUL(d, hi, d->beta[k]);
		double mx = bp ? vin[k][bp - 1] : 0;
		d->dU[k * q + d->beta[k]] = mx - 0;
		for (int x = 1; x < q; x++)
			d->dU[k * q + (x ^ d->beta[k])] = mx - vin[k][MUL(d, hi, x) - 1];
	}
	/* per deviation symbol: columns ordered by deltaU ascending (stable), TEMS_Get_Min :1836-1890 */
	for (int s = 0; s < q; s++) {
		int *ord = d->tmin + s * mdc;
		for (int k = 0; k < dc; k++) ord[k] = k;
		for (int k = 1; k < dc; k++)
			for (int j = k; j >= 1; j--) {
				if (d->dU[ord[j] * q + s] < d->dU[ord[j - 1] * q + s]) { int t = ord[j]; ord[j] = ord[j - 1]; ord[j - 1] = t; }
				else break;
			}
		for (int k = 0; k < dc; k++) d->inconf[s * mdc + k] = (s == 0);
		if (s) for (int i = 0; i < nr && i < dc; i++) d->inconf[s * mdc + ord[i]] = 1;
	}
	for (int s = 0; s < q; s++) { d->dW[s] = DBL_MAX; d->selected[s] = 0; }
	st.d = d; st.nc = d->prm.tems_nc; st.dc = dc; st.literal = (d->prm.mode == NBLO_LITERAL);
	st.sum = 0.0; st.sym = 0; st.diff = 0;
	tems_enumerate(&st, 0, 0.0);
	/* extrinsic output per edge :1075-1129 */
	for (int k = 0; k < dc; k++) {
		int hi = INV(d, h[k]), bsyn = syn ^ d->beta[k];
		double L0;
		for (int s = 0; s < q; s++) { d->lc[s] = DBL_MAX; d->upd[s] = 0; }
		for (int e = 0; e < q; e++) {
			int dev = d->eta[e * mdc + k], tgt = e ^ dev;
			double cand = d->dW[e] - d->dU[k * q + dev];
			if (d->lc[tgt] > cand) { d->lc[tgt] = cand; d->upd[tgt] = 1; }
		}
		for (int s = 0; s < q; s++)
			if (!d->upd[s]) {
				const int *ord = d->tmin + s * mdc;
				d->lc[s] = (k == ord[0]) ? d->dU[ord[1] * q + s] : d->dU[ord[0] * q + s];
			}
		L0 = -1.0 * d->lc[bsyn];
		for (int e = 0; e < q; e++) {
			int a;
			if (e == bsyn) continue;
			a = MUL(d, hi, e ^ bsyn);
			cout[k][a - 1] = shape(-1.0 * d->lc[e] - L0, d->prm.tems_factor, d->prm.tems_offset);
		}
	}
	(void)w;
}

/* ------------------------------------------------------------------------------------------------ */
/* BP / log-QSPA check node (NBLDPC.cpp:747-767, 1565-1712)                                          */
/* ------------------------------------------------------------------------------------------------ */

/* out[alpha] = src[c*alpha]: the A1==0 / A2==0 branches of LLR_BoxPlus (:1623-1642) with c = inverse of
 * the non-zero coefficient. */
static void bp_permute(nblo_decoder *d, double *out, const double *src, int c)
{
	for (int al = 1; al < d->q; al++) d->tmpv[al - 1] = src[MUL(d, c, al) - 1];
	memcpy(out, d->tmpv, sizeof(double) * d->w);
}

/* Full branch of LLR_BoxPlus (:1645-1709), literal: 80-bit accumulators under g++/x86-64, the x = 0..q-1
 * accumulation order, log(1+exp(-|d|)) form.  The seed of sum1 is evaluated in double (its operands are
 * doubles), the running updates in long double -- that is what overload resolution gives the reference. */
static void bp_boxplus_ld(nblo_decoder *d, double *out, const double *L1, const double *L2, int A1, int A2)
{
	int q = d->q, iA1 = INV(d, A1), iA2 = INV(d, A2);
	long double sum1, sum2 = 0;
	for (int x = 0; x < q; x++) {
		int v1 = x, v2 = MUL(d, iA2, MUL(d, A1, x));
		if (v1 != 0 && v2 != 0) {
			double t = L1[v1 - 1] + L2[v2 - 1];
			if (sum2 > t) sum2 = sum2 + logl(1 + expl(-1 * (sum2 - t)));
			else sum2 = t + logl(1 + expl(-1 * (t - sum2)));
		}
	}
	for (int al = 1; al < q; al++) {
		int v1 = MUL(d, iA1, al), v2 = MUL(d, iA2, al);
		double a = L1[v1 - 1], b = L2[v2 - 1];
		if (a > b) sum1 = a + log(1 + exp(-1 * (a - b)));
		else sum1 = b + log(1 + exp(-1 * (b - a)));
		for (int x = 0; x < q; x++) {
			v1 = x;
			v2 = MUL(d, iA2, al ^ MUL(d, x, A1));
			if (v1 != 0 && v2 != 0) {
				double t = L1[v1 - 1] + L2[v2 - 1];
				if (sum1 > t) sum1 = sum1 + logl(1 + expl(-1 * (sum1 - t)));
				else sum1 = t + logl(1 + expl(-1 * (t - sum1)));
			}
		}
		d->tmpv[al - 1] = (double)(sum1 - sum2);
	}
	memcpy(out, d->tmpv, sizeof(double) * d->w);
}

/* Canonical: the same sums in the same order with 64-bit accumulators (what a GPU can hold). */
static void bp_boxplus_f64(nblo_decoder *d, double *out, const double *L1, const double *L2, int A1, int A2)
{
	int q = d->q, iA1 = INV(d, A1), iA2 = INV(d, A2);
	double sum1, sum2 = 0;
	for (int x = 1; x < q; x++) {
		int v2 = MUL(d, iA2, MUL(d, A1, x));
		double t = L1[x - 1] + L2[v2 - 1];
		if (sum2 > t) sum2 = sum2 + log(1 + exp(-1 * (sum2 - t)));
		else sum2 = t + log(1 + exp(-1 * (t - sum2)));
	}
	for (int al = 1; al < q; al++) {
		int v1 = MUL(d, iA1, al), v2 = MUL(d, iA2, al);
		double a = L1[v1 - 1], b = L2[v2 - 1];
		if (a > b) sum1 = a + log(1 + exp(-1 * (a - b)));
		else sum1 = b + log(1 + exp(-1 * (b - a)));
		for (int x = 1; x < q; x++) {
			v2 = MUL(d, iA2, al ^ MUL(d, x, A1));
			if (v2 != 0) {
				double t = L1[x - 1] + L2[v2 - 1];
				if (sum1 > t) sum1 = sum1 + log(1 + exp(-1 * (sum1 - t)));
				else sum1 = t + log(1 + exp(-1 * (t - sum1)));
			}
		}
		d->tmpv[al - 1] = sum1 - sum2;
	}
	memcpy(out, d->tmpv, sizeof(double) * d->w);
}

static void bp_boxplus(nblo_decoder *d, double *out, const double *L1, const double *L2, int A1, int A2)
{
	if (d->prm.mode == NBLO_LITERAL) bp_boxplus_ld(d, out, L1, L2, A1, A2);
	else bp_boxplus_f64(d, out, L1, L2, A1, A2);
}

/* Per output edge x: sigma folds edges 0..x-1 upward (L_Back :1565-1591), rho folds edges dc-1..x+1
 * downward (L_Forward :1593-1619); the reference re-derives both for every x (:751-754). */
static void bp_check_core(nblo_decoder *d, int dc, const int *h, const double *const *vin, double *const *cout)
{
	for (int x = 0; x < dc; x++) {
		int hi = INV(d, h[x]);
		for (int l = 0; l < x; l++) {
			if (l == 0) bp_permute(d, d->sig, vin[0], INV(d, h[0]));
			else bp_boxplus(d, d->sig, d->sig, vin[l], 1, h[l]);
		}
		for (int l = dc - 1; l > x; l--) {
			if (l == dc - 1) bp_permute(d, d->rho, vin[l], INV(d, h[l]));
			else bp_boxplus(d, d->rho, d->rho, vin[l], 1, h[l]);
		}
		if (x == 0) bp_permute(d, cout[x], d->rho, INV(d, hi));          /* A1 = 0 (:757-760) */
		else if (x == dc - 1) bp_permute(d, cout[x], d->sig, INV(d, hi)); /* A2 = 0 (:761-764) */
		else bp_boxplus(d, cout[x], d->sig, d->rho, hi, hi);
	}
}

/* ------------------------------------------------------------------------------------------------ */
/* iteration loop (Decoding_BP :643-776, Decoding_EMS :778-927, Decoding_TEMS :929-1143)             */
/* ------------------------------------------------------------------------------------------------ */

static void check_update(nblo_decoder *d, int m, const double *const *vin, double *const *cout)
{
	const nblo_code *c = d->code;
	const int *h = c->c_h + c->coff[m];
	int dc = c->dc[m];
	switch (d->prm.method) {
	case NBLO_EMS: ems_check_core(d, dc, h, vin, cout); break;
	case NBLO_TEMS: tems_check_core(d, dc, h, vin, cout); break;
	default: bp_check_core(d, dc, h, vin, cout); break;
	}
}

static void check_api(nblo_decoder *d, int m, const double *v2c_in, double *c2v_out)
{
	int dc = d->code->dc[m];
	const double **vin = (const double **)alloca(sizeof(double *) * dc);
	double **cout = (double **)alloca(sizeof(double *) * dc);
	for (int k = 0; k < dc; k++) { vin[k] = v2c_in + (size_t)k * d->w; cout[k] = c2v_out + (size_t)k * d->w; }
	check_update(d, m, vin, cout);
}

void nblo_check_ems(nblo_decoder *d, int m, const double *v, double *c) { d->prm.method = NBLO_EMS; check_api(d, m, v, c); }
void nblo_check_tems(nblo_decoder *d, int m, const double *v, double *c) { d->prm.method = NBLO_TEMS; check_api(d, m, v, c); }
void nblo_check_bp(nblo_decoder *d, int m, const double *v, double *c) { d->prm.method = NBLO_BP; check_api(d, m, v, c); }

int nblo_decode(nblo_decoder *d, const double *L_ch, int *out, int *iters)
{
	const nblo_code *c = d->code;
	int w = d->w, N = c->N, M = c->M, frozen = 0, it = 0, method = d->prm.method;
	double keep_old = (method == NBLO_BP) ? 0.5 : 0.25, keep_new = (method == NBLO_BP) ? 0.5 : 0.75;
	const double **vin = (const double **)alloca(sizeof(double *) * c->maxdc);
	double **cout = (double **)alloca(sizeof(double *) * c->maxdc);

	for (int n = 0; n < N; n++)
		for (int k = 0; k < c->dv[n]; k++)
			memcpy(d->v2c + (size_t)(c->voff[n] + k) * w, L_ch + (size_t)n * w, sizeof(double) * w);
	memset(d->c2v, 0, sizeof(double) * (size_t)c->E * w);
	if (iters) *iters = d->prm.max_iter;

	while (it++ < d->prm.max_iter) {
		int ok = 1;
		/* a-posteriori sum, in the order L_ch, then the variable's edges 0..dv-1 (:676-689) */
		for (int n = 0; n < N; n++) {
			double *P = d->post + (size_t)n * w;
			memcpy(P, L_ch + (size_t)n * w, sizeof(double) * w);
			for (int k = 0; k < c->dv[n]; k++) {
				int e = c->voff[n] + k;
				const double *C = d->c2v + (size_t)(c->coff[c->v_chk[e]] + c->v_k[e]) * w;
				for (int a = 0; a < w; a++) P[a] = P[a] + C[a];
			}
			d->dec[n] = decide(P, w);
		}
		if (!frozen) memcpy(out, d->dec, sizeof(int) * N);
		/* syndrome (:693-709) */
		for (int m = 0; m < M && ok; m++) {
			int s = 0;
			for (int k = 0; k < c->dc[m]; k++) {
				int ce = c->coff[m] + k;
				s ^= MUL(d, c->c_h[ce], d->dec[c->c_var[ce]]);
			}
			if (s) ok = 0;
		}
		if (ok && !frozen) {
			frozen = 1;
			if (iters) *iters = it;
			if (!d->prm.fixed_iters) return 1;
		}
		/* variable to check (:718-744 BP, :848-857 EMS, :1029-1052 T-EMS) */
		for (int n = 0; n < N; n++) {
			const double *P = d->post + (size_t)n * w;
			for (int k = 0; k < c->dv[n]; k++) {
				int e = c->voff[n] + k;
				double *V = d->v2c + (size_t)e * w;
				const double *C = d->c2v + (size_t)(c->coff[c->v_chk[e]] + c->v_k[e]) * w;
				if (method == NBLO_EMS) {
					for (int a = 0; a < w; a++) V[a] = P[a] - C[a];
				} else {
					int before = decide(V, w), after;
					memcpy(d->old, V, sizeof(double) * w);
					for (int a = 0; a < w; a++) V[a] = P[a] - C[a];
					after = decide(V, w);
					if (after != before)
						for (int a = 0; a < w; a++) V[a] = keep_old * d->old[a] + keep_new * V[a];
				}
			}
		}
		/* check to variable */
		for (int m = 0; m < M; m++) {
			for (int k = 0; k < c->dc[m]; k++) {
				int ce = c->coff[m] + k;
				vin[k] = d->v2c + (size_t)c->c2e[ce] * w;
				cout[k] = d->c2v + (size_t)ce * w;
			}
			check_update(d, m, vin, cout);
		}
	}
	return frozen;
}

/* ------------------------------------------------------------------------------------------------ */
/* batch helper (one decoder object per thread, like one CNBLDPC per lane in the reference)          */
/* ------------------------------------------------------------------------------------------------ */

typedef struct {
	nblo_decoder *d;
	const double *L_ch;
	int b0, b1;
	int *out;
	unsigned char *conv;
	int *iters;
} batch_job;

static void *batch_worker(void *arg)
{
	batch_job *j = (batch_job *)arg;
	size_t per = (size_t)j->d->code->N * j->d->w;
	for (int b = j->b0; b < j->b1; b++) {
		int it = 0, r = nblo_decode(j->d, j->L_ch + per * b, j->out + (size_t)b * j->d->code->N, &it);
		if (j->conv) j->conv[b] = (unsigned char)r;
		if (j->iters) j->iters[b] = it;
	}
	return NULL;
}

int nblo_decode_batch(nblo_decoder *const *decs, int nthreads, const double *L_ch, int B, int *out,
                      unsigned char *converged, int *iters)
{
	pthread_t *th = (pthread_t *)alloca(sizeof(pthread_t) * nthreads);
	batch_job *jobs = (batch_job *)alloca(sizeof(batch_job) * nthreads);
	for (int t = 0; t < nthreads; t++) {
		jobs[t].d = decs[t]; jobs[t].L_ch = L_ch; jobs[t].out = out; jobs[t].conv = converged; jobs[t].iters = iters;
		jobs[t].b0 = (int)((long long)B * t / nthreads);
		jobs[t].b1 = (int)((long long)B * (t + 1) / nthreads);
	}
	if (nthreads == 1) { batch_worker(&jobs[0]); return 0; }
	for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
	for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
	return 0;
}
