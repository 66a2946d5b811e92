/*
 * qcqp_oracle.c -- TEST INFRASTRUCTURE ONLY (see qcqp_oracle.h).
 *
 * CPU restatement of the cvxgrp/qcqp hot path: QuadraticFunction algebra,
 * one-variable / one-constraint sub-solvers, coordinate descent and consensus
 * ADMM.  Written from the algorithm's behaviour, with the reference quirks of
 * SURVEY.md appendix A reproduced on purpose (they are part of "results
 * identical to the reference").  Build with -ffp-contract=off so that the
 * arithmetic is the unfused IEEE double arithmetic NumPy performs.
 */
#include "qcqp_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ problem */

typedef struct {
    int64_t nnz;
    int64_t *ptr, *idx;
    double *val, *q;
    double r;
    int relop;
} quad_t;

struct orc_prob {
    int64_t n, m;
    quad_t *f; /* m+1 entries, f[0] objective */
};

orc_prob *orc_prob_new(int64_t n, int64_t m) {
    orc_prob *p = (orc_prob *)calloc(1, sizeof(*p));
    p->n = n;
    p->m = m;
    p->f = (quad_t *)calloc((size_t)(m + 1), sizeof(quad_t));
    return p;
}

static void quad_clear(quad_t *f) {
    free(f->ptr); free(f->idx); free(f->val); free(f->q);
    memset(f, 0, sizeof(*f));
}

void orc_prob_free(orc_prob *p) {
    if (!p) return;
    for (int64_t k = 0; k <= p->m; k++) quad_clear(&p->f[k]);
    free(p->f);
    free(p);
}

int orc_prob_set(orc_prob *p, int64_t k, int64_t nnz, const int64_t *ptr,
                 const int64_t *idx, const double *val, const double *q,
                 double r, int relop) {
    if (k < 0 || k > p->m) return -1;
    quad_t *f = &p->f[k];
    quad_clear(f);
    int64_t n = p->n;
    f->nnz = nnz;
    f->ptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    f->idx = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz > 0 ? nnz : 1));
    f->val = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
    f->q = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(f->ptr, ptr, sizeof(int64_t) * (size_t)(n + 1));
    if (nnz > 0) {
        memcpy(f->idx, idx, sizeof(int64_t) * (size_t)nnz);
        memcpy(f->val, val, sizeof(double) * (size_t)nnz);
    }
    memcpy(f->q, q, sizeof(double) * (size_t)n);
    f->r = r;
    f->relop = relop;
    return 0;
}

int64_t orc_prob_n(const orc_prob *p) { return p->n; }
int64_t orc_prob_m(const orc_prob *p) { return p->m; }

/* ---------------------------------------------------------------------- RNG */

struct orc_rng {
    int mode;
    /* MT19937 (numpy legacy RandomState bit stream) */
    uint32_t key[624];
    int pos;
    /* keyed Philox context */
    uint64_t seed, restart;
    uint32_t coord, sweep_tag, iter;
    uint64_t draws;
};

static void mt_seed(orc_rng *g, uint32_t seed) {
    /* numpy _legacy_seeding(int) == Knuth init_genrand */
    g->key[0] = seed;
    for (int i = 1; i < 624; i++)
        g->key[i] = 1812433253u * (g->key[i - 1] ^ (g->key[i - 1] >> 30)) + (uint32_t)i;
    g->pos = 624;
}

static uint32_t mt_next32(orc_rng *g) {
    if (g->pos >= 624) {
        uint32_t *mt = g->key;
        int kk;
        uint32_t y;
        for (kk = 0; kk < 624 - 397; kk++) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; kk < 623; kk++) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        g->pos = 0;
    }
    uint32_t y = g->key[g->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    g->draws++;
    return y;
}

static double mt_double(orc_rng *g) {
    uint32_t a = mt_next32(g) >> 5, b = mt_next32(g) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

void orc_philox4x32(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
    uint32_t k0 = key_in[0], k1 = key_in[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static double u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

static void keyed_draw(const orc_rng *g, uint32_t out[4]) {
    uint32_t ctr[4] = {g->coord, g->sweep_tag, g->iter, (uint32_t)g->restart};
    uint32_t key[2] = {(uint32_t)g->seed, (uint32_t)(g->seed >> 32)};
    orc_philox4x32(ctr, key, out);
}

/* Box-Muller normal from the keyed stream: element `elem` of restart/sample `restart`.
 * stream tag 0xA5A5 in ctr[2] keeps it disjoint from the CD draws (iter < 2^16). */
double orc_keyed_normal(uint64_t seed, uint64_t restart, uint64_t elem) {
    uint32_t ctr[4] = {(uint32_t)(elem >> 1), (uint32_t)(elem >> 33), 0xA5A50000u,
                       (uint32_t)restart};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(restart >> 32)};
    uint32_t o[4];
    orc_philox4x32(ctr, key, o);
    double u1 = (((double)(o[0] >> 5) * 67108864.0 + (double)(o[1] >> 6)) + 0.5) /
                9007199254740992.0;
    double u2 = u53(o[2], o[3]);
    double rad = sqrt(-2.0 * log(u1));
    double ang = 6.283185307179586476925286766559 * u2;
    return (elem & 1) ? rad * sin(ang) : rad * cos(ang);
}

/* Values f_k(x_s) = x_s' P_k x_s + q_k' x_s + r_k of a SYNTHETIC function that the engine generates on the device
 * (qcqpmi_set_quad_generated; BASELINE.json configs[4]: 1025 dense 4096 x 4096 matrices that no host can hold; SURVEY.md
 * section 8(d) cfg5).  The law is restated here from include/qcqp_mi.h -- P_k[i][j] = scale w_ij N(seed, 2^48 + k,
 * min(i,j) n + max(i,j)) + diag_add [i == j], w_ii = 1, w_ij = 1/sqrt(2); q_k[j] = qscale N(seed, 2^49 + k, j) -- and
 * evaluated entry by entry without materialising the matrix (QuadraticFunction.eval, utilities.py:49-50), so that the
 * full-size tests can check single functions of the 137.6 GB problem.  X: S x n sample-major. */
void orc_generated_eval(uint64_t seed, int64_t k, int64_t n, double scale, double qscale, double diag_add, double r,
                        const double *X, int64_t S, double *out) {
    const double w = 0.70710678118654752440;
    for (int64_t s = 0; s < S; s++) out[s] = 0.0;
    for (int64_t i = 0; i < n; i++) {
        for (int64_t j = i; j < n; j++) {
            double v = (scale != 0.0) ? scale * orc_keyed_normal(seed, ((uint64_t)1 << 48) + (uint64_t)k, (uint64_t)(i * n + j)) : 0.0;
            if (i == j) {
                v += diag_add;
                if (v != 0.0)
                    for (int64_t s = 0; s < S; s++) out[s] += v * X[s * n + i] * X[s * n + i];
            } else if (v != 0.0) {
                v *= w;
                for (int64_t s = 0; s < S; s++) out[s] += 2.0 * v * X[s * n + i] * X[s * n + j];
            }
        }
        if (qscale != 0.0) {
            const double q = qscale * orc_keyed_normal(seed, ((uint64_t)2 << 48) + (uint64_t)k, (uint64_t)i);
            for (int64_t s = 0; s < S; s++) out[s] += q * X[s * n + i];
        }
    }
    for (int64_t s = 0; s < S; s++) out[s] += r;
}

orc_rng *orc_rng_new(int mode, uint64_t seed) {
    orc_rng *g = (orc_rng *)calloc(1, sizeof(*g));
    g->mode = mode;
    g->seed = seed;
    mt_seed(g, (uint32_t)seed);
    return g;
}
void orc_rng_free(orc_rng *g) { free(g); }
void orc_rng_mt_set(orc_rng *g, const uint32_t *key, int pos) {
    memcpy(g->key, key, sizeof(g->key));
    g->pos = pos;
}
void orc_rng_mt_get(const orc_rng *g, uint32_t *key, int *pos) {
    memcpy(key, g->key, sizeof(g->key));
    *pos = g->pos;
}
void orc_rng_set_restart(orc_rng *g, uint64_t restart_index) { g->restart = restart_index; }
uint64_t orc_rng_draws(const orc_rng *g) { return g->draws; }

static void rng_ctx(orc_rng *g, uint32_t coord, uint32_t sweep_tag, uint32_t iter) {
    if (!g) return;
    g->coord = coord; g->sweep_tag = sweep_tag; g->iter = iter;
}
/* the counter words of the NEXT keyed draw, set from outside: lets tests/test_oracle_golden.py pin orc_rng_uniform / orc_rng_choice
 * in keyed mode against the committed Philox table (tests/golden/g11_philox_keyed.npz) for arbitrary (coordinate, sweep tag, iteration) */
void orc_rng_set_ctx(orc_rng *g, uint32_t coord, uint32_t sweep_tag, uint32_t iter) { rng_ctx(g, coord, sweep_tag, iter); }

/* np.random.uniform(lo, hi) (legacy: lo + (hi-lo)*random_double) */
double orc_rng_uniform(orc_rng *g, double lo, double hi) {
    double u;
    if (g->mode == ORC_RNG_MT) {
        u = mt_double(g);
    } else {
        uint32_t o[4];
        keyed_draw(g, o);
        u = u53(o[0], o[1]);
    }
    return lo + (hi - lo) * u;
}

/* np.random.choice(k) == legacy randint(0,k): masked rejection on 32-bit draws,
 * and NO draw at all when k == 1. */
int64_t orc_rng_choice(orc_rng *g, int64_t k) {
    if (k <= 1) return 0;
    if (g->mode == ORC_RNG_MT) {
        uint32_t rng = (uint32_t)(k - 1), mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4;
        mask |= mask >> 8; mask |= mask >> 16;
        uint32_t v;
        do { v = mt_next32(g) & mask; } while (v > rng);
        return (int64_t)v;
    }
    uint32_t o[4];
    keyed_draw(g, o);
    return (int64_t)(((uint64_t)o[2] * (uint64_t)k) >> 32);
}

/* ------------------------------------------------------- quadratic algebra */

/* (P.dot(x) + q).dot(x) + r  (utilities.py:49-50); `skip` = coordinate treated as 0, or -1 */
static double quad_eval_skip(const quad_t *f, int64_t n, const double *x, int64_t skip) {
    double acc = 0.0;
    for (int64_t i = 0; i < n; i++) {
        double row = 0.0;
        for (int64_t jj = f->ptr[i]; jj < f->ptr[i + 1]; jj++) {
            int64_t j = f->idx[jj];
            if (j == skip) continue;
            row += f->val[jj] * x[j];
        }
        double xi = (i == skip) ? 0.0 : x[i];
        acc += (row + f->q[i]) * xi;
    }
    return acc + f->r;
}

double orc_eval(const orc_prob *p, int64_t k, const double *x) {
    return quad_eval_skip(&p->f[k], p->n, x, -1);
}

static double viol_of(double fv, int relop) {
    if (relop == ORC_RELOP_EQ) return fabs(fv);
    return fv > 0.0 ? fv : 0.0; /* max(0., f) */
}

double orc_violation(const orc_prob *p, int64_t k, const double *x) {
    return viol_of(orc_eval(p, k, x), p->f[k].relop);
}

double orc_max_violation(const orc_prob *p, const double *x) {
    double mv = -INFINITY;
    for (int64_t k = 1; k <= p->m; k++) {
        double v = orc_violation(p, k, x);
        if (v > mv) mv = v;
    }
    return mv;
}

int orc_better(const orc_prob *p, const double *x1, const double *x2, double tol) {
    /* int(max viol / tol) buckets first, strict objective second, ties -> x2 */
    long long v1 = (long long)(orc_max_violation(p, x1) / tol);
    long long v2 = (long long)(orc_max_violation(p, x2) / tol);
    double f1 = orc_eval(p, 0, x1), f2 = orc_eval(p, 0, x2);
    if (v1 < v2) return 1;
    if (v2 < v1) return 2;
    if (f1 < f2) return 1;
    return 2;
}

void orc_eval_batch(const orc_prob *p, const double *X, int64_t S, double *f0,
                    double *maxviol, double *F) {
    for (int64_t s = 0; s < S; s++) {
        const double *x = X + s * p->n;
        double mv = -INFINITY;
        for (int64_t k = 0; k <= p->m; k++) {
            double fv = orc_eval(p, k, x);
            if (F) F[k * S + s] = fv;
            if (k == 0) { if (f0) f0[s] = fv; }
            else {
                double v = viol_of(fv, p->f[k].relop);
                if (v > mv) mv = v;
            }
        }
        if (maxviol) maxviol[s] = mv;
    }
}

void orc_onevar_coeffs(const orc_prob *p, int64_t k, const double *x, int64_t c,
                       double out3[3]) {
    const quad_t *f = &p->f[k];
    double t2 = 0.0, row = 0.0;
    for (int64_t jj = f->ptr[c]; jj < f->ptr[c + 1]; jj++) {
        int64_t j = f->idx[jj];
        if (j == c) t2 += f->val[jj];
        else row += f->val[jj] * x[j];
    }
    out3[0] = t2;                          /* P[k,k] */
    out3[1] = 2.0 * row + f->q[c];         /* 2*P[k,:].dot(z) + q[k] */
    out3[2] = quad_eval_skip(f, p->n, x, c); /* (P.dot(z)+q).dot(z) + r */
}

/* ------------------------------------------------ get_feasible_intervals */

static int intervals_le(double p, double q, double rs /* r - s */, double s_minus_r,
                        double tol, double *out) {
    if (p > tol) {
        double D = q * q - 4.0 * p * rs;
        if (D >= 0.0) {
            double rD = sqrt(D);
            out[0] = (-q - rD) / (2.0 * p);
            out[1] = (-q + rD) / (2.0 * p);
            return 1;
        }
        return 0;
    } else if (p < -tol) {
        double D = q * q - 4.0 * p * rs;
        if (D >= 0.0) {
            double rD = sqrt(D);
            out[0] = -INFINITY;
            out[1] = (-q + rD) / (2.0 * p);
            out[2] = (-q - rD) / (2.0 * p);
            out[3] = INFINITY;
            return 2;
        }
        out[0] = -INFINITY; out[1] = INFINITY;
        return 1;
    } else {
        if (q > tol) { out[0] = -INFINITY; out[1] = s_minus_r / q; return 1; }
        if (q < -tol) { out[0] = s_minus_r / q; out[1] = INFINITY; return 1; }
        out[0] = -INFINITY; out[1] = INFINITY;
        return 1;
    }
}

int orc_feasible_intervals(double p, double q, double r, int relop, double s,
                           double tol, double *out) {
    if (relop == ORC_RELOP_EQ) {
        /* |p x^2 + q x + r| <= s: the two one-sided problems are solved with s folded
         * into r and the DEFAULT slack 0 / tol 1e-4 (utilities.py:200-208). */
        double a[4], b[4];
        double r1 = r - s, r2 = -r - s;
        int n1 = intervals_le(p, q, r1 - 0.0, 0.0 - r1, 1e-4, a);
        int n2 = intervals_le(-p, -q, r2 - 0.0, 0.0 - r2, 1e-4, b);
        int cnt = 0;
        for (int i = 0; i < n1; i++)
            for (int j = 0; j < n2; j++) {
                double lo = a[2 * i] > b[2 * j] ? a[2 * i] : b[2 * j];         /* max */
                double hi = a[2 * i + 1] < b[2 * j + 1] ? a[2 * i + 1] : b[2 * j + 1]; /* min */
                if (lo <= hi) { out[2 * cnt] = lo; out[2 * cnt + 1] = hi; cnt++; }
            }
        return cnt;
    }
    return intervals_le(p, q, r - s, s - r, tol, out);
}

/* ------------------------------------------------------------ onevar_qcqp */

typedef struct { double key; long long cnt; } event_t;

static int event_cmp(const void *a, const void *b) {
    double x = ((const event_t *)a)->key, y = ((const event_t *)b)->key;
    return (x > y) - (x < y);
}

/* OneVarQuadraticFunction.eval incl. the +-inf branches (utilities.py:115-120).
 * status -2 <=> the reference would raise NameError (P == 0 and q == 0 at +-inf). */
static double onevar_eval(double p, double q, double r, double x, int *status) {
    if (isinf(x)) {
        if (p != 0.0) return p * x * x;
        if (q != 0.0) return q * x;
        *status = -2;
        return NAN;
    }
    return x * (p * x + q) + r;
}

int orc_onevar_qcqp(double p0, double q0, double r0, const double *fs3,
                    const int *relops, int64_t mf, double s, orc_rng *g,
                    double *xout, double *C_out, int64_t Ccap, int64_t *nC_out) {
    int64_t cap = 4 * mf + 2;
    event_t *ev = (event_t *)malloc(sizeof(event_t) * (size_t)(2 * cap));
    int64_t ne = 0;
    ev[ne].key = -INFINITY; ev[ne++].cnt = +1;
    ev[ne].key = INFINITY; ev[ne++].cnt = -1;
    for (int64_t k = 0; k < mf; k++) {
        double iv[8];
        int c = orc_feasible_intervals(fs3[3 * k], fs3[3 * k + 1], fs3[3 * k + 2],
                                       relops[k], s, 1e-4, iv);
        for (int i = 0; i < c; i++) {
            ev[ne].key = iv[2 * i]; ev[ne++].cnt = +1;
            ev[ne].key = iv[2 * i + 1]; ev[ne++].cnt = -1;
        }
    }
    qsort(ev, (size_t)ne, sizeof(event_t), event_cmp);
    /* merge equal keys (dict semantics), drop zero net counts */
    int64_t nx = 0;
    for (int64_t i = 0; i < ne;) {
        int64_t j = i;
        long long c = 0;
        while (j < ne && ev[j].key == ev[i].key) { c += ev[j].cnt; j++; }
        if (c != 0) { ev[nx].key = ev[i].key; ev[nx].cnt = c; nx++; }
        i = j;
    }
    double *C = (double *)malloc(sizeof(double) * (size_t)(2 * (nx + 1)));
    int64_t nC = 0;
    long long tot = 0;
    for (int64_t i = 0; i < nx; i++) {
        tot += ev[i].cnt;
        if (tot == (long long)mf && ev[i].cnt == -1) {
            int64_t prev = (i > 0) ? i - 1 : nx - 1; /* python xs[i-1] */
            C[2 * nC] = ev[prev].key;
            C[2 * nC + 1] = ev[i].key;
            nC++;
        }
    }
    free(ev);
    if (nC_out) *nC_out = nC;
    if (C_out)
        for (int64_t i = 0; i < nC && i < Ccap; i++) { C_out[2 * i] = C[2 * i]; C_out[2 * i + 1] = C[2 * i + 1]; }
    if (nC == 0) { free(C); return 0; }

    int ret = 1;
    if (p0 == 0.0 && q0 == 0.0) {
        int64_t c = orc_rng_choice(g, nC);
        double lo = C[2 * c], hi = C[2 * c + 1];
        if (isinf(lo) || isinf(hi)) { free(C); return -1; } /* numpy: OverflowError */
        *xout = orc_rng_uniform(g, lo, hi);
        free(C);
        return 1;
    }
    double x0 = (p0 > 0.0) ? -q0 / (2.0 * p0) : NAN;
    double bestf = INFINITY;
    double *bestxs = (double *)malloc(sizeof(double) * (size_t)(2 * nC));
    int64_t nb = 0;
    int status = 0;
    for (int64_t i = 0; i < nC; i++) {
        double lo = C[2 * i], hi = C[2 * i + 1];
        if (lo <= x0 && x0 <= hi) { *xout = x0; free(C); free(bestxs); return 1; }
        double fl = onevar_eval(p0, q0, r0, lo, &status);
        double fr = onevar_eval(p0, q0, r0, hi, &status);
        if (status) { free(C); free(bestxs); return status; }
        if (bestf > fl) { nb = 0; bestxs[nb++] = lo; bestf = fl; }
        else if (bestf == fl) bestxs[nb++] = lo;
        if (bestf > fr) { nb = 0; bestxs[nb++] = hi; bestf = fr; }
        else if (bestf == fr) bestxs[nb++] = hi;
    }
    if (nb == 0) ret = 0;
    else *xout = bestxs[orc_rng_choice(g, nb)];
    free(C); free(bestxs);
    return ret;
}

/* ----------------------------------------------------- coordinate descent */

/* nfs = [f.get_onevar_func(x, i) for f in fs], filtered by `f.P != 0 or f.q != 0`
 * (qcqp.py:115-116, 164-166).  Returns the number kept. */
static int64_t gather_onevars(const orc_prob *p, const double *x, int64_t i,
                              double *fs3, int *relops) {
    int64_t mf = 0;
    for (int64_t k = 1; k <= p->m; k++) {
        double t[3];
        orc_onevar_coeffs(p, k, x, i, t);
        if (t[0] != 0.0 || t[1] != 0.0) {
            fs3[3 * mf] = t[0]; fs3[3 * mf + 1] = t[1]; fs3[3 * mf + 2] = t[2];
            relops[mf] = p->f[k].relop;
            mf++;
        }
    }
    return mf;
}

/* test hook: the value of x[i] after every coordinate visit of the coordinate-descent runs that follow (phase 1 and
 * phase 2 append to the same buffer; the state after any visit is x0 with the recorded values applied in order) */
static double *g_trace = NULL;
static int64_t g_trace_cap = 0, g_trace_len = 0;
void orc_cd_trace(double *buf, int64_t cap) { g_trace = buf; g_trace_cap = buf ? cap : 0; g_trace_len = 0; }
int64_t orc_cd_trace_len(void) { return g_trace_len; }
#define ORC_TRACE(v) do { if (g_trace && g_trace_len < g_trace_cap) g_trace[g_trace_len] = (v); if (g_trace) g_trace_len++; } while (0)
/* test instrumentation like the trace: stop orc_cd_phase1 / orc_cd_phase2 after this many coordinate visits (< 0: off) -- a
 * visit of a problem with 257 dense 1024 x 1024 functions costs 0.13 s (get_onevar_func forms P_k z for every function,
 * utilities.py:99-105), a sweep two minutes: the teacher-forced parity test at that size follows the first visits only */
static int64_t g_visit_limit = -1;
void orc_cd_visit_limit(int64_t visits) { g_visit_limit = visits; }

int orc_cd_phase1(const orc_prob *p, double *x, int64_t num_iters, double viol_tol,
                  double tol, orc_rng *g, int64_t *stats) {
    int64_t n = p->n;
    double *fs3 = (double *)malloc(sizeof(double) * 3 * (size_t)(p->m + 1));
    int *relops = (int *)malloc(sizeof(int) * (size_t)(p->m + 1));
    int64_t update_counter = 0, sweeps = 0, visits = 0, accepted = 0;
    double viol_last = INFINITY;
    int rc = 0;
    for (int64_t t = 0; t < num_iters && rc == 0; t++) {
        if (viol_last < viol_tol) break;
        sweeps++;
        int cut = 0;
        for (int64_t i = 0; i < n; i++) {
            if (g_visit_limit >= 0 && visits >= g_visit_limit) { cut = 1; break; }
            visits++;
            int64_t mf = gather_onevars(p, x, i, fs3, relops);
            if (mf == 0) { rc = -3; break; } /* python: max() of empty list -> ValueError */
            double viol = -INFINITY;
            for (int64_t k = 0; k < mf; k++) {
                int st = 0;
                double v = viol_of(onevar_eval(fs3[3 * k], fs3[3 * k + 1], fs3[3 * k + 2], x[i], &st),
                                   relops[k]);
                if (v > viol) viol = v;
            }
            double new_xi = x[i], new_viol = viol;
            double ss = -tol, es = viol - viol_tol;
            uint32_t it = 0;
            while (es - ss > tol) {
                double s = (ss + es) / 2.0;
                double xi;
                rng_ctx(g, (uint32_t)i, (uint32_t)t, it++);
                int got = orc_onevar_qcqp(0.0, 0.0, 0.0, fs3, relops, mf, s, g, &xi, NULL, 0, NULL);
                if (got < 0) { rc = got; break; }
                if (!got) ss = s;
                else { new_xi = xi; new_viol = s; es = s; }
            }
            if (rc) break;
            if (new_viol < viol) { x[i] = new_xi; update_counter = 0; accepted++; ORC_TRACE(x[i]); }
            else {
                ORC_TRACE(x[i]);
                update_counter++;
                if (update_counter == n) break; /* failed = True; outer loop goes on (qcqp.py:138-141) */
            }
        }
        if (rc || cut) break;
        viol_last = orc_max_violation(p, x);
    }
    if (stats) { stats[0] = sweeps; stats[1] = visits; stats[2] = accepted; }
    free(fs3); free(relops);
    return rc;
}

int orc_cd_phase2(const orc_prob *p, double *x, int64_t num_iters, double viol_tol,
                  double tol, orc_rng *g, int64_t *stats) {
    (void)viol_tol;
    int64_t n = p->n;
    double *fs3 = (double *)malloc(sizeof(double) * 3 * (size_t)(p->m + 1));
    int *relops = (int *)malloc(sizeof(int) * (size_t)(p->m + 1));
    double viol = orc_max_violation(p, x);
    int64_t update_counter = 0, sweeps = 0, visits = 0, accepted = 0;
    int converged = 0, rc = 0;
    for (int64_t t = 0; t < num_iters && !converged && rc == 0; t++) {
        sweeps++;
        for (int64_t i = 0; i < n; i++) {
            if (g_visit_limit >= 0 && visits >= g_visit_limit) { converged = 1; break; }
            visits++;
            double obj[3];
            orc_onevar_coeffs(p, 0, x, i, obj);
            int64_t mf = gather_onevars(p, x, i, fs3, relops);
            double new_xi;
            rng_ctx(g, (uint32_t)i, (uint32_t)t | 0x80000000u, 0);
            int got = orc_onevar_qcqp(obj[0], obj[1], obj[2], fs3, relops, mf, viol, g, &new_xi,
                                      NULL, 0, NULL);
            if (got < 0) { rc = got; break; }
            if (got && fabs(new_xi - x[i]) > tol) { x[i] = new_xi; update_counter = 0; accepted++; ORC_TRACE(x[i]); }
            else {
                ORC_TRACE(x[i]);
                update_counter++;
                if (update_counter == n) { converged = 1; break; }
            }
        }
    }
    if (stats) { stats[0] = sweeps; stats[1] = visits; stats[2] = accepted; }
    free(fs3); free(relops);
    return rc;
}

/* Test instrumentation: `count` coordinate visits i0, i0 + 1, ... of sweep t of phase 1 (qcqp.py:113-136) or phase 2
 * (qcqp.py:162-170; `viol2` = the slack phase 2 fixed at its start, qcqp.py:157) from the state x, with the bodies of the loops
 * above and the same keyed draws -- the oracle's own answer for ONE block of the engine's blocked kernels from an arbitrary state
 * (the block-level yardstick of the dense path's parity test: how far does the reference itself move when that state moves by
 * one ulp?).  Phase 1 does not update its stop bookkeeping here (no sweep ends inside a block). */
int orc_cd_visits(const orc_prob *p, double *x, int phase, int64_t t, int64_t i0, int64_t count, double viol2,
                  double viol_tol, double tol, orc_rng *g) {
    int64_t n = p->n;
    double *fs3 = (double *)malloc(sizeof(double) * 3 * (size_t)(p->m + 1));
    int *relops = (int *)malloc(sizeof(int) * (size_t)(p->m + 1));
    int rc = 0;
    for (int64_t i = i0; i < i0 + count && i < n && rc == 0; i++) {
        int64_t mf = gather_onevars(p, x, i, fs3, relops);
        if (phase == 1) {
            if (mf == 0) { rc = -3; break; }
            double viol = -INFINITY;
            for (int64_t k = 0; k < mf; k++) {
                int st = 0;
                double v = viol_of(onevar_eval(fs3[3 * k], fs3[3 * k + 1], fs3[3 * k + 2], x[i], &st), relops[k]);
                if (v > viol) viol = v;
            }
            double new_xi = x[i], new_viol = viol;
            double ss = -tol, es = viol - viol_tol;
            uint32_t it = 0;
            while (es - ss > tol) {
                double s = (ss + es) / 2.0;
                double xi;
                rng_ctx(g, (uint32_t)i, (uint32_t)t, it++);
                int got = orc_onevar_qcqp(0.0, 0.0, 0.0, fs3, relops, mf, s, g, &xi, NULL, 0, NULL);
                if (got < 0) { rc = got; break; }
                if (!got) ss = s;
                else { new_xi = xi; new_viol = s; es = s; }
            }
            if (rc) break;
            if (new_viol < viol) x[i] = new_xi;
        } else {
            double obj[3], new_xi;
            orc_onevar_coeffs(p, 0, x, i, obj);
            rng_ctx(g, (uint32_t)i, (uint32_t)t | 0x80000000u, 0);
            int got = orc_onevar_qcqp(obj[0], obj[1], obj[2], fs3, relops, mf, viol2, g, &new_xi, NULL, 0, NULL);
            if (got < 0) { rc = got; break; }
            if (got && fabs(new_xi - x[i]) > tol) x[i] = new_xi;
        }
    }
    free(fs3); free(relops);
    return rc;
}

/* Optimised CPU baseline of phase 2 (bench.py's second, fairer cpu_baseline): the same algorithm and the same
 * one-variable solver as orc_cd_phase2 (qcqp.py:152-178), but with the bookkeeping a CPU programmer would write
 * instead of the reference's per-call structure: g = P0 x and f0(x) are maintained incrementally (O(n) per ACCEPTED
 * move through column i of the symmetric P0, O(1) per visit), and each coordinate's constraint list is built once.
 * Requires separable constraints (every constraint touches exactly one coordinate); returns -5 otherwise.
 * P0d: dense row-major n x n copy of f0.P.  Trajectories agree with orc_cd_phase2 up to rounding of t1 / t0. */
int orc_cd_phase2_incremental(const orc_prob *p, const double *P0d, double *x, int64_t num_iters, double tol,
                              orc_rng *g, int64_t *stats) {
    const int64_t n = p->n, m = p->m;
    /* coordinate -> its constraints (order preserved) */
    int64_t *coord = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m > 0 ? m : 1));
    int64_t *cnt = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    for (int64_t k = 1; k <= m; k++) {
        const quad_t *f = &p->f[k];
        int64_t c = -1;
        int ok = 1;
        for (int64_t i = 0; i < n && ok; i++) {
            for (int64_t jj = f->ptr[i]; jj < f->ptr[i + 1]; jj++) {
                if (f->val[jj] == 0.0) continue;
                if (f->idx[jj] != i || (c >= 0 && c != i)) { ok = 0; break; }
                c = i;
            }
            if (f->q[i] != 0.0) { if (c >= 0 && c != i) ok = 0; c = i; }
        }
        if (!ok || c < 0) { free(coord); free(cnt); return -5; }
        coord[k - 1] = c;
        cnt[c + 1]++;
    }
    for (int64_t i = 0; i < n; i++) cnt[i + 1] += cnt[i];
    int64_t *list = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m > 0 ? m : 1));
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    for (int64_t i = 0; i < n; i++) fill[i] = cnt[i];
    for (int64_t k = 1; k <= m; k++) list[fill[coord[k - 1]]++] = k;
    int64_t maxc = 1;
    for (int64_t i = 0; i < n; i++) if (cnt[i + 1] - cnt[i] > maxc) maxc = cnt[i + 1] - cnt[i];
    double *fs3 = (double *)malloc(sizeof(double) * 3 * (size_t)maxc);
    int *relops = (int *)malloc(sizeof(int) * (size_t)maxc);
    double *gv = (double *)malloc(sizeof(double) * (size_t)n);
    const double *q0 = p->f[0].q;
    const double viol = orc_max_violation(p, x);
    double fcur = 0.0;
    for (int64_t i = 0; i < n; i++) {
        double acc = 0.0;
        const double *row = P0d + i * n;
        for (int64_t j = 0; j < n; j++) acc += row[j] * x[j];
        gv[i] = acc;
        fcur += (acc + q0[i]) * x[i];
    }
    fcur += p->f[0].r;
    int64_t update_counter = 0, sweeps = 0, visits = 0, accepted = 0;
    int converged = 0, rc = 0;
    for (int64_t t = 0; t < num_iters && !converged && rc == 0; t++) {
        sweeps++;
        for (int64_t i = 0; i < n; i++) {
            visits++;
            const double xi = x[i], t2 = P0d[i * n + i];
            const double t1 = 2.0 * (gv[i] - t2 * xi) + q0[i];
            const double t0 = fcur - xi * (t2 * xi + t1);
            int64_t mf = 0;
            for (int64_t e = cnt[i]; e < cnt[i + 1]; e++) {
                const quad_t *f = &p->f[list[e]];
                double pk = 0.0;
                for (int64_t jj = f->ptr[i]; jj < f->ptr[i + 1]; jj++) if (f->idx[jj] == i) pk += f->val[jj];
                if (pk != 0.0 || f->q[i] != 0.0) {
                    fs3[3 * mf] = pk; fs3[3 * mf + 1] = f->q[i]; fs3[3 * mf + 2] = f->r;
                    relops[mf++] = f->relop;
                }
            }
            double new_xi;
            rng_ctx(g, (uint32_t)i, (uint32_t)t | 0x80000000u, 0);
            int got = orc_onevar_qcqp(t2, t1, t0, fs3, relops, mf, viol, g, &new_xi, NULL, 0, NULL);
            if (got < 0) { rc = got; break; }
            if (got && fabs(new_xi - xi) > tol) {
                const double d = new_xi - xi;
                fcur += d * (t2 * d + 2.0 * gv[i] + q0[i]);
                const double *col = P0d + i * n;      /* symmetric: column i = row i */
                for (int64_t j = 0; j < n; j++) gv[j] += col[j] * d;
                x[i] = new_xi; update_counter = 0; accepted++;
            } else {
                update_counter++;
                if (update_counter == n) { converged = 1; break; }
            }
        }
    }
    if (stats) { stats[0] = sweeps; stats[1] = visits; stats[2] = accepted; }
    free(coord); free(cnt); free(list); free(fill); free(fs3); free(relops); free(gv);
    return rc;
}

int orc_improve_cd(const orc_prob *p, double *x, int64_t num_iters, double viol_tol,
                   double tol, int phase1, orc_rng *g, int64_t *stats1, int64_t *stats2) {
    int rc = 0;
    if (stats1) stats1[0] = stats1[1] = stats1[2] = 0;
    if (stats2) stats2[0] = stats2[1] = stats2[2] = 0;
    if (phase1) rc = orc_cd_phase1(p, x, num_iters, viol_tol, tol, g, stats1);
    if (rc) return rc;
    if (orc_max_violation(p, x) < viol_tol)
        rc = orc_cd_phase2(p, x, num_iters, viol_tol, tol, g, stats2);
    return rc;
}

/* ------------------------------------------------------------ onecons_qcqp */

typedef struct {
    int64_t n;
    const double *lmb;
    double *zhat, *qhat, *xhat;
    double r;
} secular_t;

static double secular_phi(secular_t *S, double nu) {
    /* xhat(nu) = -(nu*qhat - 2*zhat) / (2*(1 + nu*lmb)); phi = lmb.xhat^2 + qhat.xhat + r */
    double a = 0.0, b = 0.0;
    for (int64_t j = 0; j < S->n; j++) {
        double xh = -(nu * S->qhat[j] - 2.0 * S->zhat[j]) / (2.0 * (1.0 + nu * S->lmb[j]));
        S->xhat[j] = xh;
        a += S->lmb[j] * (xh * xh);
        b += S->qhat[j] * xh;
    }
    return a + b + S->r;
}

int orc_onecons(const orc_prob *p, int64_t k, const double *z, const double *lmb,
                const double *Q, double tol, double *out) {
    const quad_t *f = &p->f[k];
    int64_t n = p->n;
    if (f->relop == ORC_RELOP_LE && orc_eval(p, k, z) <= 0.0) {
        memcpy(out, z, sizeof(double) * (size_t)n);
        return -1;
    }
    secular_t S;
    S.n = n; S.lmb = lmb; S.r = f->r;
    S.zhat = (double *)calloc((size_t)n, sizeof(double));
    S.qhat = (double *)calloc((size_t)n, sizeof(double));
    S.xhat = (double *)calloc((size_t)n, sizeof(double));
    for (int64_t i = 0; i < n; i++) { /* Q.T.dot(z), Q.T.dot(q) */
        const double *Qi = Q + i * n;
        double zi = z[i], qi = f->q[i];
        for (int64_t j = 0; j < n; j++) { S.zhat[j] += Qi[j] * zi; S.qhat[j] += Qi[j] * qi; }
    }
    double s = -INFINITY, e = INFINITY;
    for (int64_t j = 0; j < n; j++) {
        double l = lmb[j];
        if (l > 0.0) { double c = -1.0 / l; if (c > s) s = c; }
        if (l < 0.0) { double c = -1.0 / l; if (c < e) e = c; }
    }
    int guard = 0;
    if (s == -INFINITY) { s = -1.0; while (secular_phi(&S, s) <= 0.0 && guard++ < 2000) s *= 2.0; }
    if (e == INFINITY) { e = 1.0; while (secular_phi(&S, e) >= 0.0 && guard++ < 4000) e *= 2.0; }
    int steps = 0;
    while (e - s > tol) {
        double m = (s + e) / 2.0;
        double ph = secular_phi(&S, m);
        steps++;
        if (ph > 0.0) s = m;
        else if (ph < 0.0) e = m;
        else { s = e = m; break; }
        if (steps > 100000) break;
    }
    double nu = (s + e) / 2.0;
    secular_phi(&S, nu);
    for (int64_t i = 0; i < n; i++) { /* Q.dot(xhat) */
        const double *Qi = Q + i * n;
        double acc = 0.0;
        for (int64_t j = 0; j < n; j++) acc += Qi[j] * S.xhat[j];
        out[i] = acc;
    }
    free(S.zhat); free(S.qhat); free(S.xhat);
    return steps;
}

/* -------------------------------------------------------------------- ADMM */

static void consensus_sums(int64_t n, int64_t m, const double *xs, const double *us,
                           double *sx, double *su) {
    /* python sum(list of arrays): ((0 + a0) + a1) + ... */
    for (int64_t j = 0; j < n; j++) { sx[j] = 0.0; su[j] = 0.0; }
    for (int64_t i = 0; i < m; i++)
        for (int64_t j = 0; j < n; j++) { sx[j] += xs[i * n + j]; su[j] += us[i * n + j]; }
}

int orc_admm_phase1(const orc_prob *p, double *z, const double *lmb, const double *Q,
                    double tol, int64_t num_iters, int64_t *iters_out) {
    int64_t n = p->n, m = p->m;
    double *xs = (double *)malloc(sizeof(double) * (size_t)(n * m));
    double *us = (double *)calloc((size_t)(n * m), sizeof(double));
    double *sx = (double *)malloc(sizeof(double) * (size_t)n);
    double *su = (double *)malloc(sizeof(double) * (size_t)n);
    double *v = (double *)malloc(sizeof(double) * (size_t)n);
    for (int64_t i = 0; i < m; i++) memcpy(xs + i * n, z, sizeof(double) * (size_t)n);
    int64_t t;
    for (t = 0; t < num_iters; t++) {
        if (orc_max_violation(p, z) < tol) break;
        consensus_sums(n, m, xs, us, sx, su);
        for (int64_t j = 0; j < n; j++) z[j] = (sx[j] - su[j]) / (double)m;
        for (int64_t i = 0; i < m; i++) {
            for (int64_t j = 0; j < n; j++) v[j] = z[j] + us[i * n + j];
            orc_onecons(p, i + 1, v, lmb + i * n, Q + i * n * n, 1e-6, xs + i * n);
        }
        for (int64_t i = 0; i < m; i++)
            for (int64_t j = 0; j < n; j++) us[i * n + j] += z[j] - xs[i * n + j];
    }
    if (iters_out) *iters_out = t;
    free(xs); free(us); free(sx); free(su); free(v);
    return 0;
}

static void chol_solve(int64_t n, const double *L, double *b) {
    for (int64_t i = 0; i < n; i++) { /* L y = b */
        double acc = b[i];
        for (int64_t j = 0; j < i; j++) acc -= L[i * n + j] * b[j];
        b[i] = acc / L[i * n + i];
    }
    for (int64_t i = n - 1; i >= 0; i--) { /* L^T x = y */
        double acc = b[i];
        for (int64_t j = i + 1; j < n; j++) acc -= L[j * n + i] * b[j];
        b[i] = acc / L[i * n + i];
    }
}

int orc_admm_phase2(const orc_prob *p, double *x, double rho, const double *lmb,
                    const double *Q, const double *chol, double tol, int64_t num_iters,
                    double viol_lim, int64_t *iters_out) {
    int64_t n = p->n, m = p->m;
    const quad_t *f0 = &p->f[0];
    double *xs = (double *)malloc(sizeof(double) * (size_t)(n * m));
    double *us = (double *)calloc((size_t)(n * m), sizeof(double));
    double *sx = (double *)malloc(sizeof(double) * (size_t)n);
    double *su = (double *)malloc(sizeof(double) * (size_t)n);
    double *v = (double *)malloc(sizeof(double) * (size_t)n);
    double *z = (double *)malloc(sizeof(double) * (size_t)n);
    double *last_z = (double *)malloc(sizeof(double) * (size_t)n);
    double *bestx = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(bestx, x, sizeof(double) * (size_t)n);
    for (int64_t i = 0; i < m; i++) memcpy(xs + i * n, x, sizeof(double) * (size_t)n);
    int have_last = 0;
    int64_t t;
    for (t = 0; t < num_iters; t++) {
        consensus_sums(n, m, xs, us, sx, su);
        for (int64_t j = 0; j < n; j++) z[j] = 2.0 * rho * (sx[j] - su[j]) - f0->q[j];
        chol_solve(n, chol, z);
        for (int64_t i = 0; i < m; i++) {
            for (int64_t j = 0; j < n; j++) v[j] = z[j] + us[i * n + j];
            orc_onecons(p, i + 1, v, lmb + i * n, Q + i * n * n, 1e-6, xs + i * n);
        }
        for (int64_t i = 0; i < m; i++)
            for (int64_t j = 0; j < n; j++) us[i * n + j] += z[j] - xs[i * n + j];
        if (have_last) {
            double nrm = 0.0;
            for (int64_t j = 0; j < n; j++) { double d = last_z[j] - z[j]; nrm += d * d; }
            if (sqrt(nrm) < tol) { t++; break; }
        }
        memcpy(last_z, z, sizeof(double) * (size_t)n);
        have_last = 1;
        double maxviol = orc_max_violation(p, z);
        if (maxviol > viol_lim) { t++; break; }
        if (orc_better(p, z, bestx, 1e-4) == 1) memcpy(bestx, z, sizeof(double) * (size_t)n);
    }
    memcpy(x, bestx, sizeof(double) * (size_t)n);
    if (iters_out) *iters_out = t;
    free(xs); free(us); free(sx); free(su); free(v); free(z); free(last_z); free(bestx);
    return 0;
}
