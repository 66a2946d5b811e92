/*
 * selftest.c -- TEST INFRASTRUCTURE ONLY.  Driver for the sanitizer build of the oracle
 * (`make -C oracle asan`): walks every entry point of qcqp_oracle.h on a small Boolean least squares
 * problem so that AddressSanitizer / UBSan see the code paths the parity tests rely on.
 * Exit code 0 = no sanitizer report and the cross-checks below hold.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "qcqp_oracle.h"

#define N 12
#define MROWS 9

static double lcg(uint64_t *s) {
    *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
    return ((double)((*s >> 11) & ((1ULL << 53) - 1)) / (double)(1ULL << 53)) * 2.0 - 1.0;
}

int main(void) {
    uint64_t seed = 12345;
    static double A[MROWS][N], b[MROWS], P0[N * N], q0[N];
    for (int i = 0; i < MROWS; i++) { for (int j = 0; j < N; j++) A[i][j] = lcg(&seed); b[i] = lcg(&seed); }
    double r0 = 0.0;
    for (int i = 0; i < MROWS; i++) r0 += b[i] * b[i];
    for (int j = 0; j < N; j++) {
        q0[j] = 0.0;
        for (int i = 0; i < MROWS; i++) q0[j] -= 2.0 * A[i][j] * b[i];
        for (int k = 0; k < N; k++) {
            double s = 0.0;
            for (int i = 0; i < MROWS; i++) s += A[i][j] * A[i][k];
            P0[j * N + k] = s;
        }
    }
    for (int j = 0; j < N; j++) for (int k = j + 1; k < N; k++) { double v = 0.5 * (P0[j * N + k] + P0[k * N + j]); P0[j * N + k] = P0[k * N + j] = v; }
    orc_prob *p = orc_prob_new(N, N);
    {   /* objective: dense CSR */
        int64_t ptr[N + 1], idx[N * N];
        for (int i = 0; i <= N; i++) ptr[i] = (int64_t)i * N;
        for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) idx[i * N + j] = j;
        if (orc_prob_set(p, 0, N * N, ptr, idx, P0, q0, r0, ORC_RELOP_NONE)) return 2;
    }
    for (int k = 0; k < N; k++) {   /* x_k^2 == 1 */
        int64_t ptr[N + 1], idx[1] = {k};
        double val[1] = {1.0}, q[N];
        for (int i = 0; i <= N; i++) ptr[i] = i > k ? 1 : 0;
        for (int i = 0; i < N; i++) q[i] = 0.0;
        if (orc_prob_set(p, k + 1, 1, ptr, idx, val, q, -1.0, ORC_RELOP_EQ)) return 3;
    }
    if (orc_prob_n(p) != N || orc_prob_m(p) != N) return 4;
    double x[N], y[N];
    for (int j = 0; j < N; j++) x[j] = orc_keyed_normal(7, 3, (uint64_t)j);
    double F[(N + 1) * 1], f0, mv;
    orc_eval_batch(p, x, 1, &f0, &mv, F);
    if (!(fabs(f0 - orc_eval(p, 0, x)) <= 1e-12 * (1 + fabs(f0))) || !(mv == orc_max_violation(p, x))) return 5;
    (void)orc_better(p, x, x, 1e-4);
    double t3[3];
    orc_onevar_coeffs(p, 0, x, 5, t3);
    /* feasible intervals over a small grid (every branch of utilities.py:198-232) */
    const double ps[] = {-2.0, -5e-5, 0.0, 5e-5, 1.0}, qs[] = {-1.0, 0.0, 5e-5, 2.0}, rs[] = {-1.0, 0.0, 3.0};
    for (unsigned a = 0; a < 5; a++) for (unsigned c = 0; c < 4; c++) for (unsigned d = 0; d < 3; d++)
        for (int rel = 1; rel <= 2; rel++) { double out[8]; (void)orc_feasible_intervals(ps[a], qs[c], rs[d], rel, 0.3, 1e-4, out); }
    /* coordinate descent, keyed stream; then the incremental baseline from the same phase-1 point */
    orc_rng *g = orc_rng_new(ORC_RNG_KEYED, 99);
    orc_rng_set_restart(g, 3);
    int64_t s1[3], s2[3], s3[3];
    if (orc_cd_phase1(p, x, 50, 1e-2, 1e-4, g, s1)) return 6;
    for (int j = 0; j < N; j++) y[j] = x[j];
    if (orc_cd_phase2(p, x, 50, 1e-2, 1e-4, g, s2)) return 7;
    if (orc_cd_phase2_incremental(p, P0, y, 50, 1e-4, g, s3)) return 8;
    for (int j = 0; j < N; j++) if (!(fabs(x[j] - y[j]) < 1e-9)) { fprintf(stderr, "incremental mismatch %d %.17g %.17g\n", j, x[j], y[j]); return 9; }
    if (s2[1] != s3[1] || s2[2] != s3[2]) return 10;
    if (!(orc_max_violation(p, x) < 1e-2)) return 11;
    (void)orc_rng_draws(g);
    orc_rng_free(g);
    /* MT stream + full driver */
    g = orc_rng_new(ORC_RNG_MT, 5);
    for (int j = 0; j < N; j++) x[j] = 2.0 * orc_rng_uniform(g, -1.0, 1.0);
    (void)orc_rng_choice(g, 3);
    if (orc_improve_cd(p, x, 30, 1e-2, 1e-4, 1, g, s1, s2)) return 12;
    orc_rng_free(g);
    /* onecons + ADMM phase 1 with the exact eigenpairs of e_k e_k^T: identity basis */
    static double lmb[N * N], Q[N * N * N], out[N];
    for (int k = 0; k < N; k++) for (int j = 0; j < N; j++) {
        lmb[k * N + j] = (j == k) ? 1.0 : 0.0;
        for (int i = 0; i < N; i++) Q[(k * N + i) * N + j] = (i == j) ? 1.0 : 0.0;
    }
    for (int j = 0; j < N; j++) x[j] = 0.3 + 0.1 * j;
    (void)orc_onecons(p, 2, x, lmb + 1 * N, Q + 1 * N * N, 1e-6, out);
    if (!(fabs(fabs(out[1]) - 1.0) < 1e-5)) return 13;
    int64_t it = 0;
    if (orc_admm_phase1(p, x, lmb, Q, 1e-2, 50, &it)) return 14;
    if (!(orc_max_violation(p, x) == orc_max_violation(p, x))) return 15;   /* finite */
    /* phase 2 with the Cholesky factor of 2 (P0 + rho m I) */
    static double M[N * N], Lc[N * N];
    const double rho = 1.0;
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) M[i * N + j] = 2.0 * (P0[i * N + j] + (i == j ? rho * N : 0.0));
    for (int i = 0; i < N; i++) for (int j = 0; j <= i; j++) {
        double s = M[i * N + j];
        for (int k = 0; k < j; k++) s -= Lc[i * N + k] * Lc[j * N + k];
        Lc[i * N + j] = (i == j) ? sqrt(s) : s / Lc[j * N + j];
    }
    if (orc_admm_phase2(p, x, rho, lmb, Q, Lc, 1e-2, 40, 1e4, &it)) return 16;
    orc_prob_free(p);
    printf("oracle selftest ok\n");
    return 0;
}
