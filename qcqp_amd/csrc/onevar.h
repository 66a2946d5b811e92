// One-variable sub-solver of coordinate descent, as straight-line per-lane code.
//
// Behaviour follows the reference exactly, quirks included:
//   get_feasible_intervals   utilities.py:198-232
//   onevar_qcqp              utilities.py:241-288  (event sweep + scalar minimiser)
// but the data structures are GPU ones: a coordinate's constraints are at most MAXC
// (p, q, r, relop) quadruples, the interval end points are a fixed-size register array and the
// reference's dict/sort sweep is an O(E^2) branch-free rank computation (E = 2 + 4 MAXC).
#pragma once
#include <hip/hip_runtime.h>
#include "philox.h"

namespace qcqpmi {

enum { RELOP_NONE = 0, RELOP_LE = 1, RELOP_EQ = 2 };

#define QM_INF (__builtin_inf())
#define QM_NAN (__builtin_nan(""))

struct Seg2 { int n; double lo0, hi0, lo1, hi1; };

// p x^2 + q x + rs <= 0 with rs = r - s, smr = s - r  (utilities.py:209-231)
__device__ inline Seg2 intervals_le(double p, double q, double rs, double smr, double tol) {
    Seg2 o;
    o.n = 0; o.lo0 = o.hi0 = o.lo1 = o.hi1 = 0.0;
    if (p > tol) {
        double D = q * q - 4.0 * p * rs;
        if (D >= 0.0) {
            double rD = sqrt(D);
            o.n = 1; o.lo0 = (-q - rD) / (2.0 * p); o.hi0 = (-q + rD) / (2.0 * p);
        }
    } else if (p < -tol) {
        double D = q * q - 4.0 * p * rs;
        if (D >= 0.0) {
            double rD = sqrt(D);
            o.n = 2;
            o.lo0 = -QM_INF; o.hi0 = (-q + rD) / (2.0 * p);
            o.lo1 = (-q - rD) / (2.0 * p); o.hi1 = QM_INF;
        } else {
            o.n = 1; o.lo0 = -QM_INF; o.hi0 = QM_INF;
        }
    } else {
        o.n = 1;
        if (q > tol) { o.lo0 = -QM_INF; o.hi0 = smr / q; }
        else if (q < -tol) { o.lo0 = smr / q; o.hi0 = QM_INF; }
        else { o.lo0 = -QM_INF; o.hi0 = QM_INF; }
    }
    return o;
}

// Feasible intervals of ONE constraint at slack s; at most 2 intervals for either relop.
__device__ inline Seg2 feasible_intervals(double p, double q, double r, int relop, double s) {
    const double tol = 1e-4;
    if (relop != RELOP_EQ) return intervals_le(p, q, r - s, s - r, tol);
    // |f| <= s: intersect f - s <= 0 with -f - s <= 0, each at the DEFAULT slack 0
    double r1 = r - s, r2 = -r - s;
    Seg2 a = intervals_le(p, q, r1 - 0.0, 0.0 - r1, tol);
    Seg2 b = intervals_le(-p, -q, r2 - 0.0, 0.0 - r2, tol);
    double alo[2] = {a.lo0, a.lo1}, ahi[2] = {a.hi0, a.hi1};
    double blo[2] = {b.lo0, b.lo1}, bhi[2] = {b.hi0, b.hi1};
    // up to 4 pairwise intersections can be non-empty only in degenerate (touching) cases;
    // keep the first 4 in the reference's (i, j) order and let the sweep merge them.
    double lo[4], hi[4];
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            double l = alo[i] > blo[j] ? alo[i] : blo[j];
            double h = ahi[i] < bhi[j] ? ahi[i] : bhi[j];
            bool ok = (i < a.n) && (j < b.n) && (l <= h);
            // compact without dynamic indexing
            if (ok) {
                if (cnt == 0) { lo[0] = l; hi[0] = h; }
                else if (cnt == 1) { lo[1] = l; hi[1] = h; }
                else if (cnt == 2) { lo[2] = l; hi[2] = h; }
                else { lo[3] = l; hi[3] = h; }
                cnt++;
            }
        }
    Seg2 o;
    o.n = cnt > 2 ? 2 : cnt;   // >2 only when intervals degenerate to shared points; see sweep
    o.lo0 = cnt > 0 ? lo[0] : 0.0; o.hi0 = cnt > 0 ? hi[0] : 0.0;
    o.lo1 = cnt > 1 ? lo[1] : 0.0; o.hi1 = cnt > 1 ? hi[1] : 0.0;
    return o;
}

// Feasible set C of a coordinate: the reference's counting sweep over interval end points.
template <int MAXC>
struct FeasSet {
    int n;
    double lo[MAXC + 1], hi[MAXC + 1];
};

template <int MAXC>
__device__ inline void feasible_set(const double *__restrict__ cp, const double *__restrict__ cq,
                                    const double *__restrict__ cr, const int *__restrict__ crel,
                                    int mf, double s, FeasSet<MAXC> &out) {
    constexpr int E = 2 + 4 * MAXC;
    double key[E];
    int cnt[E];
    key[0] = -QM_INF; cnt[0] = +1;
    key[1] = QM_INF; cnt[1] = -1;
#pragma unroll
    for (int k = 0; k < MAXC; k++) {
        Seg2 iv;
        iv.n = 0; iv.lo0 = iv.hi0 = iv.lo1 = iv.hi1 = 0.0;
        if (k < mf) iv = feasible_intervals(cp[k], cq[k], cr[k], crel[k], s);
        key[2 + 4 * k + 0] = iv.lo0; cnt[2 + 4 * k + 0] = iv.n > 0 ? +1 : 0;
        key[2 + 4 * k + 1] = iv.hi0; cnt[2 + 4 * k + 1] = iv.n > 0 ? -1 : 0;
        key[2 + 4 * k + 2] = iv.lo1; cnt[2 + 4 * k + 2] = iv.n > 1 ? +1 : 0;
        key[2 + 4 * k + 3] = iv.hi1; cnt[2 + 4 * k + 3] = iv.n > 1 ? -1 : 0;
    }
    // per event: net count of its key, running total up to and including its key, first-occurrence
    int net[E], tot[E];
    bool first[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        int nn = 0, tt = 0;
        bool fo = true;
#pragma unroll
        for (int f = 0; f < E; f++) {
            bool same = (key[f] == key[e]);
            nn += same ? cnt[f] : 0;
            tt += (key[f] <= key[e]) ? cnt[f] : 0;
            if (f < e && same && cnt[f] != 0) fo = false;
        }
        net[e] = nn; tot[e] = tt; first[e] = fo && cnt[e] != 0;
    }
    out.n = 0;
#pragma unroll
    for (int j = 0; j <= MAXC; j++) { out.lo[j] = 0.0; out.hi[j] = 0.0; }
#pragma unroll
    for (int e = 0; e < E; e++) {
        bool is_end = first[e] && net[e] == -1 && tot[e] == mf;
        // left end = previous key with non-zero net count; rank = #segment ends to the left
        double left = -QM_INF;
        int rank = 0;
#pragma unroll
        for (int f = 0; f < E; f++) {
            bool nz = first[f] && net[f] != 0 && key[f] < key[e];
            if (nz && key[f] > left) left = key[f];
            bool endf = first[f] && net[f] == -1 && tot[f] == mf && key[f] < key[e];
            rank += endf ? 1 : 0;
        }
        if (is_end) {
#pragma unroll
            for (int j = 0; j <= MAXC; j++)
                if (rank == j) { out.lo[j] = left; out.hi[j] = key[e]; }
            out.n++;
        }
    }
}

// The same feasible set for the common case of ONE constraint on the coordinate (mf == 1), derived
// from the counting sweep in closed form: the constraint's intervals, with touching intervals
// merged (hi0 == lo1: the shared key nets to zero), zero-width intervals dropped (lo == hi nets to
// zero) and intervals that end at +inf dropped (the base interval also ends there: the event is -2,
// not -1 -- SURVEY.md A.6).
template <int MAXC>
__device__ inline void feasible_set_single(double p, double q, double r, int relop, double s,
                                           FeasSet<MAXC> &out) {
    Seg2 iv = feasible_intervals(p, q, r, relop, s);
    double lo0 = iv.lo0, hi0 = iv.hi0, lo1 = iv.lo1, hi1 = iv.hi1;
    int n = iv.n;
    if (n == 2 && lo0 == lo1 && hi0 == hi1) {
        // two identical intervals: +2 / -2 events, never a -1: nothing is recorded
        n = 0;
    } else if (n == 2 && hi0 == lo1) { hi0 = hi1; n = 1; }           // [a,b] [b,c] -> [a,c]
    else if (n == 2 && hi1 == lo0) { lo0 = lo1; n = 1; }              // (same, other order)
    bool k0 = n >= 1 && lo0 != hi0 && hi0 != QM_INF;
    bool k1 = n >= 2 && lo1 != hi1 && hi1 != QM_INF;
    // a left end at -inf coincides with the base interval's start: the count there is +2, harmless
    // -- unless BOTH intervals start at -inf (cannot happen for one constraint).
    if (k0 && k1 && lo1 < lo0) {   // keep ascending order
        double t;
        t = lo0; lo0 = lo1; lo1 = t;
        t = hi0; hi0 = hi1; hi1 = t;
    }
    out.n = 0;
#pragma unroll
    for (int j = 0; j <= MAXC; j++) { out.lo[j] = 0.0; out.hi[j] = 0.0; }
    if (k0) { out.lo[0] = lo0; out.hi[0] = hi0; out.n = 1; }
    if (k1) {
        if (out.n == 0) { out.lo[0] = lo1; out.hi[0] = hi1; }
        else { out.lo[1] = lo1; out.hi[1] = hi1; }
        out.n++;
    }
}

// OneVarQuadraticFunction.eval with the +-inf branches (utilities.py:115-120); the reference
// raises NameError for P == q == 0 at +-inf, signalled here through *err.
__device__ inline double onevar_eval(double p, double q, double r, double x, int *err) {
    if (__builtin_isinf(x)) {
        if (p != 0.0) return p * x * x;
        if (q != 0.0) return q * x;
        *err = 1;
        return QM_NAN;
    }
    return x * (p * x + q) + r;
}

// Scalar minimiser over the feasible set (utilities.py:257-288).
// returns 1 and *xout on success, 0 for None, <0 where the reference raises.
struct DrawKey {
    uint64_t seed, restart;
    uint32_t coord, sweep_tag, iter;
};

template <int MAXC>
__device__ inline int onevar_minimise(double p0, double q0, double r0, const FeasSet<MAXC> &C,
                                      const DrawKey &dk, double *xout) {
    if (C.n == 0) return 0;
    if (p0 == 0.0 && q0 == 0.0) {
        // the Philox draw is only materialised on the (rare in phase 2) paths that consume it
        U4 rnd = cd_draw(dk.seed, dk.restart, dk.coord, dk.sweep_tag, dk.iter);
        int c = draw_choice(rnd, C.n);
        double lo = C.lo[0], hi = C.hi[0];
#pragma unroll
        for (int j = 1; j <= MAXC; j++)
            if (c == j) { lo = C.lo[j]; hi = C.hi[j]; }
        if (__builtin_isinf(lo) || __builtin_isinf(hi)) return -1;  // numpy: OverflowError
        *xout = draw_uniform(rnd, lo, hi);
        return 1;
    }
    double x0 = (p0 > 0.0) ? -q0 / (2.0 * p0) : QM_NAN;
    double bestf = QM_INF;
    int nb = 0, err = 0;
    // first pass: best value and number of ties
    double fl[MAXC + 1], fr[MAXC + 1];
    bool hit = false;
#pragma unroll
    for (int j = 0; j <= MAXC; j++) {
        if (j < C.n && !hit) {
            if (C.lo[j] <= x0 && x0 <= C.hi[j]) { hit = true; }
            else {
                fl[j] = onevar_eval(p0, q0, r0, C.lo[j], &err);
                fr[j] = onevar_eval(p0, q0, r0, C.hi[j], &err);
                if (bestf > fl[j]) { nb = 1; bestf = fl[j]; } else if (bestf == fl[j]) nb++;
                if (bestf > fr[j]) { nb = 1; bestf = fr[j]; } else if (bestf == fr[j]) nb++;
            }
        }
    }
    if (hit) { *xout = x0; return 1; }
    if (err) return -2;
    if (nb == 0) return 0;
    // second pass: the idx-th candidate whose value equals the best one (np.random.choice(bestxs))
    int idx = 0, seen = 0;
    if (nb > 1) {
        U4 rnd = cd_draw(dk.seed, dk.restart, dk.coord, dk.sweep_tag, dk.iter);
        idx = draw_choice(rnd, nb);
    }
    double pick = 0.0;
#pragma unroll
    for (int j = 0; j <= MAXC; j++) {
        if (j < C.n) {
            if (fl[j] == bestf) { if (seen == idx) pick = C.lo[j]; seen++; }
            if (fr[j] == bestf) { if (seen == idx) pick = C.hi[j]; seen++; }
        }
    }
    *xout = pick;
    return 1;
}

}  // namespace qcqpmi
