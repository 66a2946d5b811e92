// One coordinate visit of coordinate-descent PHASE 1 (qcqp.py:113-136) for separable constraints -- shared by
// cd_phase1_sep_kernel (kernels.hip: one launch per population) and by the lifecycle mode of the slot-queue kernel
// (cd_queue.hip: a restart's suggest + phase 1 + gate + phase 2 inside one persistent launch), so that both make the same
// moves bit for bit.
//
// A coordinate's local violation and its update depend on x_i alone (the objective is identically zero in phase 1,
// qcqp.py:114): the reference's bisection on the achievable slack s in [-tol, viol - viol_tol] (qcqp.py:122-131) with the
// exact interval sweep of onevar_qcqp per step.  Only the LAST successful bisection step decides the point (qcqp.py:126-131
// overwrite new_xi each time) and every draw of the counter-based stream is independent of the others, so the Philox draw is
// deferred: a successful step just remembers its set and its draw index.  A set with an unbounded piece draws at once (the
// reference may raise there).
#pragma once
#include "kernels.h"
#include "onevar.h"

namespace qcqpmi {

// QuadraticFunction.violation (utilities.py:56-62)
__device__ inline double viol_of(double f, int relop) {
    if (relop == RELOP_EQ) return fabs(f);
    return f > 0.0 ? f : 0.0;
}

struct P1Visit {
    bool visited;     // false: the coordinate carries no constraint (python: max([]) -> ValueError; status -3)
    bool moved;       // the update was accepted (new_viol < viol, qcqp.py:132): xi holds the new value
    double vafter;    // max violation of the coordinate's constraints at xi after the visit (feeds qcqp.py:142)
    int status;       // 0, or < 0 where the reference would raise
};

// the visit for a coordinate whose mf >= 1 constraints (p, q, r, relop) are given
template <int MAXC>
__device__ inline void p1_sep_visit_core(int mf, const double (&cp)[MAXC], const double (&cq)[MAXC], const double (&cr)[MAXC],
                                         const int (&crel)[MAXC], int64_t i, double &xi, double tol, double viol_tol, uint64_t seed,
                                         uint64_t grestart, int64_t t, P1Visit &V) {
    V.visited = true; V.moved = false; V.vafter = -QM_INF; V.status = 0;
    double viol = -QM_INF;
#pragma unroll
    for (int k = 0; k < MAXC; k++)
        if (k < mf) {
            double v = viol_of(xi * (cp[k] * xi + cq[k]) + cr[k], crel[k]);
            viol = v > viol ? v : viol;
        }
    double new_xi = xi, new_viol = viol;
    double ss = -tol, es = viol - viol_tol;
    uint32_t it = 0;
    FeasSet<MAXC> Cp;
    uint32_t itp = 0;
    bool pending = false;
    // Boolean-type constraint p x^2 + r == 0 (p > tol, no linear term): |f| <= s is the band
    // a <= |x| <= b with a = sqrt(D2)/(2p), b = sqrt(D1)/(2p), D1 = 4p(s - r), D2 = -4p(r + s)
    // (utilities.py:209-231 with q = 0).  The set is non-empty iff D1 > 0 and (D2 < 0 or D2 < D1)
    // -- square root and division are monotone, touching / zero-width pieces vanish in the sweep --
    // so the bisection only needs the two discriminants; the set itself is built once, for the
    // last successful slack.
    const bool band = mf == 1 && cq[0] == 0.0 && crel[0] == RELOP_EQ && cp[0] > 1e-4;
    double sp = 0.0;
    while (es - ss > tol) {
        double s = (ss + es) / 2.0;
        if (band) {
            const uint32_t itb = it++;
            const double D1 = 0.0 - 4.0 * cp[0] * (cr[0] - s);     // q*q - 4 p rs, as the reference forms it
            const double D2 = 0.0 - 4.0 * (-cp[0]) * (-cr[0] - s);
            const bool nonempty = D1 > 0.0 && (D2 < 0.0 || D2 < D1);
            if (!nonempty) { ss = s; continue; }
            sp = s; itp = itb; pending = true;
            new_viol = s; es = s;
            continue;
        }
        FeasSet<MAXC> C;
        if (mf == 1) feasible_set_single<MAXC>(cp[0], cq[0], cr[0], crel[0], s, C);
        else feasible_set<MAXC>(cp, cq, cr, crel, mf, s, C);
        const uint32_t itc = it++;
        if (C.n == 0) { ss = s; continue; }
        bool unb = false;
#pragma unroll
        for (int j = 0; j <= MAXC; j++) unb = unb || (j < C.n && (__builtin_isinf(C.lo[j]) || __builtin_isinf(C.hi[j])));
        if (unb) {
            DrawKey dk{seed, grestart, (uint32_t)i, (uint32_t)t, itc};
            double xn;
            int got = onevar_minimise<MAXC>(0.0, 0.0, 0.0, C, dk, &xn);
            if (got < 0) { V.status = got; pending = false; break; }
            new_xi = xn; pending = false;
        } else {
            Cp = C; itp = itc; pending = true;
        }
        new_viol = s; es = s;
    }
    if (pending && band) feasible_set_single<MAXC>(cp[0], cq[0], cr[0], crel[0], sp, Cp);
    if (pending) {
        DrawKey dk{seed, grestart, (uint32_t)i, (uint32_t)t, itp};
        double xn = xi;
        const int got = onevar_minimise<MAXC>(0.0, 0.0, 0.0, Cp, dk, &xn);
        // the band test decides non-emptiness on the discriminants; the set rebuilt from rounded
        // end points can collapse to nothing: then the step is infeasible at this slack, as in
        // the reference (onevar_qcqp returns None -> the move is not made)
        if (got == 1) new_xi = xn;
        else { new_viol = viol; if (got < 0) V.status = got; }
    }
    if (new_viol < viol) { xi = new_xi; V.moved = true; }
    // violation of the constraints on x_i after the update (feeds qcqp.py:142)
#pragma unroll
    for (int k = 0; k < MAXC; k++)
        if (k < mf) {
            double v = viol_of((cp[k] * xi + cq[k]) * xi + cr[k], crel[k]);
            V.vafter = v > V.vafter ? v : V.vafter;
        }
}

// The same visit for the constraint class the pipelined kernels are built for -- ONE constraint p x^2 + r == 0 on the
// coordinate, p > 1e-4, no linear term ("band": |f| <= s is a <= |x| <= b) -- with the generic machinery of onevar.h resolved
// by hand: feasible_intervals' two intervals_le calls + their pairwise intersection, feasible_set_single's merge / drop rules
// and onevar_minimise's zero-objective branch, each expression the one the generic code evaluates on the same values in the
// same order (the lifecycle kernel runs this one, cd_phase1_sep_kernel the generic one: the GPU tests compare the two bit for
// bit on millions of visits).  q is passed although it is zero: the generic expressions contain it.
__device__ inline void p1_band_visit(double p, double q, double r, int64_t i, double &xi, double tol, double viol_tol, uint64_t seed,
                                     uint64_t grestart, int64_t t, P1Visit &V) {
    V.visited = true; V.moved = false; V.status = 0;
    const double viol = fabs(xi * (p * xi + q) + r);
    double new_xi = xi, new_viol = viol;
    double ss = -tol, es = viol - viol_tol, sp = 0.0;
    uint32_t it = 0, itp = 0;
    bool pending = false;
    // The bisection's test on the two discriminants D1 = 4 p (s - r), D2 = 4 p (-r - s) is, in exact arithmetic,
    // s > r and (s > -r or s > 0).  D1 > 0 and D2 < 0 are decided by the SIGN of one rounded difference (exact); D2 < D1 compares
    // two rounded products that differ by 8 p s.  For r well below zero (the Boolean family: r = -1; every slack of the
    // bisection is >= -tol > r) the test is therefore just s > 0 unless s is within rounding of 0 -- there, and for any other
    // r, the discriminants are formed as the reference forms them.  13 double-precision operations per step become 4: under
    // the matrix instructions of the neighbouring workgroup every one of them waits for a slot of the pipe.
    const bool plain = r < -1e-3;
    const double guard = 1e-9 * (1.0 - r);
    if (plain) {
        // branch-free body (selects): the lanes of a wave run the loop in lock step for as long as the slowest one needs it
        while (es - ss > tol) {
            const double s = (ss + es) / 2.0;
            const uint32_t itb = it++;
            bool nonempty = s > 0.0;
            if (__builtin_expect(!(fabs(s) > guard), 0)) {       // within rounding of the threshold: the discriminants as the reference forms them
                const double D1 = 0.0 - 4.0 * p * (r - s);
                const double D2 = 0.0 - 4.0 * (-p) * (-r - s);
                nonempty = D1 > 0.0 && (D2 < 0.0 || D2 < D1);
            }
            ss = nonempty ? ss : s;
            es = nonempty ? s : es;
            sp = nonempty ? s : sp;
            itp = nonempty ? itb : itp;
            pending = pending || nonempty;
        }
        if (pending) new_viol = sp;
    } else {
        while (es - ss > tol) {
            const double s = (ss + es) / 2.0;
            const uint32_t itb = it++;
            const double D1 = 0.0 - 4.0 * p * (r - s);
            const double D2 = 0.0 - 4.0 * (-p) * (-r - s);
            const bool nonempty = D1 > 0.0 && (D2 < 0.0 || D2 < D1);
            if (!nonempty) { ss = s; continue; }
            sp = s; itp = itb; pending = true;
            new_viol = s; es = s;
        }
    }
    if (pending) {
        // feasible_intervals(p, q, r, ==, sp): A = {f - s <= 0} (convex: at most one interval), B = {-f - s <= 0} (concave: two rays, or the line)
        const double r1 = r - sp, r2 = -r - sp;
        const double rsA = r1 - 0.0, rsB = r2 - 0.0;
        const double DA = q * q - 4.0 * p * rsA;
        const bool hasA = DA >= 0.0;
        const double rDA = sqrt(hasA ? DA : 0.0);
        // q is +-0 for this class (the caller checks): -q -+ rD is -+rD exactly and IEEE division is odd in its numerator, so
        // the reference's four quotients (-q - rD) / (2p), (-q + rD) / (2p), (-qB + rDB) / (2pB), (-qB - rDB) / (2pB) are
        // -tA, tA, -tB, tB with TWO divisions, bit for bit
        const double pB = -p, qB = -q;
        const double DB = qB * qB - 4.0 * pB * rsB;
        const bool twoB = DB >= 0.0;
        const double rDB = sqrt(twoB ? DB : 0.0);
        const double tA = rDA / (2.0 * p), tB = rDB / (2.0 * p);
        const double alo = -tA, ahi = tA;
        const double bhi0 = twoB ? -tB : QM_INF;       // B = (-inf, bhi0] u [blo1, +inf), or the whole line
        const double blo1 = tB;
        // pairwise intersections in the reference's (i, j) order: (A, first piece of B), (A, second piece of B)
        const double l0 = alo > -QM_INF ? alo : -QM_INF, h0 = ahi < bhi0 ? ahi : bhi0;
        const bool ok0 = hasA && l0 <= h0;
        const double l1 = alo > blo1 ? alo : blo1, h1 = ahi < QM_INF ? ahi : QM_INF;
        const bool ok1 = hasA && twoB && l1 <= h1;
        int n = (ok0 ? 1 : 0) + (ok1 ? 1 : 0);
        double lo0 = ok0 ? l0 : (ok1 ? l1 : 0.0), hi0 = ok0 ? h0 : (ok1 ? h1 : 0.0);
        double lo1 = (ok0 && ok1) ? l1 : 0.0, hi1 = (ok0 && ok1) ? h1 : 0.0;
        // feasible_set_single: identical intervals vanish, touching ones merge, zero-width ones and ones that end at +inf are dropped
        if (n == 2 && lo0 == lo1 && hi0 == hi1) n = 0;
        else if (n == 2 && hi0 == lo1) { hi0 = hi1; n = 1; }
        else if (n == 2 && hi1 == lo0) { lo0 = lo1; n = 1; }
        const bool k0 = n >= 1 && lo0 != hi0 && hi0 != QM_INF;
        const bool k1 = n >= 2 && lo1 != hi1 && hi1 != QM_INF;
        if (k0 && k1 && lo1 < lo0) {
            double tt;
            tt = lo0; lo0 = lo1; lo1 = tt;
            tt = hi0; hi0 = hi1; hi1 = tt;
        }
        const int cn = (k0 ? 1 : 0) + (k1 ? 1 : 0);
        const double c0lo = k0 ? lo0 : lo1, c0hi = k0 ? hi0 : hi1;      // first kept interval; the second one is (lo1, hi1) when both are kept
        // onevar_minimise with the objective identically zero: a uniform point of a random interval (utilities.py:266-267)
        int got = 0;
        double xn = xi;
        if (cn > 0) {
            const U4 rnd = cd_draw(seed, grestart, (uint32_t)i, (uint32_t)t, itp);
            const int c = draw_choice(rnd, cn);
            const double lo = (c == 1) ? lo1 : c0lo, hi = (c == 1) ? hi1 : c0hi;
            if (__builtin_isinf(lo) || __builtin_isinf(hi)) got = -1;
            else { xn = draw_uniform(rnd, lo, hi); got = 1; }
        }
        if (got == 1) new_xi = xn;
        else { new_viol = viol; if (got < 0) V.status = got; }
    }
    if (new_viol < viol) { xi = new_xi; V.moved = true; }
    V.vafter = fabs((p * xi + q) * xi + r);
}

// NV visits of the band class at once, their dependency chains side by side in the same basic blocks (round 5): a lone wave
// executes a visit as one long chain of dependent double-precision instructions -- ~20 cycles each with a single wave on the
// SIMD -- and two independent chains interleave almost for free.  Element k of the result is p1_band_visit on element k, bit for
// bit: the bisection loops are fused (an element that is done idles in the body: selects), the tail that builds the set of the
// last successful slack and draws the point is computed for every element and applied where a step succeeded.  r < -1e-3 only
// (the Boolean family; see p1_band_visit for the test on the slack); `on[k]` = false: element k is padding.
template <int NV>
__device__ inline void p1_band_visit_n(double p, double q, double r, const int64_t (&i)[NV], double (&xi)[NV], const bool (&on)[NV], double tol,
                                       double viol_tol, uint64_t seed, uint64_t grestart, int64_t t, P1Visit (&V)[NV]) {
    double viol[NV], ss[NV], es[NV], sp[NV];
    uint32_t it[NV], itp[NV];
    bool pending[NV];
    const double guard = 1e-9 * (1.0 - r);
#pragma unroll
    for (int k = 0; k < NV; k++) {
        V[k].visited = true; V[k].moved = false; V[k].status = 0;
        viol[k] = fabs(xi[k] * (p * xi[k] + q) + r);
        ss[k] = -tol; es[k] = on[k] ? viol[k] - viol_tol : -tol; sp[k] = 0.0;
        it[k] = 0; itp[k] = 0; pending[k] = false;
    }
    for (;;) {
        bool any = false;
#pragma unroll
        for (int k = 0; k < NV; k++) any = any || (es[k] - ss[k] > tol);
        if (!any) break;
#pragma unroll
        for (int k = 0; k < NV; k++) {
            const bool act = es[k] - ss[k] > tol;
            const double s = (ss[k] + es[k]) / 2.0;
            const uint32_t itb = it[k];
            it[k] += act ? 1u : 0u;
            bool nonempty = s > 0.0;
            if (__builtin_expect(act && !(fabs(s) > guard), 0)) {       // within rounding of the threshold: the reference's discriminants
                const double D1 = 0.0 - 4.0 * p * (r - s);
                const double D2 = 0.0 - 4.0 * (-p) * (-r - s);
                nonempty = D1 > 0.0 && (D2 < 0.0 || D2 < D1);
            }
            const bool up = act && nonempty, dn = act && !nonempty;
            ss[k] = dn ? s : ss[k];
            es[k] = up ? s : es[k];
            sp[k] = up ? s : sp[k];
            itp[k] = up ? itb : itp[k];
            pending[k] = pending[k] || up;
        }
    }
#pragma unroll
    for (int k = 0; k < NV; k++) {
        // (the expressions of p1_band_visit's tail, evaluated whether or not a step succeeded)
        const double spk = sp[k];
        const double r1 = r - spk, r2 = -r - spk;
        const double rsA = r1 - 0.0, rsB = r2 - 0.0;
        const double DA = q * q - 4.0 * p * rsA;
        const bool hasA = DA >= 0.0;
        const double rDA = sqrt(hasA ? DA : 0.0);
        const double pB = -p, qB = -q;
        const double DB = qB * qB - 4.0 * pB * rsB;
        const bool twoB = DB >= 0.0;
        const double rDB = sqrt(twoB ? DB : 0.0);
        const double tA = rDA / (2.0 * p), tB = rDB / (2.0 * p);
        const double alo = -tA, ahi = tA;
        const double bhi0 = twoB ? -tB : QM_INF;
        const double blo1 = tB;
        const double l0 = alo > -QM_INF ? alo : -QM_INF, h0 = ahi < bhi0 ? ahi : bhi0;
        const bool ok0 = hasA && l0 <= h0;
        const double l1 = alo > blo1 ? alo : blo1, h1 = ahi < QM_INF ? ahi : QM_INF;
        const bool ok1 = hasA && twoB && l1 <= h1;
        int n = (ok0 ? 1 : 0) + (ok1 ? 1 : 0);
        double lo0 = ok0 ? l0 : (ok1 ? l1 : 0.0), hi0 = ok0 ? h0 : (ok1 ? h1 : 0.0);
        double lo1 = (ok0 && ok1) ? l1 : 0.0, hi1 = (ok0 && ok1) ? h1 : 0.0;
        const bool same = n == 2 && lo0 == lo1 && hi0 == hi1;
        const bool m01 = !same && n == 2 && hi0 == lo1;
        const bool m10 = !same && !m01 && n == 2 && hi1 == lo0;
        hi0 = m01 ? hi1 : hi0;
        lo0 = m10 ? lo1 : lo0;
        n = same ? 0 : ((m01 || m10) ? 1 : n);
        const bool k0 = n >= 1 && lo0 != hi0 && hi0 != QM_INF;
        const bool k1 = n >= 2 && lo1 != hi1 && hi1 != QM_INF;
        const bool sw = k0 && k1 && lo1 < lo0;
        const double a0 = sw ? lo1 : lo0, b0 = sw ? hi1 : hi0, a1 = sw ? lo0 : lo1, b1 = sw ? hi0 : hi1;
        const int cn = (k0 ? 1 : 0) + (k1 ? 1 : 0);
        const double c0lo = k0 ? a0 : a1, c0hi = k0 ? b0 : b1;
        const U4 rnd = cd_draw(seed, grestart, (uint32_t)i[k], (uint32_t)t, itp[k]);
        const int c = draw_choice(rnd, cn);
        const double lo = (c == 1) ? a1 : c0lo, hi = (c == 1) ? b1 : c0hi;
        const bool unb = __builtin_isinf(lo) || __builtin_isinf(hi);
        const double xn = draw_uniform(rnd, lo, hi);
        const int got = (cn > 0) ? (unb ? -1 : 1) : 0;
        double new_xi = xi[k], new_viol = viol[k];
        if (pending[k]) {
            if (got == 1) { new_xi = xn; new_viol = spk; }
            else if (got < 0) V[k].status = got;
        }
        if (on[k] && new_viol < viol[k]) { xi[k] = new_xi; V[k].moved = true; }
        V[k].vafter = fabs((p * xi[k] + q) * xi[k] + r);
    }
}

template <int MAXC>
__device__ inline void p1_sep_visit(const DevProblem &P, int64_t i, double &xi, double tol, double viol_tol, uint64_t seed,
                                    uint64_t grestart, int64_t t, P1Visit &V) {
    V.visited = false; V.moved = false; V.vafter = -QM_INF; V.status = 0;
    const int e0 = P.cptr[i], mf = P.cptr[i + 1] - e0;
    if (mf == 0) { V.status = -3; return; }
    double cp[MAXC], cq[MAXC], cr[MAXC];
    int crel[MAXC];
#pragma unroll
    for (int k = 0; k < MAXC; k++) {
        bool ok = k < mf;
        cp[k] = ok ? P.cp[e0 + k] : 0.0; cq[k] = ok ? P.cq[e0 + k] : 0.0;
        cr[k] = ok ? P.cr[e0 + k] : 0.0; crel[k] = ok ? P.crel[e0 + k] : RELOP_LE;
    }
    p1_sep_visit_core<MAXC>(mf, cp, cq, cr, crel, i, xi, tol, viol_tol, seed, grestart, t, V);
}

}  // namespace qcqpmi
