// Device-side data model of the engine (gfx950 only).
//
// Candidate points ("population": restarts of improve(), samples of suggest()) live in HBM in
// TILE-MAJOR layout:  X[tile][j][16]  -- tile = r / 16, 16 consecutive candidates side by side,
// j = 0..n16-1 (n padded to a multiple of 16 with zeros).  One row j of a tile is 128 contiguous
// bytes and four consecutive rows are exactly the B operand of one v_mfma_f64_16x16x4_f64
// (lane l <- X[4kk + (l>>4)][l&15]), so a wave reads it with one fully coalesced 512-byte load.
//
// The dense objective matrix P0 is kept twice: row-major padded (n16 x n16) and PRE-PACKED in MFMA
// A-fragment order  Apack[b][kk][l] = P0[16b + (l&15)][4kk + (l>>4)]  so that the A operand of
// every MFMA is also one coalesced 512-byte wave load.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qcqpmi {

struct DevProblem {
    int64_t n, n16, NB, KS, m;
    const double *Apack;  // [NB][KS][64]
    const double *Apack2; // [NB][KS/2][64][2]: the same fragments, two consecutive k-steps per lane (16-byte loads)
    const double *P0;     // [n16][n16] row-major, zero padded
    const double *q0;     // [n16]
    double r0;
    // separable constraints (each touches exactly one coordinate): per-coordinate lists
    int sep;              // 1 if every constraint is separable
    int maxc;             // max #constraints on one coordinate
    const int *cptr;      // [n16 + 1]
    const double *cp, *cq, *cr;  // per entry: p = P_k[i,i], q = q_k[i], r = r_k
    const int *crel;      // per entry relop
    const int *cidx;      // per entry: constraint index k (1-based)
    // constraint classes: coordinates with bit-identical constraint lists share a class
    int K;                // number of classes
    const int *krep;      // [K] a representative coordinate of each class
    const int *cls;       // [n16] class of each coordinate
    const double *rcp2d;  // [n16] 1 / (2 P0[i,i]) where P0[i,i] > 0, else 0
    // general constraints (COO of P_k, dense q_k) -- used by eval when !sep
    const int64_t *gptr;  // [m + 1] entry ranges
    const int *gi, *gj;   // entry row / col
    const double *gv;     // entry value
    const double *gq;     // [m][n16]
    const double *gr;     // [m]
    const int *grel;      // [m]
};

struct EvalArgs {
    DevProblem P;
    const double *X;   // tile-major population
    int64_t R;         // live candidates
    double *f0;        // [Rpad]
    double *maxviol;   // [Rpad]
    double *F;         // optional (m+1) x Rpad row-major, or nullptr
    int64_t Rpad;
    const double *planes;   // optional: x'P0x already computed as nplanes partial sums [nplanes][Rpad]
    int nplanes;            // (dense_products_kernel<2>); the kernel then only adds q0'x + r0 and the constraints
};

// debug trace of the profiled phase-2 kernel: words appended to the prof buffer (8 waves x 64 entries x 4 stamps)
#define QCQPMI_TRACE_WORDS 2048

struct CdArgs {
    DevProblem P;
    double *X;
    int64_t R;
    const double *f0cur;    // [Rpad] objective at phase-2 start
    const double *slack;    // [Rpad] max violation at phase-2 start (the fixed slack, qcqp.py:157)
    int64_t num_iters;
    double viol_tol, tol;
    uint64_t seed, first_index;
    int64_t *visits;        // [Rpad] coordinate visits performed (qcqp.py:162 loop bodies)
    int64_t *accepted;      // [Rpad] accepted coordinate updates
    int64_t *sweeps;        // [Rpad] sweeps started
    int *status;            // [Rpad] 0 ok, <0 where the reference would raise
    uint8_t *flag;          // [Rpad] in: run this restart (phase 2) / out: phase-1 feasible
    long long *prof;        // optional [tiles][8] cycle counters of wave 0 (debug), or nullptr
    int dbg;                // debug switches for timing experiments (results invalid when != 0)
    double *f0out;          // optional [Rpad]: phase 2 writes the tracked objective of the restarts it ran ...
    double *mvout;          // ... and their max violation at the final point (saves the evaluation pass afterwards)
};

}  // namespace qcqpmi
