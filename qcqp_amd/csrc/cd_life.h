// Lifecycle kernel, second generation (round 5): interface between capi.hip and cd_life.hip (own translation unit).
//
// improve_coord_descent (qcqp.py:181-192) for a QUEUE of restarts -- suggest(RANDOM) (qcqp.py:381-382) or resident start
// points, phase 1 (qcqp.py:101-149), the gate (qcqp.py:189), phase 2 (qcqp.py:152-178), objective and max violation of the
// result -- inside ONE persistent launch, like the lifecycle mode of cd_phase2_qs_kernel (cd_queue.h), rebuilt around what
// round 5 measured (tools/ubench/ubench5.hip): the fp64 matrix pipes of all four SIMDs of a CU sustain their data-sheet
// rate (the "47 TFLOP/s power limit" of rounds 1-4 was the code the compiler made of that microbenchmark), so the kernel
// that kept ONE sequential chain per CU on the critical path (6.4 k cycles per block of 16 coordinates against 4.2 k of
// matrix work) left a third of the chip idle.  Here
//   * a workgroup is FOUR waves -- one chain wave + three multiplying waves -- and keeps its X tile in a private tile of
//     global memory (L2) instead of LDS: only the four blocks committed last live in LDS (a ring the multiplying waves
//     refresh their register-resident B operands from).  35 KB of LDS and 256 registers per wave: TWO workgroups per CU,
//     each SIMD 1-3 carries two independent product streams, SIMD 0 two chains -- the matrix pipes, not a chain, set the pace;
//     the column build (normals, phase 1) of one workgroup runs under the products of its neighbour;
//   * roles are dealt by the SIMD a wave landed on (HW_ID), not by its index: the dispatcher rotates the waves of a
//     four-wave workgroup over the SIMDs;
//   * 1024 < n <= 2304 (MAXCUT n = 2000 of BASELINE.json configs[2]): eight waves, seven multiplying waves split the
//     contraction, one workgroup per CU;
//   * n need not be a multiple of 16 (zero rows / columns up to n16; the padded coordinates are never visited);
//   * three step kinds: the mirrored band of p x^2 + r == 0 on a positive diagonal (Boolean least squares), any single
//     class with at most two intervals on a positive diagonal (box / disc / one-sided), and a ZERO diagonal (MAXCUT: the
//     scalar objective is linear, the minimiser an end point of the feasible set).
// Per restart the arithmetic is that of the serial path (cd_phase1_sep.h, the blocked Gauss-Seidel step of cd_phase2_q.h,
// near-ties replayed in the reference's arithmetic through onevar_minimise); results do not depend on the slot, the
// workgroup, the episode boundaries or the number of populations in the launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "cd_queue.h"

namespace qcqpmi {

// step kinds.  GENK / LINK (round 6): GEN / LIN for problems with SEVERAL constraint classes (coordinates with different
// constraint lists: up to four classes) and up to two constraints per coordinate -- the slots keep a feasible set per class, the
// chain looks its columns' classes up per block
enum { L2_KIND_BAND = 0, L2_KIND_GEN = 1, L2_KIND_LIN = 2, L2_KIND_GENK = 3, L2_KIND_LINK = 4 };

struct CdLife2Args {
    DevProblem P;
    CdBatch b;                   // outputs of all Rtotal restarts (and, for generate = 0, the start points)
    int64_t num_iters;
    double tol;
    const CdLife *life;          // parameters of the run in DEVICE memory
    double *scratch;             // [workgroups][tiles per workgroup][n16][16]: the workgroups' X tiles
    const double *Dpack;         // [NB][256]: strictly upper triangle of the diagonal blocks of P0 (zeros elsewhere)
    const double *Spack;         // [NB][48]: q0 / 2, 1 / P0[i,i] (0 where the diagonal is 0), P0[i,i] of the block's coordinates
    int *abort;                  // [0] set by a wave whose wait ran into the watchdog (a bug, never the data): the launch unwinds
    int *cuslot;                 // [4096] zeroed before the launch, or NULL: arrival counter per compute unit (key: XCC, SE, CU of HW_ID) -- the
                                 // second four-wave workgroup of a CU turns its roles by two SIMDs, so that the two chains of a CU do not share a SIMD
    double fbound;               // sum |P0| + sum |q0| + |r0|: scale of the objective for the near-tie test of the linear kind
    // factored objective (P0 = L L^T, L n x r; cd_life2_pack_factor): fragments of L for the products / the updates of Y = L^T X;
    // RB = blocks of 16 rows of Y (0: not factored)
    const double *Gpack, *Upack;
    int RB;
    int rotmode;                 // with cuslot: 1 = the second workgroup of a CU turns ALL roles by two SIMDs (chains on different SIMDs: measured slower),
                                 // 2 = it turns only its multiplying roles by one among SIMDs 1-3 (the wave with six blocks of Y of either workgroup on its own SIMD)
    int nclass;                  // multi-class kinds: classes among the real coordinates (<= 4; class k = DevProblem::krep[k])
    int dbg;                     // timing experiments (results INVALID when != 0): 1 = every block row reads the fragments of rows 0..7 (an L2-resident stream)
};

// does the kernel take this problem?  nmw / cs / kind: the instantiation (multiplying waves 3 | 7, chain share, step kind)
// factor_rb: blocks of 16 rows of an objective factor the caller WILL hand over (0: none) -- the factored instantiation keeps Y, not X,
// in registers, so with a factor the kernel also takes 2304 < n <= 4096 (and prefers three multiplying waves from n = 1040 on)
bool cd_life2_config(const DevProblem &P, int Kreal, int objclass, bool symcls, int factor_rb, int *nmw, int *cs, int *kind);
size_t cd_life2_lds_bytes(int nmw, int cs, int tiles, int lr, int kind);
bool cd_life2_factor_ok(const DevProblem &P, int64_t r);
int cd_life2_pack_factor(const double *Lrow, double *Gpack, double *Upack, int NB, int RB, hipStream_t st);
int cd_life2_max_wgs(int nmw, int cus, int tiles);
// tiles of 16 slots per workgroup (1 | 2) for a run of `restarts` restarts; requested: 0 = automatic
int cd_life2_tiles(const DevProblem &P, int nmw, int cs, int64_t restarts, int cus, int requested, int lr);
// masked diagonal blocks + per-block scalars (device, once per problem)
int cd_life2_pack(const DevProblem &P, double *Dpack, double *Spack, hipStream_t st);
int cd_life2_launch(const CdLife2Args &a, int nmw, int cs, int kind, int tiles, int wgs, hipStream_t st);
const char *cd_life2_name(int nmw, int kind, int tiles, int lr);

}  // namespace qcqpmi
