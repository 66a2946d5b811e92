// Coordinate descent phase 2 (qcqp.py:152-178) with RESTART-LEVEL scheduling (round 3): interface between capi.hip and
// cd_queue.hip (own translation unit).
//
// cd_phase2_q_kernel binds 16 restarts to a workgroup for the whole launch: the workgroup runs until its slowest restart
// has converged (lanes of converged restarts keep multiplying), and the launch until the slowest workgroup has.  Here a
// workgroup owns 16 SLOTS: a restart that has converged is written out at the next sweep boundary and its slot takes the
// next restart from a device-side queue, so the matrix pipes keep working on live columns.  Per restart the arithmetic is
// that of cd_phase2_q_kernel bit for bit (a column's products depend on that column only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace qcqpmi {

struct CdBatch {                 // the restarts a launch works on: one population, or (lifecycle mode) all populations of a run
    double *X;                   // tile-major [tile][n16][16], in / out
    const double *f0cur;         // [Rpad] objective at phase-2 start               (not read in lifecycle mode)
    const double *slack;         // [Rpad] max violation at phase-2 start (the fixed slack of phase 2, qcqp.py:157)   (ditto)
    const uint8_t *flag;         // [Rpad] passed the gate of improve_coord_descent (qcqp.py:189)                    (ditto)
    int64_t *visits, *accepted, *sweeps;
    int *status;
    double *f0out, *mvout;       // objective / max violation of the final point, written for the restarts that ran
    int64_t R;
    uint64_t seed, first_index;
    int *next;                   // queue head: next restart index to hand out (zeroed before the launch)
};

// Lifecycle mode (round 4): the launch runs a restart's WHOLE improve step -- suggest(RANDOM) (keyed normals), phase 1,
// the gate of improve_coord_descent, phase 2, objective and max violation of the result -- for a queue of Rtotal restarts
// that belong to Rtotal / Rpop populations of Rpop restarts each (population p: seed + p seed_stride, global restart
// indices first_index + p first_stride + [0, Rpop)).  No preparation kernels, no second stream, no CU partition: a slot
// that becomes free draws the next restart index and builds the column itself.
struct CdLife {
    int on;
    int generate;                // 1: x0 = keyed normals (suggest RANDOM, qcqp.py:381-382); 0: the columns of b.X
    int phase1;                  // run phase 1 (qcqp.py:186-187)
    int64_t Rtotal, Rpop;
    uint64_t seed, seed_stride, first_index, first_stride;
    double viol_tol;
    int64_t *sweeps1;            // [Rtotal] phase-1 sweeps
    int *status1;                // [Rtotal] phase-1 status
    uint8_t *ran2;               // [Rtotal] passed the gate (qcqp.py:189)
    long long *prof;             // optional [8]: ticks (s_memtime) summed over the workgroups -- 0 column build, 1 whole launch, 2 episodes,
                                 // 3 columns built, 4 the normals' share of 0, 5 the roles (episodes proper); nullptr: off
};

struct CdQueueArgs {
    DevProblem P;
    CdBatch b;
    int64_t num_iters;
    double tol;
    const CdLife *life;          // lifecycle mode: parameters in DEVICE memory (b holds the outputs of all Rtotal restarts); nullptr: off.
                                 // (By value they cost scalar registers for the whole kernel -- the multiplying waves sit exactly at the
                                 // register limit and spill when the spilled scalars take two more VGPRs.)
    int life_on;
};

// LDS bytes of the kernel for this problem (0: does not fit / not eligible)
size_t cd_queue_lds_bytes(const DevProblem &P);
// launches ceil(R / 16) workgroups at most `max_wgs`; cs = blocks of the contraction the chain wave multiplies (0, 2, 4, 6)
int cd_queue_launch(const CdQueueArgs &a, int cs, int max_wgs, hipStream_t st);

}  // namespace qcqpmi
