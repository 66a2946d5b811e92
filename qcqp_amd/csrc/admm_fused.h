// Fused, persistent consensus-ADMM iteration (round 3): improve_admm (qcqp.py:254-285) -- phase 1 (qcqp.py:195-212),
// `better`, phase 2 (qcqp.py:215-251), `better` -- for a tile of 16 restarts inside ONE kernel, no host in the loop.
// Interface between capi_admm.inc (host side, in the big translation unit) and admm_fused.hip (own translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qcqpmi {

constexpr int AF_MAXC = 16;         // workgroups that may share one tile of restarts
constexpr int AF_MAXRP = 8;         // columns of a reduced basis

struct AdmmFusedArgs {
    // ---- problem (reduced bases, diagonal P0)
    int n, n16, m, rp, Mh16;        // Mh16 = m rp padded to a multiple of 16
    int R, ntiles;
    const double *WTpk;             // [Mh16/16][n16/4][64]  A fragments of W^T   (ZQ = W^T z)
    const double *Wpk;              // [n16/16][Mh16/4][64]  A fragments of W     (T = W d)
    const double *lam, *qhat;       // [m][rp]
    const double *rk, *slo, *ehi;   // [m]
    const int *relop;               // [m]
    const double *q0, *pdiag, *dinv;   // [n16]: linear term, P0_ii, 1 / (2 (P0_ii + rho m))
    double r0, rho;
    int qzero;                      // every qhat is zero (constraints without linear terms): shorter secular function
    // ---- run parameters (improve_admm's arguments)
    int phase1, num_iters;
    double tol, viol_lim, sec_tol;
    // ---- launch geometry
    int C;                          // workgroups per tile: rows of z (blocks of 16) and constraints are split C ways
    int nt;                         // threads per workgroup: 512 (eight waves, one workgroup per CU: the default) or 256 (four waves,
                                    // two workgroups of different tiles per CU)
    int G;                          // resident clusters; cluster g walks tiles g, g + G, ...
    // ---- population, tile-major [ntiles][n16][16]: in = x0, out = improve_admm's result
    double *X;
    double *BEST;                   // work: bestx of phase 2, same layout
    // ---- exchange buffers of the clusters (global memory, agent-scope atomics), indexed by tile
    double *xb1;                    // [ntiles][C][Mh16 + 2][16]: partial W^T z of every member + partial ||dz||^2 and f0
    double *xb2;                    // [ntiles][Mh16 + C][16]: operand rows d (disjoint per member) + partial max violations
    unsigned *flags;                // [ntiles][2][C] sequence numbers (zeroed before the launch)
    int *abort_flag;                // set when a spin wait times out (the host reports an error instead of hanging)
    // ---- outputs, R entries each
    int64_t *iters1, *iters2;
    double *f0_out, *mv_out;
    long long *prof;                // optional: 16 cycle counters of member 0 of the first tile (s_memtime per stage), or nullptr
};

// dynamic LDS the kernel needs for this geometry (0 if it does not fit 160 KB)
size_t admm_fused_lds_bytes(const AdmmFusedArgs &a);
// resident clusters the device can hold for this geometry (cooperative launch limit / C), at least 1
int admm_fused_max_clusters(const AdmmFusedArgs &a, int device);
// cooperative launch on `st`; returns a hipError_t
int admm_fused_launch(const AdmmFusedArgs &a, hipStream_t st);

}  // namespace qcqpmi
