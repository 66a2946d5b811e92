// SDP relaxation for the unit-diagonal family (replaces the cvxpy call of
// solve_sdr, qcqp.py:72-97, for problems whose constraints are x_i^2 = d_i -- Boolean least squares,
// MAXCUT, partitioning):
//        minimise <C, X>   s.t.  X_ii = 1 (i = 0..N-1),  X >= 0 (PSD),      N = n + 1 (homogenised)
// solved in the Burer-Monteiro form X = V V^T, V in R^{N x K} with unit rows, by the MIXING METHOD
// (Wang & Kolter 2017): cyclic exact block-coordinate minimisation
//        g_i = sum_{j != i} C_ij v_j,      v_i <- -g_i / ||g_i||,
// which decreases <C, V V^T> monotonically and converges to the SDP optimum for K > sqrt(2N).
// Third-party solver in the reference => parity is unpinned by construction; validated by optimality
// conditions (dual certificate lambda_min(C + diag(y)) >= -eps, tests/test_gpu_api.py) and brute force.
//
// The method is a strictly sequential chain of N small matrix-vector products per sweep.  Layout on the
// chip: COMPONENT-SLICED -- workgroup kk (one wavefront, 64 workgroups on 64 CUs) owns component kk of
// every row, V[:, kk] = N doubles in its LDS, so g_i[kk] = sum_j C_ij V[j][kk] is workgroup-local: the
// only traffic per coordinate is row i of C (the same 8 N bytes for everybody, prefetched one step
// ahead into registers) and ONE all-reduce of 64 numbers for ||g_i|| (and v_i . g_i for the objective):
// every workgroup publishes its contribution, bumps an arrival counter and spins until all 64 have
// arrived, then sums the 64 values in a fixed order -- every workgroup computes bit-identical norms, and
// the result does not depend on timing.  Launched cooperatively (co-residency is guaranteed); the spin
// is bounded and raises an abort flag instead of hanging.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qcqpmi {

constexpr int SDR_K = 64;          // rank of the factor = number of workgroups
constexpr int SDR_NMAX = 4096;     // N <= 64 lanes x 64 registers per row of C
constexpr int SDR_NPL = SDR_NMAX / 64;

struct SdrWork {                   // global scratch of the all-reduce
    double vals[2][2][SDR_K];      // [step parity][g^2 | g v_old][workgroup]
    unsigned arrive;
    int abort;
};

__device__ inline double sdr_wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// hist[0] = objective of the initial V, hist[t] = objective after sweep t (tracked by the exact decrease
// of every update), hist[sweeps + 1] = objective recomputed from scratch at the end.
__global__ __launch_bounds__(64) void sdr_mixing_kernel(const double *__restrict__ C, double *__restrict__ V, int N,
                                                        int max_sweeps, double tol, double *__restrict__ hist,
                                                        int *__restrict__ sweeps_done, SdrWork *__restrict__ wk) {
    extern __shared__ double sdr_lds[];
    double *vs = sdr_lds, *dg = sdr_lds + N;   // V[:, kk] and diag(C)
    const int lane = threadIdx.x, kk = blockIdx.x;
    for (int j = lane; j < N; j += 64) { vs[j] = V[(int64_t)j * SDR_K + kk]; dg[j] = C[(int64_t)j * N + j]; }
    double ra[SDR_NPL], rb[SDR_NPL];
#pragma unroll
    for (int q = 0; q < SDR_NPL; q++) {
        const int jj = q * 64 + lane;
        ra[q] = (q * 64 < N) ? C[jj < N ? jj : N - 1] : 0.0;
        rb[q] = 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    double f = 0.0, facc = 0.0, dsweep = 0.0;
    int sweeps = 0;
    unsigned stepno = 0;
    bool aborted = false, update = false;

    // one coordinate: `cur` holds row i of C, row i+1 (cyclically) is fetched into `nxt`
    auto step = [&](int i, double (&cur)[SDR_NPL], double (&nxt)[SDR_NPL]) {
        const double *Cn = C + (int64_t)((i + 1 < N) ? i + 1 : 0) * N;
#pragma unroll
        for (int q = 0; q < SDR_NPL; q++) {
            const int jj = q * 64 + lane;
            if (q * 64 < N) nxt[q] = Cn[jj < N ? jj : N - 1];   // in flight behind the all-reduce
        }
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int q = 0; q < SDR_NPL; q += 4) {
            if (q * 64 < N) {
                const int j0 = q * 64 + lane;
                a0 = __builtin_fma(j0 < N ? cur[q] : 0.0, vs[j0 < N ? j0 : 0], a0);
                a1 = __builtin_fma(j0 + 64 < N ? cur[q + 1] : 0.0, vs[j0 + 64 < N ? j0 + 64 : 0], a1);
                a2 = __builtin_fma(j0 + 128 < N ? cur[q + 2] : 0.0, vs[j0 + 128 < N ? j0 + 128 : 0], a2);
                a3 = __builtin_fma(j0 + 192 < N ? cur[q + 3] : 0.0, vs[j0 + 192 < N ? j0 + 192 : 0], a3);
            }
        }
        const double vold = vs[i], cii = dg[i];
        const double g = sdr_wave_sum((a0 + a1) + (a2 + a3)) - cii * vold;   // j != i
        // ---- all-reduce over the 64 components
        const unsigned par = stepno & 1u;
        if (lane == 0) {
            __hip_atomic_store(&wk->vals[par][0][kk], g * g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&wk->vals[par][1][kk], g * vold, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&wk->arrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned target = (unsigned)SDR_K * (stepno + 1u);
        unsigned spins = 0;
        for (;;) {
            const unsigned seen = __hip_atomic_load(&wk->arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)(seen - target) >= 0) break;
            if (++spins > (1u << 24) || __hip_atomic_load(&wk->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(&wk->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                aborted = true;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        const double n2 = sdr_wave_sum(__hip_atomic_load(&wk->vals[par][0][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const double gv = sdr_wave_sum(__hip_atomic_load(&wk->vals[par][1][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (update) {
            const double nrm = sqrt(n2);
            if (nrm > 0.0) {
                if (lane == 0) vs[i] = -g / nrm;
                dsweep += -2.0 * (nrm + gv);     // exact change of <C, V V^T>
            }
        } else {
            facc += gv + cii;                    // v_i . g_i + C_ii  (unit rows)
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        stepno++;
    };

    // pass -1: objective only; passes 0..: updates; final pass: objective only
    for (int pass = -1; pass <= max_sweeps && !aborted; pass++) {
        update = pass >= 0 && pass < max_sweeps;
        facc = 0.0; dsweep = 0.0;
        for (int i = 0; i < N && !aborted; i++) {
            if ((stepno & 1u) == 0u) step(i, ra, rb); else step(i, rb, ra);
        }
        bool conv = false;
        if (!update) {
            f = facc;
            if (kk == 0 && lane == 0) hist[pass < 0 ? 0 : sweeps + 1] = f;
        } else {
            f += dsweep;
            sweeps++;
            if (kk == 0 && lane == 0) hist[sweeps] = f;
            conv = fabs(dsweep) <= tol * (1.0 + fabs(f));
        }
        if (pass == max_sweeps) break;
        if (conv) pass = max_sweeps - 1;   // converged: jump to the final objective pass
    }
    for (int j = lane; j < N; j += 64) V[(int64_t)j * SDR_K + kk] = vs[j];
    if (kk == 0 && lane == 0) *sweeps_done = aborted ? -1 : sweeps;
}

}  // namespace qcqpmi
