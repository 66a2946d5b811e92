// Counter-based RNG shared by every device kernel (and restated bit-for-bit by the oracle's
// ORC_RNG_KEYED mode): Philox4x32-10 keyed on the run seed, counter = (coordinate, sweep tag,
// bisection iteration, GLOBAL restart/sample index).  Keying on the global index makes every
// result independent of how restarts are sharded over GPUs.
//
// Replaces the reference's use of the global NumPy MT19937 stream inside the improve path
// (utilities.py:266-267, 288) and np.random.randn / multivariate_normal in suggest
// (qcqp.py:382, 396).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qcqpmi {

struct U4 { uint32_t x, y, z, w; };

__host__ __device__ inline U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                             uint32_t k0, uint32_t k1) {
#pragma unroll
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
    return U4{c0, c1, c2, c3};
}

__host__ __device__ inline double u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

// Draw for one onevar_qcqp call of coordinate descent.
__host__ __device__ inline U4 cd_draw(uint64_t seed, uint64_t restart, uint32_t coord,
                                      uint32_t sweep_tag, uint32_t iter) {
    return philox4x32_10(coord, sweep_tag, iter, (uint32_t)restart, (uint32_t)seed,
                         (uint32_t)(seed >> 32));
}

__host__ __device__ inline double draw_uniform(const U4 &o, double lo, double hi) {
    return lo + (hi - lo) * u53(o.x, o.y);
}

__host__ __device__ inline int draw_choice(const U4 &o, int k) {
    return k <= 1 ? 0 : (int)(((uint64_t)o.z * (uint64_t)k) >> 32);
}

// Box-Muller standard normal: element `elem` of restart/sample `restart` (stream tag 0xA5A50000).
__host__ __device__ inline double keyed_normal(uint64_t seed, uint64_t restart, uint64_t elem) {
    U4 o = philox4x32_10((uint32_t)(elem >> 1), (uint32_t)(elem >> 33), 0xA5A50000u,
                         (uint32_t)restart, (uint32_t)seed,
                         (uint32_t)(seed >> 32) ^ (uint32_t)(restart >> 32));
    double u1 = (((double)(o.x >> 5) * 67108864.0 + (double)(o.y >> 6)) + 0.5) /
                9007199254740992.0;
    double u2 = u53(o.z, o.w);
    double rad = sqrt(-2.0 * log(u1));
    double ang = 6.283185307179586476925286766559 * u2;
    return (elem & 1) ? rad * sin(ang) : rad * cos(ang);
}

}  // namespace qcqpmi
