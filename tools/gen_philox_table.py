#!/usr/bin/env python
"""Fixture g11: a table of Philox4x32-10 outputs and of the keyed draws derived from them, computed by an implementation that
shares NOTHING with the repository's C / HIP code -- arbitrary-precision Python integers -- and checked against the published
known-answer vectors of Random123 (philox4x32_10 in kat_vectors) before anything is written.  Pins the KEYED branch of the
oracle's random numbers (oracle/qcqp_oracle.c: orc_philox4x32, keyed_draw, orc_rng_uniform / orc_rng_choice, orc_keyed_normal),
on which every GPU parity test of phase 1 rests and which no fixture of the reference can pin (the reference draws from NumPy's
MT19937; the counter-based stream is this build's documented deviation 1).      usage: python tools/gen_philox_table.py"""
import math
import os

import numpy as np

M0, M1, W0, W1, MASK = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF


def philox4x32_10(ctr, key):
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c3 ^ k1) & MASK, p0 & MASK
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c0, c1, c2, c3


KAT = [   # Random123 kat_vectors, philox4x32 10 rounds: counter, key, expected
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def u53(a, b):
    return ((a >> 5) * 67108864.0 + (b >> 6)) / 9007199254740992.0


def main():
    for ctr, key, want in KAT:
        got = philox4x32_10(ctr, key)
        assert got == want, (ctr, key, [hex(v) for v in got])
    rs = np.random.RandomState(20260929)
    N = 96
    seed = rs.randint(0, 2 ** 62, size=N, dtype=np.int64).astype(np.uint64)
    seed[:4] = [0, 1, 2 ** 32 - 1, 2 ** 32 + 5]
    restart = rs.randint(0, 2 ** 40, size=N, dtype=np.int64).astype(np.uint64)
    restart[:4] = [0, 1, 2 ** 32 - 1, 2 ** 32]             # (the CD draws use the low 32 bits only; the normals fold the high ones into the key)
    coord = rs.randint(0, 5000, size=N).astype(np.uint32)
    sweep = rs.randint(0, 1000, size=N).astype(np.uint32)
    sweep[::2] |= np.uint32(0x80000000)                      # phase-2 tag
    it = rs.randint(0, 40, size=N).astype(np.uint32)
    k = rs.randint(2, 9, size=N).astype(np.int64)
    lo = rs.randn(N)
    hi = lo + rs.rand(N) * 3.0
    elem = rs.randint(0, 2 ** 34, size=N, dtype=np.int64).astype(np.uint64)
    words = np.zeros((N, 4), dtype=np.uint32)
    uni = np.zeros(N)
    cho = np.zeros(N, dtype=np.int64)
    nwords = np.zeros((N, 4), dtype=np.uint32)
    nrm = np.zeros(N)
    for j in range(N):
        sd, rr = int(seed[j]), int(restart[j])
        o = philox4x32_10((int(coord[j]), int(sweep[j]), int(it[j]), rr & MASK), (sd & MASK, (sd >> 32) & MASK))      # cd_draw (philox.h)
        words[j] = o
        uni[j] = lo[j] + (hi[j] - lo[j]) * u53(o[0], o[1])                                                              # draw_uniform
        cho[j] = (o[2] * int(k[j])) >> 32                                                                                # draw_choice
        e = int(elem[j])
        w = philox4x32_10(((e >> 1) & MASK, (e >> 33) & MASK, 0xA5A50000, rr & MASK), (sd & MASK, ((sd >> 32) ^ (rr >> 32)) & MASK))   # keyed_normal
        nwords[j] = w
        u1 = (((w[0] >> 5) * 67108864.0 + (w[1] >> 6)) + 0.5) / 9007199254740992.0
        u2 = u53(w[2], w[3])
        rad, ang = math.sqrt(-2.0 * math.log(u1)), 6.283185307179586476925286766559 * u2
        nrm[j] = rad * math.sin(ang) if (e & 1) else rad * math.cos(ang)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'g11_philox_keyed.npz')
    np.savez_compressed(out, seed=seed, restart=restart, coord=coord, sweep=sweep, it=it, k=k, lo=lo, hi=hi, elem=elem, words=words,
                        uniform=uni, choice=cho, normal_words=nwords, normal=nrm,
                        kat_ctr=np.array([c for c, _, _ in KAT], dtype=np.uint32), kat_key=np.array([kk for _, kk, _ in KAT], dtype=np.uint32),
                        kat_out=np.array([w for _, _, w in KAT], dtype=np.uint32))
    print('wrote', out, os.path.getsize(out), 'bytes')


if __name__ == '__main__':
    main()
