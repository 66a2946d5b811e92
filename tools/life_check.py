#!/usr/bin/env python
"""cd_life_kernel (round 5) against the serial path (qcqpmi_pop_randn + qcqpmi_cd_run per population), the round-4 lifecycle
kernel where it applies and, for a few restarts, the oracle; and its rate.
Usage: python tools/life_check.py family n R K [num_iters] [oracle_restarts]      family: bls | box | maxcut | maxcutw"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcqp_amd import problems  # noqa: E402
from qcqp_amd.engine import Engine, EngineError  # noqa: E402
from qcqp_amd.form import QCQPForm  # noqa: E402


def build(fam, n):
    if fam == 'bls':
        return problems.boolean_least_squares(n, max(4, n // 4), seed=1)[0]
    if fam == 'box':
        return problems.box_least_squares(n, max(4, n // 2), bound=1.0, seed=1)[0]
    if fam == 'maxcut':
        return problems.maxcut(n, 0.5, seed=1)[0]
    if fam == 'maxcutw':
        return problems.maxcut(n, 0.5, seed=1, weighted=True)[0]
    if fam in ('box3', 'ann2', 'lin2', 'cut2'):
        return problems.multi_class(fam, n)
    raise SystemExit('family?')


def main():
    fam = sys.argv[1] if len(sys.argv) > 1 else 'bls'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    R = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    K = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    iters = int(sys.argv[5]) if len(sys.argv) > 5 else 1000
    norc = int(sys.argv[6]) if len(sys.argv) > 6 else 2
    serial = os.environ.get('LIFE_SERIAL', '1') == '1'
    funcs = build(fam, n)
    form = QCQPForm.from_arrays(funcs)
    seed0, first0, fstride = 1000, 7, 100000
    es = Engine(form)
    es.cd_life_version(int(os.environ.get('LIFE_VER', '2')))
    if os.environ.get('LIFE_FACTOR', '0') == '1':          # factored objective (qcqpmi_cd_set_objective_factor)
        from qcqp_amd import lowrank
        P0 = funcs[0][0]
        P0 = P0.toarray() if hasattr(P0, 'toarray') else np.asarray(P0)
        t0 = time.perf_counter()
        Lf = lowrank.objective_factor(P0)
        print('objective factor:', None if Lf is None else Lf.shape, '%.2f s' % (time.perf_counter() - t0), flush=True)
        es.cd_set_objective_factor(Lf)
    dbg = int(os.environ.get('LIFE_DBG', '0'))
    if dbg:
        es.L.qcqpmi_debug_profile(es.h, dbg << 4, None)
    o = None
    for rep in range(3):
        t0 = time.perf_counter()
        o = es.cd_stream_run(K, R, num_iters=iters, seed=seed0, seed_stride=1, first_index=first0, first_stride=fstride)
        dt = time.perf_counter() - t0
        ms = es.kernel_ms(2)
        sweeps = float(o['visits2'].sum()) / n
        print('%s n=%d R=%d K=%d run %d: wall %.2f ms, kernel %.3f ms, %.3e restart-sweeps/s, frac %.3f; sweeps per restart %.2f; feasible %d' % (
            fam, n, R, K, rep, 1e3 * dt, ms, sweeps / (ms * 1e-3), sweeps * 2.0 * n * n / (ms * 1e-3) / 78.6e12, sweeps / (K * R),
            int(o['ran_phase2'].sum())), flush=True)
    print('kernel:', es.last_cd_kernel(), flush=True)
    if os.environ.get('LIFE_PROF', '0') == '1':
        import ctypes as C
        es.L.qcqpmi_debug_profile(es.h, 1 | (dbg << 4), None)
        es.cd_stream_run(K, R, num_iters=iters, seed=seed0, seed_stride=1, first_index=first0, first_stride=fstride)
        pr = np.zeros(24, dtype=np.int64)
        es.L.qcqpmi_debug_life_profile(es.h, pr.ctypes.data_as(C.POINTER(C.c_int64)))
        es.L.qcqpmi_debug_profile(es.h, dbg << 4, None)
        print('profile (%.3f ms): column build %.1f %% of the workgroups\' time (normals %.1f %%), roles %.1f %%, rest %.1f %%; %d episodes, %d columns; workgroups with evenly spread waves: %d' % (
            es.kernel_ms(2), 100.0 * pr[0] / max(pr[1], 1), 100.0 * pr[4] / max(pr[1], 1), 100.0 * pr[5] / max(pr[1], 1),
            100.0 * (pr[1] - pr[0] - pr[5]) / max(pr[1], 1), pr[2], pr[3], pr[6]), flush=True)
        print('  column build: normals %.1f %%, phase-1 sweeps %.1f %%, slack + gate + set + padding %.1f %% of the workgroups\' time' % (
            100.0 * pr[4] / max(pr[1], 1), 100.0 * (pr[17] - pr[4]) / max(pr[1], 1), 100.0 * (pr[0] - pr[17]) / max(pr[1], 1)), flush=True)
        nwg = max(int(pr[6]), 1)
        print('  longest workgroup %.3e ticks = %.3f GHz tick rate; the average workgroup was alive %.1f %% of the launch' % (
            pr[16], pr[16] / (es.kernel_ms(2) * 1e-3) / 1e9, 100.0 * pr[1] / nwg / max(pr[16], 1)), flush=True)
        ni = max(int(pr[11]), 1)
        print('  per block interval (s_memtime ticks): roles %.1f, chain waits for partials %.1f, multiplying wave 0 waits: commit %.1f, slot %.1f; intervals %d, with a near-tie replay %d' % (
            pr[5] / ni, pr[8] / ni, pr[9] / ni, pr[10] / ni, pr[11], pr[12]), flush=True)
        print('  per EPISODE (ticks): chain prologue %.0f, after the chain left its loop %.0f; per interval: requests %.1f' % (pr[18] / max(pr[2], 1), pr[20] / max(pr[2], 1), pr[19] / ni), flush=True)
        print('  chain per interval: sum + requests %.1f, 16 steps %.1f, block end + commit %.1f, fix-up + share + staging %.1f' % (
            pr[13] / ni, pr[14] / ni, pr[15] / ni, pr[7] / ni), flush=True)
        print('  factored objective: Y of the starting columns %.1f %% of the workgroups\' time' % (100.0 * pr[21] / max(pr[1], 1)), flush=True)
    X = es.download()
    f0e, mve = es.eval()
    print('reported vs fresh evaluation: rel df0 %.2e, d maxviol %.2e' % (np.max(np.abs(o['f0'] - f0e) / (1 + np.abs(f0e))), np.max(np.abs(o['maxviol'] - mve))), flush=True)
    if serial:
        e = Engine(form)
        worst = 0.0
        for p in range(K):
            e.randn(R, seed=seed0 + p, first_index=first0 + p * fstride)
            outr = e.cd_run(phase1=True, num_iters=iters, seed=seed0 + p, first_index=first0 + p * fstride)
            Xr = e.download()
            best = e.select_best()
            sl = slice(p * R, (p + 1) * R)
            d = np.max(np.abs(X[:, sl] - Xr), axis=0)
            worst = max(worst, d.max())
            same = {k: bool(np.array_equal(o[k][sl], outr[k])) for k in ('sweeps1', 'sweeps2', 'visits2', 'accepted2', 'ran_phase2', 'status1', 'status2')}
            rf = np.max(np.abs(o['f0'][sl] - outr['f0']) / (1 + np.abs(outr['f0'])))
            rv = np.max(np.abs(o['maxviol'][sl] - outr['maxviol']))
            print('population %d vs serial (%s): max |dx| %.2e (restarts off by > 1e-9: %d), counters equal: %s, rel df0 %.2e, d maxviol %.2e; best %d (serial %d)' % (
                p, e.last_cd_kernel(), d.max(), int((d > 1e-9).sum()), same, rf, rv, o['best_index'][p], best[0]), flush=True)
        print('worst |dx| vs serial', worst)
    try:
        e1 = Engine(form)
        e1.cd_life_version(1)
        o1 = e1.cd_stream_run(K, R, num_iters=iters, seed=seed0, seed_stride=1, first_index=first0, first_stride=fstride)
        X1 = e1.download()
        print('vs round-4 lifecycle kernel (%s, %.3f ms): max |dx| %.2e, counters equal %s' % (
            e1.last_cd_kernel(), e1.kernel_ms(2), np.max(np.abs(X - X1)),
            all(np.array_equal(o[k], o1[k]) for k in ('sweeps1', 'sweeps2', 'visits2', 'accepted2', 'ran_phase2'))), flush=True)
    except EngineError as ex:
        print('round-4 lifecycle kernel: refused (%s)' % str(ex)[:80])
    if norc > 0:
        from oracle import oracle as orc
        prob = orc.Problem(funcs)
        e = Engine(form)
        for p in range(min(K, 2)):
            sd, fi = seed0 + p, first0 + p * fstride
            e.randn(R, seed=sd, first_index=fi)
            X0 = e.download()
            for r in list(range(R))[:norc]:
                rng = orc.Rng(orc.RNG_KEYED, sd)
                rng.set_restart(fi + r)
                t0 = time.perf_counter()
                x, s1, s2 = prob.improve_cd(X0[:, r], num_iters=iters, rng=rng)
                k = p * R + r
                print('oracle population %d restart %d (%.1f s): max |dx| %.2e; sweeps1 %d/%d visits2 %d/%d accepted2 %d/%d; f0 %.10g / %.10g' % (
                    p, r, time.perf_counter() - t0, np.max(np.abs(X[:, k] - x)), o['sweeps1'][k], s1[0], o['visits2'][k], s2[1], o['accepted2'][k], s2[2],
                    o['f0'][k], prob.eval(0, x)), flush=True)


if __name__ == '__main__':
    main()
