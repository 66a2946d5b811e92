"""Randomised shake-out of the lifecycle launch (qcqpmi_cd_stream_run) against the serial path (qcqpmi_pop_randn / upload +
qcqpmi_cd_run per population): random n, row counts, population sizes, numbers of populations, sweep limits, with and without
phase 1, generated and uploaded starts (uploaded ones scaled so that part of them fails the gate).  Not a test (runtime): prints
every mismatch.   usage: fuzz_stream.py [cases=40] [seed=0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import problems
from qcqp_amd.engine import Engine, EngineError
from qcqp_amd.form import QCQPForm

COUNTERS = ('sweeps1', 'sweeps2', 'visits2', 'accepted2', 'ran_phase2', 'status1', 'status2')
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(cases):
    n = 16 * int(rs.choice([3, 4, 5, 7, 8, 12, 16, 21, 32, 33, 48, 64]))
    m_rows = max(2, int(n * rs.choice([0.25, 0.5, 1.0, 1.5])))
    R = int(rs.choice([1, 2, 15, 16, 17, 100, 257, 600, 4096 if n <= 256 else 300]))
    K = int(rs.choice([1, 2, 3, 5]))
    iters = int(rs.choice([0, 1, 2, 5, 1000]))
    phase1 = bool(rs.rand() < 0.7)
    generate = bool(rs.rand() < 0.6)
    funcs, _, _ = problems.boolean_least_squares(n, m_rows, seed=int(rs.randint(1 << 30)))
    form = QCQPForm.from_arrays(funcs)
    es, e = Engine(form), Engine(form)
    seed0, sstride, first0, fstride = int(rs.randint(1 << 20)), int(rs.randint(0, 4)), int(rs.randint(100)), int(rs.choice([0, R, 100000]))
    tag = 'case %d n=%d rows=%d R=%d K=%d iters=%d phase1=%d generate=%d' % (case, n, m_rows, R, K, iters, phase1, generate)
    try:
        if generate:
            o = es.cd_stream_run(K, R, generate=True, phase1=phase1, num_iters=iters, seed=seed0, seed_stride=sstride,
                                 first_index=first0, first_stride=fstride)
            X0 = None
        else:
            # uploaded starts: near +-1 (pass the gate without phase 1), a share of them far off (fail it)
            X0 = np.sign(rs.randn(n, K * R)) * (1.0 + 2e-3 * rs.rand(n, K * R))
            far = rs.rand(K * R) < 0.3
            X0[:, far] *= 1.0 + rs.rand(int(far.sum()))
            es.upload(X0)
            o = es.cd_stream_run(K, R, generate=False, phase1=phase1, num_iters=iters, seed=seed0, seed_stride=sstride,
                                 first_index=first0, first_stride=fstride)
    except EngineError as err:
        print(tag, ': engine error', err); bad += 1
        es.close(); e.close()
        continue
    X = es.download()
    worst = 0.0
    for p in range(K):
        sd, fi = seed0 + p * sstride, first0 + p * fstride
        sl = slice(p * R, (p + 1) * R)
        if generate:
            e.randn(R, seed=sd, first_index=fi)
        else:
            e.upload(X0[:, sl])
        outr = e.cd_run(phase1=phase1, num_iters=iters, seed=sd, first_index=fi)
        Xr = e.download()
        d = float(np.max(np.abs(X[:, sl] - Xr)))
        worst = max(worst, d)
        msgs = []
        if not d <= 1e-12 * (1 + np.max(np.abs(Xr))):
            msgs.append('points %.3e' % d)
        for key in COUNTERS:
            if not np.array_equal(o[key][sl], outr[key]):
                msgs.append('%s (%d restarts)' % (key, int(np.sum(o[key][sl] != outr[key]))))
        rf = float(np.max(np.abs(o['f0'][sl] - outr['f0']) / (1 + np.abs(outr['f0']))))
        rv = float(np.max(np.abs(o['maxviol'][sl] - outr['maxviol'])))
        if not rf <= 1e-10:
            msgs.append('f0 %.3e' % rf)
        if not rv <= 1e-12:
            msgs.append('maxviol %.3e' % rv)
        idx = e.select_best(1e-4)[0]
        if o['best_index'][p] != idx:
            msgs.append('best %d / %d' % (o['best_index'][p], idx))
        if msgs:
            print(tag, 'population', p, ':', '; '.join(msgs)); bad += 1
    print(tag, ': ok' if not bad else ': (mismatches so far %d)' % bad, 'worst |dx| %.1e' % worst, 'passed the gate %d of %d' % (int(o['ran_phase2'].sum()), K * R))
    es.close(); e.close()
print('%d cases, %d mismatching populations' % (cases, bad))
