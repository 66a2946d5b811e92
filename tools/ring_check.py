"""Ring mode (one persistent slot-queue launch for several contexts): correctness against serial runs and step time.
usage: ring_check.py [n=1024] [R=4096] [steps=40] [p2_cus=192] [N=4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import problems
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
p2 = int(sys.argv[4]) if len(sys.argv) > 4 else 192
N = int(sys.argv[5]) if len(sys.argv) > 5 else 4
seed = 2024
funcs, _, _ = problems.boolean_least_squares(n, n // 4, seed=1)
form = QCQPForm.from_arrays(funcs)
engs = [Engine(form) for _ in range(N)]
ref = Engine(form); ref.cd_queue(0)
refs = []
for k in range(min(steps, 6)):
    ref.randn(R, seed=seed + k); o = ref.cd_run(seed=seed + k); refs.append((ref.download(), o))
for e in engs:                      # warm-up: every buffer exists before the persistent launch starts
    e.randn(R, seed=1); e.cd_run(seed=1)
print('starting ring', flush=True)
Engine.ring_start(engs, phase2_cus=p2)
print('ring started', flush=True)
def submit(k):
    e = engs[k % N]; e.randn(R, seed=seed + k); e.ring_submit(seed=seed + k)
t0 = time.perf_counter()
for k in range(min(N - 1, steps)):
    submit(k)
got, sw = [], 0.0
for k in range(steps):
    e = engs[k % N]
    try:
        o = e.ring_collect()
    except Exception as ex:
        import ctypes as C
        print('collect of step %d failed: %s' % (k, ex))
        for i, g in enumerate(engs):
            v = np.zeros(10, dtype=np.int64)
            g.L.qcqpmi_debug_cd_ring_state(g.h, v.ctypes.data_as(C.POINTER(C.c_int64)))
            print('   member %d state' % i, v.tolist())
        raise
    sw += o['visits2'].sum() / float(n)
    if k < 8 or k % 10 == 0:
        print('collected step %d at %.1f ms' % (k, (time.perf_counter() - t0) * 1e3), flush=True)
    if k < len(refs):
        got.append((e.download(), o))
    e.select_best(1e-4)
    if k + N - 1 < steps:
        submit(k + N - 1)
dt = time.perf_counter() - t0
engs[0].ring_stop()
print('ring: n=%d R=%d, %d contexts, phase 2 on %d CUs: %d steps in %.1f ms = %.3f ms/step, %.4g restart-sweeps/s, %.3f of 78.6 TFLOP/s'
      % (n, R, N, p2, steps, dt * 1e3, dt * 1e3 / steps, sw / dt, sw * 2.0 * n * n / dt / 78.6e12))
ok = True
for k, ((X, o), (rX, ro)) in enumerate(zip(got, refs)):
    d = np.max(np.abs(X - rX))
    same = all(np.array_equal(o[key], ro[key]) for key in ('sweeps1', 'sweeps2', 'visits2', 'accepted2', 'ran_phase2', 'status1', 'status2'))
    fe = np.max(np.abs(o['f0'] - ro['f0']) / (1 + np.abs(ro['f0'])))
    ok = ok and d < 1e-12 and same and fe < 1e-11 and np.array_equal(o['maxviol'], ro['maxviol'])
    print('   step %d: max|dx| %.1e, counters identical %s, f0 rel %.1e, maxviol identical %s' % (k, d, same, fe, np.array_equal(o['maxviol'], ro['maxviol'])))
print('RING_OK' if ok else 'RING_MISMATCH')
