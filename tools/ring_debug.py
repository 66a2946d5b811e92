import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from qcqp_amd import problems
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm
n, R, p2 = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
funcs, _, _ = problems.boolean_least_squares(n, n // 4, seed=1)
form = QCQPForm.from_arrays(funcs)
engs = [Engine(form) for _ in range(3)]
for e in engs:
    e.randn(R, seed=1); e.cd_run(seed=1)
def state(e, tag):
    v = np.zeros(10, dtype=np.int64)
    e.L.qcqpmi_debug_cd_ring_state(e.h, v.ctypes.data_as(C.POINTER(C.c_int64)))
    print(tag, v.tolist(), flush=True)
Engine.ring_start(engs, phase2_cus=p2)
time.sleep(0.2); state(engs[0], 'after start')
engs[0].randn(R, seed=5); engs[0].ring_submit(seed=5)
for t in range(6):
    time.sleep(0.3); state(engs[0], 'after submit %.1fs' % (0.3 * (t + 1)))
engs[0].ring_stop()
state(engs[0], 'after stop')
