"""In-kernel cycle breakdown of the phase-2 coordinate-descent kernel (debug counters)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import problems, _ffi
from qcqp_amd.engine import Engine, _ip
from qcqp_amd.form import QCQPForm

n, R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
funcs, _, _ = problems.boolean_least_squares(n, n // 4, seed=1)
e = Engine(QCQPForm.from_arrays(funcs))
e.randn(R, seed=1)
e.cd_run()
e.L.qcqpmi_debug_profile(e.h, 1, None)
e.randn(R, seed=2)
out = e.cd_run()
s = np.zeros(8, dtype=np.int64)
e.L.qcqpmi_debug_profile(e.h, 0, _ip(s))
tiles = (R + 15) // 16
names = ['mfma', 'feasible-sets', 'barrier1', 'sequential', 'barrier2', 'blocks']
print('phase2 kernel ms', e.kernel_ms(2), 'phase1 ms', e.kernel_ms(1), 'eval ms', e.kernel_ms(0))
print('sweeps/restart', out['visits2'].mean() / n, 'max', out['visits2'].max() / n, 'accepted/restart', out['accepted2'].mean())
blocks = s[5] / tiles
print('blocks per tile', blocks, '= sweeps', blocks / (n / 16))
print('generic-path blocks', s[6], 'of', s[5])
for k in range(5):
    print('%-14s %10.0f cycles/block' % (names[k], s[k] / s[5]))
