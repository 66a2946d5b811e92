"""In-kernel cycle breakdown of the phase-2 coordinate-descent kernels (debug counters).

usage: python tools/phase_profile.py [n] [R] [generic]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from qcqp_amd import problems
from qcqp_amd.engine import Engine, _ip
from qcqp_amd.form import QCQPForm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
generic = len(sys.argv) > 3 and sys.argv[3] == 'generic'
funcs, _, _ = problems.boolean_least_squares(n, n // 4, seed=1)
e = Engine(QCQPForm.from_arrays(funcs))
mode = 2 if generic else 0
if len(sys.argv) > 4:
    mode |= int(sys.argv[4]) << 4
e.L.qcqpmi_debug_profile(e.h, mode, None)
e.randn(R, seed=1)
e.cd_run()
e.L.qcqpmi_debug_profile(e.h, mode | 1, None)
e.randn(R, seed=2)
out = e.cd_run()
s = np.zeros(16, dtype=np.int64)
e.L.qcqpmi_debug_profile(e.h, mode, _ip(s))
tiles = (R + 15) // 16
print('kernel:', 'general (4 waves)' if generic else 'role-split (8 waves)')
print('phase2 kernel ms', e.kernel_ms(2), 'phase1 ms', e.kernel_ms(1), 'eval ms', e.kernel_ms(0))
print('sweeps/restart', out['visits2'].mean() / n, 'max', out['visits2'].max() / n,
      'accepted/restart', out['accepted2'].mean())
blocks = s[5] / tiles
print('blocks per tile', blocks, '= sweeps', blocks / (n / 16))
print('generic-path blocks', s[6], 'of', s[5])
quad = (not generic) and n % 16 == 0 and n <= 1024 and not ((mode >> 4) & 64)
if quad:
    print('kernel variant: cd_phase2_q_kernel (quad chain)')
    for k, nm in enumerate(['chain: loop top', 'chain: wait for partial tiles', 'chain: sum partials + loads', 'chain: 16 steps',
                            'chain: block end + commit', '-', '-', 'chain: fix-up + own share']):
        if nm != '-':
            print('%-34s %10.0f cycles/block' % (nm, s[k] / max(s[5], 1)))
    print('%-34s %10.0f' % ('generic-path blocks', s[6]))
    for w, base in (('2 (elder)', 8), ('6 (younger)', 12)):
        for k, nm in enumerate(['store + publish', 'commit wait + refresh', 'product', 'slot-release wait']):
            print('%-34s %10.0f cycles/block' % ('mfma wave %s: %s' % (w, nm), s[base + k] / max(s[5], 1)))
    sys.exit(0)
names = (['mfma', 'stage', 'barrier1', 'sequential', 'barrier2'] if generic
         else ['chain', 'barrier wait', 'fix-up + preload', '-'])
for k in range(len(names)):
    print('%-18s %10.0f cycles/block' % (names[k], s[k] / max(s[5], 1)))
if not generic:
    for k, nm in enumerate(['mfma wave: store+b1 wait', 'mfma wave: prefetch issue', 'mfma wave: b2 wait', 'mfma wave: B preload + mfma', 'mfma wave:   of which B preload']):
        print('%-26s %10.0f cycles/block' % (nm, s[8 + k] / max(s[5], 1)))
