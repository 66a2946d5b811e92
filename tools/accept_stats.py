"""Acceptance statistics of phase 2 on the headline workload: how many visits move a coordinate,
as a function of the sweep number (deterministic restarts: same seed, growing sweep cap)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import problems
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
funcs, _, _ = problems.boolean_least_squares(n, n // 4, seed=1)
e = Engine(QCQPForm.from_arrays(funcs))
pa = pv = 0
for cap in [3, 4, 5, 6, 8, 10, 12, 16, 20, 30]:
    e.randn(R, seed=1)
    o = e.cd_run(num_iters=cap)
    a, v = o['accepted2'].sum(), o['visits2'].sum()
    live = (o['sweeps2'] >= cap).sum()
    print('cap %2d: ran %d live %4d  visits %9d accepted %8d | this span: visits %9d accepted %7d rate %.4f'
          % (cap, o['ran_phase2'].sum(), live, v, a, v - pv, a - pa, (a - pa) / max(1, v - pv)))
    pa, pv = a, v
# per-block structure at the end: how many of a tile's 16 restarts move inside one block of 16
