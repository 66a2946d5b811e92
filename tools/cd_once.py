"""One improve(COORD_DESCENT) on the headline workload (for rocprofv3 runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcqp_amd import problems
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
funcs, _, _ = problems.boolean_least_squares(n, n // 4, seed=1)
e = Engine(QCQPForm.from_arrays(funcs))
for k in range(reps):
    e.randn(R, seed=1 + k)
    out = e.cd_run()
print('phase2 ms', e.kernel_ms(2), 'sweeps', out['visits2'].sum() / n)
