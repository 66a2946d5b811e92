"""Host-side timeline of the chained four-context loop of bench.py (where does the host block?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import problems
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm
p2 = int(sys.argv[1]) if len(sys.argv) > 1 else 192
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n, R, seed = 1024, 4096, 2024
funcs, _, _ = problems.boolean_least_squares(n, 256, seed=1)
form = QCQPForm.from_arrays(funcs)
engs = [Engine(form) for _ in range(NC)]
for e in engs:
    e.cd_queue(1); e.cd_partition(p2)
T0 = time.perf_counter()
log = []
def stamp(tag, k):
    log.append((time.perf_counter() - T0, tag, k))
def prepare(e, k):
    e.randn(R, seed=seed + k); e.cd_begin(phase1=True, seed=seed + k)
def finish(j):
    e = engs[j % NC]
    stamp('fetch>', j); o = e.cd_fetch(); stamp('fetch<', j)
    e.select_best(1e-4); stamp('best<', j)
    return o
count = 32
LA, DL = 2, NC - 4
for j in range(LA + 1):
    prepare(engs[j % NC], j)
for k in range(count):
    cur = engs[k % NC]
    for p_ in range(1, LA + 1):
        cur.cd_chain(engs[(k + p_) % NC] if k + p_ < count else None, R, seed + k + p_, 0, pos=p_)
    stamp('launch>', k); cur.cd_phase2(); stamp('launch<', k)
    if k + LA + 1 < count:
        prepare(engs[(k + LA + 1) % NC], k + LA + 1); stamp('prep<', k + LA + 1)
    if k >= DL:
        finish(k - DL)
for j in range(max(count - DL, 0), count):
    finish(j)
tot = time.perf_counter() - T0
for t, tag, k in log:
    if 14 <= k <= 17:
        print('%9.3f ms  %-8s %d' % (t * 1e3, tag, k))
print('p2 cus %d, %d contexts: total %.1f ms for %d steps = %.3f ms/step; run ahead %d' % (p2, NC, tot * 1e3, count, tot * 1e3 / count, sum(e.cd_pulled() for e in engs)))
