#!/usr/bin/env python
"""gpurun_out/prof_dense_<tag>/ (tools/profile_dense.sh) -> profiles/<tag>_dense_summary.md, <tag>_dense_kernel_stats.csv"""
import collections, csv, glob, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
src = os.path.join(REPO, 'gpurun_out', 'prof_dense_' + tag)
dst = os.path.join(REPO, 'profiles')


def find(sub, suffix):
    hits = glob.glob(os.path.join(src, sub, '**', '*' + suffix), recursive=True)
    return hits[0] if hits else None


def short(name):
    return name.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('qcqpmi::', '')[:60]


def pmc(sub, counter):
    f = find(sub, 'counter_collection.csv')
    agg = collections.defaultdict(lambda: [0.0, 0])
    if f:
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                a = agg[short(r['Kernel_Name'])]
                a[0] += float(r['Counter_Value']); a[1] += 1
    return agg


L = ['# rocprofv3 summary %s: coordinate descent on the dense-constraint path (BASELINE.json configs[4] family)' % tag, '',
     'Command: `python tools/dense_rate.py 1024 256 512 4096` (n = 1024, m = 256 generated on the device; 512 restarts, then 4096; a',
     'warm-up sweep + 2 sweeps per phase each) under `rocprofv3 --kernel-trace --stats`, then separate `--pmc` passes',
     '(tools/profile_dense.sh).', '', '```']
for line in open(os.path.join(src, 'stats.log')):
    if line.startswith('n '):
        L.append(line.rstrip())
L += ['```', '']
ks = find('stats', 'kernel_stats.csv')
if ks:
    open(os.path.join(dst, tag + '_dense_kernel_stats.csv'), 'w').write(open(ks).read())
    L += ['| kernel | calls | total ms | avg us | % |', '|---|---|---|---|---|']
    for r in csv.DictReader(open(ks)):
        if float(r['Percentage']) < 0.05:
            continue
        L.append('| %s | %s | %.3f | %.1f | %.1f |' % (short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, float(r['Percentage'])))
    L.append('')
busy, act = pmc('mfma', 'SQ_VALU_MFMA_BUSY_CYCLES'), pmc('mfma', 'GRBM_GUI_ACTIVE')
fe, wr = pmc('fetch', 'FETCH_SIZE'), pmc('write', 'WRITE_SIZE')
L += ['| kernel | launches | MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) | HBM bytes / launch = (2 x FETCH_SIZE + WRITE_SIZE) KB |', '|---|---|---|---|']
for k in sorted(busy):
    if 'dense_products' not in k and 'dense_chain' not in k:
        continue
    b, n = busy[k]
    a = act[k][0]
    frac = b / (1024.0 * a / 8.0) if a else float('nan')
    hb = (2.0 * fe[k][0] / max(fe[k][1], 1) + wr[k][0] / max(wr[k][1], 1)) * 1024.0 if k in fe else float('nan')
    L.append('| %s | %d | %.3f | %.3e |' % (k, n, frac, hb))
L += ['', 'All launches of a kernel are pooled (both population sizes; dense_products_kernel<0> includes the one-stage fix-up launches).',
      'FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md); it counts Infinity-Cache hits too.',
      'Algorithmic work of a block visit: 2 (m + 1) 16 n flops per restart (4.3 GFLOP at 512 restarts); the packed matrices of one block',
      'are 33.7 MB, read once per group of 128 candidates.']
open(os.path.join(dst, tag + '_dense_summary.md'), 'w').write('\n'.join(L) + '\n')
print('\n'.join(L))
