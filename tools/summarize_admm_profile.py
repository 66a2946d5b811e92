#!/usr/bin/env python
"""gpurun_out/prof_admm_<tag>/ (tools/profile_admm.sh) -> profiles/<tag>_admm_summary.md, <tag>_admm_kernel_stats.csv"""
import collections, csv, glob, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
src = os.path.join(REPO, 'gpurun_out', 'prof_admm_' + tag)
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


L = ['# rocprofv3 summary %s: improve(ADMM) at BASELINE.json configs[3] size' % tag, '',
     'Command: `python tools/admm_fused_rate.py 1024 1000` under `rocprofv3 --kernel-trace --stats`, then separate `--pmc` passes',
     '(tools/profile_admm.sh): beamforming with 512 antennas (n = 1024), 16 + 64 constraints, reduced bases (rp = 2), 1024 restarts,',
     'rho = 1, num_iters = 1000 (the reference default; the runs end by the reference\'s stop rules).  The script runs the fused',
     'persistent kernel (csrc/admm_fused.hip, clusters of 4 workgroups per tile of 16 restarts), the multi-launch path (the',
     'round-2 scheme: 7 launches per iteration), and the fused kernel again.', '', '```']
for line in open(os.path.join(src, 'stages.log')):
    if line.startswith('n=') or line.startswith('admm_') or line.startswith('   per'):
        L.append(line.rstrip()[:330])
L += ['```', '']
ks = find('stats', 'kernel_stats.csv')
if ks:
    open(os.path.join(dst, tag + '_admm_kernel_stats.csv'), 'w').write(open(ks).read())
    L += ['| kernel | calls | total ms | avg us | % |', '|---|---|---|---|---|']
    for r in csv.DictReader(open(ks)):
        L.append('| %s | %s | %.3f | %.1f | %.1f |' % (short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, float(r['Percentage'])))
    L.append('')
busy, act = pmc('mfma', 'SQ_VALU_MFMA_BUSY_CYCLES'), pmc('mfma', 'GRBM_GUI_ACTIVE')
fe, wr = pmc('fetch', 'FETCH_SIZE'), pmc('write', 'WRITE_SIZE')
L += ['| kernel | launches | MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) | HBM bytes / launch = (2 x FETCH_SIZE + WRITE_SIZE) KB |', '|---|---|---|---|']
for k in sorted(busy):
    if 'admm' not in k and 'gemm' not in k:
        continue
    b, n = busy[k]
    a = act[k][0]
    frac = b / (1024.0 * a / 8.0) if a else float('nan')
    hb = (2.0 * fe[k][0] / max(fe[k][1], 1) + wr[k][0] / max(wr[k][1], 1)) * 1024.0 if k in fe else float('nan')
    L.append('| %s | %d | %.3f | %.3e |' % (k, n, frac, hb))
L += ['', 'FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md); it counts Infinity-Cache hits too.',
      'The fused kernel is one launch per run: its stage split comes from the in-kernel cycle counters above (s_memtime of member 0 of tile 0).']
open(os.path.join(dst, tag + '_admm_summary.md'), 'w').write('\n'.join(L) + '\n')
print('\n'.join(L[-16:]))
