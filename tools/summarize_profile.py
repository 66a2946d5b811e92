#!/usr/bin/env python
"""Turn gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the committed summaries under profiles/."""
import collections
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src = os.path.join(REPO, 'gpurun_out', 'prof_' + tag)
dst = os.path.join(REPO, 'profiles')
os.makedirs(dst, exist_ok=True)


def find(sub, suffix):
    hits = glob.glob(os.path.join(src, sub, '**', '*' + suffix), recursive=True)
    return hits[0] if hits else None


def short(name):
    return name.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('qcqpmi::', '')[:60]


import datetime
import subprocess

out = {'tag': tag, 'date': datetime.date.today().isoformat(),
       'launch': 'one launch of the lifecycle kernel = 20 steps of 4096 restarts (bench.py --scheme stream --steps 20 --warmup 20)'}
try:
    out['git_commit'] = subprocess.check_output(['git', '-C', REPO, 'rev-parse', '--short', 'HEAD']).decode().strip()
except Exception:
    out['git_commit'] = None
# ---- kernel statistics
ks = find('stats', 'kernel_stats.csv')
lines = []
if ks:
    rows = list(csv.DictReader(open(ks)))
    lines.append('| kernel | calls | total ms | avg us | min us | max us | % |')
    lines.append('|---|---|---|---|---|---|---|')
    for r in rows:
        lines.append('| %s | %s | %.3f | %.1f | %.1f | %.1f | %.1f |' % (
            short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3,
            float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, float(r['Percentage'])))
        if 'cd_phase2' in r['Name'] or 'cd_life_kernel' in r['Name']:
            out['cd_phase2_avg_ms'] = float(r['AverageNs']) / 1e6
            out['kernel'] = short(r['Name']).split('<')[0]
            out['cd_phase2_calls'] = int(r['Calls'])
    open(os.path.join(dst, tag + '_kernel_stats.csv'), 'w').write(open(ks).read())


def pmc(sub, counter):
    f = find(sub, 'counter_collection.csv')
    agg = collections.defaultdict(lambda: [0.0, 0])
    if not f:
        return agg
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter:
            a = agg[short(r['Kernel_Name'])]
            a[0] += float(r['Counter_Value'])
            a[1] += 1
    return agg


fetch, write = pmc('fetch', 'FETCH_SIZE'), pmc('write', 'WRITE_SIZE')
hit, miss = pmc('l2', 'TCC_HIT_sum'), pmc('l2', 'TCC_MISS_sum')
lines.append('')
lines.append('| kernel | launches | FETCH_SIZE KB/launch (raw) | WRITE_SIZE KB/launch | HBM bytes/launch (2*FETCH + WRITE)*1024 | L2 hit rate |')
lines.append('|---|---|---|---|---|---|')
for k in sorted(set(fetch) | set(write)):
    fe = fetch[k][0] / max(fetch[k][1], 1)
    wr = write[k][0] / max(write[k][1], 1)
    h, m = hit[k][0], miss[k][0]
    hbm = (2.0 * fe + wr) * 1024.0
    lines.append('| %s | %d | %.1f | %.1f | %.3e | %s |' % (k, fetch[k][1], fe, wr, hbm,
                                                          ('%.3f' % (h / (h + m))) if h + m else 'n/a'))
    if 'cd_phase2' in k or 'cd_life_kernel' in k:
        out['cd_phase2_hbm_bytes_per_launch'] = hbm
        out['cd_phase2_fetch_kb_raw'] = fe
        out['cd_phase2_write_kb'] = wr
        out['cd_phase2_l2_hit_rate'] = h / (h + m) if h + m else None
# pass 5: matrix-pipe occupancy.  SQ_VALU_MFMA_BUSY_CYCLES sums, over all 1024 SIMDs, the cycles the MFMA pipe is
# busy (64 per v_mfma_f64_16x16x4_f64); GRBM_GUI_ACTIVE sums the active cycles of the 8 XCDs.
mb, ga = pmc('mfma', 'SQ_VALU_MFMA_BUSY_CYCLES'), pmc('mfma', 'GRBM_GUI_ACTIVE')
if mb:
    lines.append('')
    lines.append('| kernel | launches | SQ_VALU_MFMA_BUSY_CYCLES / launch | GRBM_GUI_ACTIVE / launch (8 XCDs) | MFMA pipe busy = BUSY / (1024 SIMDs x GUI_ACTIVE / 8) |')
    lines.append('|---|---|---|---|---|')
    for k in sorted(mb):
        if mb[k][0] <= 0:
            continue
        b_ = mb[k][0] / max(mb[k][1], 1)
        g_ = ga[k][0] / max(ga[k][1], 1)
        frac = b_ / (1024.0 * g_ / 8.0) if g_ else 0.0
        lines.append('| %s | %d | %.3e | %.3e | %.3f |' % (k, mb[k][1], b_, g_, frac))
        if 'cd_phase2' in k or 'cd_life_kernel' in k:
            out['cd_phase2_mfma_busy_frac'] = frac
bj = os.path.join(src, 'bench_under_profiler.json')
if os.path.exists(bj) and os.path.getsize(bj):
    out['bench_under_profiler'] = json.load(open(bj))
    out['engine_kernel'] = (out['bench_under_profiler'].get('roofline') or {}).get('kernel')     # the name the engine reports (bench.py matches on it)
hdr = ['# rocprofv3 summary %s' % tag, '',
       'Command: `python bench.py --scheme stream --steps 20 --warmup 20 --no-cpu-baseline --no-secondary` (tools/profile_round.sh):',
       'two launches of the lifecycle kernel (round 6: `cd_life_kernel<3, 0, 0, 1, 1>` = three multiplying waves, no chain share, band kind, one tile per workgroup, FACTORED objective; round 5: `cd_life_kernel<3, 4, 0>`; round 4: `cd_phase2_qs_kernel<4, true>`), each = 20 steps of 4096 restarts --',
       'suggest, phase 1, gate, phase 2, objective of 81920 restarts inside the launch.',
       'Pass 1 `--kernel-trace --stats`; passes 2-4 `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` / `--pmc TCC_HIT_sum TCC_MISS_sum`',
       '(separate runs, as MI355X_MICROARCH.md prescribes). FETCH_SIZE is doubled (gfx950 reports half the bytes of',
       'wide coalesced reads); WRITE_SIZE is taken as is (uncalibrated).  Pass 5 `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`:',
       'fraction of the SIMD-cycles of the launch during which the matrix pipe is busy (all MFMAs issued, useful or not).', '']
notes = os.path.join(dst, tag + '_notes.md')          # hand-written findings of the round (resource usage, stage split, experiments): appended
tail = ['', open(notes).read().rstrip()] if os.path.exists(notes) else []
open(os.path.join(dst, tag + '_summary.md'), 'w').write('\n'.join(hdr + lines + tail) + '\n')
json.dump(out, open(os.path.join(dst, tag + '_summary.json'), 'w'), indent=1, sort_keys=True)
print('\n'.join(lines))
print(json.dumps(out)[:600])
