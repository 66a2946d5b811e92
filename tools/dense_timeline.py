"""Kernel timeline of the dense path's block loop from a rocprofv3 --kernel-trace csv (steady state of the last phase-2 sweep).
usage: python tools/dense_timeline.py <dir with *_kernel_trace.csv> [events=26]"""
import csv
import glob
import sys

d = sys.argv[1]
nev = int(sys.argv[2]) if len(sys.argv) > 2 else 26
f = sorted(glob.glob(d + '/**/*kernel_trace.csv', recursive=True))[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void qcqpmi::', '').replace('qcqpmi::', ''),
             r.get('Stream_Id', '?'), r.get('Grid_Size_Z') or '') for r in rows)
idx = [i for i, e in enumerate(ev) if e[2].startswith('dense_chain') and '<2' in e[2]]
i0 = idx[-20]
t0 = ev[i0][0]
for e in ev[i0 - 3:i0 + nev]:
    print('%9.1f %9.1f  %7.1f us  %-44s stream %s z %s' % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2], e[3], e[4]))
ch = [e for e in ev if e[2].startswith('dense_chain') and '<2' in e[2]]
if len(ch) > 40:
    per = (ch[-2][0] - ch[-34][0]) / 32e3
    print('block period (last 32 phase-2 blocks): %.1f us' % per)
