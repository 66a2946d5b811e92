"""Prints a timeline (start, duration in ms) of the kernels in a rocprofv3 --kernel-trace CSV: usage trace_timeline.py file.csv [t0_ms] [t1_ms]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
t_lo = float(sys.argv[2]) if len(sys.argv) > 2 else None
t_hi = float(sys.argv[3]) if len(sys.argv) > 3 else None
ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', '?'))) for r in rows]
ks.sort()
base = ks[0][0]
qs = [k for k in ks if 'cd_phase2_q' in k[2]]
print('%d kernels, %d phase-2 kernels' % (len(ks), len(qs)))
mid = qs[len(qs) // 2][0] if qs else base
lo = (mid - base) / 1e6 - 1.0 if t_lo is None else t_lo
hi = lo + 12.0 if t_hi is None else t_hi
for s, e, nm, q in ks:
    ts = (s - base) / 1e6
    if lo <= ts <= hi:
        short = nm.split('(')[0].replace('qcqpmi::', '').replace('(anonymous namespace)::', '')[:40]
        print('%9.3f  +%7.3f ms  q%-4s %s' % (ts, (e - s) / 1e6, q, short))
