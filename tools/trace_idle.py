#!/usr/bin/env python3
"""GPU idle time inside the bench's timed sequence, from a rocprofv3 --kernel-trace csv (same command as tools/layer_times.py):
the union of all kernel intervals against the span they cover, and how much of the busy time has two kernels in flight.
   python3 tools/trace_idle.py OUT"""
import csv, glob, sys
ev = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
ev.sort()
# keep the dense part: the last 60 % of the launches (steady-state replays)
ev = ev[int(len(ev) * 0.4):]
t0, t1 = ev[0][0], max(e for _, e, _ in ev)
pts = sorted([(s, 1) for s, e, _ in ev] + [(e, -1) for s, e, _ in ev])
busy = over = 0; depth = 0; last = pts[0][0]
for t, d in pts:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    depth += d; last = t
span = t1 - t0
print(f'{len(ev)} launches over {span * 1e-6:.2f} ms: no kernel in flight {100 * (span - busy) / span:.2f} % of the time, two or more in flight {100 * over / span:.1f} %')
