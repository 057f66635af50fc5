#!/usr/bin/env python3
"""Duration of the coupling kernels by their position in a sweep, from a rocprofv3 --kernel-trace csv:
   rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 bench.py --steps 3 --warmup 1 --regions 5 --no-cpu-baseline
   python3 tools/layer_times.py OUT
Per queue the launches are ordered by start time; the k-th forward (backward) launch of a run of consecutive forward (backward)
launches is layer k (7 - k) of a sweep."""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((r.get('Queue_Id', '0'), int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
byq = collections.defaultdict(list)
for q, s, e, n in rows:
    byq[q].append((s, e, n))
acc = collections.defaultdict(lambda: [0.0, 0])
for q, v in byq.items():
    v.sort()
    run, kind = 0, None
    for s, e, n in v:
        k = 'fwd' if 'k_flow_fwd<' in n else 'bwd' if 'k_flow_bwd_gather<' in n else None
        if k is None:
            run, kind = 0, None
            continue
        if k != kind:
            run, kind = 0, k
        mu = n.split('<')[1].split(',')[4 if k == 'fwd' else 3].strip()
        a = acc[(k, run, mu)]; a[0] += (e - s) * 1e-3; a[1] += 1
        run += 1
for (k, run, mu), (t, c) in sorted(acc.items()):
    if c >= 8:
        print(f'{k} position {run:2d} mu {mu}: {t / c:7.2f} us  ({c} launches)')
