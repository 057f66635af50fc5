#!/usr/bin/env python3
"""Mean counter value per launch and kernel from rocprofv3 --pmc csv output: python3 tools/pmc_by_kernel.py DIR [DIR ...]"""
import csv, glob, re, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r'(k_\w+(<[^>]*>)?)', r['Kernel_Name'])
            k = (m.group(1) if m else r['Kernel_Name'][:60], r['Counter_Name'])
            acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print(f'{k:62s} {c:24s} launches {n:5d} mean {s / n:16.1f}')
