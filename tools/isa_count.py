#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -S listing, split at s_barrier (stage boundaries).
usage: isa_count.py file.s kernel_substring"""
import re, sys
from collections import Counter
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and key in l)
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
def cls(m):
    if m.startswith('v_mfma'): return 'mfma'
    if m.startswith('ds_'): return 'lds'
    if m.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if m.startswith('s_waitcnt'): return 'wait'
    if m.startswith('s_'): return 'salu'
    if m.startswith('v_') and ('f64' in m or m in ('v_rcp_f64', 'v_ldexp_f64')): return 'valu64'
    if m.startswith('v_'): return 'valu'
    return 'other'
stage, tot = Counter(), Counter()
k = 0
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if not t or t.startswith(('.', ';', '/')) or t.endswith(':'): continue
    m = t.split()[0]
    if m == 's_barrier':
        print(f'stage {k}:', dict(sorted(stage.items())), 'total', sum(stage.values()))
        k += 1; stage = Counter(); continue
    stage[cls(m)] += 1; tot[cls(m)] += 1
print(f'stage {k}:', dict(sorted(stage.items())), 'total', sum(stage.values()))
print('all:', dict(sorted(tot.items())), 'total', sum(tot.values()))
