#!/usr/bin/env python3
"""Per-stage cycle breakdown of the MFMA coupling-layer kernels at the bench shape."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fthmc_amd import ops
from oracle import ref_cpu as R

import sys as _s
B, L = (int(_s.argv[1]) if len(_s.argv) > 1 else 128), 64
gen = torch.Generator().manual_seed(1331)
flow = R.default_flow(1, gen)
w = ops.pack_weights(flow, device='cuda')
x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
names = {'flow_fwd': ['', 'plaq+sincos', 'conv1', 'conv2', 'conv3', 'transform1', 'finish+store'],
         'flow_bwd': ['', 'load+xform', 'conv3T', 'conv2T', 'conv1T', 'store']}
for kind in ('flow_fwd', 'flow_bwd'):
    for mu in (0, 1):
        cyc = ops.profile_stages(kind, x, w, mu=mu, off=1, beta=6.0)
        tot = sum(cyc[:7]) if kind == 'flow_fwd' else sum(cyc[:6])
        print(f'{kind} mu={mu}: total {tot:.0f} cycles/WG; ' + ', '.join(f'{n} {c:.0f}' for n, c in zip(names[kind][1:], cyc[1:]) if n))
        if kind == 'flow_bwd':
            print('    per-wave arrival at the first barrier:', ' '.join(f'{c:.0f}' for c in cyc[6:14]))
    print(kind, 'ms/launch', ops.time_kernel(kind, x, w, mu=0, off=1, beta=6.0, reps=20))
