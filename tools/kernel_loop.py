#!/usr/bin/env python3
"""Launch the two hot kernels a few times at the bench shape (for rocprofv3 --pmc passes)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fthmc_amd import ops
from oracle import ref_cpu as R
B, L = 128, 64
gen = torch.Generator().manual_seed(1331)
w = ops.pack_weights(R.default_flow(1, gen), device='cuda')
x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
for kind in ('flow_fwd', 'flow_bwd'):
    print(kind, ops.time_kernel(kind, x, w, mu=0, off=1, beta=6.0, reps=10))
# the plain-HMC leapfrog step: row-strip kernel (L % 64 == 0) and the 16 x 16-tile kernel
for rows in ('1', '0'):
    os.environ['FTHMC_LEAP_ROWS'] = rows
    ops.set_variant(1)
    print('leap_step rows=' + rows, ops.time_kernel('leap_step', x, beta=6.0, reps=10))
os.environ['FTHMC_LEAP_ROWS'] = '1'
ops.set_variant(1)
