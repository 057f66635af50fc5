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
