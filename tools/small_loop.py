#!/usr/bin/env python3
"""A few BASELINE config-2 trajectories on the small-lattice fused path (for rocprofv3 passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fthmc_amd import ops
import bench
gen = torch.Generator().manual_seed(1)
L, nl, B, beta = 16, 4, 32, 4.0
w = ops.pack_weights(bench.make_flow(gen, nl), device='cuda')
x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * 0.3).cuda()
v = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64).cuda()
u = torch.rand(B, generator=gen, dtype=torch.float64).cuda()
out = ops.ft_trajectory(x, v, u, w, nl, beta, 0.1, 10)
st = out['state'].clone()
for _ in range(3):
    ops.ft_trajectory(x, v, u, w, nl, beta, 0.1, 10, out=out, state_in=st)
torch.cuda.synchronize()
