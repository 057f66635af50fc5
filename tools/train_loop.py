#!/usr/bin/env python3
"""A few train_grad calls at the config-5 shard shape (B=32, L=256, 16 layers) for rocprofv3 passes."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fthmc_amd import ops
import bench
gen = torch.Generator().manual_seed(1)
B, L, nl, beta = 32, 256, 16, 7.0
w = ops.pack_weights(bench.make_flow(gen, nl), device='cuda')
xi = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
for _ in range(4):
    r = ops.train_grad(xi, w, nl, beta, groups=ops.default_groups(B, L))
torch.cuda.synchronize()
