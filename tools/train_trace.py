#!/usr/bin/env python3
"""A few train_grad calls at a training-sized shape for rocprofv3 --kernel-trace --stats: python3 tools/train_trace.py L B n_layers [reps] [chain groups]"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fthmc_amd import ops
import bench
L, B, nl = (int(t) for t in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
groups = int(sys.argv[5]) if len(sys.argv) > 5 else ops.default_groups(B, L)
gen = torch.Generator().manual_seed(1)
w = ops.pack_weights(bench.make_flow(gen, nl), device='cuda')
xi = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
for _ in range(reps):
    r = ops.train_grad(xi, w, nl, 4.0, groups=groups)
torch.cuda.synchronize()
