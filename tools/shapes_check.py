#!/usr/bin/env python3
"""Correctness + timing sanity at the other BASELINE shapes (configs 2 and 5 per-GPU shard)."""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fthmc_amd import ops
from oracle import ref_cpu as R

for (B, L, nl, beta, check) in [(32, 16, 4, 4.0, True), (32, 256, 16, 7.0, False), (128, 64, 8, 6.0, False)]:
    gen = torch.Generator().manual_seed(1331)
    flow = R.default_flow(nl, gen)
    w = ops.pack_weights(flow, device='cuda')
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi)
    xg = x.cuda()
    F = ops.ft_force(xg, w, nl, beta); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): F = ops.ft_force(xg, w, nl, beta)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    msg = f'B={B} L={L} layers={nl}: ft_force {dt*1e3:.3f} ms  ({3744*L*L*nl*B/dt/1e12:.2f} TFLOP/s dense-algorithmic)'
    if check:
        Fc = R.ft_force(x, flow, beta)
        msg += f'  max|F-F_oracle| = {float((F.cpu()-Fc).abs().max()):.2e}'
    else:
        nb = 1
        Fc = R.ft_force(x[:nb], flow, beta)
        msg += f'  chain0 max|F-F_oracle| = {float((F[:nb].cpu()-Fc).abs().max()):.2e}'
    print(msg, flush=True)
