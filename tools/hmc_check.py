#!/usr/bin/env python3
"""Plain-HMC trajectory timing: fused single-launch kernel vs per-step launches."""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fthmc_amd import ops
for (B, L, beta) in [(1, 8, 2.0), (128, 64, 6.0), (1024, 64, 6.0)]:
    gen = torch.Generator().manual_seed(1)
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * 0.3).cuda()
    v = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64).cuda()
    u = torch.rand(B, generator=gen, dtype=torch.float64).cuda()
    for var in (1, 0):
        ops.set_variant(var)
        r = ops.hmc_trajectory(x, v, u, beta, 0.1, 10); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): r = ops.hmc_trajectory(x, v, u, beta, 0.1, 10)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f'B={B} L={L} variant={"fused" if var else "steps"}: {dt*1e6:.1f} us / trajectory (10 steps) '
              f'= {64.0*L*L*B*10/dt/1e9:.0f} GB/s algorithmic, acc={float(r["acc"].mean()):.2f}', flush=True)
    ops.set_variant(1)
