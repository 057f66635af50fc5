#!/usr/bin/env python3
"""Timing of the fused training gradient at BASELINE config 2 and the config-5 per-GPU shard."""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fthmc_amd import ops
from oracle import ref_cpu as R
for (B, L, nl, beta) in [(32, 16, 4, 4.0), (128, 64, 8, 6.0), (32, 256, 16, 7.0)]:
    gen = torch.Generator().manual_seed(1331)
    flow = R.default_flow(nl, gen)
    w = ops.pack_weights(flow, device='cuda')
    xi = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    for G in (1, 2):
        r = ops.train_grad(xi, w, nl, beta, groups=G); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): r = ops.train_grad(xi, w, nl, beta, groups=G)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f'train_grad B={B} L={L} layers={nl} groups={G}: {dt*1e3:.2f} ms  ({5616*L*L*nl*B/dt/1e12:.2f} TFLOP/s dense-algorithmic fwd+dgrad+wgrad)', flush=True)
