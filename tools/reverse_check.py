#!/usr/bin/env python3
"""Time the inverse flow (fthmc_flow_reverse) and the forward at the bench shape; check reverse(forward(x)) = x."""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fthmc_amd import ops
from oracle import ref_cpu as R

B, L, nl = 128, 64, 8
gen = torch.Generator().manual_seed(1331)
flow = R.default_flow(nl, gen)
w = ops.pack_weights(flow, device='cuda')
x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
y, ld = ops.flow_forward(x, w, nl)
for name, fn in (('flow_forward', lambda: ops.flow_forward(x, w, nl)), ('flow_reverse', lambda: ops.flow_reverse(y, w, nl, tol=1e-12))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = fn()
    torch.cuda.synchronize()
    print(f'{name}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms for {nl} layers, B={B}, L={L}')
xb, ldb = ops.flow_reverse(y, w, nl, tol=1e-12)
d = (xb - x + math.pi) % (2 * math.pi) - math.pi
print('reverse(forward(x)) - x: max', float(d.abs().max()), ' logdet sum max', float((ld + ldb).abs().max()))
