#!/usr/bin/env python3
"""Cycles per stage of one BASELINE config-2 trajectory (L=16, 4 layers, 32 chains, nstep 10) on the small-lattice fused
path, and its wall time.  FTHMC_LIB selects another build of the library."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fthmc_amd import ops  # noqa: E402
import bench  # noqa: E402

dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1)
L, nl, B, beta = (int(sys.argv[1]) if len(sys.argv) > 1 else 16), 4, 32, 4.0
flow = bench.make_flow(gen, nl)
w = ops.pack_weights(flow, device=dev)
x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * 0.3).to(dev)
v = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64).to(dev)
u = torch.rand(B, generator=gen, dtype=torch.float64).to(dev)
out = ops.ft_trajectory(x, v, u, w, nl, beta, 0.1, 10)
st = out['state'].clone()
torch.cuda.synchronize()
reps = 50
t0 = time.perf_counter()
for _ in range(reps):
    ops.ft_trajectory(x, v, u, w, nl, beta, 0.1, 10, out=out, state_in=st)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
cyc = ops.small_profile(x, v, u, w, nl, beta, 0.1, 10)
names = {0: 'fwd weights+plaq+sincos', 1: 'fwd conv1', 2: 'fwd conv2', 3: 'fwd conv3', 4: 'fwd transform', 5: 'fwd newP/logJ', 6: 'fwd link update',
         8: 'bwd loads+transform adj', 9: 'bwd conv3T', 10: 'bwd conv2T', 11: 'bwd conv1T', 12: 'bwd channel sum', 13: 'bwd gP update',
         16: 'copy latent', 17: 'sync+wilson seed', 18: 'kick/drift', 19: 'action/charge'}
tot = sum(cyc)
nf, nb = nl * 11, nl * 10
print(f'trajectory {ms:.3f} ms eager; thread-0 cycles {tot:.0f}; per layer: ' +
      '  '.join(f'{k}:{cyc[k] / (nf if k < 8 else nb):.0f}' for k in names if k < 16) +
      '  | ' + '  '.join(f'{k}:{cyc[k]:.0f}' for k in names if k >= 16))
