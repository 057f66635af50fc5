#!/usr/bin/env python3
"""fthmc_train_grad timed at training shapes with the library FTHMC_LIB names (A/B runs in one call on one device):
    python3 tools/train_ab.py [L B n_layers reps] ...      default: the config-5 shard and L=16 / batch 512 / 8 layers"""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fthmc_amd import ops
import bench
a = [int(t) for t in sys.argv[1:]]
shapes = [tuple(a[i:i + 4]) for i in range(0, len(a), 4)] or [(256, 32, 16, 10), (16, 512, 8, 200)]
out = []
for L, B, nl, reps in shapes:
    gen = torch.Generator().manual_seed(1)
    w = ops.pack_weights(bench.make_flow(gen, nl), device='cuda')
    xi = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    for G in sorted({1, ops.default_groups(B, L), ops.default_train_groups(B, L)}):
        for _ in range(3):
            r = ops.train_grad(xi, w, nl, 4.0, groups=G)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            r = ops.train_grad(xi, w, nl, 4.0, groups=G)
        torch.cuda.synchronize()
        out.append(f'L={L} B={B} nl={nl} groups={G}: {(time.perf_counter() - t0) / reps * 1e3:.4f} ms; gw checksum {float(r["gw"].sum()):.12e}')
print(os.environ.get('FTHMC_LIB', 'default library').split('/')[-1], ' | '.join(out), flush=True)
