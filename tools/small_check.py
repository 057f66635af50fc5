#!/usr/bin/env python3
"""Small-lattice fused path (csrc/flow_small.hip) against the tiled kernels on the same inputs, and its speed.
    python tools/small_check.py [reps]"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fthmc_amd import ops  # noqa: E402
import bench  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda', 0)


def both(fn):
    ops.set_small_path(True); a = fn(); torch.cuda.synchronize()
    ops.set_small_path(False); b = fn(); torch.cuda.synchronize()
    ops.set_small_path(True)
    return a, b


def md(a, b):
    return float((a - b).abs().max())


def amd(a, b):
    d = (a - b + math.pi) % (2 * math.pi) - math.pi
    return float(d.abs().max())


worst = 0.0
for (L, nl, B, beta, act) in [(16, 4, 32, 4.0, 'silu'), (8, 2, 3, 2.0, 'silu'), (12, 8, 5, 3.0, 'silu'), (16, 16, 2, 4.0, 'relu'),
                              (8, 8, 1, 2.0, 'leaky_relu'), (16, 1, 7, 4.0, 'silu'), (16, 3, 4, 5.0, 'silu')]:
    gen = torch.Generator().manual_seed(100 + L + nl)
    flow = bench.make_flow(gen, nl)
    w = ops.pack_weights(flow, device=dev)
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).to(dev)
    v = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64).to(dev)
    u = torch.rand(B, generator=gen, dtype=torch.float64).to(dev)
    (ya, lda), (yb, ldb) = both(lambda: ops.flow_forward(x, w, nl, act))
    (Sa, la, pa, qa), (Sb, lb, pb, qb) = both(lambda: ops.ft_action(x, w, nl, beta, act))
    Fa, Fb = both(lambda: ops.ft_force(x, w, nl, beta, act))
    xs = (0.3 * x).contiguous()
    (xa, va), (xb_, vb) = both(lambda: ops.ft_leapfrog(xs, v, w, nl, beta, 0.05, 4, act))
    ra, rb = both(lambda: ops.ft_trajectory(xs, v, u, w, nl, beta, 0.05, 6, act))
    st = ra['state'].clone()
    rc = ops.ft_trajectory(ra['x_new'].clone(), v, u, w, nl, beta, 0.05, 6, act, state_in=st)
    rd = ops.ft_trajectory(ra['x_new'].clone(), v, u, w, nl, beta, 0.05, 6, act)
    errs = {'fwd': amd(ya, yb), 'logdet': md(lda, ldb) / L ** 2, 'S_eff': md(Sa, Sb) / float(Sb.abs().max()), 'plaq': md(pa, pb), 'Q': md(qa, qb),
            'force': md(Fa, Fb) / float(Fb.abs().max()), 'lf_x': amd(xa, xb_), 'lf_v': md(va, vb),
            'H0': md(ra['H0'], rb['H0']) / float(rb['H0'].abs().max()), 'H1': md(ra['H1'], rb['H1']) / float(rb['H1'].abs().max()),
            'dH': md(ra['dH'], rb['dH']), 'x_new': amd(ra['x_new'], rb['x_new']), 'acc': md(ra['acc'], rb['acc']),
            'state': md(ra['state'], rb['state']), 'chained_vs_stateless_bits': 0.0 if all(torch.equal(rc[k], rd[k]) for k in ('x_new', 'dH', 'H0', 'H1', 'acc', 'state')) else 1.0}
    bad = {k: e for k, e in errs.items() if not e < 1e-9}
    worst = max(worst, max(errs.values()))
    print(f'L={L} nl={nl} B={B} act={act}: max err {max(errs.values()):.2e}' + (f'  BAD {bad}' if bad else ''), flush=True)

# speed at BASELINE config 2: L=16, 4 layers, 32 chains, nstep 10
gen = torch.Generator().manual_seed(1)
L, nl, B, beta = 16, 4, 32, 4.0
flow = bench.make_flow(gen, nl)
w = ops.pack_weights(flow, device=dev)
x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * 0.3).to(dev)
v = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64).to(dev)
u = torch.rand(B, generator=gen, dtype=torch.float64).to(dev)
for small in (True, False):
    ops.set_small_path(small)
    out = ops.ft_trajectory(x, v, u, w, nl, beta, 0.1, 10)
    st = out['state'].clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.ft_trajectory(x, v, u, w, nl, beta, 0.1, 10, out=out, state_in=st)
    torch.cuda.synchronize()
    print(f'config 2 trajectory, small={small}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms (eager launches)', flush=True)
    for name, fn in (('ft_force', lambda: ops.ft_force(x, w, nl, beta)), ('ft_action', lambda: ops.ft_action(x, w, nl, beta))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        print(f'   {name}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms', flush=True)
ops.set_small_path(True)
cyc = ops.small_profile(x, v, u, w, nl, beta, 0.1, 10)
names = {0: 'fwd weights+plaq+sincos', 1: 'fwd conv1', 2: 'fwd conv2', 3: 'fwd conv3', 4: 'fwd transform', 5: 'fwd newP/logJ', 6: 'fwd link update',
         8: 'bwd loads+transform adj', 9: 'bwd conv3T', 10: 'bwd conv2T', 11: 'bwd conv1T', 12: 'bwd channel sum', 13: 'bwd gP update',
         16: 'copy latent', 17: 'sync+wilson seed', 18: 'kick/drift', 19: 'action/charge'}
tot = sum(cyc)
print(f'profile (cycles of thread 0, one trajectory = 44 fwd + 40 bwd layers, total {tot:.0f}):')
for k, nme in names.items():
    per = cyc[k] / (44 if k < 8 else 40 if k < 16 else 1)
    print(f'   {k:2d} {nme:28s} {cyc[k]:10.0f}  ({100 * cyc[k] / tot:4.1f} %)  per layer {per:8.0f}')
print('WORST', worst)
sys.exit(0 if worst < 1e-9 else 1)
