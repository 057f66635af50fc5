#!/usr/bin/env python3
"""Kernel-level A/B bench at the config-3 launch shapes: forward / backward coupling-layer kernels (HIP events,
several rounds, min and median), stage stamps, and a quick parity check against the oracle on a small case."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fthmc_amd import ops
from oracle import ref_cpu as R

def check():
    gen = torch.Generator().manual_seed(7)
    B, L, nl, beta = 3, 24, 8, 3.0
    flow = R.default_flow(nl, gen)
    x = (torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi
    w = ops.pack_weights(flow, device='cuda')
    y, ld = ops.flow_forward(x.cuda(), w, nl)
    yc, ldc = R.flow_forward(x, flow)
    F = ops.ft_force(x.cuda(), w, nl, beta).cpu(); Fc = R.ft_force(x, flow, beta)
    d = ((y.cpu() - yc + math.pi) % (2 * math.pi) - math.pi).abs().max()
    out, grads = R.train_grads(x, flow, beta)
    r = ops.train_grad(x.cuda(), w, nl, beta)
    gw = ops.unpack_weight_grads(r['gw'], nl)
    ge = max(float((gw[li][pi].cpu() - grads[li][pi]).abs().max()) for li in range(nl) for pi in range(6))
    print(f'parity: fwd {float(d):.2e} logdet {float((ld.cpu() - ldc).abs().max()):.2e} force {float((F - Fc).abs().max()):.2e} wgrad {ge:.2e}', flush=True)
    assert d < 1e-10 and (F - Fc).abs().max() < 1e-8 and ge < 1e-9

def main():
    check()
    L = 64
    gen = torch.Generator().manual_seed(1331)
    w = ops.pack_weights(R.default_flow(1, gen), device='cuda')
    for B in (64, 128):
        x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
        for kind in ('flow_fwd', 'flow_bwd'):
            for act in ('silu', 'relu'):
                ts = sorted(ops.time_kernel(kind, x, w, mu=m, off=1, act=act, beta=6.0, reps=30) for m in (0, 1, 0, 1, 0, 1))
                fl = 1872 * L * L * B / 1e12
                print(f'B={B:3d} {kind} {act}: min {ts[0]*1e3:7.2f} us  median {ts[3]*1e3:7.2f} us  -> {fl / (ts[3]*1e-3) / 78.6 * 100:5.1f} % of fp64 peak', flush=True)
    x = ((torch.rand(128, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    names = {'flow_fwd': ['', 'plaq+sincos', 'conv1', 'conv2', 'conv3', 'transform1', 'finish+store'],
             'flow_bwd': ['', 'load+xform', 'conv3T', 'conv2T', 'conv1T', 'store']}
    for kind in ('flow_fwd', 'flow_bwd'):
        cyc = ops.profile_stages(kind, x, w, mu=0, off=1, beta=6.0)
        tot = sum(cyc[:7]) if kind == 'flow_fwd' else sum(cyc[:6])
        print(f'{kind}: total {tot:.0f} cycles/WG; ' + ', '.join(f'{n} {c:.0f}' for n, c in zip(names[kind][1:], cyc[1:]) if n), flush=True)
        if kind == 'flow_fwd':
            print('   extra stamps (slot k minus slot k-1):', ' '.join(f'{k}:{c:.0f}' for k, c in enumerate(cyc) if k >= 7), flush=True)

if __name__ == '__main__':
    main()
