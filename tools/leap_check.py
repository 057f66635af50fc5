#!/usr/bin/env python3
"""Row-strip leapfrog kernel (k_leap_rows, L % 64 == 0) against the 16 x 16-tile kernel (k_force<1>): same results, and
the achieved algorithmic bandwidth (64 B per site and step) of both."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fthmc_amd import ops


def use_rows(on):
    os.environ['FTHMC_LEAP_ROWS'] = '1' if on else '0'
    ops.set_variant(1)


gen = torch.Generator().manual_seed(5)
for (B, L) in [(3, 64), (2, 128), (2, 256), (5, 192)]:
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    p = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64).cuda()
    use_rows(True); xa, pa = ops.leapfrog(x, p, 3.0, 0.1, 5)
    use_rows(False); xb, pb = ops.leapfrog(x, p, 3.0, 0.1, 5)
    print(f'B={B} L={L}: |dx| {float((xa - xb).abs().max()):.2e} |dp| {float((pa - pb).abs().max()):.2e}', flush=True)
for (B, L) in [(128, 64), (1024, 64), (32, 256)]:
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    for on in (True, False):
        use_rows(on)
        ms = ops.time_kernel('leap_step', x, beta=6.0, reps=50)
        print(f'B={B} L={L} rows={on}: {ms * 1e3:.2f} us  {64.0 * L * L * B / (ms * 1e-3) / 1e12:.2f} TB/s algorithmic', flush=True)
use_rows(True)
