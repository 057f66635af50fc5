"""Stage-0 timeline of k_flow_bwd_gather from a -DFT_DIAG2 build (FTHMC_LIB=experiments/lib_diag2.so): mean cycles since the
workgroup's first stamp.  usage: FTHMC_LIB=... python3 tools/bwd_stage0.py [B ...]"""
import math, os, sys, torch
sys.path.insert(0, os.getcwd())
from fthmc_amd import ops
from oracle import ref_cpu as R
L = 64
gen = torch.Generator().manual_seed(1331)
w = ops.pack_weights(R.default_flow(1, gen), device='cuda')
for B in [int(a) for a in sys.argv[1:]] or (16, 64, 128):
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    for mu in (0, 1):
        cyc = ops.profile_stages('flow_bwd', x, w, mu=mu, off=1, beta=6.0)
        print(f'B={B:3d} mu={mu}: stages', ' '.join(f'{c:.0f}' for c in cyc[1:6]),
              '| wave0: weights issued', f'{cyc[6]:.0f}', 'all loads issued', f'{cyc[7]:.0f}', 'weights in LDS', f'{cyc[8]:.0f}',
              'at barrier', f'{cyc[10]:.0f}', '| wave7: tc/ag/cs landed', f'{cyc[9]:.0f}', 'at barrier', f'{cyc[11]:.0f}', 'all its loads landed', f'{cyc[12]:.0f}', flush=True)
