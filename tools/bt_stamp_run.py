import math, os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from fthmc_amd import ops
import bench
L, B, nl = 256, 32, 2
gen = torch.Generator().manual_seed(1)
w = ops.pack_weights(bench.make_flow(gen, nl), device='cuda')
xi = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
for _ in range(2):
    r = ops.train_grad(xi, w, nl, 4.0, groups=1)
torch.cuda.synchronize()
