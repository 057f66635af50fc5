"""Rehearsal of the captured training step under a one-rank nccl group (FTHMC_FORCE_PG=1): python3 tools/pg_capture_probe.py L B n_layers steps"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fthmc_amd import parallel, train as T
from fthmc_amd.config import TrainConfig
L, B, nl, steps = (int(a) for a in sys.argv[1:5])
parallel.init()
tc = TrainConfig(L=L, beta=4.0, n_layers=nl, batch_size=B, base_lr=1e-3, print_freq=0)
torch.manual_seed(3)
model = T.get_model(tc)
# some eager collectives first, as bench.py has issued by the time it builds a trainer
if parallel.have_group():
    t = torch.ones(4, dtype=torch.float64, device='cuda')
    for _ in range(20):
        torch.distributed.all_reduce(t, async_op=True)
tr = T.GraphTrainer(model, tc, T.make_optimizer(model, tc), B, seed=5)
for _ in range(3):
    tr.step()
tr.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step()
tr.synchronize()
dt = (time.perf_counter() - t0) / steps * 1e3
print(json.dumps({'group': parallel.have_group(), 'captured': tr.captured, 'capture_error': getattr(tr, 'capture_error', None),
                  'ms_per_step': round(dt, 4), 'loss': float(tr.metrics()['loss_dkl'])}), flush=True)
if parallel.have_group():
    torch.distributed.destroy_process_group()
