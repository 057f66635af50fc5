"""Rehearsal of the captured training step under a one-rank nccl group (FTHMC_FORCE_PG=1) with eager asynchronous collectives of
the DEFAULT group in flight right up to the capture (the C1 traffic of a sampler next to a trainer; the captured C2 collectives
run on their own group, parallel.capture_group): python3 tools/pg_capture_probe.py L B n_layers steps [eager collectives in flight]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fthmc_amd import parallel, train as T
from fthmc_amd.config import TrainConfig
L, B, nl, steps = (int(a) for a in sys.argv[1:5])
NFLY = int(sys.argv[5]) if len(sys.argv) > 5 else 200
parallel.init()
tc = TrainConfig(L=L, beta=4.0, n_layers=nl, batch_size=B, base_lr=1e-3, print_freq=0)
torch.manual_seed(3)
model = T.get_model(tc)
tr = T.GraphTrainer(model, tc, T.make_optimizer(model, tc), B, seed=5)
works = []
if parallel.have_group():
    # eager asynchronous collectives in flight when the first step() captures: behind a long-running kernel on their stream, so
    # that none of them has finished (and the watchdog is polling every one of them) while the capture runs
    side = torch.cuda.Stream()
    t = torch.ones(4, dtype=torch.float64, device='cuda')
    big = torch.zeros(1 << 26, dtype=torch.float64, device='cuda')
    with torch.cuda.stream(side):
        for _ in range(40):
            big.add_(1.0)
        for _ in range(NFLY):
            works.append(torch.distributed.all_reduce(t, async_op=True))
# ... and a second thread that KEEPS issuing them while the first step() runs its eager pass, synchronises and captures
# (a sampler's C1 next to a trainer): new work reaches the watchdog's list during the capture itself
import threading
stop, issued = threading.Event(), [0]


def c1_traffic():
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    tt = torch.ones(4, dtype=torch.float64, device='cuda')
    mine = []
    with torch.cuda.stream(st):
        while not stop.is_set():
            mine.append(torch.distributed.all_reduce(tt, async_op=True)); issued[0] += 1
            if len(mine) >= 64:
                mine.pop(0).wait()
            time.sleep(0.0005)
        for wk in mine:
            wk.wait()


th = threading.Thread(target=c1_traffic) if parallel.have_group() else None
if th is not None:
    th.start()
for _ in range(3):
    tr.step()
tr.synchronize()
if th is not None:
    stop.set(); th.join()
for wk in works:
    wk.wait()
tr.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step()
tr.synchronize()
dt = (time.perf_counter() - t0) / steps * 1e3
print(json.dumps({'group': parallel.have_group(), 'eager_collectives_in_flight': len(works), 'issued_by_the_second_thread_meanwhile': issued[0], 'captured': tr.captured, 'capture_error': getattr(tr, 'capture_error', None),
                  'ms_per_step': round(dt, 4), 'loss': float(tr.metrics()['loss_dkl'])}), flush=True)
if parallel.have_group():
    torch.distributed.destroy_process_group()
