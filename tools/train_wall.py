#!/usr/bin/env python3
"""Wall time of a whole training step (prior draw, fthmc_train_grad, metrics, Adam) on the GPU box:
the step-by-step `train_step` (one host synchronisation per step) against `GraphTrainer` (one captured hipGraph per
step, no synchronisation), at the shapes the reference trains at (L = 8, 16) and at the config-5 shard.
    python3 tools/train_wall.py [L B n_layers [steps]] ..."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fthmc_amd import ops, train as T
from fthmc_amd.config import TrainConfig
from fthmc_amd.utils import qed_helpers as qed

def wall(L, B, nl, steps):
    beta = {8: 2.0, 16: 4.0}.get(L, 7.0)
    tc = TrainConfig(L=L, beta=beta, n_layers=nl, batch_size=B, base_lr=1e-3, print_freq=0)
    out = {'L': L, 'batch': B, 'n_layers': nl, 'steps': steps}
    torch.manual_seed(3)
    # (a) train_step, step by step (Adam as the reference builds it, train.py:297)
    model = T.get_model(tc)
    opt = torch.optim.Adam(model.layers.parameters(), lr=tc.base_lr)
    act = qed.BatchAction(beta)
    for _ in range(3):
        T.train_step(model, tc, act, opt, B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        T.train_step(model, tc, act, opt, B)
    torch.cuda.synchronize(); out['train_step_ms'] = (time.perf_counter() - t0) / steps * 1e3
    # (b) GraphTrainer, eager and captured
    for name, graph in (('trainer_eager_ms', False), ('trainer_graph_ms', True)):
        model = T.get_model(tc)
        opt = T.make_optimizer(model, tc)
        tr = T.GraphTrainer(model, tc, opt, B, use_graph=graph)
        for _ in range(3):
            tr.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            tr.step()
        torch.cuda.synchronize(); out[name] = (time.perf_counter() - t0) / steps * 1e3
        m = tr.metrics()
        out[name.replace('_ms', '_loss')] = float(m['loss_dkl'])
    # (c) compute only: fthmc_train_grad on a fixed draw
    xi = model.prior.sample_n(B)
    w = qed.flow_weights(model.layers, xi.device)
    for G in (ops.default_groups(B, L), 1, 2):
        for _ in range(3):
            ops.train_grad(xi, w, nl, beta, groups=G)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            ops.train_grad(xi, w, nl, beta, groups=G)
        torch.cuda.synchronize()
        out['train_grad_only_ms' if G == ops.default_groups(B, L) and 'train_grad_only_ms' not in out else f'train_grad_groups{G}_ms'] = (time.perf_counter() - t0) / steps * 1e3
    ops.release_workspaces()
    return {k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}

if __name__ == '__main__':
    a = [int(t) for t in sys.argv[1:]]
    shapes = [tuple(a[i:i + 4]) for i in range(0, len(a), 4)] if a else [(8, 512, 8, 200), (16, 512, 8, 200), (16, 64, 8, 200), (256, 32, 16, 10)]
    for sh in shapes:
        print(json.dumps(wall(*sh)), flush=True)
