"""fthmc/train.py on the HIP path: reverse-KL training of the flow.

`train_step` keeps the reference signature.  Its compute (flow forward, Wilson action, backward
wrt every conv weight) is one fused call, `ops.train_grad` (C ABI fthmc_train_grad); Adam /
ReduceLROnPlateau stay torch optimizers acting on the `nn.Conv2d` parameters.  With more than
one rank the weight gradients and the ESS / loss pieces are all-reduced (fthmc_amd.parallel C2).
"""
from __future__ import annotations

import os
import time
from math import pi as PI
from typing import Any, Callable, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from . import ops, parallel
from .config import DTYPE, FlowModel, Param, TrainConfig, device
from .utils import qed_helpers as qed
from .utils.distributions import MultivariateUniform, calc_dkl, calc_ess
from .utils.layers import (flow_activation, flow_weights, get_nets, make_net_from_layers, make_u1_equiv_layers,
                           net_weights, set_weights)
from .utils.samplers import apply_flow_to_prior

TWO_PI = 2 * PI


def grab(x: torch.Tensor):
    return x.detach().cpu().numpy()


def get_model(config: TrainConfig) -> FlowModel:
    """train.py:57-74: U(-pi, pi) prior + n_layers coupling layers (Conv2d default init: the
    reference's set_weights(layers) call is a no-op)."""
    dev = device()
    prior = MultivariateUniform(-PI * torch.ones((2, *config.lat), dtype=DTYPE, device=dev),
                                PI * torch.ones(tuple(config.lat), dtype=DTYPE, device=dev))
    layers = make_u1_equiv_layers(lattice_shape=tuple(config.lat), n_layers=config.n_layers,
                                  n_mixture_comps=config.n_s_nets, hidden_sizes=config.hidden_sizes,
                                  kernel_size=config.kernel_size, activation_fn=config.activation_fn)
    set_weights(layers)
    return FlowModel(prior=prior, layers=layers)


def _plain(obj):
    """numpy arrays / scalars inside a history -> tensors / Python numbers, so that the checkpoint unpickles with
    `weights_only=True` (no arbitrary code on load)."""
    import numpy as np
    if isinstance(obj, dict):
        return {k: _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_plain(v) for v in obj)
    if isinstance(obj, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(obj))
    if isinstance(obj, np.generic):
        return obj.item()
    return obj


def restore_model_from_checkpoint(infile, train_config: TrainConfig, trusted: bool = False):
    """train.py:77-92 (same checkpoint dict keys as io.save_checkpoint, io.py:148-170).  Checkpoints written by
    `save_checkpoint` here hold tensors and plain numbers only and load with `weights_only=True`.  A checkpoint
    written by the reference pickles numpy arrays in its history: pass `trusted=True` for such a file -- unpickling
    it can run arbitrary code, so only for files you wrote yourself."""
    try:
        checkpoint = torch.load(infile, map_location=device(), weights_only=True)
    except Exception as e:                                      # pickle.UnpicklingError and friends
        if not trusted:
            raise RuntimeError(f'{infile} does not load with weights_only=True ({type(e).__name__}); if you wrote '
                               f'this file yourself, call restore_model_from_checkpoint(..., trusted=True)') from e
        checkpoint = torch.load(infile, map_location=device(), weights_only=False)
    model = get_model(train_config)
    optimizer = optim.AdamW(model.layers.parameters(), lr=train_config.base_lr, weight_decay=1e-5)
    model.layers.load_state_dict(checkpoint['model_state_dict'])
    optimizer.load_state_dict(checkpoint['optimizer_state_dict'])
    return {'model': model, 'optimizer': optimizer}


def save_checkpoint(era: int, epoch: int, model: nn.Module, optimizer, history: dict, outdir: str):
    """io.py:114-172: ckpt-era{e}-epoch{n}.tar with the reference's keys."""
    os.makedirs(outdir, exist_ok=True)
    path = os.path.join(outdir, f'ckpt-era{era}-epoch{epoch}.tar')
    torch.save({'era': era, 'epoch': epoch, 'model_state_dict': model.state_dict(),
                'optimizer_state_dict': optimizer.state_dict(), 'history': _plain(history)}, path)
    return path


ActionFn = Callable[[torch.Tensor], torch.Tensor]


def train_step(model: FlowModel, config: TrainConfig, action: ActionFn, optimizer: optim.Optimizer,
               batch_size: int, scheduler: Any = None, scaler: Any = None, pre_model: FlowModel = None,
               dkl_factor: float = 1., xi: torch.Tensor = None, fused: bool = True):
    """train.py:162-228.  `batch_size` is this rank's share of the global batch.

    fused=True : one HIP call computes x, logq, logp and d(loss)/d(weights) (needs `action` to be
                 the Wilson `BatchAction(config.beta)`, which is what train.py:291 passes);
    fused=False: the layers run one by one through autograd (any `action` callable)."""
    t0 = time.time()
    optimizer.zero_grad()
    if scaler is not None:
        raise NotImplementedError('GradScaler (fp16 autocast) does not apply to the fp64 HIP path')
    if pre_model is not None:
        pre_xi = pre_model.prior.sample_n(batch_size)
        x_pre = qed.ft_flow(pre_model.layers, pre_xi)
        xi = qed.ft_flow_inv(pre_model.layers, x_pre)
    layers = model.layers
    world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    n_global = batch_size * world
    if fused and isinstance(action, qed.BatchAction):
        if xi is None:
            xi = model.prior.sample_n(batch_size)
        xi = xi.to(DTYPE)
        r = ops.train_grad(xi, flow_weights(layers, xi.device), len(layers), action.beta, flow_activation(layers),
                           groups=ops.default_groups(xi.shape[0], xi.shape[-1]))
        x, logq, logp = r['x'], r['logq'], r['logp']
        gw = r['gw'] * (dkl_factor / world)          # kernel seeds 1/B_local; loss is the global mean
        parallel.allreduce_grads(gw)
        for layer, gl in zip(layers, ops.unpack_weight_grads(gw, len(layers), arch=ops.arch_of(r['gw']))):
            for p, g in zip(net_weights(layer.plaq_coupling.net), gl):
                p.grad = g.clone()
        loss_dkl = dkl_factor * parallel.global_mean(logq - logp, n_global)
    else:
        x, xi, logq = apply_flow_to_prior(model.prior, layers, xi=xi, batch_size=batch_size)
        logp = (-1.) * action(x)
        loss_local = dkl_factor * (logq - logp).sum() / n_global
        loss_local.backward()
        if world > 1:
            for p in layers.parameters():
                parallel.allreduce_grads(p.grad)
        loss_dkl = dkl_factor * parallel.global_mean((logq - logp).detach(), n_global)
    logw = (logp - logq).detach()
    ess = torch.exp(2 * parallel.global_logsumexp(logw) - parallel.global_logsumexp(2 * logw)) / n_global
    qi = qed.batch_charges(xi)
    q = qed.batch_charges(x.detach())
    plaq = logp.detach() / (config.beta * config.volume)
    dq = torch.sqrt((q - qi) ** 2)
    optimizer.step()
    if scheduler is not None:
        scheduler.step(loss_dkl)
    return {'dt': time.time() - t0, 'ess': grab(ess), 'logp': grab(logp), 'logq': grab(logq),
            'loss_dkl': grab(loss_dkl), 'q': grab(q), 'dq': grab(dq), 'plaq': grab(plaq)}


def train(config: TrainConfig, model: Optional[FlowModel] = None, pre_model: FlowModel = None,
          figsize=None, dpi: int = 120, scheduler_config=None, dkl_factor: float = 1., save: bool = False,
          verbose: bool = True):
    """train.py:236-431 without plots / tensorboard: n_era x n_epoch steps, one checkpoint per era
    (save=True).  Returns dict(model, optimizer, history, ckpt_files)."""
    if model is None:
        model = get_model(config)
    optimizer = optim.Adam(model.layers.parameters(), lr=config.base_lr)
    scheduler = None
    if scheduler_config is not None:
        sc = {k: v for k, v in vars(scheduler_config).items() if k != 'verbose'}
        scheduler = optim.lr_scheduler.ReduceLROnPlateau(optimizer, **sc)
    action = qed.BatchAction(config.beta)
    history, ckpts = {}, []
    step = 0
    for era in range(config.n_era):
        for epoch in range(config.n_epoch):
            metrics = train_step(model, config, action, optimizer, config.batch_size, scheduler=scheduler,
                                 pre_model=pre_model, dkl_factor=dkl_factor)
            for k, v in metrics.items():
                history.setdefault(k, []).append(v)
            if verbose and config.print_freq and step % config.print_freq == 0:
                print(f"era {era} epoch {epoch}: loss_dkl={float(metrics['loss_dkl']):.4f} "
                      f"ess={float(metrics['ess']):.4f} plaq={float(np.mean(metrics['plaq'])):.5f}", flush=True)
            step += 1
        if save:
            ckpts.append(save_checkpoint(era, config.n_epoch, model.layers, optimizer, history,
                                         config.update_logdirs(config.logdir)['ckpts']))
    return {'model': model, 'optimizer': optimizer, 'history': history, 'ckpt_files': ckpts, 'action': action}


def transfer_to_new_lattice(L: int, layers: nn.ModuleList, param_init: Param = None):
    """train.py:434-455: reuse the (translation-equivariant) conv nets on an L x L lattice."""
    lattice_shape = (L, L)
    dev = device()
    prior = MultivariateUniform(-PI * torch.ones((2, *lattice_shape), dtype=DTYPE, device=dev),
                                PI * torch.ones(lattice_shape, dtype=DTYPE, device=dev))
    new_layers = make_net_from_layers(nets=get_nets(layers), lattice_shape=lattice_shape)
    return FlowModel(prior=prior, layers=new_layers)
