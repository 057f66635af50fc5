"""fthmc/utils/qed_helpers.py on the HIP kernels: plaquettes, Wilson action, topological
charge, plain force / leapfrog / HMC, flow drivers, ft_action, ft_force, and the notebook's
physical-field ftHMC wrapper (ipynb/ft_hmc.py:420-513: ft_hmc, ft_run, flow_resize)."""
from __future__ import annotations

import time
from dataclasses import dataclass
from math import pi as PI
from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from .layers import flow_activation, flow_weights, weights_generation

TWO_PI = 2 * PI


def grab(var):
    return var.detach().cpu().numpy()


def regularize(f):
    """qed_helpers.py:40-42."""
    return ops.regularize(f)


def torch_mod(x):
    """qed_helpers.py:45-46: remainder(x, 2 pi) in [0, 2 pi) (not the flow's [-pi, pi) map)."""
    return torch.remainder(x, TWO_PI)


def torch_wrap(x):
    """qed_helpers.py:49-50: [-pi, pi)."""
    return ops.wrap(x)


def _batched(x):
    return x if x.dim() == 4 else x[None]


def compute_u1_plaq(links, mu=0, nu=1):
    """qed_helpers.py:80-90."""
    assert (mu, nu) == (0, 1)
    P = ops.plaquettes(_batched(links))
    return P if links.dim() == 4 else P[0]


def batch_plaqs(x, mu: int = 0, nu: int = 1):
    """qed_helpers.py:94-105."""
    return compute_u1_plaq(x, mu, nu)


def u1_plaq(x, mu: int, nu: int):
    """qed_helpers.py:66-70 (same plaquette, other summation order)."""
    return compute_u1_plaq(x, mu, nu)


def plaq_phase(f, mu=0, nu=1):
    """qed_helpers.py:246-257 (squeezes a leading batch of 1)."""
    f = torch.squeeze(f)
    return compute_u1_plaq(f, mu, nu)


def batch_charges(x: torch.Tensor = None, plaqs: torch.Tensor = None):
    """qed_helpers.py:108-116."""
    if plaqs is None:
        if x is None:
            raise ValueError('Either `x` or `plaq` must be specified.')
        return ops.wilson_action_charge(_batched(x), 1.0)[1]
    return ops.wrap(plaqs).flatten(1).sum(1) / TWO_PI


def topo_charge(x):
    """qed_helpers.py:73-77."""
    return batch_charges(x=x)


class _WilsonActionFn(torch.autograd.Function):
    """S_b(x) with its gradient dS_b/dx = fthmc_wilson_force (so flows trained through
    `action(x)` by autograd, train.py:202-210, see the Wilson term)."""

    @staticmethod
    def forward(ctx, x, beta):
        ctx.save_for_backward(x)
        ctx.beta = beta
        return ops.wilson_action_charge(x, beta)[0]

    @staticmethod
    def backward(ctx, gS):
        (x,) = ctx.saved_tensors
        return ops.wilson_force(x, ctx.beta) * gS[:, None, None, None], None


class BatchAction:
    """qed_helpers.py:166-186: S_b = -beta sum cos P."""

    def __init__(self, beta):
        self.beta = beta

    def __call__(self, x: torch.Tensor):
        if x.requires_grad:
            return _WilsonActionFn.apply(x, self.beta)
        return ops.wilson_action_charge(x, self.beta)[0]


@dataclass
class LatticeMetrics:
    beta: float
    plaqs: torch.Tensor
    action: torch.Tensor
    charges: torch.Tensor


class BatchObservables:
    """qed_helpers.py:130-163."""

    def __init__(self, beta: float = 1.):
        self.beta = beta

    def get_plaqs(self, x):
        return torch.cos(batch_plaqs(x))

    def get_action(self, x=None, plaqs=None):
        if x is None:
            raise ValueError('`x` must be specified.')
        return ops.wilson_action_charge(x, self.beta)[0]

    def get_charges(self, x=None, plaqs=None):
        return batch_charges(x=x, plaqs=plaqs)

    def get_observables(self, x):
        S, Q, _ = ops.wilson_action_charge(x, self.beta)
        return LatticeMetrics(self.beta, self.get_plaqs(x), S, Q)


# ---------------------------------------------------------------- flow drivers
def ft_flow(flow: nn.ModuleList, x: torch.Tensor):
    """qed_helpers.py:191-198: x -> F(x)."""
    return ops.flow_forward(x, flow_weights(flow, x.device), len(flow), flow_activation(flow))[0]


def ft_flow_inv(flow: nn.ModuleList, x: torch.Tensor, tol: float = 1e-12):
    """qed_helpers.py:201-209: x -> F^-1(x)."""
    return ops.flow_reverse(x, flow_weights(flow, x.device), len(flow), flow_activation(flow), tol=tol)[0]


def ft_action(param, flow, x: torch.Tensor):
    """qed_helpers.py:212-223: S_W(F(x)) - sum_l logJ_l, per chain."""
    return ops.ft_action(x, flow_weights(flow, x.device), len(flow), param.beta, flow_activation(flow))[0]


def ft_force(param, flow, field: torch.Tensor, create_graph=False):
    """qed_helpers.py:226-242: d(sum_b S_eff)/dx."""
    if create_graph:
        raise NotImplementedError('second-order graphs through the force are not part of the HIP path')
    return ops.ft_force(field, flow_weights(flow, field.device), len(flow), param.beta, flow_activation(flow))


# ---------------------------------------------------------------- plain HMC
def action(param, x: torch.Tensor):
    """qed_helpers.py:261-262: one scalar for the whole tensor."""
    return ops.wilson_action_charge(_batched(x), param.beta)[0].sum()


def force(param, x: torch.Tensor):
    """qed_helpers.py:265-272."""
    return ops.wilson_force(_batched(x), param.beta).reshape(x.shape)


def leapfrog(param, x: torch.Tensor, p: torch.Tensor, verbose: bool = True):
    """qed_helpers.py:275-295."""
    xo, po = ops.leapfrog(_batched(x), _batched(p), param.beta, param.dt, param.nstep)
    return xo.reshape(x.shape), po.reshape(p.shape)


def hmc(param, x, verbose=True, v: Optional[torch.Tensor] = None, u: Optional[torch.Tensor] = None):
    """qed_helpers.py:298-311: one trajectory; the whole tensor is ONE system (one H, one
    accept), as in the reference.  `v`, `u` may be supplied for reproducible checks."""
    xb = _batched(x)
    if v is None:
        v = torch.randn_like(xb)
    if u is None:
        u = torch.rand([], dtype=torch.float64, device=x.device)
    vb = _batched(v)
    if xb.shape[0] == 1:
        r = ops.hmc_trajectory(xb, vb, u.reshape(1), param.beta, param.dt, param.nstep)
        dH = r['dH'][0]
        acc = r['acc'][0] > 0.5
        return dH, torch.exp(-dH), acc, r['x_new'].reshape(x.shape)
    h0 = ops.wilson_action_charge(xb, param.beta)[0].sum() + 0.5 * ops.kinetic(vb).sum()
    x_, v_ = ops.leapfrog(xb, vb, param.beta, param.dt, param.nstep)
    xr = ops.regularize(x_)
    dH = ops.wilson_action_charge(xr, param.beta)[0].sum() + 0.5 * ops.kinetic(v_).sum() - h0
    exp_mdH = torch.exp(-dH)
    acc = u < exp_mdH
    newx = xr if bool(acc) else xb
    return dH, exp_mdH, acc, newx.reshape(x.shape)


# ---------------------------------------------------------------- ftHMC on the physical field
def ft_hmc(param, flow, field: torch.Tensor, v: Optional[torch.Tensor] = None, u: Optional[torch.Tensor] = None,
           tol: float = 1e-12):
    """ipynb/ft_hmc.py:420-435: one ftHMC trajectory that starts and ends on the PHYSICAL field:
    x = F^-1(field); momenta; leapfrog in the latent field with ft_force (`:394-418`); xr = regularize(x_);
    accept on u < exp(-dH); return F(newx).  Like the notebook the whole tensor is ONE system (one H,
    one accept; the notebook always passes [1, 2, L, L]).  -> (dH, exp_mdH, acc, newfield).

    `v`, `u` may be supplied for reproducible checks (the notebook draws randn_like(x), then rand([])).
    `tol`: tolerance of the inverse flow (the reference bisects to a global 1e-6)."""
    fb = _batched(field)
    w, nl, act = flow_weights(flow, fb.device), len(flow), flow_activation(flow)
    x = ops.flow_reverse(fb, w, nl, act, tol=tol)[0]
    if v is None:
        v = torch.randn_like(x)
    if u is None:
        u = torch.rand([], dtype=torch.float64, device=x.device)
    vb = _batched(v)
    if x.shape[0] == 1:
        r = ops.ft_trajectory(x, vb, u.reshape(1), w, nl, param.beta, param.dt, param.nstep, act, mode='md')
        dH, acc, newx = r['dH'][0], r['acc'][0] > 0.5, r['x_new']
    else:
        h0 = ops.ft_action(x, w, nl, param.beta, act)[0].sum() + 0.5 * ops.kinetic(vb).sum()
        x_, v_ = ops.ft_leapfrog(x, vb, w, nl, param.beta, param.dt, param.nstep, act)
        xr = ops.regularize(x_)
        dH = ops.ft_action(xr, w, nl, param.beta, act)[0].sum() + 0.5 * ops.kinetic(v_).sum() - h0
        acc = u < torch.exp(-dH)
        newx = xr if bool(acc) else x
    exp_mdH = torch.exp(-dH)
    newfield = ops.flow_forward(newx, w, nl, act)[0]
    return float(dH), float(exp_mdH), acc, newfield.reshape(field.shape)


def ft_run(param, flow, field: Optional[torch.Tensor] = None, logfile: Optional[str] = None, verbose: bool = False,
           use_graph: Optional[bool] = None):
    """ipynb/ft_hmc.py:437-487: `param.nrun` x `param.ntraj` physical-field ftHMC trajectories of one
    configuration [2, L, L].  Returns (field, history) with per-trajectory dH, exp_mdH, acc, plaq, topo
    (the notebook returns the field and keeps the histories in module globals); the status lines it prints
    go to `logfile` when given.

    On the device the trajectory sequence (inverse flow, momenta and the uniform from torch's generator, the fused trajectory,
    forward flow, observables) is captured once in a hipGraph and replayed (`use_graph`, default on; FTHMC_RUN_GRAPH=0 turns it
    off): the notebook's loop synchronises five times per trajectory, this one not at all; the histories and the log are
    written when the loop has run.  Same draws, identical numbers."""
    import os
    if field is None:
        field = param.initializer()[0]
    history = {k: [] for k in ('dH', 'exp_mdH', 'acc', 'plaq', 'topo')}
    out = open(logfile, 'w') if logfile else None
    if use_graph is None:
        use_graph = os.environ.get('FTHMC_RUN_GRAPH', '1') not in ('', '0')

    def put(s):
        if out is not None:
            out.write(s)
        if verbose:
            print(s, end='', flush=True)

    def line(k, dH, exp_mdH, acc, plaq, topo):
        return (f'Traj: {k:4}  {"ACCEPT" if acc else "REJECT"}  dH: {dH:< 12.8}  '
                f'exp(-dH): {exp_mdH:< 12.8}  plaq: {plaq:< 12.8}  topo: {topo:< 3.3}\n')
    try:
        S, Q, plaq = ops.wilson_action_charge(_batched(field), param.beta)
        put(f'Initial configuration:  plaq: {float(plaq[0])}  topo: {float(Q[0])} {tuple(field.shape)}\n')
        ntot = param.nrun * param.ntraj
        if use_graph and field.is_cuda and ntot > 0:
            field, rows = _ft_run_captured(param, flow, field, ntot)
            for k, (dH, acc, plaq_, topo, exp_mdH) in enumerate(rows.tolist()):
                for key, val in zip(history, (dH, exp_mdH, float(acc > 0.5), plaq_, topo)):
                    history[key].append(val)
                put(line(k + 1, dH, exp_mdH, acc > 0.5, plaq_, topo))
        else:
            for n in range(param.nrun):
                for i in range(param.ntraj):
                    dH, exp_mdH, acc, field_run = ft_hmc(param, flow, field.reshape((1,) + tuple(field.shape[-3:])))
                    field = field_run[0]
                    S, Q, plaq = ops.wilson_action_charge(field_run, param.beta)
                    for k, val in zip(history, (dH, exp_mdH, float(bool(acc)), float(plaq[0]), float(Q[0]))):
                        history[k].append(val)
                    put(line(n * param.ntraj + i + 1, dH, exp_mdH, bool(acc), float(plaq[0]), float(Q[0])))
    finally:
        if out is not None:
            out.close()
    return field, history


def _ft_run_captured(param, flow, field: torch.Tensor, ntot: int, tol: float = 1e-12):
    """the loop of ft_run as one captured sequence per trajectory -> (final field [2, L, L], rows [ntot, 5] on the host:
    dH, acc, plaq, Q, exp(-dH))"""
    from ..graph_loop import GraphLoop
    dev = field.device
    fs = field.detach().reshape((1,) + tuple(field.shape[-3:])).clone()
    L = fs.shape[-1]
    w, nl, act = flow_weights(flow, dev), len(flow), flow_activation(flow)
    # the weights do not change inside this loop: every launch states one content version, the library expands them once
    # (the first call) and checks the stamp on the device afterwards (include/fthmc_hip.h "Weight versions")
    wkey = ('ft_run', id(flow), w.data_ptr(), w._version, weights_generation(w), time.monotonic_ns())
    v = torch.empty_like(fs)
    u = torch.empty(1, dtype=torch.float64, device=dev)
    row = torch.empty(5, dtype=torch.float64, device=dev)
    res = {'x_new': torch.empty_like(fs), 'dH': row[0:1], 'acc': row[1:2], 'plaq': torch.empty(1, dtype=torch.float64, device=dev),
           'Q': torch.empty(1, dtype=torch.float64, device=dev), 'H0': torch.empty(1, dtype=torch.float64, device=dev),
           'H1': torch.empty(1, dtype=torch.float64, device=dev)}
    obs = {'S': torch.empty(1, dtype=torch.float64, device=dev), 'Q': row[3:4], 'plaq': row[2:3]}

    def enqueue():
        x = ops.flow_reverse(fs, w, nl, act, tol=tol, wkey=wkey)[0]      # x = F^-1(field)
        v.normal_()                                                      # randn_like(x), then rand([]): ipynb/ft_hmc.py:423-424
        u.uniform_()
        ops.ft_trajectory(x, v, u, w, nl, param.beta, param.dt, param.nstep, act, mode='md', out=res, wkey=wkey)
        fs.copy_(ops.flow_forward(res['x_new'], w, nl, act, wkey=wkey)[0])   # newfield = F(newx)
        ops.wilson_action_charge(fs, param.beta, out=obs)
        torch.exp(torch.neg(row[0:1]), out=row[4:5])
    loop = GraphLoop(enqueue, row, use_graph=True)
    for k in range(ntot):
        loop.step()
    rows = loop.rows()
    loop.join()
    return fs[0].clone(), rows


def flow_resize(flow: nn.ModuleList, lat_new):
    """ipynb/ft_hmc.py:489-513: the same (translation-equivariant) conv nets with masks of another lattice."""
    from .layers import get_nets, make_net_from_layers
    return make_net_from_layers(lattice_shape=tuple(lat_new), nets=get_nets(flow))
