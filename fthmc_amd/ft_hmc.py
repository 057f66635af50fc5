"""fthmc/ft_hmc.py on the HIP path: `FieldTransformation` (HMC in the latent field of a trained
flow) and `run_ftHMC`.  Same constructor, methods and metric keys as the reference; no
tensorboard / plotting side effects.

Two reference quirks are explicit switches, defaults = what the method describes:
  leapfrog_mode = 'md'               real MD integrator (ipynb/ft_hmc.py:394-418);
                  'reference_literal' reproduces ft_hmc.py:180-188 (evolution discarded:
                                      proposal = x + dt/2 v with the initial v)
  energy_mode   = 'per_chain'        H_b = S_eff,b + v_b^2 / 2;
                  'reference_literal' reproduces calc_energy ft_hmc.py:177-178
                                      (no 1/2, kinetic term summed over the whole batch)
"""
from __future__ import annotations

import time
from math import pi as PI
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .config import DTYPE, TrainConfig, device, lfConfig
from .utils import qed_helpers as qed
from .utils.layers import flow_activation, flow_weights

TWO_PI = 2. * PI


class FieldTransformation(nn.Module):
    def __init__(self, flow: nn.ModuleList, config: TrainConfig, lfconfig: lfConfig,
                 leapfrog_mode: str = 'md', energy_mode: str = 'per_chain'):
        super().__init__()
        assert leapfrog_mode in ('md', 'reference_literal') and energy_mode in ('per_chain', 'reference_literal')
        self.flow = flow
        self.config = config
        self.lfconfig = lfconfig
        self.dt, self.tau, self.nstep = lfconfig.dt, lfconfig.tau, lfconfig.nstep
        self.leapfrog_mode, self.energy_mode = leapfrog_mode, energy_mode
        self._denom = self.config.beta * self.config.volume
        self._w = None
        self._w_versions = None
        self._params = None
        self._carry = None            # (field tensor, its version, weights, state [3, B]) of the last batch trajectory's result

    # ---- weights: packed once, refreshed when a parameter changed in place -----------
    def weights(self, dev) -> torch.Tensor:
        # the parameter list is gathered once (Module.parameters() walks the module tree: it was two thirds of the host time of a
        # trajectory at L = 16); a flow whose modules are swapped afterwards wants a new FieldTransformation
        if self._params is None:
            self._params = list(self.flow.parameters())
        vers = tuple(p._version for p in self._params)
        if self._w is None or self._w_versions != vers or self._w.device != dev:
            self._w = flow_weights(self.flow, dev)
            self._w_versions = vers
        return self._w

    @property
    def _act(self):
        return flow_activation(self.flow)

    # ---- reference methods ----------------------------------------------------------
    def action(self, x: torch.Tensor):
        """ft_hmc.py:135-141: S_W(F(x)) - sum logJ, per chain."""
        return ops.ft_action(x, self.weights(x.device), len(self.flow), self.config.beta, self._act)[0]

    def flow_forward(self, x: torch.Tensor):
        """ft_hmc.py:143-150 -> (F(x), logdet)."""
        return ops.flow_forward(x, self.weights(x.device), len(self.flow), self._act)

    def flow_backward(self, x: torch.Tensor, tol: float = 1e-12):
        """ft_hmc.py:152-160 -> (F^-1(x), logdet)."""
        return ops.flow_reverse(x, self.weights(x.device), len(self.flow), self._act, tol=tol)

    def force(self, x: torch.Tensor):
        """ft_hmc.py:162-171 (the reference only works for B = 1; any B here)."""
        return ops.ft_force(x.detach(), self.weights(x.device), len(self.flow), self.config.beta, self._act)

    @staticmethod
    def wrap(x: torch.Tensor):
        """ft_hmc.py:173-175."""
        return ops.wrap(x)

    def calc_energy(self, x: torch.Tensor, v: torch.Tensor):
        """ft_hmc.py:177-178 (literal) or the per-chain Hamiltonian."""
        if self.energy_mode == 'reference_literal':
            return self.action(x) + ops.kinetic(v).sum()
        return self.action(x) + 0.5 * ops.kinetic(v)

    def leapfrog(self, x: torch.Tensor, v: torch.Tensor):
        """ft_hmc.py:180-188."""
        if self.leapfrog_mode == 'reference_literal':
            return x + 0.5 * self.dt * v, v
        return ops.ft_leapfrog(x, v, self.weights(x.device), len(self.flow), self.config.beta, self.dt,
                               self.nstep, self._act)

    def _mode(self):
        return 'md' if self.leapfrog_mode == 'md' else 'literal'

    def hmc(self, x: torch.Tensor, step: int = None, v: Optional[torch.Tensor] = None,
            u: Optional[torch.Tensor] = None):
        """ft_hmc.py:190-224: the tensor is one system (one scalar H, one accept).  For the usual
        [1, 2, L, L] field this is a single fused trajectory launch sequence."""
        x = x.to(device()) if not x.is_cuda else x
        t0 = time.time()
        metrics = {}
        if step is not None:
            metrics['traj'] = step
        if v is None:
            v = torch.randn_like(x)
        if u is None:
            u = torch.rand([], dtype=torch.float64, device=x.device)
        if x.shape[0] == 1:
            w = self.weights(x.device)
            c = self._carry                                              # see _batch_hmc
            state = c[3] if c is not None and c[0] is x and c[1] == x._version and c[2] is w else None
            r = ops.ft_trajectory(x, v, u.reshape(1), w, len(self.flow), self.config.beta,
                                  self.dt, self.nstep, self._act, mode=self._mode(), state_in=state)
            # the packaged code maps the end point with wrap, the notebook with regularize: same set
            xnew, acc, dh = r['x_new'], r['acc'][0] > 0.5, r['dH'][0]
            self._carry = (xnew, xnew._version, w, r['state'])
            metrics.update({'_plaq': r['plaq'], '_q': r['Q']})
        else:
            h0 = self.action(x).sum() + 0.5 * ops.kinetic(v).sum()
            x_, v_ = self.leapfrog(x, v)
            x_ = self.wrap(x_)
            dh = self.action(x_).sum() + 0.5 * ops.kinetic(v_).sum() - h0
            acc = u < torch.exp(-dh)
            xnew = x_ if bool(acc) else x
        metrics.update({'dt': time.time() - t0, 'acc': acc.detach(), 'dh': dh.detach()})
        return xnew, metrics

    def _batch_hmc(self, x: torch.Tensor, step: int = None, v: Optional[torch.Tensor] = None,
                   u: Optional[torch.Tensor] = None):
        """ft_hmc.py:226-257: independent chains, per-chain accept (dead code in the reference
        for B > 1 because its force only works for B = 1)."""
        t0 = time.time()
        metrics = {}
        if step is not None:
            metrics['traj'] = step
        if v is None:
            v = torch.randn_like(x)
        if u is None:
            u = torch.rand(x.shape[0], dtype=torch.float64, device=x.device)
        if self.energy_mode == 'per_chain':
            # (S_eff, plaq, Q) of x is carried over when x IS the field the previous call returned, untouched, under the same
            # weights: that call's H1 sweep computed exactly what this call's H0 sweep would (bit-identical; bench.py's
            # `stateless` figure is the price of recomputing it, as the reference does at ft_hmc.py:205)
            w = self.weights(x.device)
            c = self._carry
            state = c[3] if c is not None and c[0] is x and c[1] == x._version and c[2] is w else None
            r = ops.ft_trajectory(x, v, u, w, len(self.flow), self.config.beta, self.dt,
                                  self.nstep, self._act, mode=self._mode(), state_in=state,
                                  groups=ops.default_groups(x.shape[0], x.shape[-1]))
            x_, dh, acc = r['x_new'], r['dH'], r['acc']
            x_ = x_.detach()
            self._carry = (x_, x_._version, w, r['state'])
            # plaq / Q of the flowed accepted field come with the trajectory: run() need not flow x again for its metrics
            metrics.update({'_plaq': r['plaq'], '_q': r['Q']})
        else:
            h = self.calc_energy(x, v)
            xp, v_ = self.leapfrog(x, v)
            xp = self.wrap(xp)
            dh = self.calc_energy(xp, v_) - h
            acc = (u < torch.exp(-dh)).to(DTYPE)
            x_ = torch.where(acc[:, None, None, None] > 0.5, xp, x)
        metrics.update({'dt': time.time() - t0, 'acc': acc, 'dh': dh, 'exp_mdh': torch.exp(-dh)})
        return x_.detach() if x_.requires_grad else x_, metrics

    def initializer(self, rand: bool = True):
        """ft_hmc.py:259-264: U(0, 2 pi) latent start."""
        x = torch.zeros([self.config.nd] + self.config.lat, dtype=DTYPE, device=device())
        if rand:
            x = x.uniform_(0, TWO_PI)
        return x[None, :]

    def lattice_metrics(self, x: torch.Tensor, qold: torch.Tensor):
        """ft_hmc.py:266-270."""
        S, q, p = ops.wilson_action_charge(x, self.config.beta)
        return {'plaq': p, 'q': q, 'dq': torch.sqrt((q - qold) ** 2)}

    def run(self, x: torch.Tensor = None, nprint: int = 25, nplot: int = 25, window: int = 10,
            num_trajs: int = 1024, writer=None, plotdir: str = None, batch: bool = False, **kwargs):
        """ft_hmc.py:272-346 without plotting: returns the history dict of per-trajectory metrics.
        batch=True advances a batch of independent chains with per-chain accepts."""
        if x is None:
            x = self.initializer()
        history = {}
        q = qed.batch_charges(self.flow_forward(x)[0]) if batch else qed.batch_charges(x)
        for i in range(num_trajs):
            x, metrics_ = (self._batch_hmc(x, step=i) if batch else self.hmc(x, step=i))
            qold = history['q'][-1] if 'q' in history else q
            if '_plaq' in metrics_:                                       # the trajectory's own observables (ft_hmc.py:266-270 on F(x))
                p_, q_ = metrics_.pop('_plaq'), metrics_.pop('_q')
                metrics = {**metrics_, 'plaq': p_, 'q': q_, 'dq': torch.sqrt((q_ - qold) ** 2)}
            else:
                x_phys, _ = self.flow_forward(x)
                metrics = {**metrics_, **self.lattice_metrics(x_phys, qold)}
            for key, val in metrics.items():
                history.setdefault(key, []).append(val)
            if nprint and i % nprint == 0:
                print(f"traj {i}: acc={float(torch.as_tensor(metrics['acc'], dtype=torch.float64).mean()):.3f} "
                      f"dh={float(metrics['dh'].mean()):.4g} plaq={float(metrics['plaq'].mean()):.6f} "
                      f"q={float(metrics['q'].mean()):.3f}", flush=True)
        self.x_last = x
        return history


def run_ftHMC(flow: torch.nn.Module, config: TrainConfig, tau: float, nstep: int, num_trajs: int = 1024,
              nprint: int = 50, **kwargs):
    """ft_hmc.py:349-380 -> (ft, history, dirs)."""
    lfconfig = lfConfig(tau=tau, nstep=nstep)
    flow.eval()
    ft = FieldTransformation(flow=flow, config=config, lfconfig=lfconfig, **kwargs)
    history = ft.run(nprint=nprint, num_trajs=num_trajs)
    fthmcdir = f"{config.logdir}/ftHMC/{lfconfig.uniquestr()}"
    dirs = {'logdir': fthmcdir, 'plotsdir': f'{fthmcdir}/plots', 'summarydir': f'{fthmcdir}/summaries'}
    return ft, history, dirs
