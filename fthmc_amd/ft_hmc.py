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

import os

from . import ops
from .config import DTYPE, TrainConfig, device, lfConfig
from .graph_loop import GraphLoop
from .utils import qed_helpers as qed
from .utils.layers import flow_activation, flow_weights, weights_generation

TWO_PI = 2. * PI


class LazyHistory(dict):
    """The history dict of a captured run: {metric: [per-trajectory tensors]} as the eager loop builds it, filled from the
    device ring the first time anybody looks -- `run()` itself returns when everything is ENQUEUED."""

    def __init__(self, fill):
        super().__init__()
        self._fill_fn = fill

    def _fill(self):
        if self._fill_fn is not None:
            fn, self._fill_fn = self._fill_fn, None
            super().update(fn())

    # every way of looking at, copying or changing a dict fills first: a caller written against the reference's plain history dict
    # (ft_hmc.py:272-346) never sees (or overwrites) an empty one
    def __getitem__(self, k): self._fill(); return super().__getitem__(k)
    def __setitem__(self, k, v): self._fill(); return super().__setitem__(k, v)
    def __delitem__(self, k): self._fill(); return super().__delitem__(k)
    def __contains__(self, k): self._fill(); return super().__contains__(k)
    def __iter__(self): self._fill(); return super().__iter__()
    def __reversed__(self): self._fill(); return super().__reversed__()
    def __len__(self): self._fill(); return super().__len__()
    def __bool__(self): self._fill(); return super().__len__() > 0
    def __or__(self, other): self._fill(); return dict(super().items()) | dict(other)
    def __ror__(self, other): self._fill(); return dict(other) | dict(super().items())
    def __ior__(self, other): self._fill(); super().update(other); return self
    def get(self, k, d=None): self._fill(); return super().get(k, d)
    def keys(self): self._fill(); return super().keys()
    def values(self): self._fill(); return super().values()
    def items(self): self._fill(); return super().items()
    def setdefault(self, k, d=None): self._fill(); return super().setdefault(k, d)
    def pop(self, *a): self._fill(); return super().pop(*a)
    def popitem(self): self._fill(); return super().popitem()
    def update(self, *a, **kw): self._fill(); return super().update(*a, **kw)
    def clear(self): self._fill_fn = None; return super().clear()
    def copy(self): self._fill(); return dict(super().items())
    def __repr__(self): self._fill(); return super().__repr__()
    def __eq__(self, other): self._fill(); return super().__eq__(other)
    __hash__ = None
    def __reduce__(self): self._fill(); return (dict, (dict(super().items()),))     # pickles / copies as the plain dict it stands for


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
        self._flow_key = None
        # (field tensor, its version, weight key, beta, state [3, B]) of the last trajectory's result
        self._carry = None
        self._last_obs = None         # (plaq, Q) of the flowed accepted field of the last trajectory (run() reads them)
        self._loop = None             # the captured run loop (GraphLoop) and what it was captured for
        self._loop_streams = None     # its stream and the chain groups' side streams: made once, reused by every re-capture
        self.use_graph = os.environ.get('FTHMC_RUN_GRAPH', '1') not in ('', '0')

    # ---- weights: packed once, refreshed when a parameter changed in place -----------
    def weights(self, dev) -> torch.Tensor:
        # the parameter list is gathered once per flow (Module.parameters() walks the module tree: it was two thirds of the host
        # time of a trajectory at L = 16) and again when the flow object, its length or one of its conv nets was replaced
        fkey = (id(self.flow), tuple(id(layer.plaq_coupling.net) for layer in self.flow))
        if self._params is None or self._flow_key != fkey:
            self._params = list(self.flow.parameters())
            self._flow_key = fkey
            self._w = None
        vers = tuple(p._version for p in self._params)
        if self._w is None or self._w_versions != vers or self._w.device != dev:
            self._w = flow_weights(self.flow, dev)
            self._w_versions = vers
        return self._w

    def _wkey(self, w: torch.Tensor):
        """What the weights' CONTENT is keyed by: the flow's identity, every parameter's version (torch optimizers,
        load_state_dict), the flat buffer's generation (FlatAdam / graph replays write it through raw pointers) and its own
        version.  The tensor's identity says nothing: a flattened flow hands out the same buffer before and after a step."""
        return (self._flow_key, self._w_versions, w.data_ptr(), w._version, weights_generation(w))

    def _carried_state(self, x: torch.Tensor, wkey):
        """(S_eff, plaq, Q) of x if x IS the field the previous trajectory returned, untouched, under the same weights and beta"""
        c = self._carry
        if c is not None and c[0] is x and c[1] == x._version and c[2] == wkey and c[3] == self.config.beta:
            return c[4]
        return None

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
        self._last_obs = None
        if x.shape[0] == 1:
            w = self.weights(x.device)
            wkey = self._wkey(w)
            state = self._carried_state(x, wkey)                         # see _batch_hmc
            r = ops.ft_trajectory(x, v, u.reshape(1), w, len(self.flow), self.config.beta,
                                  self.dt, self.nstep, self._act, mode=self._mode(), state_in=state, wkey=wkey)
            # the packaged code maps the end point with wrap, the notebook with regularize: same set
            xnew, acc, dh = r['x_new'], r['acc'][0] > 0.5, r['dH'][0]
            self._carry = (xnew, xnew._version, wkey, self.config.beta, r['state'])
            self._last_obs = (r['plaq'], r['Q'])
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
        self._last_obs = None
        if self.energy_mode == 'per_chain':
            # (S_eff, plaq, Q) of x is carried over when x IS the field the previous call returned, untouched, under the same
            # weights (by content: _wkey) and beta: that call's H1 sweep computed exactly what this call's H0 sweep would
            # (bit-identical; bench.py's `stateless` figure is the price of recomputing it, as the reference does at ft_hmc.py:205)
            w = self.weights(x.device)
            wkey = self._wkey(w)
            state = self._carried_state(x, wkey)
            r = ops.ft_trajectory(x, v, u, w, len(self.flow), self.config.beta, self.dt,
                                  self.nstep, self._act, mode=self._mode(), state_in=state,
                                  groups=ops.default_groups(x.shape[0], x.shape[-1]), wkey=wkey)
            x_, dh, acc = r['x_new'], r['dH'], r['acc']
            x_ = x_.detach()
            self._carry = (x_, x_._version, wkey, self.config.beta, r['state'])
            # plaq / Q of the flowed accepted field come with the trajectory: run() need not flow x again for its metrics
            # (kept out of the returned metrics: those are the reference's keys, ft_hmc.py:244-257)
            self._last_obs = (r['plaq'], r['Q'])
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
            num_trajs: int = 1024, writer=None, plotdir: str = None, batch: bool = False, use_graph: bool = None, **kwargs):
        """ft_hmc.py:272-346 without plotting: returns the history dict of per-trajectory metrics.
        batch=True advances a batch of independent chains with per-chain accepts.

        On the device the trajectory sequence (momenta and uniforms from torch's generator, the fused trajectory with the
        carried state, in place) is captured once in a hipGraph and replayed (`use_graph`, default on; FTHMC_RUN_GRAPH=0
        turns it off): the host issues one graph launch and one small copy per trajectory, the history is read back when it
        is first looked at.  Same draws and bit-identical histories as the eager loop."""
        if x is None:
            x = self.initializer()
        use_graph = self.use_graph if use_graph is None else bool(use_graph)
        fused = (self.energy_mode == 'per_chain') if batch else (x.shape[0] == 1)
        if use_graph and fused and x.is_cuda and num_trajs > 0:
            return self._run_captured(x, nprint, num_trajs, batch)
        history = {}
        q = qed.batch_charges(self.flow_forward(x)[0]) if batch else qed.batch_charges(x)
        for i in range(num_trajs):
            x, metrics_ = (self._batch_hmc(x, step=i) if batch else self.hmc(x, step=i))
            qold = history['q'][-1] if 'q' in history else q
            if self._last_obs is not None:                                # the trajectory's own observables (ft_hmc.py:266-270 on F(x))
                p_, q_ = self._last_obs
                metrics = {**metrics_, 'plaq': p_, 'q': q_, 'dq': torch.sqrt((q_ - qold) ** 2)}
            else:
                x_phys, _ = self.flow_forward(x)
                metrics = {**metrics_, **self.lattice_metrics(x_phys, qold)}
            for key, val in metrics.items():
                history.setdefault(key, []).append(val)
            if nprint and i % nprint == 0:
                self._print_line(i, metrics['acc'], metrics['dh'], metrics['plaq'], metrics['q'])
        self.x_last = x
        return history

    @staticmethod
    def _print_line(i, acc, dh, plaq, q):
        print(f"traj {i}: acc={float(torch.as_tensor(acc, dtype=torch.float64).mean()):.3f} "
              f"dh={float(torch.as_tensor(dh).mean()):.4g} plaq={float(torch.as_tensor(plaq).mean()):.6f} "
              f"q={float(torch.as_tensor(q).mean()):.3f}", flush=True)

    # ---- the captured loop ----------------------------------------------------------
    def _run_captured(self, x: torch.Tensor, nprint: int, num_trajs: int, batch: bool):
        dev = x.device
        caller = torch.cuda.current_stream(dev)
        B, L = x.shape[0], x.shape[-1]
        w = self.weights(dev)
        wkey = self._wkey(w)
        carried = self._carried_state(x, wkey)                            # by the identity of the tensor the caller passed
        xd = x.detach()
        nl, act, mode, beta = len(self.flow), self._act, self._mode(), self.config.beta
        G = ops.default_groups(B, L) if batch else 1
        # the weights' content version is part of what a capture is FOR: its launches carry it to the library, which checks
        # it on the device against the stamps in the workspaces (include/fthmc_hip.h "Weight versions")
        sig = (tuple(x.shape), dev, w.data_ptr(), nl, act, mode, beta, self.dt, self.nstep, G, batch,
               ops.get_variant(), ops.get_small_path(), ops.weights_version(wkey))
        if self._loop is not None and self._loop.get('pending') is not None:
            self._loop['pending']()                                       # an unread history of the previous run: read it before its ring is reused
            self._loop['pending'] = None
        if self._loop is None or self._loop['sig'] != sig:
            self._loop = self._make_loop(xd, w, nl, act, mode, beta, G, batch, sig, wkey)

        def prepare(lp):
            """start field, start state and the weight expansion on the loop's stream -> (q0, workspace token)"""
            loop = lp['loop']
            loop.stream.wait_stream(caller)
            with torch.cuda.stream(loop.stream):
                if carried is not None:
                    lp['state'].copy_(carried.reshape(3, B))
                else:
                    S, _, p_, q_ = ops.ft_action(xd, w, nl, beta, act)
                    torch.stack([S, p_, q_], out=lp['state'])
                # the start of the q history: Q of the flowed start field for a batch (ft_hmc.py:311-313), of the field itself otherwise
                q0_ = lp['state'][2].clone() if batch else qed.batch_charges(xd)
                lp['x'].copy_(xd)
                # the workspaces of the loop's streams at their final size, holding the expansion of these weights: the
                # replays' own expansion launches find the stamps and leave at once
                return q0_, ops.trajectory_workspaces(lp['x'], w, nl, groups=G, side_streams=lp['sides'], wkey=wkey)
        q0, token = prepare(self._loop)
        if self._loop['loop'].captured and self._loop['token'] != token:
            # a workspace moved since the capture (grown by another caller of these streams): capture again
            self._loop = self._make_loop(xd, w, nl, act, mode, beta, G, batch, sig, wkey)
            q0, token = prepare(self._loop)
        lp = self._loop
        loop, xs, state = lp['loop'], lp['x'], lp['state']
        lp['token'] = token
        loop.reset_history()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(loop.stream)
        for i in range(num_trajs):
            was_captured = loop.captured
            loop.step()
            if loop.captured and not was_captured:
                # the eager first step may have grown a workspace: the capture that followed saw the final ones
                with torch.cuda.stream(loop.stream):
                    lp['token'] = ops.trajectory_workspaces(xs, w, nl, groups=G, side_streams=lp['sides'], wkey=wkey)
            if nprint and i % nprint == 0:
                r = torch.from_numpy(loop.last()).view(4, B)
                self._print_line(i, r[0], r[1], r[2], r[3])
        ev1.record(loop.stream)
        with torch.cuda.stream(loop.stream):
            x_out = xs.clone()
            st_out = state.clone()
        loop.join()
        self._carry = (x_out, x_out._version, wkey, beta, st_out)
        self._last_obs = None
        self.x_last = x_out
        n = num_trajs
        rows_cache = [None]

        def loop_rows():
            if rows_cache[0] is None:
                rows_cache[0] = loop.rows()
            return rows_cache[0]
        lp['pending'] = loop_rows

        def fill():
            H = torch.from_numpy(loop_rows()).to(dev).view(n, 4, B)
            ev1.synchronize()
            dt = ev0.elapsed_time(ev1) * 1e-3 / n                         # per trajectory, on the device's clock
            qs = torch.cat([q0.reshape(1, -1).to(H.dtype), H[:, 3]], 0)
            dq = torch.sqrt((qs[1:] - qs[:-1]) ** 2)
            h = {'traj': list(range(n)), 'dt': [dt] * n}
            if batch:
                h.update({'acc': [H[i, 0] for i in range(n)], 'dh': [H[i, 1] for i in range(n)],
                          'exp_mdh': [torch.exp(-H[i, 1]) for i in range(n)]})
            else:
                h.update({'acc': [H[i, 0, 0] > 0.5 for i in range(n)], 'dh': [H[i, 1, 0] for i in range(n)]})
            h.update({'plaq': [H[i, 2] for i in range(n)], 'q': [H[i, 3] for i in range(n)], 'dq': [dq[i] for i in range(n)]})
            return h
        return LazyHistory(fill)

    def _make_loop(self, x, w, nl, act, mode, beta, G, batch, sig, wkey):
        dev, B = x.device, x.shape[0]
        xs = torch.empty_like(x)
        v = torch.empty_like(x)
        u = torch.empty(B, dtype=torch.float64, device=dev)
        row = torch.empty(4, B, dtype=torch.float64, device=dev)          # acc, dH, plaq, Q of the trajectory
        state = torch.empty(3, B, dtype=torch.float64, device=dev)
        out = {'x_new': xs, 'acc': row[0], 'dH': row[1], 'plaq': row[2], 'Q': row[3], 'state': state,
               'H0': torch.empty(B, dtype=torch.float64, device=dev), 'H1': torch.empty(B, dtype=torch.float64, device=dev)}

        # the loop's own streams (its own and one per chain group beyond the first), made once per FieldTransformation and
        # reused by every re-capture (a beta scan re-captures per beta: a new stream each time would leave a workspace of
        # 0.3-0.4 GB behind per stream until torch's stream pool wraps)
        if self._loop_streams is None or self._loop_streams[0] != dev:
            self._loop_streams = (dev, torch.cuda.Stream(device=dev), [])
        _, lstream, pool = self._loop_streams
        while len(pool) < max(G, 1) - 1:
            pool.append(torch.cuda.Stream(device=dev))
        sides = pool[:max(G, 1) - 1]

        def enqueue():
            v.normal_()                                                   # = torch.randn_like(x), then torch.rand(B): ft_hmc.py:204, 243
            u.uniform_()
            # in place: the accepted field replaces x, its (S_eff, plaq, Q) the carried state (both read before they are written)
            ops.ft_trajectory(xs, v, u, w, nl, beta, self.dt, self.nstep, act, mode=mode, out=out, state_in=state, groups=G,
                              side_streams=sides, wkey=wkey)
        loop = GraphLoop(enqueue, row, use_graph=True, stream=lstream)
        return {'sig': sig, 'loop': loop, 'x': xs, 'state': state, 'row': row, 'token': None, 'pending': None, 'sides': sides}


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
