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

from . import graph_loop, ops, parallel
from .config import DTYPE, FlowModel, Param, TrainConfig, device
from .utils import qed_helpers as qed
from .utils.distributions import MultivariateUniform, calc_dkl, calc_ess
from .utils.layers import (attach_grads, bump_weights_generation, flatten_flow, flow_activation, flow_grad_buffer, flow_grad_ext, flow_weights, get_nets,
                           make_net_from_layers, make_u1_equiv_layers, net_weights, set_weights)
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
    import pickle
    try:
        checkpoint = torch.load(infile, map_location=device(), weights_only=True)
    except pickle.UnpicklingError as e:                         # what the weights-only unpickler raises on a refused global;
        if not trusted:                                         # a missing or unreadable file (OSError) propagates as it is
            raise RuntimeError(f'{infile} does not load with weights_only=True ({type(e).__name__}: a checkpoint written by '
                               f'the reference pickles numpy arrays in its history); if you wrote this file yourself, call '
                               f'restore_model_from_checkpoint(..., trusted=True)') from e
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
                'optimizer_state_dict': _portable_optimizer_state(optimizer), 'history': _plain(history)}, path)
    return path


def _portable_optimizer_state(optimizer) -> dict:
    """optimizer.state_dict() in the layout a plain (non-capturable) torch optimizer writes and reads: the learning rate a
    Python float and the step counts CPU scalars -- a capturable optimizer (make_optimizer) keeps both on the device."""
    sd = optimizer.state_dict()
    groups = []
    for g in sd['param_groups']:
        g = dict(g)
        if torch.is_tensor(g.get('lr')):
            g['lr'] = float(g['lr'])
        g['capturable'] = False; g['fused'] = None; g['foreach'] = None
        groups.append(g)
    state = {}
    for k, st in sd['state'].items():
        st = dict(st)
        if torch.is_tensor(st.get('step')):
            st['step'] = st['step'].detach().to('cpu', torch.float32)
        state[k] = st
    return {'state': state, 'param_groups': groups}


ActionFn = Callable[[torch.Tensor], torch.Tensor]

METRIC_KEYS = ('ess', 'logp', 'logq', 'loss_dkl', 'q', 'dq', 'plaq')


def _fused_step_device(model: FlowModel, action, batch_size: int, dkl_factor: float, xi: torch.Tensor, row: torch.Tensor = None,
                       groups: int = None, pgroup=None):
    """The device side of one fused training step, enqueue only (no host synchronisation, graph-capturable without a
    process group): d(loss)/d(weights) straight into the flat gradient buffer every conv parameter's .grad is a view of
    (no unpacking, no copies), and the stacked metrics row (ops.train_metrics).  -> (row, x)."""
    layers = model.layers
    flat = flatten_flow(layers)
    gflat = flow_grad_buffer(layers)
    world = torch.distributed.get_world_size() if parallel.have_group() else 1
    B, L = xi.shape[0], xi.shape[-1]
    r = ops.train_grad(xi, flat, len(layers), action.beta, flow_activation(layers),
                       groups=ops.default_train_groups(B, L) if groups is None else groups, out_gw=gflat)
    scale = dkl_factor / world                   # kernel seeds 1 / B_local; the loss is the global mean
    if scale != 1.0:
        gflat.mul_(scale)
    row = ops.train_metrics(xi, r['x'], r['logq'], r['logp'], action.beta, dkl_factor, out=row)
    if parallel.have_group():
        # C2 in two collectives: ONE SUM all-reduce of [gradients, sum_b (logq - logp)] and ONE all-gather of three doubles
        # for the ESS (parallel.global_ess) -- the messages are tiny, so the step pays per collective, not per byte
        n_global = B * world
        gext = flow_grad_ext(layers)
        d = r['logq'] - r['logp']
        n = gflat.numel()
        torch.sum(d, dim=0, keepdim=True, out=gext[n:n + 1])
        parallel.allreduce_grads(gext, group=pgroup)
        row[0] = dkl_factor * gext[n] / n_global
        row[1] = parallel.global_ess(-d, n_global, group=pgroup)
    return row, r['x']


def _metrics_dict(row_host: np.ndarray, B: int) -> dict:
    m = ops.split_metrics(row_host, B)
    return {k: np.asarray(m[k]) for k in METRIC_KEYS}


def train_step(model: FlowModel, config: TrainConfig, action: ActionFn, optimizer: optim.Optimizer,
               batch_size: int, scheduler: Any = None, scaler: Any = None, pre_model: FlowModel = None,
               dkl_factor: float = 1., xi: torch.Tensor = None, fused: bool = True):
    """train.py:162-228.  `batch_size` is this rank's share of the global batch.

    fused=True : one HIP call computes x, logq, logp and d(loss)/d(weights) (needs `action` to be
                 the Wilson `BatchAction(config.beta)`, which is what train.py:291 passes); the gradient lands in the
                 flat buffer the parameters' .grad are views of, the metrics come back in ONE device-to-host copy;
    fused=False: the layers run one by one through autograd (any `action` callable).
    `scaler` (a torch GradScaler, train.py:206-209, 321-324): the reference's scale / step / update sequence on the autograd
    route; on the fp64 path it changes nothing but the order of two multiplications (there is no fp16 to protect).
    A whole training loop without any per-step host synchronisation: `GraphTrainer` (what `train` uses)."""
    t0 = time.time()
    if scaler is not None:
        fused = False        # train.py:206-209: scaler.scale(loss).backward() needs the loss as an autograd tensor: the layer-wise route
    if pre_model is not None:
        pre_xi = pre_model.prior.sample_n(batch_size)
        x_pre = qed.ft_flow(pre_model.layers, pre_xi)
        xi = qed.ft_flow_inv(pre_model.layers, x_pre)
    layers = model.layers
    world = torch.distributed.get_world_size() if parallel.have_group() else 1
    n_global = batch_size * world
    if fused and isinstance(action, qed.BatchAction):
        if xi is None:
            xi = model.prior.sample_n(batch_size)
        xi = xi.to(DTYPE)
        row, _ = _fused_step_device(model, action, batch_size, dkl_factor, xi)
        attach_grads(layers)                     # the fused call overwrites every gradient: no zero_grad pass
        optimizer.step()
        host = row.cpu().numpy()                 # the one synchronisation of the step
        out = _metrics_dict(host, xi.shape[0])
        if scheduler is not None:
            scheduler.step(float(out['loss_dkl']))
        out['dt'] = time.time() - t0
        return out
    optimizer.zero_grad()
    x, xi, logq = apply_flow_to_prior(model.prior, layers, xi=xi, batch_size=batch_size)
    logp = (-1.) * action(x)
    loss_local = dkl_factor * (logq - logp).sum() / n_global
    if scaler is not None:
        scaler.scale(loss_local).backward()      # the scaled gradients are summed over the ranks below, unscaled inside scaler.step
    else:
        loss_local.backward()
    if parallel.have_group():
        for p in layers.parameters():
            parallel.allreduce_grads(p.grad)
    loss_dkl = dkl_factor * parallel.global_mean((logq - logp).detach(), n_global)
    logw = (logp - logq).detach()
    ess = torch.exp(2 * parallel.global_logsumexp(logw) - parallel.global_logsumexp(2 * logw)) / n_global
    qi = qed.batch_charges(xi)
    q = qed.batch_charges(x.detach())
    plaq = logp.detach() / (config.beta * config.volume)
    dq = torch.sqrt((q - qi) ** 2)
    if scaler is not None:
        scaler.step(optimizer)
        scaler.update()
    else:
        optimizer.step()
    if scheduler is not None:
        scheduler.step(loss_dkl)
    # one stacked copy instead of seven synchronising ones
    B = x.shape[0]
    host = torch.cat([loss_dkl.reshape(1), ess.reshape(1), logp.detach(), logq.detach(), q, dq, plaq]).cpu().numpy()
    out = _metrics_dict(host, B)
    out['dt'] = time.time() - t0
    return out


class GraphTrainer:
    """The training loop of fthmc/train.py:352-407 without the host in it: one step = prior draw (Philox on the device,
    keyed by global chain id and step) -> fthmc_train_grad into the flat gradient buffer -> metrics row -> optimizer
    step, captured ONCE in a hipGraph and replayed; the metrics of every step are kept on the device (one small
    device-to-device copy per step) and come to the host when somebody looks (`metrics()`, `history()`).

    Needs an optimizer whose step is capturable (`FlatAdam` = `make_optimizer`, or a torch one built with capturable=True).  With a process
    group (more than one rank, or FTHMC_FORCE_PG=1) the C2 collectives (gradient all-reduce, global loss mean, MAX + SUM of
    the ESS) are part of the captured sequence -- RCCL collectives are graph-capturable -- so the 8-GPU step is the same
    replayed graph as the 1-GPU one; the ranks agree (one eager MIN all-reduce) on whether every capture succeeded and all
    fall back to the eager sequence otherwise (FTHMC_GRAPH_COLLECTIVES=0 forces that).  The per-chain seeds of a step are
    formed on the device from (seed, global chain id, a device step counter): a step costs the host no copy.  A
    ReduceLROnPlateau scheduler needs the loss on the host after every step and so brings one synchronisation per step back."""

    def __init__(self, model: FlowModel, config: TrainConfig, optimizer: optim.Optimizer, batch_size: int,
                 dkl_factor: float = 1., scheduler: Any = None, seed: int = 1234, use_graph: bool = True, chunk: int = 256):
        self.model, self.config, self.optimizer, self.scheduler = model, config, optimizer, scheduler
        self.B, self.dkl_factor, self.seed = int(batch_size), float(dkl_factor), int(seed)
        self.action = qed.BatchAction(config.beta)
        self.dev = next(model.layers.parameters()).device
        L = tuple(config.lat)[-1]
        self.rank, self.world = (torch.distributed.get_rank(), torch.distributed.get_world_size()) if parallel.have_group() else (0, 1)
        self.lo = self.rank * self.B                                   # global chain ids of this rank's batch
        self.seeds = torch.empty(self.B, dtype=torch.int64, device=self.dev)
        self.counter = torch.zeros(1, dtype=torch.int64, device=self.dev)      # steps taken, on the device: keys the step's draws
        self.xi = torch.empty(self.B, 2, L, L, dtype=DTYPE, device=self.dev)
        self.row = torch.empty(2 + 5 * self.B, dtype=DTYPE, device=self.dev)
        self.chunk = max(1, int(chunk))
        self.hist_dev = torch.empty(self.chunk, self.row.numel(), dtype=DTYPE, device=self.dev)
        self.hist_host = []                                            # flushed chunks (numpy)
        self.nstep = 0
        self.graph = None
        self.stream = torch.cuda.Stream(device=self.dev)
        flatten_flow(model.layers); flow_grad_buffer(model.layers); attach_grads(model.layers)
        # the steps run on the trainer's own stream (a capture needs one): it starts behind whatever initialised the model and
        # the optimizer on the caller's stream; metrics() / history() / synchronize() are where the caller waits for it
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        # with a process group the step is captured only over RCCL ("nccl": its collectives are stream-ordered device work; gloo
        # moves the tensors through the host and cannot be captured) and unless FTHMC_GRAPH_COLLECTIVES=0 asks for the eager sequence
        group_ok = (not parallel.have_group() or
                    (torch.distributed.get_backend() == 'nccl' and os.environ.get('FTHMC_GRAPH_COLLECTIVES', '1') not in ('', '0')))
        self.use_graph = bool(use_graph) and group_ok
        # the captured collectives run on a process group of their own, on which nothing is ever issued eagerly
        # (parallel.capture_group: every rank creates it here, communicator connected)
        self.pgroup = parallel.capture_group(self.dev) if (self.use_graph and parallel.have_group()) else None
        if self.use_graph and not (getattr(optimizer, 'graph_safe', False) or all(g.get('capturable', False) for g in optimizer.param_groups)):
            raise ValueError('GraphTrainer captures optimizer.step(): pass a FlatAdam (train.make_optimizer) or a torch '
                             'optimizer built with capturable=True, or use_graph=False')

    def _enqueue(self, pgroup=None):
        # seeds of step `counter` for this rank's global chain ids (= parallel.chain_seeds(seed, lo, lo + B, step)); counter += 1
        ops.chain_seeds(self.seed, self.lo, self.B, counter=self.counter, advance=True, out=self.seeds)
        ops.random_uniform(self.seeds, self.xi.shape, -PI, PI, out=self.xi)        # MultivariateUniform(-pi, pi).sample_n
        _fused_step_device(self.model, self.action, self.B, self.dkl_factor, self.xi, row=self.row, pgroup=pgroup)
        self.optimizer.step()

    def _capture(self):
        """capture one step; under a process group every rank learns whether every rank's capture succeeded"""
        ok = 1.0
        g = torch.cuda.CUDAGraph()
        # No fence: the captured collectives run on parallel.capture_group(), on which nothing was ever issued eagerly -- its
        # internal stream carries no event the watchdog could be polling when that stream joins the capture (an event query on
        # a stream that takes part in a capture is hipErrorCapturedEvent; the eager first step ran on the default group, whose
        # stream stays outside).  Round 5 slept 0.5 s here.
        try:
            with graph_loop.capture(g, self.stream):
                self._enqueue(self.pgroup)
        except Exception as e:                                             # noqa: BLE001 -- whatever the capture objects to
            if not parallel.have_group():
                raise                                                      # no group to agree with
            if torch.cuda.is_current_stream_capturing():
                # the capture cannot be left: no collective can be issued to tell the peers -- take the group down instead of
                # leaving them blocked in the agreement below
                try:
                    torch.distributed.distributed_c10d._abort_process_group()
                finally:
                    raise
            ok, self.capture_error = 0.0, repr(e)
        if parallel.have_group():
            flag = torch.tensor([ok], dtype=torch.float64, device=self.dev)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            ok = float(flag)
        if ok > 0.5:
            self.graph = g
        else:
            self.use_graph = False                                          # every rank: the eager sequence with its collectives

    @property
    def captured(self) -> bool:
        return self.graph is not None

    def _flush(self):
        k = self.nstep % self.chunk or (self.chunk if self.nstep else 0)
        if k:
            self.hist_host.append(self.hist_dev[:k].cpu().numpy())

    def step(self):
        """enqueue one training step; returns nothing and waits for nothing (unless a scheduler is attached)"""
        with torch.cuda.stream(self.stream):
            if self.use_graph and self.graph is None:
                # first step: once eagerly (allocator, workspaces, optimizer state, RCCL's communicator), then capture the SAME
                # sequence; the eager step counts as step 0 and the capture run is not replayed into the statistics (capture
                # does not execute)
                self._enqueue()
                attach_grads(self.model.layers)
                self.stream.synchronize()
                self._capture()
            elif self.use_graph:
                if hasattr(self.optimizer, 'sync_lr'):
                    self.optimizer.sync_lr()                   # a scheduler's new rate -> the device scalar the replay reads
                self.graph.replay()
                if hasattr(self.optimizer, 'count_step'):
                    self.optimizer.count_step()
            else:
                self._enqueue()
                attach_grads(self.model.layers)
            bump_weights_generation(flatten_flow(self.model.layers))     # a replay writes the weights behind every version counter
            self.hist_dev[self.nstep % self.chunk].copy_(self.row)
            self.nstep += 1
            if self.nstep % self.chunk == 0:
                self._flush()
            if self.scheduler is not None:
                self.scheduler.step(float(self.row[0]))

    def synchronize(self):
        """wait for every step enqueued so far (before reading the weights on another stream)"""
        self.stream.synchronize()

    def metrics(self) -> dict:
        """metrics of the last step on the host (synchronises)"""
        self.stream.synchronize()
        return _metrics_dict(self.row.cpu().numpy(), self.B)

    def history(self) -> dict:
        """{key: [per-step arrays]} of every step so far (synchronises)"""
        self.stream.synchronize()
        chunks = list(self.hist_host)
        k = self.nstep % self.chunk
        if k:
            chunks.append(self.hist_dev[:k].cpu().numpy())
        if not chunks:
            return {k_: [] for k_ in METRIC_KEYS}
        allrows = np.concatenate(chunks, axis=0)
        out = {k_: [] for k_ in METRIC_KEYS}
        for r in allrows:
            m = _metrics_dict(r, self.B)
            for k_ in METRIC_KEYS:
                out[k_].append(m[k_])
        return out


class FlatAdam(optim.Adam):
    """torch.optim.Adam (or AdamW: decoupled=True) over the conv parameters of a flattened flow whose `step()` is ONE HIP
    launch on the flat buffers (C ABI fthmc_adam_step: torch's single-tensor update formulas in fp64).  Step count and
    learning rate live on the device, so a captured step can be replayed (GraphTrainer) and a scheduler's new rate reaches
    the next replay; `state_dict()` / `load_state_dict()` keep torch's Adam layout (per-parameter step / exp_avg / exp_avg_sq),
    so checkpoints move between this class and a plain torch optimizer either way."""

    graph_safe = True

    def __init__(self, layers: nn.ModuleList, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 decoupled: bool = False):
        self._flat = flatten_flow(layers)
        self._gflat = flow_grad_buffer(layers)
        super().__init__(list(layers.parameters()), lr=float(lr), betas=betas, eps=eps, weight_decay=weight_decay, foreach=False)
        self._layers, self._decoupled = layers, bool(decoupled)
        self._m = torch.zeros_like(self._flat)
        self._v = torch.zeros_like(self._flat)
        self._hyper = torch.zeros(3, dtype=torch.float64, device=self._flat.device)      # [steps taken, lr, ticket]
        self._hyper[1] = float(lr)
        self._lr_dev = float(lr)
        self._nstep = 0                                                                  # host mirror of _hyper[0]
        self._view_state()

    def _view_state(self):
        o = 0
        for p in self.param_groups[0]['params']:
            n = p.numel()
            self.state[p] = {'step': torch.tensor(float(self._nstep)), 'exp_avg': self._m[o:o + n].view(p.shape),
                             'exp_avg_sq': self._v[o:o + n].view(p.shape)}
            o += n

    def sync_lr(self):
        """param_groups[0]['lr'] (what a scheduler writes) -> the device scalar the kernel reads; a launch only when it changed"""
        lr = float(self.param_groups[0]['lr'])
        if lr != self._lr_dev:
            self._hyper[1:2].fill_(lr)
            self._lr_dev = lr

    def count_step(self, n: int = 1):
        """a step the host did not launch itself (graph replay): keep the host mirror of the step count in line"""
        self._nstep += n
        bump_weights_generation(self._flat)

    @torch.no_grad()
    def step(self, closure=None):
        if flatten_flow(self._layers) is not self._flat:
            raise RuntimeError('the flow\'s parameters left their flat buffer (.to() / .cuda() after the optimizer was built): '
                               'build the optimizer again')
        g = self.param_groups[0]
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
            attach_grads(self._layers)
        ops.adam_step(self._flat, self._gflat, self._m, self._v, self._hyper, betas=g['betas'], eps=g['eps'],
                      weight_decay=g['weight_decay'], decoupled=self._decoupled)
        if not torch.cuda.is_current_stream_capturing():
            self._nstep += 1
            bump_weights_generation(self._flat)          # the kernel wrote through raw pointers: no tensor version moved

    def state_dict(self):
        for st in self.state.values():
            st['step'] = torch.tensor(float(self._nstep))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)                # torch's checks and casts; leaves copies in self.state
        o, nstep = 0, 0
        for p in self.param_groups[0]['params']:
            st, n = self.state.get(p, {}), p.numel()
            if 'exp_avg' in st:
                self._m[o:o + n].copy_(st['exp_avg'].reshape(-1)); self._v[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
                nstep = int(float(st['step']))
            o += n
        self._nstep = nstep
        self._hyper[0] = float(nstep)
        self._lr_dev = None
        self.sync_lr()
        self._view_state()


def make_optimizer(model: FlowModel, config: TrainConfig, capturable: bool = True) -> optim.Optimizer:
    """optim.Adam(model.layers.parameters(), lr=config.base_lr) (train.py:297): on the GPU as `FlatAdam` -- one launch per
    step on the flat parameter buffer, capturable -- otherwise the plain torch optimizer."""
    params = list(model.layers.parameters())
    if capturable and params and params[0].is_cuda:
        return FlatAdam(model.layers, lr=config.base_lr)
    return optim.Adam(params, lr=config.base_lr)


def train(config: TrainConfig, model: Optional[FlowModel] = None, pre_model: FlowModel = None,
          figsize=None, dpi: int = 120, scheduler_config=None, dkl_factor: float = 1., save: bool = False,
          verbose: bool = True, use_graph: bool = True, seed: int = 1234, use_scaler: bool = False):
    """train.py:236-431 without plots / tensorboard: n_era x n_epoch steps, one checkpoint per era
    (save=True).  Returns dict(model, optimizer, history, ckpt_files).

    The steps run through `GraphTrainer` (one captured hipGraph per step, metrics kept on the device until the end or
    the next `print_freq` line); `pre_model` (a prior passed through another flow and back) and `use_scaler` (train.py:250,
    321-324: a torch GradScaler around the step) keep the step-by-step `train_step` route."""
    if model is None:
        model = get_model(config)
    stepwise = pre_model is not None or use_scaler
    optimizer = make_optimizer(model, config, capturable=not stepwise)
    scaler = torch.amp.GradScaler('cuda') if (use_scaler and torch.cuda.is_available()) else None
    scheduler = None
    if scheduler_config is not None:
        sc = {k: v for k, v in vars(scheduler_config).items() if k != 'verbose'}
        scheduler = optim.lr_scheduler.ReduceLROnPlateau(optimizer, **sc)
    action = qed.BatchAction(config.beta)
    history, ckpts = {}, []
    step = 0
    t0 = time.time()
    trainer = None if stepwise else GraphTrainer(model, config, optimizer, config.batch_size, dkl_factor=dkl_factor,
                                                              scheduler=scheduler, seed=seed, use_graph=use_graph)
    for era in range(config.n_era):
        for epoch in range(config.n_epoch):
            show = verbose and config.print_freq and step % config.print_freq == 0
            if trainer is not None:
                trainer.step()
                metrics = trainer.metrics() if show else None
            else:
                metrics = train_step(model, config, action, optimizer, config.batch_size, scheduler=scheduler, scaler=scaler,
                                     pre_model=pre_model, dkl_factor=dkl_factor)
                for k, v in metrics.items():
                    history.setdefault(k, []).append(v)
            if show:
                print(f"era {era} epoch {epoch}: loss_dkl={float(metrics['loss_dkl']):.4f} "
                      f"ess={float(metrics['ess']):.4f} plaq={float(np.mean(metrics['plaq'])):.5f}", flush=True)
            step += 1
        if save:
            if trainer is not None:
                history = trainer.history()
            ckpts.append(save_checkpoint(era, config.n_epoch, model.layers, optimizer, history,
                                         config.update_logdirs(config.logdir)['ckpts']))
    if trainer is not None:
        history = trainer.history()
        history['dt'] = [(time.time() - t0) / max(step, 1)] * step       # per-step wall time: the loop never stops to measure one
    return {'model': model, 'optimizer': optimizer, 'history': history, 'ckpt_files': ckpts, 'action': action}


def transfer_to_new_lattice(L: int, layers: nn.ModuleList, param_init: Param = None):
    """train.py:434-455: reuse the (translation-equivariant) conv nets on an L x L lattice."""
    lattice_shape = (L, L)
    dev = device()
    prior = MultivariateUniform(-PI * torch.ones((2, *lattice_shape), dtype=DTYPE, device=dev),
                                PI * torch.ones(lattice_shape, dtype=DTYPE, device=dev))
    new_layers = make_net_from_layers(nets=get_nets(layers), lattice_shape=lattice_shape)
    return FlowModel(prior=prior, layers=new_layers)
