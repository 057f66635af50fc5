"""Multi-GPU layer: independent Markov chains sharded over one process per GPU.

The path shards by chain: no data-path collective exists.  The only
exchange is the small SUM all-reduce of run statistics (C1: acceptance and
observables, 8 doubles) and, for training, of weight gradients plus the
logsumexp pieces of the ESS (C2).  `torch.distributed` backend "nccl" is RCCL
over xGMI on ROCm; the same code runs on "gloo" for the CPU tests.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist

STAT_KEYS = ('n', 'acc', 'plaq', 'q', 'q2', 'absdq', 'dh', 'exp_mdh')


def world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1 process = 1 GPU)."""
    return (int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)),
            int(os.environ.get('LOCAL_RANK', 0)))


def force_group() -> bool:
    """FTHMC_FORCE_PG=1: create a process group even for ONE rank, so that every collective of this module (and of
    bench.py / train.py) takes its RCCL branch on a single GPU -- communicator creation, the async C1 handle living
    across hipGraph replays, barrier(device_ids=...), the watchdog thread next to a graph capture: the 8-GPU code
    path minus the peers.  A rehearsal switch; results are bit-identical to the run without a group."""
    return os.environ.get('FTHMC_FORCE_PG', '0') not in ('', '0')


def have_group() -> bool:
    """A process group exists (of any size): collectives are issued whenever this is true, not only for world > 1."""
    return dist.is_available() and dist.is_initialized()


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    rank, ws, local = world()
    if (ws > 1 or force_group()) and not dist.is_initialized():
        if backend is None:
            # FTHMC_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsals on a 1-GPU box;
            # RCCL refuses two ranks on the same device)
            backend = os.environ.get('FTHMC_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if ws == 1:
            os.environ.setdefault('MASTER_PORT', str(29500 + os.getpid() % 2000))   # a lone rank outside torchrun
        dist.init_process_group(backend=backend, rank=rank, world_size=ws)
    return rank, ws, local


_CAPTURE_GROUP = [None, None]      # (the default group it belongs to, the capture group)


def capture_group(device=None):
    """The process group of the CAPTURED collectives (C2 inside GraphTrainer's hipGraph): all ranks, the same backend as the
    default group, its own RCCL communicator and its own internal stream -- created on first use, by every rank at the same
    point (GraphTrainer's constructor), the communicator connected there and then (`device_id`: ncclCommInitRank must not
    happen inside a capture).  NOTHING is ever issued on it eagerly: its first collective is the captured one.

    Why (tools/event_query_probe.py on the MI355X, profiles/r06_event_query_probe.txt): the HIP runtime answers
    hipErrorCapturedEvent to a query of ANY event whose last record was on a stream that takes part in a capture -- the stream
    that originates it AND a stream that joined it -- also when the event was recorded there eagerly, long before.  A process
    group's watchdog thread queries the end events of its EAGER collectives until it has retired them (a polling round every
    100 ms), and those events live on the group's internal stream, which joins the capture under a captured collective: an
    un-retired eager collective of the same group is a process abort waiting for the watchdog's next round (what round 5
    fenced with a 0.5 s sleep).  Work issued DURING a capture is never handed to the watchdog.  So: the group whose
    collectives are captured has no eager collectives, ever -- the eager first step of a trainer, C1, barriers and the capture
    agreement run on the default group, whose stream takes no part in any capture and whose events stay queryable
    (tools/pg_capture_probe.py: 200 asynchronous collectives in flight and a second thread issuing more during the capture).
    Nothing here waits for a thread's polling round."""
    if not have_group():
        return None
    if _CAPTURE_GROUP[0] is not dist.group.WORLD:          # first use, or the default group was torn down and made again
        kw = {}
        if dist.get_backend() == 'nccl' and device is not None:
            kw['device_id'] = torch.device(device)
        try:
            g = dist.new_group(backend=dist.get_backend(), **kw)
        except TypeError:                                   # a torch without new_group(device_id=...): connects at first use
            g = dist.new_group(backend=dist.get_backend())
        if kw:
            # new_group connects the communicator now only where the backend "supports splitting"; ask for it in any case
            # (ncclCommInitRank inside a capture is what this avoids; a second request for an existing communicator is a lookup)
            try:
                g._get_backend(kw['device_id']).eager_connect_single_device(kw['device_id'])
            except (AttributeError, RuntimeError):
                pass
        _CAPTURE_GROUP[:] = [dist.group.WORLD, g]
    return _CAPTURE_GROUP[1]


def shard_range(n_chains: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of global chain ids owned by `rank` (remainder to the
    first ranks), so that a chain's id -- and with it its RNG stream -- never depends
    on the number of GPUs."""
    base, rem = divmod(n_chains, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def chain_seeds(seed: int, lo: int, hi: int, traj: int) -> torch.Tensor:
    """One 63-bit seed per (global chain id, trajectory): SplitMix64 of the triple, so
    the momenta / uniform draws of a chain are the same on 1 or 8 GPUs."""
    import numpy as np
    M = (1 << 64) - 1
    base = ((seed * 2 + 1) * 0x2545F4914F6CDD1D + traj * 0xBF58476D1CE4E5B9) & M
    with np.errstate(over='ignore'):
        z = np.arange(lo, hi, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(base)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return torch.from_numpy((z & np.uint64(0x7FFFFFFFFFFFFFFF)).astype(np.int64))


class RunStats:
    """Running sums over chains and trajectories.  `vec` holds this rank's sums; `reduce()`
    all-reduces a snapshot of it into `glob` (C1), so it can be issued asynchronously every
    trajectory without ever double counting."""

    def __init__(self, vec: torch.Tensor):
        self.vec = vec                # [len(STAT_KEYS)] float64 on the compute device
        self.glob = None

    @classmethod
    def zeros(cls, device):
        return cls(torch.zeros(len(STAT_KEYS), dtype=torch.float64, device=device))

    def add_device(self, acc, plaq, q, qold, dh):
        """The same accumulation as `add` with dq = q - qold, in ONE launch of the HIP library (fthmc_stats_accumulate);
        also moves qold on to q.  Device tensors [B_local]; `qold` is updated in place."""
        from . import ops
        ops.stats_accumulate(acc, plaq, q, qold, dh, self.vec)

    def add(self, acc, plaq, q, dq, dh):
        """Accumulate one trajectory's per-chain results (device tensors [B_local])."""
        # one stacked reduction instead of seven: this runs once per trajectory behind ~200 dependent launches
        rows = torch.stack([torch.ones_like(acc), acc, plaq, q, q * q, dq.abs(), dh, torch.exp(-dh)])
        self.vec += rows.sum(dim=1)

    def reduce(self, async_op: bool = False):
        """C1: SUM all-reduce of a snapshot over the ranks of the process group (plain copy without one)."""
        if not have_group():
            self.glob = None              # one process: `means()` reads the running sums themselves (no snapshot, no copy per trajectory)
            return None
        self.glob = self.vec.clone()
        return dist.all_reduce(self.glob, op=dist.ReduceOp.SUM, async_op=async_op)

    def means(self) -> dict:
        v = (self.glob if self.glob is not None else self.vec).detach().cpu()
        n = max(float(v[0]), 1.0)
        out = {k: float(v[i]) / n for i, k in enumerate(STAT_KEYS) if k != 'n'}
        out['n'] = float(v[0])
        out['chi_q'] = out['q2'] - out['q'] ** 2
        return out


def allreduce_grads(gw: torch.Tensor, world_size: Optional[int] = None, group=None) -> torch.Tensor:
    """C2: SUM all-reduce of the flat weight-gradient buffer (955 * n_layers doubles)."""
    if have_group():
        dist.all_reduce(gw, op=dist.ReduceOp.SUM, group=group)
    return gw


def global_logsumexp(logw: torch.Tensor) -> torch.Tensor:
    """logsumexp over the chains of all ranks: MAX all-reduce, then SUM all-reduce."""
    m = logw.max()
    if have_group():
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        s = torch.exp(logw - m).sum()
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        return m + torch.log(s)
    return m + torch.log(torch.exp(logw - m).sum())


def global_ess(logw: torch.Tensor, n_global: int, group=None) -> torch.Tensor:
    """calc_ess (fthmc/utils/distributions.py:27-37) over the chains of all ranks with ONE collective: every rank contributes
    (m, sum exp(logw - m), sum exp(2 (logw - m))) with its own maximum m, an all-gather of the three doubles lets every rank
    rescale them to the common maximum: ESS = (sum w)^2 / (n sum w^2).  (global_logsumexp twice = four all-reduces.)"""
    m = logw.max()
    z = logw - m
    loc = torch.stack([m, torch.exp(z).sum(), torch.exp(2 * z).sum()])
    if have_group():
        parts = [torch.empty_like(loc) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, loc, group=group)
        allv = torch.stack(parts)
    else:
        allv = loc[None]
    f = torch.exp(allv[:, 0] - allv[:, 0].max())
    s1, s2 = (allv[:, 1] * f).sum(), (allv[:, 2] * f * f).sum()
    return s1 * s1 / s2 / n_global


def global_mean(t: torch.Tensor, n_global: int) -> torch.Tensor:
    s = t.sum()
    if have_group():
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return s / n_global
