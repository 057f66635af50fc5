"""Boundary types of the hot path: re-statement of fthmc/config.py (Param :194-258,
lfConfig :260-280, TrainConfig :283-385, FlowModel :112-115, PLAQ_EXACT :37-47) without the
reference's import-time side effects (default-dtype switch, directory creation, logging)."""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from functools import reduce
from math import pi as PI
from typing import Any, List

import torch

TWO_PI = 2 * PI
DTYPE = torch.float64        # the HIP path computes in fp64 only (reference default: fp32, config.py:25-31,61)


def device() -> torch.device:
    """The HIP device this process drives (one process per GPU)."""
    if not torch.cuda.is_available():
        raise RuntimeError('fthmc_amd needs an MI355X: no HIP device visible and there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


# <cos P> of the infinite-volume 2D U(1) theory, I1(beta)/I0(beta)   (config.py:37-47)
PLAQ_EXACT = {
    1.0: 0.44638990, 1.5: 0.59613320, 2.0: 0.69777477, 2.5: 0.76499665,
    3.0: 0.80998540, 3.5: 0.84110373, 4.0: 0.86352290, 4.5: 0.88033150,
    5.0: 0.89338326, 5.5: 0.90381753, 6.0: 0.91235965, 6.5: 0.91948840,
    7.0: 0.92553246, 7.5: 0.93072510, 8.0: 0.93523590, 8.5: 0.93919160,
    9.0: 0.94268996, 9.5: 0.94580620,
}

LOGS_DIR = os.path.join(os.getcwd(), 'logs')


@dataclass
class FlowModel:
    prior: Any
    layers: torch.nn.ModuleList


@dataclass
class Param:
    beta: float = 6.0       # inverse coupling
    L: int = 8              # lattice extent (L x L)
    tau: float = 2.0        # trajectory length
    nstep: int = 10         # leapfrog steps per trajectory
    ntraj: int = 256        # trajectories per run
    nrun: int = 4           # independent runs
    nprint: int = 256
    seed: int = 11 * 13
    randinit: bool = False  # start from U(-pi, pi) instead of zeros

    def __post_init__(self):
        self.lat = [self.L, self.L]
        self.nd = len(self.lat)
        self.shape = [self.nd, *self.lat]
        self.volume = reduce(lambda x, y: x * y, self.lat)
        self.dt = self.tau / self.nstep
        lat = 'x'.join(str(x) for x in self.lat)
        self.logdir = os.path.join(LOGS_DIR, 'hmc', f'lat{lat}', f'beta{self.beta}', self.uniquestr())

    def initializer(self):
        x = torch.zeros([self.nd] + self.lat, dtype=DTYPE, device=device())
        if self.randinit:
            x.uniform_(-PI, PI)
        return x[None, :]

    def uniquestr(self):
        lat = 'x'.join(str(x) for x in self.lat)
        return '_'.join([f't{lat}', f'b{self.beta}', f'n{self.ntraj}', f't{self.tau}', f's{self.nstep}'])

    def titlestr(self):
        return ', '.join(['x'.join(str(x) for x in self.lat), f'beta: {self.beta}', f'tau: {self.tau}',
                          f'nstep: {self.nstep}', f'dt: {self.dt}'])

    def to_json(self):
        return dict(self.__dict__)

    def __repr__(self):
        return '\n'.join(['Param:', 16 * '-'] + [f'{k}={v}' for k, v in self.__dict__.items()])


@dataclass
class lfConfig:
    tau: float
    nstep: int

    def __post_init__(self):
        self.dt = self.tau / self.nstep

    def uniquestr(self):
        return '_'.join([f't{self.tau}', f's{self.nstep}', f'dt{self.dt}'])

    def titlestr(self):
        return ', '.join([f'tau: {self.tau}', f'nstep: {self.nstep}', f'dt: {self.dt}'])

    def __repr__(self):
        return json.dumps(dict(self.__dict__), indent=4)


@dataclass
class SchedulerConfig:
    factor: float = 0.98
    mode: str = 'min'
    patience: int = 10
    threshold: float = 1e-4
    threshold_mode: str = 'rel'
    cooldown: int = 0
    min_lr: float = 1e-5
    verbose: bool = False


@dataclass
class TrainConfig:
    L: int
    beta: float
    restore: bool = False
    activation_fn: str = 'silu'
    n_era: int = 10
    n_epoch: int = 100
    batch_size: int = 64
    base_lr: float = 0.001
    n_s_nets: int = 2            # mixture components of the tan transform
    n_layers: int = 24           # coupling layers
    kernel_size: int = 3
    with_force: bool = False
    print_freq: int = 50
    plot_freq: int = 50
    log_freq: int = 50
    debug: bool = False
    hidden_sizes: List[int] = field(default_factory=lambda: [8, 8])

    def __post_init__(self):
        self.lat = [self.L, self.L]
        self.nd = len(self.lat)
        self.shape = [self.nd, *self.lat]
        self.latstr = 'x'.join(str(x) for x in self.lat)
        self.volume = reduce(lambda x, y: x * y, self.lat)
        base = os.path.join(os.getcwd(), 'debug') if self.debug else os.path.join(LOGS_DIR, 'models')
        self.logdir = os.path.join(base, f'lat{self.latstr}', f'beta{self.beta}', self.uniquestr())
        # unlike the reference (config.py:304-345) nothing is created on disk here;
        # call update_logdirs() when checkpoints are really wanted
        self.dirs = {'logdir': self.logdir,
                     'training': os.path.join(self.logdir, 'training'),
                     'inference': os.path.join(self.logdir, 'inference'),
                     'ckpts': os.path.join(self.logdir, 'training', 'checkpoints')}

    def update_logdirs(self, logdir: str):
        self.logdir = logdir
        dtrain = os.path.join(logdir, 'training')
        self.dirs = {'logdir': logdir, 'training': dtrain, 'inference': os.path.join(logdir, 'inference'),
                     'ckpts': os.path.join(dtrain, 'checkpoints')}
        for d in self.dirs.values():
            os.makedirs(d, exist_ok=True)
        return self.dirs

    def uniquestr(self):
        hstr = ''.join(str(i) for i in self.hidden_sizes)
        return '_'.join([f'L{self.L}', f'b{self.beta}', f'nb{self.batch_size}', f'act{self.activation_fn}',
                         f'nh{self.n_layers}', f'ns{self.n_s_nets}', f'ks{self.kernel_size}', f'hl{hstr}',
                         f'lr{self.base_lr}', f'era{self.n_era}', f'epoch{self.n_epoch}'])

    def to_json(self):
        return dict(self.__dict__)

    def __repr__(self):
        return json.dumps({k: v for k, v in self.__dict__.items()}, indent=4, default=str)
