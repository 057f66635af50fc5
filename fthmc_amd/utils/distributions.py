"""fthmc/utils/distributions.py: uniform prior, reverse KL, ESS, bootstrap."""
from __future__ import annotations

from math import log, pi as PI

import numpy as np
import torch
import torch.nn as nn


def bootstrap(x: np.ndarray, *, nboot: int, binsize: int):
    """distributions.py:13-20 (host statistics)."""
    boots = []
    x = x.reshape(-1, binsize, *x.shape[1:])
    for _ in range(nboot):
        boots.append(np.mean(x[np.random.randint(len(x), size=len(x))], axis=(0, 1)))
    return np.mean(boots), np.std(boots)


def calc_dkl(logp: torch.Tensor, logq: torch.Tensor):
    """distributions.py:23-24."""
    return (logq - logp).mean()


def calc_ess(logp: torch.Tensor, logq: torch.Tensor):
    """distributions.py:27-37."""
    logw = logp - logq
    log_ess = 2 * torch.logsumexp(logw, dim=0) - torch.logsumexp(2 * logw, dim=0)
    return torch.exp(log_ess) / len(logw)


class BasePrior(nn.Module):
    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def sample_n(self, batch_size: int) -> torch.Tensor:
        raise NotImplementedError


class MultivariateUniform(BasePrior):
    """Uniform on [a, b] per link (distributions.py:65-76); tensors live where `a` lives."""

    def __init__(self, a: torch.Tensor, b: torch.Tensor):
        super().__init__()
        a = a.to(torch.float64)
        b = torch.broadcast_to(b.to(torch.float64).to(a.device), a.shape).clone()
        self.a, self.b = a, b

    def log_prob(self, x: torch.Tensor):
        lp = -torch.log(self.b - self.a)                      # [2, L, L]
        return lp.sum().expand(x.shape[0]).clone()

    def sample_n(self, batch_size: int):
        u = torch.rand((batch_size,) + tuple(self.a.shape), dtype=torch.float64, device=self.a.device)
        return self.a + (self.b - self.a) * u
