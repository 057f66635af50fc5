"""fthmc/utils/distributions.py: uniform prior, reverse KL, ESS, bootstrap."""
from __future__ import annotations

from math import log, pi as PI

import numpy as np
import torch
import torch.nn as nn


def bootstrap(x: np.ndarray, *, nboot: int, binsize: int):
    """Binned bootstrap of the mean -> (mean of the resampled means, their std); distributions.py:13-20.
    Bins are averaged first, all resampling indices are drawn in one call (same draws, in the same order, as
    one call per resample)."""
    x = np.asarray(x)
    bins = x.reshape(-1, binsize, *x.shape[1:]).mean(axis=1)
    pick = np.random.randint(len(bins), size=(nboot, len(bins)))
    means = bins[pick].mean(axis=1)                          # [nboot, *trailing dims of x]
    return means.mean(), means.std()


def calc_dkl(logp: torch.Tensor, logq: torch.Tensor):
    """Reverse KL estimate E_q[log q - log p] (distributions.py:23-24)."""
    return torch.mean(logq - logp)


def calc_ess(logp: torch.Tensor, logq: torch.Tensor):
    """Effective sample size per configuration, (sum w)^2 / (N sum w^2) with w = p / q, in log space
    (distributions.py:27-37)."""
    logw = logp - logq
    n = logw.shape[0]
    return (2 * torch.logsumexp(logw, 0) - torch.logsumexp(2 * logw, 0)).exp() / n


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
