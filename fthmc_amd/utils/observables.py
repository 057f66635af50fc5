"""Host-side analysis of sampling histories: topological-charge tunnelling versus MD-time lag and the
topological susceptibility.  numpy only; the histories come from
`FieldTransformation.run` / `run_hmc` (lists or arrays of per-trajectory charges, optionally per chain).

Reference: ipynb/ft_hmc.py:16-53 (`average`, `sigma`, `sub_avg`, `block_list`, `change_sqr`, `change_sqr_vs_dt`),
:168-176 (`save_topo_change_sqr`); hmc_2dU1.py:661 (susceptibility 1.23 +- 0.02 at L=8, beta=2).  Pinned to the
outputs of those reference functions on seeded charge histories (tests/golden/observables.npz, written by
tests/golden/make_golden.py section 10 and compared in the CPU test suite)."""
from typing import Optional, Sequence

import numpy as np

from .distributions import bootstrap

N_BLOCK = 16          # blocks used for the error of a mean (the reference's `n_block`)


def block_means(v: np.ndarray, n_block: int = N_BLOCK) -> np.ndarray:
    """Means of `n_block` equal blocks of `v` along axis 0; a remainder is dropped from the FRONT (the
    oldest samples), a series shorter than `n_block` falls back to blocks of one."""
    v = np.asarray(v, dtype=np.float64)
    n = v.shape[0]
    if n == 0:
        return v[:0]
    size = n // n_block
    if size < 1:
        size, n_block = 1, n
    start = n - n_block * size
    return v[start:].reshape(n_block, size, *v.shape[1:]).mean(axis=1)


def sub_avg(v) -> np.ndarray:
    """v - mean(v) (ipynb/ft_hmc.py:25-27)."""
    v = np.asarray(v, dtype=np.float64)
    return v - v.mean(axis=0)


def sigma(v, reference_literal: bool = False) -> float:
    """Error of the mean of `v`.  The standard error sqrt(var / (n - 1)) (var = population variance); with
    `reference_literal` what ipynb/ft_hmc.py:20-23 returns: var / sqrt(n - 1), a variance where a standard
    deviation is meant (pinned by tests/golden/observables.npz)."""
    v = np.asarray(v, dtype=np.float64)
    if len(v) < 2:
        return float('nan')
    var = float(np.mean((v - v.mean()) ** 2))
    return var / np.sqrt(len(v) - 1) if reference_literal else float(np.sqrt(var / (len(v) - 1)))


def change_sqr(q: np.ndarray, lag: int, n_block: int = N_BLOCK, reference_literal: bool = False):
    """<(Q(t + lag) - Q(t))^2> over the history (axis 0 = trajectory; further axes = chains, averaged) and the
    error of that mean from `n_block` block means -> (mean, sigma).  ipynb/ft_hmc.py:42-50; `reference_literal`
    selects the reference's error formula (see `sigma`)."""
    q = np.asarray(q, dtype=np.float64)
    if lag < 1 or q.shape[0] <= lag:
        return float('nan'), float('nan')
    d2 = (q[lag:] - q[:-lag]) ** 2
    if d2.ndim > 1:
        d2 = d2.reshape(d2.shape[0], -1).mean(axis=1)
    bm = block_means(d2, n_block)
    return float(bm.mean()), sigma(bm, reference_literal)


def change_sqr_vs_dt(q: np.ndarray, dt_range: int = 10, n_block: int = N_BLOCK, reference_literal: bool = False):
    """[[lag, mean, sigma], ...] for lag = 1..dt_range: how fast the topological charge decorrelates in units
    of trajectories (the figure of merit of arXiv:2112.01586; ipynb/ft_hmc.py:52-53)."""
    return [[lag, *change_sqr(q, lag, n_block, reference_literal)] for lag in range(1, dt_range + 1)]


def topological_susceptibility(q: np.ndarray, volume: int, nboot: int = 100, binsize: int = 16):
    """chi_Q = (<Q^2> - <Q>^2) / V with a binned-bootstrap error -> (chi, err)."""
    q = np.asarray(q, dtype=np.float64).reshape(-1)
    q2_mean, q2_err = bootstrap(q ** 2, nboot=nboot, binsize=binsize)
    chi = (q2_mean - q.mean() ** 2) / volume
    return float(chi), float(q2_err / volume)


def save_topo_change_sqr(fn: str, q_history: Sequence, drop_len: Optional[int] = None, dt_range: int = 10,
                         reference_literal: bool = False):
    """Write `lag mean sigma` lines for the history with its first `drop_len` entries (default: a third,
    `len // 3`, ipynb/ft_hmc.py:168-176) dropped as thermalisation."""
    q = np.asarray(q_history, dtype=np.float64)
    rows = change_sqr_vs_dt(q[len(q) // 3 if drop_len is None else drop_len:], dt_range,
                            reference_literal=reference_literal)
    with open(fn, 'w') as f:
        for lag, mean, sig in rows:
            f.write(f'{lag} {mean} {sig}\n')
    return rows
