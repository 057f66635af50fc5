"""fthmc/hmc.py::run_hmc on the HIP path (plain HMC experiment loop, no plots)."""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from .config import DTYPE, Param
from .utils import qed_helpers as qed


def run_hmc(param: Param, x: torch.Tensor = None, plot_metrics: bool = False, figsize=None, use_title: bool = True,
            save_data: bool = False, nplot: int = 10):
    """hmc.py:57-175: `param.nrun` experiments of `param.ntraj` trajectories each.
    Returns (fields_arr, histories) with the reference's metric keys."""
    action = qed.BatchAction(param.beta)
    histories, fields_arr, run_times = {}, [], []
    for n in range(param.nrun):
        t0 = time.time()
        x = param.initializer()
        q = qed.batch_charges(x)
        xarr, history = [], {}
        for i in range(param.ntraj):
            t1 = time.time()
            dH, exp_mdH, acc, x = qed.hmc(param, x, verbose=False)
            qold = history['q'][-1] if 'q' in history else q
            qnew = qed.batch_charges(x)
            dq = torch.sqrt((qnew - qold) ** 2)
            plaq = (-1.) * action(x) / (param.beta * param.volume)
            xarr.append(x)
            metrics = {'traj': n * param.ntraj + i + 1, 'dt': time.time() - t1, 'acc': acc.to(DTYPE), 'dH': dH,
                       'plaq': plaq, 'q': qnew, 'dq': dq}
            for k, v in metrics.items():
                history.setdefault(k, []).append(v)
            if param.nprint and (i - 1) % param.nprint == 0:
                print(f"run {n} traj {i}: acc={float(acc):.0f} dH={float(dH):.4g} plaq={float(plaq):.6f} "
                      f"q={float(qnew):.2f}", flush=True)
        run_times.append(time.time() - t0)
        histories[n] = history
        fields_arr.append(xarr)
    if save_data:
        os.makedirs(param.logdir, exist_ok=True)
        np.savez(os.path.join(param.logdir, 'hmc_histories.npz'),
                 **{f'run{n}_{k}': np.array([float(torch.as_tensor(v).reshape(-1)[0]) for v in vals])
                    for n, h in histories.items() for k, vals in h.items()})
    return fields_arr, histories
