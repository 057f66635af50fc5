"""fthmc/utils/samplers.py on the HIP path: apply_flow_to_prior (used by the training step) and the
flow-proposal independence Metropolis ensemble (make_mcmc_ensemble, generate_ensemble)."""
from __future__ import annotations

import torch
import torch.nn as nn


def apply_flow_to_prior(prior, coupling_layers: nn.ModuleList, *, batch_size: int, xi: torch.Tensor = None):
    """samplers.py:40-56: x = F(xi), logq = log_prob(xi) - sum_l logJ_l."""
    if xi is None:
        xi = prior.sample_n(batch_size)
    x = xi.clone().to(torch.float64)
    logq = prior.log_prob(x)
    for layer in coupling_layers:
        x, logJ = layer.forward(x)
        logq = logq - logJ
    return x, xi, logq


# ---------------------------------------------------------------- independence Metropolis
def serial_sample_generator(model, action, batch_size, N_samples, reference_literal: bool = False):
    """samplers.py:123-137: proposals one at a time; every `batch_size` samples a new batch is pushed
    through the flow on the GPU (one fused forward sweep, no autograd).  Yields (x, logq, logp, q).

    reference_literal=True reproduces the generator as packaged: it unpacks `_, x, logq =
    apply_flow_to_prior(...)` although that function returns (x, xi, logq), so its proposals are the
    PRIOR draws xi weighted with logp = -S(xi) (and logq of the flowed sample).  The default proposes the
    flowed sample x = F(xi) with logp = -S(x), which is what the method describes (inference.py:85-153
    builds on the same generator)."""
    from .. import ops
    from . import qed_helpers as qed
    from .layers import flow_activation, flow_weights
    layers, prior = (model['layers'], model['prior']) if isinstance(model, dict) else (model.layers, model.prior)
    layers.eval()
    x = logq = logp = q = None
    for i in range(N_samples):
        bi = i % batch_size
        if bi == 0:
            with torch.no_grad():
                xi = prior.sample_n(batch_size)
                xf, logdet = ops.flow_forward(xi, flow_weights(layers, xi.device), len(layers), flow_activation(layers))
                x = xi if reference_literal else xf
                logq = (prior.log_prob(xi) - logdet).cpu()
                logp = (-action(x)).cpu()
                q = qed.batch_charges(x).cpu()
        yield x[bi], logq[bi], logp[bi], q[bi]


def make_mcmc_ensemble(model, action_fn, batch_size, num_samples, writer=None, keep_x: bool = False,
                       proposals=None, uniforms=None, reference_literal: bool = False):
    """samplers.py:182-259: flow-proposal independence Metropolis.  Proposals are generated and scored in
    batches on the GPU; the accept chain is inherently serial and runs on the host: the first proposal is
    accepted, then `draw < min(1, exp((logp' - logq') - (logp - logq)))` with one uniform per step.
    Returns numpy histories q, dqsq, logq, logp, acc (float64; the reference's are rounded to float32 by
    its `torch.Tensor(v)`), and the configurations with keep_x.

    `proposals`: iterable of (x, logq, logp, q) replacing the flow generator; `uniforms`: iterable of the
    draws of steps 1, 2, ... replacing torch.rand(1) -- both for reproducible checks against recorded
    reference chains."""
    import numpy as np
    history = {k: [] for k in ('q', 'dqsq', 'logq', 'logp', 'acc')}
    xarr = []
    x_old = q_old = None
    gen = proposals if proposals is not None else serial_sample_generator(model, action_fn, batch_size, num_samples,
                                                                            reference_literal=reference_literal)
    draws = iter(uniforms) if uniforms is not None else None
    for step, (x_new, logq_new, logp_new, q_new) in enumerate(gen):
        if step >= num_samples:
            break
        if not history['logp']:
            accepted = True                                   # the chain has to start somewhere
            q_prev = q_new
        else:
            q_prev = q_old
            logp_old, logq_old = history['logp'][-1], history['logq'][-1]
            p_accept = min(1.0, float(torch.exp(torch.as_tensor((logp_new - logq_new) - (logp_old - logq_old)))))
            draw = float(next(draws)) if draws is not None else float(torch.rand(1))
            accepted = draw < p_accept
            if not accepted:
                x_new, q_new, logp_new, logq_new = x_old, q_old, logp_old, logq_old
        x_old, q_old = x_new, q_new
        if keep_x:
            xarr.append(x_new)
        history['q'].append(q_new)
        history['dqsq'].append((q_new - q_prev) ** 2)
        history['logp'].append(logp_new)
        history['logq'].append(logq_new)
        history['acc'].append(float(accepted))
        if writer is not None:
            for k in history:
                writer.add_scalar(f'inference/{k}', float(history[k][-1]), global_step=step)
    out = {k: np.array([float(v) for v in vals]) for k, vals in history.items()}
    if keep_x:
        out['x'] = torch.stack(xarr) if xarr else None
    return out


def generate_ensemble(model, action, ensemble_size: int = 1024, batch_size: int = 64, nboot: int = 100,
                      binsize: int = 16):
    """samplers.py:80-104: topological susceptibility <Q^2> of a flow-proposal ensemble (bootstrap)."""
    from .distributions import bootstrap
    history = make_mcmc_ensemble(model, action, batch_size, ensemble_size)
    qsq_mean, qsq_err = bootstrap(history['q'] ** 2, nboot=nboot, binsize=binsize)
    return {'history': history, 'suscept_mean': qsq_mean, 'suscept_err': qsq_err}
