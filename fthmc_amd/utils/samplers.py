"""fthmc/utils/samplers.py: only the piece the training step needs (apply_flow_to_prior)."""
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
