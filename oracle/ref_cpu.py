"""oracle/ref_cpu.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU fp64 restatement (PyTorch CPU ops + autograd) of the reference's ftHMC hot
path for 2D U(1).  It exists only to *check* the HIP path and to be timed as the
CPU baseline ("port") by bench.py.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it; nothing under fthmc_amd/ does.

Parity pin: every function here is checked against golden vectors generated
from the real reference (tests/golden/make_golden.py -> tests/golden/*.npz) by
tests/test_oracle_golden.py.

All functions are pure: lattice fields are `[B, 2, L, L]` float64 tensors
(angle of the U(1) link, mu-major), flow weights are explicit lists
`[(w0, b0, w1, b1, w2, b2), ...]` in PyTorch conv layout `[Cout, Cin, kh, kw]`.
Citations are `file:line` relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import Callable, List, Sequence, Tuple

import torch
import torch.nn.functional as F

PI = math.pi
TWO_PI = 2.0 * math.pi

Weights = Sequence[torch.Tensor]          # (w0, b0, w1, b1, ..., wn, bn)


# --------------------------------------------------------------------------
# angle maps                                                   (a4)
# --------------------------------------------------------------------------
def wrap(x: torch.Tensor) -> torch.Tensor:
    """[-pi, pi) map used by the packaged flow and by topological charge.

    fthmc/utils/layers.py:41-43 (torch_mod), fthmc/utils/qed_helpers.py:49-50
    (torch_wrap), fthmc/ft_hmc.py:173-175 (FieldTransformation.wrap).
    """
    return torch.remainder(x + PI, TWO_PI) - PI


def regularize(f: torch.Tensor) -> torch.Tensor:
    """fthmc/utils/qed_helpers.py:40-42 (also hmc_2dU1.py:127-129)."""
    f_ = (f - PI) / TWO_PI
    return TWO_PI * (f_ - torch.floor(f_) - 0.5)


# --------------------------------------------------------------------------
# plaquette / Wilson action / topological charge          (a1-a3)
# --------------------------------------------------------------------------
def plaq(x: torch.Tensor) -> torch.Tensor:
    """P[i,j] = x0[i,j] - x1[i,j] - x0[i,j+1] + x1[i+1,j], periodic.

    fthmc/utils/qed_helpers.py:94-105 (batch_plaqs), :80-90 (compute_u1_plaq),
    :170-175 (BatchAction._u1_plaq: same terms, different order of summation).
    """
    x0, x1 = x[:, 0], x[:, 1]
    return x0 - x1 - torch.roll(x0, -1, 2) + torch.roll(x1, -1, 1)


def _plaq_action_order(x: torch.Tensor) -> torch.Tensor:
    """Summation order of BatchAction._u1_plaq / plaq_phase
    (qed_helpers.py:170-175, :246-257): x0 + roll(x1) - roll(x0) - x1."""
    x0, x1 = x[:, 0], x[:, 1]
    return x0 + torch.roll(x1, -1, 1) - torch.roll(x0, -1, 2) - x1


def action(x: torch.Tensor, beta: float) -> torch.Tensor:
    """Per-chain Wilson action S_b = -beta sum_ij cos P   (qed_helpers.py:177-186)."""
    return (-beta) * torch.cos(_plaq_action_order(x)).sum(dim=(1, 2))


def charge(x: torch.Tensor) -> torch.Tensor:
    """Per-chain topological charge (qed_helpers.py:108-116)."""
    return wrap(plaq(x)).sum(dim=(1, 2)) / TWO_PI


def plaq_mean(x: torch.Tensor, beta: float) -> torch.Tensor:
    """<cos P> per chain = -S / (beta L^2)  (fthmc/hmc.py:103,125; ft_hmc.py:130-132)."""
    L = x.shape[-1]
    return -action(x, beta) / (beta * L * L)


def wilson_force(x: torch.Tensor, beta: float) -> torch.Tensor:
    """dS/dx by autograd, exactly as qed_helpers.py:265-272 does it."""
    xg = x.detach().clone().requires_grad_(True)
    s = action(xg, beta).sum()
    (g,) = torch.autograd.grad(s, xg)
    return g


def wilson_force_analytic(x: torch.Tensor, beta: float) -> torch.Tensor:
    """Closed form of the same gradient."""
    s = torch.sin(plaq(x))
    f0 = beta * (s - torch.roll(s, 1, 2))
    f1 = beta * (-s + torch.roll(s, 1, 1))
    return torch.stack((f0, f1), dim=1)


# --------------------------------------------------------------------------
# leapfrog + plain HMC                                    (a6, a7)
# --------------------------------------------------------------------------
def leapfrog(x: torch.Tensor, p: torch.Tensor,
             force_fn: Callable[[torch.Tensor], torch.Tensor],
             dt: float, nstep: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """qed_helpers.py:275-295 / hmc_2dU1.py:132-141 / ipynb/ft_hmc.py:394-418."""
    x_ = x + 0.5 * dt * p
    p_ = p + (-dt) * force_fn(x_)
    for _ in range(nstep - 1):
        x_ = x_ + dt * p_
        p_ = p_ + (-dt) * force_fn(x_)
    x_ = x_ + 0.5 * dt * p_
    return x_, p_


def hmc(x: torch.Tensor, v: torch.Tensor, u: torch.Tensor, beta: float,
        dt: float, nstep: int, joint: bool = False):
    """Plain HMC trajectory with supplied momenta `v` and uniforms `u`.

    qed_helpers.py:298-311.  joint=True reproduces the reference literally: the
    whole tensor is one system, one scalar H and one accept (`u` is a
    0-d tensor).  joint=False treats chains independently (`u` is `[B]`).
    Returns (dH, exp(-dH), acc, newx).
    """
    red = (lambda t: t.sum()) if joint else (lambda t: t.flatten(1).sum(1))
    act = (lambda y: action(y, beta).sum()) if joint else (lambda y: action(y, beta))
    h0 = act(x) + 0.5 * red(v * v)
    x_, v_ = leapfrog(x, v, lambda y: wilson_force(y, beta), dt, nstep)
    xr = regularize(x_)
    h1 = act(xr) + 0.5 * red(v_ * v_)
    dH = h1 - h0
    exp_mdH = torch.exp(-dH)
    acc = u < exp_mdH
    if joint:
        newx = xr if bool(acc) else x
    else:
        newx = torch.where(acc[:, None, None, None], xr, x)
    return dH, exp_mdH, acc, newx


# --------------------------------------------------------------------------
# stripe masks                                                 (a8)
# --------------------------------------------------------------------------
def layer_mu_off(i: int) -> Tuple[int, int]:
    """layers.py:409-412: mu = i % 2, off = (i // 2) % 4."""
    return i % 2, (i // 2) % 4


def stripe_masks(L: int, mu: int, off: int):
    """Active / frozen / passive plaquette masks and the active-link mask.

    layers.py:213-292.  mu=0: stripes are columns (axis "2", index j); mu=1:
    rows (axis "1", index i).  active: idx % 4 == off; frozen: off+1, off+2;
    passive: off+3.  Link mask: channel `mu` on the active stripe only.
    """
    idx = torch.arange(L)
    sel = (idx - off) % 4
    a1, f1, p1 = (sel == 0), ((sel == 1) | (sel == 2)), (sel == 3)
    if mu == 0:
        expand = lambda m: m[None, :].expand(L, L)
    else:
        expand = lambda m: m[:, None].expand(L, L)
    mA, mF, mP = (expand(m).to(torch.float64) for m in (a1, f1, p1))
    mL = torch.zeros(2, L, L, dtype=torch.float64)
    mL[mu] = mA
    return mA, mF, mP, mL


# --------------------------------------------------------------------------
# s/t network + tan-mixture transform                     (a9, a10)
# --------------------------------------------------------------------------
_ACT = {
    'silu': F.silu, 'swish': F.silu, None: F.silu,
    'relu': F.relu, 'leaky_relu': F.leaky_relu,
}


def conv_net(inp: torch.Tensor, w: Weights, act: str = 'silu') -> torch.Tensor:
    """layers.py:138-167: Conv2d(k, circular pad k//2) + act, no final act -- or a final tanh (use_final_tanh,
    layers.py:163-164), asked for here by an activation name that ends in '+tanh' (every caller hands `act` through)."""
    final_tanh = isinstance(act, str) and act.endswith('+tanh')
    if final_tanh:
        act = act[:-5]
    fn = _ACT[act]
    n = len(w) // 2
    h = inp
    for li in range(n):
        wt, b = w[2 * li], w[2 * li + 1]
        pad = wt.shape[-1] // 2
        h = F.conv2d(F.pad(h, (pad, pad, pad, pad), mode='circular'), wt, b)
        if li != n - 1:
            h = fn(h)
    return torch.tanh(h) if final_tanh else h


def tan_transform(x: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """layers.py:66-69."""
    return wrap(2 * torch.atan(torch.exp(s) * torch.tan(x / 2.)))


def tan_transform_logJ(x: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """layers.py:72-76."""
    return -torch.log(torch.exp(-s) * torch.cos(x / 2) ** 2
                      + torch.exp(s) * torch.sin(x / 2) ** 2)


def mixture_tan_transform(x, s):
    """layers.py:79-82."""
    return torch.mean(tan_transform(x, s), dim=1, keepdim=True)


def mixture_tan_transform_logJ(x, s):
    """layers.py:85-90."""
    return torch.logsumexp(tan_transform_logJ(x, s), dim=1) - math.log(s.shape[1])


# --------------------------------------------------------------------------
# coupling layer forward / reverse                     (a11-a13)
# --------------------------------------------------------------------------
def plaq_coupling_forward(P, w, mu, off, act='silu'):
    """NCPPlaqCouplingLayer.forward, layers.py:348-371."""
    L = P.shape[-1]
    mA, mF, mP, _ = stripe_masks(L, mu, off)
    x2 = mF * P
    net_out = conv_net(torch.stack((torch.cos(x2), torch.sin(x2)), dim=1), w, act)
    s, t = net_out[:, :-1], net_out[:, -1]
    x1 = (mA * P).unsqueeze(1)
    local_logJ = mA * mixture_tan_transform_logJ(x1, s)
    logJ = local_logJ.sum(dim=(1, 2))
    fx1 = mA * mixture_tan_transform(x1, s).squeeze(1)
    fx = mA * wrap(fx1 + t) + mP * P + mF * P
    return fx, logJ


def layer_forward(x, w, mu, off, act='silu'):
    """GaugeEquivCouplingLayer.forward, layers.py:196-202 -> (x', logJ[B])."""
    L = x.shape[-1]
    mL = stripe_masks(L, mu, off)[3]
    P = plaq(x)
    newP, logJ = plaq_coupling_forward(P, w, mu, off, act)
    d = newP - P
    dl = torch.stack((d, -d), dim=1)
    fx = mL * wrap(dl + x) + (1 - mL) * x
    return fx, logJ


def invert_transform_bisect(y, f, tol, max_iter, a=-PI, b=PI):
    """layers.py:294-320 (global-max stop rule)."""
    min_x = a * torch.ones_like(y)
    max_x = b * torch.ones_like(y)
    mid_x = (min_x + max_x) / 2
    with torch.no_grad():
        for _ in range(max_iter):
            mid_x = (min_x + max_x) / 2
            mid_val = f(mid_x)
            greater = (y > mid_val).to(y.dtype)
            err = torch.max(torch.abs(y - mid_val))
            if err < tol:
                return mid_x
            if torch.all((mid_x == min_x) + (mid_x == max_x)):
                return mid_x
            min_x = greater * mid_x + (1 - greater) * min_x
            max_x = (1 - greater) * mid_x + greater * max_x
    return mid_x


def plaq_coupling_reverse(fP, w, mu, off, act='silu', tol=1e-6, max_iter=1000):
    """NCPPlaqCouplingLayer.reverse, layers.py:373-396."""
    L = fP.shape[-1]
    mA, mF, mP, _ = stripe_masks(L, mu, off)
    fx2 = mF * fP
    net_out = conv_net(torch.stack((torch.cos(fx2), torch.sin(fx2)), dim=1), w, act)
    s, t = net_out[:, :-1], net_out[:, -1]
    x1 = wrap(mA * (fP - t).unsqueeze(1))
    x1 = invert_transform_bisect(
        x1, f=lambda z: mA * mixture_tan_transform(z, s), tol=tol, max_iter=max_iter)
    local_logJ = mA * mixture_tan_transform_logJ(x1, s)
    logJ = -local_logJ.sum(dim=(1, 2))
    x1 = x1.squeeze(1)
    P = mA * x1 + mP * fP + mF * fx2
    return P, logJ


def layer_reverse(fx, w, mu, off, act='silu', tol=1e-6, max_iter=1000):
    """GaugeEquivCouplingLayer.reverse, layers.py:204-210."""
    L = fx.shape[-1]
    mL = stripe_masks(L, mu, off)[3]
    newP = plaq(fx)
    P, logJ = plaq_coupling_reverse(newP, w, mu, off, act, tol, max_iter)
    d = P - newP
    dl = torch.stack((d, -d), dim=1)
    x = mL * wrap(dl + fx) + (1 - mL) * fx
    return x, logJ


# --------------------------------------------------------------------------
# flow, effective action, force                         (a14, a15)
# --------------------------------------------------------------------------
def flow_forward(x, flow: List[Weights], act='silu'):
    """ft_hmc.py:143-150 / qed_helpers.py:191-198 -> (x_phys, logdet[B])."""
    logdet = torch.zeros(x.shape[0], dtype=x.dtype)
    for i, w in enumerate(flow):
        mu, off = layer_mu_off(i)
        x, lj = layer_forward(x, w, mu, off, act)
        logdet = logdet + lj
    return x, logdet


def flow_reverse(x, flow: List[Weights], act='silu', tol=1e-6, max_iter=1000):
    """ft_hmc.py:152-160 / qed_helpers.py:201-209."""
    logdet = torch.zeros(x.shape[0], dtype=x.dtype)
    for i in reversed(range(len(flow))):
        mu, off = layer_mu_off(i)
        x, lj = layer_reverse(x, flow[i], mu, off, act, tol, max_iter)
        logdet = logdet + lj
    return x, logdet


def ft_action(x, flow, beta, act='silu'):
    """S_eff = S_W(F(x)) - sum_l logJ_l   (qed_helpers.py:212-223, ft_hmc.py:135-141)."""
    y, logdet = flow_forward(x, flow, act)
    return action(y, beta) - logdet


def ft_force(x, flow, beta, act='silu'):
    """d(sum_b S_eff)/dx by autograd (qed_helpers.py:226-242)."""
    xg = x.detach().clone().requires_grad_(True)
    s = ft_action(xg, flow, beta, act).sum()
    (g,) = torch.autograd.grad(s, xg)
    return g


# --------------------------------------------------------------------------
# ftHMC trajectories                                         (a16)
# --------------------------------------------------------------------------
def ft_hmc(x, v, u, flow, beta, dt, nstep, act='silu', mode='md', joint=False):
    """One ftHMC trajectory in the latent field `x` with supplied `v`, `u`.

    mode='md'      : intended integrator (ipynb/ft_hmc.py:394-435 minus the flow
                     inverse wrapper; same structure as qed_helpers.py:275-295),
                     end point mapped with `regularize` as the notebook does.
    mode='literal' : FieldTransformation.hmc as packaged (ft_hmc.py:180-224):
                     the MD evolution is computed and thrown away, the proposal
                     is x + dt/2 v with the *initial* v, end point
                     mapped with `wrap`.
    joint=True     : one scalar H for the whole tensor (reference, B=1);
    joint=False    : per-chain H and accept.
    Returns (dH, exp(-dH), acc, newx, H0, H1).
    """
    red = (lambda t: t.sum()) if joint else (lambda t: t.flatten(1).sum(1))
    act_fn = ((lambda y: ft_action(y, flow, beta, act).sum()) if joint
              else (lambda y: ft_action(y, flow, beta, act)))
    with torch.no_grad():
        h0 = act_fn(x) + 0.5 * red(v * v)
    if mode == 'md':
        x_, v_ = leapfrog(x, v, lambda y: ft_force(y, flow, beta, act), dt, nstep)
        xr = regularize(x_)
    elif mode == 'literal':
        x_, v_ = x + 0.5 * dt * v, v
        xr = wrap(x_)
    else:
        raise ValueError(mode)
    with torch.no_grad():
        h1 = act_fn(xr) + 0.5 * red(v_ * v_)
    dH = h1 - h0
    exp_mdH = torch.exp(-dH)
    acc = u < exp_mdH
    if joint:
        newx = xr if bool(acc) else x
    else:
        newx = torch.where(acc[:, None, None, None], xr, x)
    return dH, exp_mdH, acc, newx, h0, h1


def ft_hmc_phys(field, v, u, flow, beta, dt, nstep, act='silu', tol=1e-6, max_iter=1000):
    """One ftHMC trajectory on the PHYSICAL field (ipynb/ft_hmc.py:420-435): x = F^-1(field) by the
    reference's bisection (`tol` = layer.inv_prec, global stop rule), momenta `v` and uniform `u` as drawn
    there (randn_like(x), rand([])), trajectory of the whole tensor as ONE system, newfield = F(newx).
    Returns (dH, exp(-dH), acc, newfield, x)."""
    with torch.no_grad():
        x = flow_reverse(field, flow, act, tol, max_iter)[0]
    dH, exp_mdH, acc, newx, _, _ = ft_hmc(x, v, u, flow, beta, dt, nstep, act, mode='md', joint=True)
    with torch.no_grad():
        newfield = flow_forward(newx, flow, act)[0]
    return dH, exp_mdH, acc, newfield, x


# --------------------------------------------------------------------------
# flow-proposal independence Metropolis                       (3)
# --------------------------------------------------------------------------
def mcmc_chain(proposals, uniforms):
    """Accept chain of samplers.make_mcmc_ensemble (samplers.py:182-259) on recorded proposals
    [(x, logq, logp)] and the uniforms of steps 1, 2, ...: the first proposal is accepted, afterwards
    `draw < min(1, exp((logp' - logq') - (logp - logq)))`; q = topological charge of the current
    configuration, dqsq = (q - q_previous)^2.  Returns dict of float64 lists q, dqsq, logq, logp, acc."""
    hist = {k: [] for k in ('q', 'dqsq', 'logq', 'logp', 'acc')}
    xarr = []
    draws = iter(uniforms)
    for x_new, logq_new, logp_new in proposals:
        if not hist['logp']:
            accepted = True
            q_old = charge(x_new[None])
        else:
            q_old = charge(xarr[-1][None])
            logp_old, logq_old = hist['logp'][-1], hist['logq'][-1]
            p_accept = min(1, torch.exp((logp_new - logq_new) - (logp_old - logq_old)))
            if next(draws) < p_accept:
                accepted = True
            else:
                accepted = False
                x_new, logp_new, logq_new = xarr[-1], logp_old, logq_old
        q_new = charge(x_new[None])
        xarr.append(x_new)
        for k, val in (('q', q_new), ('dqsq', (q_new - q_old) ** 2), ('logp', logp_new), ('logq', logq_new),
                       ('acc', float(accepted))):
            hist[k].append(val)
    return {k: [float(v) for v in vals] for k, vals in hist.items()}


# --------------------------------------------------------------------------
# training step math                                         (a17)
# --------------------------------------------------------------------------
def prior_log_prob(x):
    """MultivariateUniform(-pi, pi).log_prob summed over the lattice
    (distributions.py:65-76; train.py:64-65): -2 L^2 log(2 pi)."""
    n = x[0].numel()
    return torch.full((x.shape[0],), -n * math.log(TWO_PI), dtype=x.dtype)


def calc_dkl(logp, logq):
    """distributions.py:23-24."""
    return (logq - logp).mean()


def calc_ess(logp, logq):
    """distributions.py:27-37."""
    logw = logp - logq
    log_ess = 2 * torch.logsumexp(logw, dim=0) - torch.logsumexp(2 * logw, dim=0)
    return torch.exp(log_ess) / len(logw)


def train_loss(xi, flow, beta, act='silu'):
    """Forward half of train.train_step (train.py:191-202) for a fixed prior
    draw `xi`: returns dict(loss_dkl, ess, logp, logq, x, q, qi, plaq)."""
    x, logdet = flow_forward(xi, flow, act)          # samplers.py:40-56
    logq = prior_log_prob(xi) - logdet
    logp = -action(x, beta)
    L = xi.shape[-1]
    return {
        'loss_dkl': calc_dkl(logp, logq),
        'ess': calc_ess(logp, logq),
        'logp': logp, 'logq': logq, 'x': x,
        'q': charge(x), 'qi': charge(xi),
        'plaq': logp / (beta * L * L),
    }


def train_grads(xi, flow, beta, act='silu'):
    """loss_dkl.backward() of train.py:210 wrt every conv weight/bias."""
    leaves = [[t.detach().clone().requires_grad_(True) for t in w] for w in flow]
    out = train_loss(xi, leaves, beta, act)
    flat = [t for w in leaves for t in w]
    grads = torch.autograd.grad(out['loss_dkl'], flat)
    it = iter(grads)
    return out, [[next(it) for _ in w] for w in leaves]


# --------------------------------------------------------------------------
# helpers for tests / bench
# --------------------------------------------------------------------------
def default_flow(n_layers, gen, hidden=(8, 8), n_mix=2, k=3):
    """Conv2d default init (Kaiming-uniform a=sqrt(5) == U(-1/sqrt(fan_in), ..);
    SURVEY Q6: set_weights is a no-op in the reference) drawn from `gen`."""
    sizes = [2, *hidden, n_mix + 1]
    flow = []
    for _ in range(n_layers):
        w = []
        for ci, co in zip(sizes[:-1], sizes[1:]):
            bound = 1.0 / math.sqrt(ci * k * k)
            w.append((torch.rand(co, ci, k, k, generator=gen, dtype=torch.float64) * 2 - 1) * bound)
            w.append((torch.rand(co, generator=gen, dtype=torch.float64) * 2 - 1) * bound)
        flow.append(tuple(w))
    return flow
