"""`torch.ops.fthmc_hip.*`: the torch operator boundary of SURVEY.md §8(b), level 2.

Every operator is a thin dispatcher entry over the C ABI (include/fthmc_hip.h):
inputs are contiguous fp64 tensors on the HIP device, outputs are allocated by torch, the launch
goes to the current HIP stream and nothing synchronises.  Only the device dispatch key is
registered: a CPU tensor raises NotImplementedError from the dispatcher (there is no CPU fallback).
The differentiable operators carry autograd formulas that call the hand-written backward kernels,
so `loss.backward()` through `torch.ops.fthmc_hip.flow_layer_fwd` never builds an ATen graph.

Two registrations of the SAME schemas (`BACKEND` says which one this process uses):
  'compiled': `libfthmc_torch.so` -- a compiled TORCH_LIBRARY(fthmc_hip) (csrc/torch_library.cpp, built by csrc/Makefile next to
              libfthmc_hip.so): definitions and device implementations in C++, straight onto the C ABI; this module attaches
              the shape functions for tracing and the autograd formulas to them.  Used whenever the library is there.
  'python'  : the same operators defined with torch.library.custom_op over `ops` (ctypes), when the compiled library has not
              been built (or FTHMC_TORCH_OPS=python asks for it: the two are tested against each other).

Operators (reference call site each one replaces):
  wilson_action_charge(x, beta) -> (S, Q, plaq)      qed_helpers.py:94-116,177-186
  wilson_force(x, beta) -> F                         qed_helpers.py:265-272
  hmc_trajectory(x, v, u, beta, dt, nstep) -> (x_new, dH, acc)          qed_helpers.py:275-311
  flow_layer_fwd(x, w, mu, off, n_mix, act, hidden=None, kernel_size=3) -> (y, logJ)      layers.py:196-202,348-371
  flow_layer_bwd_x(x, gy, glogJ, w, mu, off, n_mix, act, hidden, kernel_size) -> gx        (autograd of the above)
  flow_layer_bwd_w(x, gy, glogJ, w, mu, off, n_mix, act, hidden, kernel_size) -> gw[params]
  flow_layer_rev(y, w, mu, off, n_mix, act, tol, hidden, kernel_size) -> (x, logJ)         layers.py:204-210,294-320,373-396
  ft_action_force(x, w_all, n_layers, beta, act, n_mix=2, hidden, kernel_size) -> (S_eff, logdet, F)  qed_helpers.py:212-242
  fthmc_trajectory(x, v, u, w_all, n_layers, beta, dt, nstep, mode, act, n_mix=2, hidden, kernel_size)
                   -> (x_new, dH, acc, plaq, Q)                        ft_hmc.py:180-224 / ipynb/ft_hmc.py:394-435
  train_grad(xi, w_all, n_layers, beta, act, n_mix=2, hidden, kernel_size) -> (x, logq, logp, gw)     train.py:162-228
`act` is the integer code of fthmc_hip.h (0 silu/swish, 1 relu, 2 leaky_relu); `mode` 0 = MD
semantics, 1 = literal reference leapfrog.  The s/t net's shape travels IN the schema, as plain
integers: `n_mix` mixture components, `hidden` = hidden_sizes (None = the reference default [8, 8]), `kernel_size` --
the C ABI's fthmc_arch_t of the call; a weight tensor is a plain tensor here and carries no shape of its own.
"""
import os
from typing import Optional, Sequence

import torch

from . import _lib, ops
from ._lib import FthmcError

_COMPILED_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libfthmc_torch.so')
BACKEND = 'compiled' if os.path.exists(_COMPILED_PATH) and os.environ.get('FTHMC_TORCH_OPS', '') != 'python' else 'python'
if BACKEND == 'compiled':
    _lib.load()                                  # libfthmc_hip.so first: the compiled operators link against that same object
    torch.ops.load_library(_COMPILED_PATH)

_ACT = {0: 'silu', 1: 'relu', 2: 'leaky_relu'}
_MODE = {0: 'md', 1: 'literal'}
_DEV = 'cuda'       # ROCm devices live under torch's "cuda" device type / dispatch key


def _act(code: int) -> str:
    if code not in _ACT:
        raise FthmcError(f'act: unknown activation code {code}')
    return _ACT[code]


def _arch(n_mix: int, hidden, kernel_size: int):
    """(hidden_sizes, kernel_size, n_mix) of a call from the schema's integers (a net that ends in a tanh -- never built by the
    reference -- is served through `ops` with `arch=(hidden, k, n_mix, True)`, not through these schemas)"""
    return (tuple(int(h) for h in hidden) if hidden is not None else (8, 8), int(kernel_size), int(n_mix))


if BACKEND == 'python':
    @torch.library.custom_op('fthmc_hip::wilson_action_charge', mutates_args=(), device_types=_DEV)
    def wilson_action_charge(x: torch.Tensor, beta: float) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        return ops.wilson_action_charge(x, beta)


    @torch.library.custom_op('fthmc_hip::wilson_force', mutates_args=(), device_types=_DEV)
    def wilson_force(x: torch.Tensor, beta: float) -> torch.Tensor:
        return ops.wilson_force(x, beta)


    @torch.library.custom_op('fthmc_hip::hmc_trajectory', mutates_args=(), device_types=_DEV)
    def hmc_trajectory(x: torch.Tensor, v: torch.Tensor, u: torch.Tensor, beta: float, dt: float,
                       nstep: int) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        r = ops.hmc_trajectory(x, v, u, beta, dt, nstep)
        return r['x_new'], r['dH'], r['acc']


    @torch.library.custom_op('fthmc_hip::flow_layer_fwd', mutates_args=(), device_types=_DEV)
    def flow_layer_fwd(x: torch.Tensor, w: torch.Tensor, mu: int, off: int, n_mix: int, act: int,
                       hidden: Optional[Sequence[int]] = None, kernel_size: int = 3) -> tuple[torch.Tensor, torch.Tensor]:
        return ops.flow_layer_fwd(x, w, mu, off, _act(act), arch=_arch(n_mix, hidden, kernel_size))


    @torch.library.custom_op('fthmc_hip::flow_layer_bwd_x', mutates_args=(), device_types=_DEV)
    def flow_layer_bwd_x(x: torch.Tensor, gy: torch.Tensor, glogJ: torch.Tensor, w: torch.Tensor, mu: int, off: int,
                         n_mix: int, act: int, hidden: Optional[Sequence[int]] = None, kernel_size: int = 3) -> torch.Tensor:
        return ops.flow_layer_bwd(x, w, gy, glogJ, mu, off, _act(act), need_gw=False, arch=_arch(n_mix, hidden, kernel_size))[0]


    @torch.library.custom_op('fthmc_hip::flow_layer_bwd_w', mutates_args=(), device_types=_DEV)
    def flow_layer_bwd_w(x: torch.Tensor, gy: torch.Tensor, glogJ: torch.Tensor, w: torch.Tensor, mu: int, off: int,
                         n_mix: int, act: int, hidden: Optional[Sequence[int]] = None, kernel_size: int = 3) -> torch.Tensor:
        return ops.flow_layer_bwd(x, w, gy, glogJ, mu, off, _act(act), need_gw=True, arch=_arch(n_mix, hidden, kernel_size))[1]


    @torch.library.custom_op('fthmc_hip::flow_layer_bwd', mutates_args=(), device_types=_DEV)
    def flow_layer_bwd(x: torch.Tensor, gy: torch.Tensor, glogJ: torch.Tensor, w: torch.Tensor, mu: int, off: int,
                       n_mix: int, act: int, hidden: Optional[Sequence[int]] = None,
                       kernel_size: int = 3) -> tuple[torch.Tensor, torch.Tensor]:
        """Both gradients from one launch (what the autograd formula of flow_layer_fwd uses)."""
        gx, gw = ops.flow_layer_bwd(x, w, gy, glogJ, mu, off, _act(act), need_gw=True, arch=_arch(n_mix, hidden, kernel_size))
        return gx, gw


    @torch.library.custom_op('fthmc_hip::flow_layer_rev', mutates_args=(), device_types=_DEV)
    def flow_layer_rev(y: torch.Tensor, w: torch.Tensor, mu: int, off: int, n_mix: int, act: int, tol: float,
                       hidden: Optional[Sequence[int]] = None, kernel_size: int = 3) -> tuple[torch.Tensor, torch.Tensor]:
        return ops.flow_layer_rev(y, w, mu, off, _act(act), tol, arch=_arch(n_mix, hidden, kernel_size))


    @torch.library.custom_op('fthmc_hip::ft_action_force', mutates_args=(), device_types=_DEV)
    def ft_action_force(x: torch.Tensor, w_all: torch.Tensor, n_layers: int, beta: float, act: int, n_mix: int = 2,
                        hidden: Optional[Sequence[int]] = None,
                        kernel_size: int = 3) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        a = _arch(n_mix, hidden, kernel_size)
        S, logdet, _, _ = ops.ft_action(x, w_all, n_layers, beta, _act(act), arch=a)
        return S, logdet, ops.ft_force(x, w_all, n_layers, beta, _act(act), arch=a)


    @torch.library.custom_op('fthmc_hip::fthmc_trajectory', mutates_args=(), device_types=_DEV)
    def fthmc_trajectory(x: torch.Tensor, v: torch.Tensor, u: torch.Tensor, w_all: torch.Tensor, n_layers: int,
                         beta: float, dt: float, nstep: int, mode: int, act: int, n_mix: int = 2,
                         hidden: Optional[Sequence[int]] = None,
                         kernel_size: int = 3) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        if mode not in _MODE:
            raise FthmcError(f'mode: expected 0 (md) or 1 (literal), got {mode}')
        r = ops.ft_trajectory(x, v, u, w_all, n_layers, beta, dt, nstep, _act(act), _MODE[mode], arch=_arch(n_mix, hidden, kernel_size))
        return r['x_new'], r['dH'], r['acc'], r['plaq'], r['Q']


    @torch.library.custom_op('fthmc_hip::train_grad', mutates_args=(), device_types=_DEV)
    def train_grad(xi: torch.Tensor, w_all: torch.Tensor, n_layers: int, beta: float, act: int, n_mix: int = 2,
                   hidden: Optional[Sequence[int]] = None,
                   kernel_size: int = 3) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        r = ops.train_grad(xi, w_all, n_layers, beta, _act(act), arch=_arch(n_mix, hidden, kernel_size))
        return r['x'], r['logq'], r['logp'], r['gw']


# ---------------------------------------------------------------- shapes for tracing (meta tensors)
def _b(x):
    return x.new_empty(x.shape[0])


@torch.library.register_fake('fthmc_hip::wilson_action_charge')
def _(x, beta):
    return _b(x), _b(x), _b(x)


@torch.library.register_fake('fthmc_hip::wilson_force')
def _(x, beta):
    return torch.empty_like(x)


@torch.library.register_fake('fthmc_hip::hmc_trajectory')
def _(x, v, u, beta, dt, nstep):
    return torch.empty_like(x), _b(x), _b(x)


@torch.library.register_fake('fthmc_hip::flow_layer_fwd')
def _(x, w, mu, off, n_mix, act, hidden=None, kernel_size=3):
    return torch.empty_like(x), _b(x)


@torch.library.register_fake('fthmc_hip::flow_layer_bwd_x')
def _(x, gy, glogJ, w, mu, off, n_mix, act, hidden=None, kernel_size=3):
    return torch.empty_like(x)


@torch.library.register_fake('fthmc_hip::flow_layer_bwd_w')
def _(x, gy, glogJ, w, mu, off, n_mix, act, hidden=None, kernel_size=3):
    return x.new_empty(w.numel())


@torch.library.register_fake('fthmc_hip::flow_layer_bwd')
def _(x, gy, glogJ, w, mu, off, n_mix, act, hidden=None, kernel_size=3):
    return torch.empty_like(x), x.new_empty(w.numel())


@torch.library.register_fake('fthmc_hip::flow_layer_rev')
def _(y, w, mu, off, n_mix, act, tol, hidden=None, kernel_size=3):
    return torch.empty_like(y), _b(y)


@torch.library.register_fake('fthmc_hip::ft_action_force')
def _(x, w_all, n_layers, beta, act, n_mix=2, hidden=None, kernel_size=3):
    return _b(x), _b(x), torch.empty_like(x)


@torch.library.register_fake('fthmc_hip::fthmc_trajectory')
def _(x, v, u, w_all, n_layers, beta, dt, nstep, mode, act, n_mix=2, hidden=None, kernel_size=3):
    return torch.empty_like(x), _b(x), _b(x), _b(x), _b(x)


@torch.library.register_fake('fthmc_hip::train_grad')
def _(xi, w_all, n_layers, beta, act, n_mix=2, hidden=None, kernel_size=3):
    return torch.empty_like(xi), _b(xi), _b(xi), xi.new_empty(w_all.numel())


# ---------------------------------------------------------------- autograd formulas
def _wilson_setup(ctx, inputs, output):
    x, beta = inputs
    ctx.save_for_backward(x)
    ctx.beta = beta


def _wilson_backward(ctx, gS, gQ, gplaq):
    # S = -beta sum cos P; plaq = -S / (beta L^2); Q is piecewise constant (qed_helpers.py:108-116)
    (x,) = ctx.saved_tensors
    L = x.shape[-1]
    g = torch.zeros(x.shape[0], dtype=x.dtype, device=x.device)
    if gS is not None:
        g = g + gS
    if gplaq is not None:
        g = g - gplaq / (ctx.beta * L * L)
    return torch.ops.fthmc_hip.wilson_force(x, ctx.beta) * g.view(-1, 1, 1, 1), None


torch.library.register_autograd('fthmc_hip::wilson_action_charge', _wilson_backward, setup_context=_wilson_setup)


def _layer_setup(ctx, inputs, output):
    x, w, mu, off, n_mix, act, hidden, kernel_size = inputs
    ctx.save_for_backward(x, w)
    ctx.args = (mu, off, n_mix, act, hidden, kernel_size)


def _layer_backward(ctx, gy, glogJ):
    x, w = ctx.saved_tensors
    B = x.shape[0]
    gy = torch.zeros_like(x) if gy is None else gy.contiguous()
    glogJ = torch.zeros(B, dtype=x.dtype, device=x.device) if glogJ is None else glogJ.contiguous()
    if ctx.needs_input_grad[1]:
        gx, gw = torch.ops.fthmc_hip.flow_layer_bwd(x, gy, glogJ, w, *ctx.args)
        return gx, gw.view_as(w), None, None, None, None, None, None
    return torch.ops.fthmc_hip.flow_layer_bwd_x(x, gy, glogJ, w, *ctx.args), None, None, None, None, None, None, None


torch.library.register_autograd('fthmc_hip::flow_layer_fwd', _layer_backward, setup_context=_layer_setup)

if BACKEND == 'compiled':                        # the module's names are the dispatcher's operators themselves
    for _n in ('wilson_action_charge', 'wilson_force', 'hmc_trajectory', 'flow_layer_fwd', 'flow_layer_bwd_x', 'flow_layer_bwd_w',
               'flow_layer_bwd', 'flow_layer_rev', 'ft_action_force', 'fthmc_trajectory', 'train_grad'):
        globals()[_n] = getattr(torch.ops.fthmc_hip, _n)

__all__ = ['wilson_action_charge', 'wilson_force', 'hmc_trajectory', 'flow_layer_fwd', 'flow_layer_bwd_x',
           'flow_layer_bwd_w', 'flow_layer_bwd', 'flow_layer_rev', 'ft_action_force', 'fthmc_trajectory', 'train_grad']
