"""Gauge-equivariant coupling layers with the API of fthmc/utils/layers.py, running on the
HIP kernels (fthmc_amd/csrc/flow*.hip) through the C ABI.

The conv nets stay `nn.Conv2d` parameter containers so that `state_dict` keeps the reference
layout `"{i}.plaq_coupling.net.{0,2,4}.{weight,bias}"` and `get_nets` /
`make_net_from_layers` (transfer to a larger lattice) keep working; the numbers are computed
by the fused layer kernels, forward and backward (autograd.Function below).
"""
from __future__ import annotations

from math import pi as PI
from typing import List, Sequence

import numpy as np
import torch
from torch import nn

from .. import ops
from ..config import DTYPE, device

TWO_PI = 2 * PI
TOL = 1e-6            # reference inverse tolerance (layers.py:32); the HIP inverse defaults tighter


def torch_mod(x: torch.Tensor):
    """layers.py:41-43: remainder(x + pi, 2 pi) - pi."""
    return ops.wrap(x)


def torch_wrap(x: torch.Tensor):
    """layers.py:46-47 (kept literally: torch_mod(x + pi) - pi; unused by the reference)."""
    return ops.wrap(x + PI) - PI


def grab(var: torch.Tensor):
    return var.detach().cpu().numpy()


def get_nets(layers: nn.ModuleList) -> List[nn.Module]:
    """layers.py:54-55."""
    return [layer.plaq_coupling.net for layer in layers]


ACTIVATION_FNS = {'relu': nn.ReLU, 'silu': nn.SiLU, 'swish': nn.SiLU, 'leaky_relu': nn.LeakyReLU}


def _act_name(actfn) -> str:
    key = str(actfn).lower() if actfn is not None else 'silu'
    return key if key in ACTIVATION_FNS else 'silu'       # reference falls back to SiLU (layers.py:124-135)


def make_conv_net(*, hidden_sizes: Sequence[int], kernel_size: int, in_channels: int, out_channels: int,
                  use_final_tanh: bool = False, activation_fn: str = None):
    """layers.py:138-167: Conv2d(k, circular padding) + activation, no final activation -- or a final tanh
    (`use_final_tanh`, layers.py:163-164; the reference always passes False, :419).  A net with the tanh runs on the plain
    kernels (csrc/flow_generic.hip), like any net shape other than the default."""
    assert kernel_size % 2 == 1, 'kernel size must be odd'
    act = _act_name(activation_fn)
    sizes = [in_channels] + list(hidden_sizes) + [out_channels]
    net = []
    for i in range(len(sizes) - 1):
        net.append(nn.Conv2d(sizes[i], sizes[i + 1], kernel_size, stride=1, padding=kernel_size // 2,
                             padding_mode='circular', dtype=DTYPE, device=device()))
        if i != len(sizes) - 2:
            net.append(ACTIVATION_FNS[act]())
    if use_final_tanh:
        net.append(nn.Tanh())
    seq = nn.Sequential(*net)
    seq.activation_fn = act
    seq.final_tanh = bool(use_final_tanh)
    return seq


def set_weights(m):
    """layers.py:170-174 (a no-op when called on a ModuleList, as train.get_model does: SURVEY Q6)."""
    if hasattr(m, 'weight') and m.weight is not None:
        nn.init.normal_(m.weight, mean=1, std=2)
    if hasattr(m, 'bias') and m.bias is not None:
        m.bias.data.fill_(-1)


def gauge_transform(links, alpha):
    """layers.py:177-180."""
    for mu in range(len(links.shape[2:])):
        links[:, mu] = alpha + links[:, mu] - torch.roll(alpha, -1, mu + 1)
    return links


def random_gauge_transform(x):
    """layers.py:183-185."""
    return gauge_transform(x, TWO_PI * torch.rand((x.shape[0],) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device))


# ---------------------------------------------------------------- stripe masks (host logic)
# Index form, as the kernels state them (ft_stripe, csrc/common.h): line k of the axis across the stripes (columns for mu = 0, rows for
# mu = 1) belongs to stripe class ((k - off) mod n) mod 4 -- 0 active, 1 and 2 frozen (for the plaquette masks with off + 1),
# 3 passive.
def _stripe_class(n: int, off: int) -> np.ndarray:
    return ((np.arange(n) - off) % n) % 4


def _stripes(shape, mu: int, off: int, classes) -> np.ndarray:
    """[L0, L1] uint8: 1 on the lines across axis 1 - mu whose stripe class is in `classes`"""
    assert len(shape) == 2 and mu in (0, 1)
    across = 1 - mu                                   # mu = 0: the class is a function of the column index, mu = 1: of the row
    line = np.isin(_stripe_class(shape[across], off), classes).astype(np.uint8)
    return np.ascontiguousarray(np.broadcast_to(line[None, :] if across == 1 else line[:, None], shape))


def make_2d_link_active_stripes(shape, mu, off):
    """layers.py:213-237: the links of direction `mu` on every 4th line across the other axis, first one `off`: float32 [2, L, L]."""
    assert len(shape) == 3 and shape[0] == 2, 'need a (2, L, L) shape'
    assert mu in (0, 1)
    mask = np.zeros(shape, dtype=np.float32)
    mask[mu] = _stripes(shape[1:], mu, off, (0,))
    return mask


def make_single_stripes(shape, mu, off):
    """layers.py:240-259: every 4th line, first one `off`."""
    return _stripes(shape, mu, off, (0,))


def make_double_stripes(shape, mu, off):
    """layers.py:261-284: two adjacent lines of every four, first pair at `off`."""
    return _stripes(shape, mu, off, (0, 1))


def make_plaq_masks(mask_shape, mask_mu, mask_off):
    """layers.py:287-292.  The kernels derive these from (mu, off) on the fly; the arrays are
    kept for inspection and for the host-side tests."""
    mask = {'frozen': make_double_stripes(mask_shape, mask_mu, mask_off + 1),
            'active': make_single_stripes(mask_shape, mask_mu, mask_off)}
    mask['passive'] = 1 - mask['frozen'] - mask['active']
    return mask


# ---------------------------------------------------------------- autograd bridge
def net_weights(net: nn.Module) -> List[torch.Tensor]:
    return [p for m in net if isinstance(m, nn.Conv2d) for p in (m.weight, m.bias)]


class _CouplingFn(torch.autograd.Function):
    """y, logJ = layer(x): forward = fthmc_flow_layer_fwd, backward = fthmc_flow_layer_bwd."""

    @staticmethod
    def forward(ctx, x, mu, off, act, *params):
        act, final_tanh = act if isinstance(act, tuple) else (act, False)
        w = ops.pack_weights([params], device=x.device, final_tanh=final_tanh)
        need_gw = any(p.requires_grad for p in params)
        stash = None
        if x.requires_grad or need_gw:
            # keep the layer's activations for the backward (fthmc_flow_layer_fwd_stash): nothing is run twice
            y, logJ, stash = ops.flow_layer_fwd_stash(x, w, mu, off, act)
        else:
            y, logJ = ops.flow_layer_fwd(x, w, mu, off, act)
        ctx.save_for_backward(x, w, *([stash] if stash is not None else []))
        ctx.meta = (mu, off, act, [p.shape for p in params], need_gw, ops.arch_of(w))
        return y, logJ

    @staticmethod
    def backward(ctx, gy, glogJ):
        x, w, *rest = ctx.saved_tensors
        mu, off, act, shapes, need_gw, arch = ctx.meta
        if arch != ops.DEFAULT_ARCH:
            w._fthmc_arch = arch                                  # saved_tensors hands back a plain tensor
        gy = torch.zeros_like(x) if gy is None else gy.contiguous()
        glogJ = torch.zeros(x.shape[0], dtype=x.dtype, device=x.device) if glogJ is None else glogJ.contiguous()
        if rest:
            gx, gw = ops.flow_layer_bwd_stash(rest[0], x.shape, w, gy, glogJ, mu, off, act, need_gw=need_gw)
        else:                                                     # no stash on this kernel variant: forward again inside
            gx, gw = ops.flow_layer_bwd(x, w, gy, glogJ, mu, off, act, need_gw=need_gw)
        gparams = list(ops.unpack_weight_grads(gw, 1, arch=arch)[0]) if need_gw else [None] * len(shapes)
        return (gx, None, None, None, *gparams)


class _PlaqCouplingFn(torch.autograd.Function):
    """fP, logJ = NCPPlaqCouplingLayer.forward(P) on a plaquette field: forward = fthmc_plaq_coupling_fwd, backward =
    fthmc_plaq_coupling_bwd (the link-level backward kernels with the upstream gradient dressed as a link gradient)."""

    @staticmethod
    def forward(ctx, P, mu, off, act, *params):
        act, final_tanh = act if isinstance(act, tuple) else (act, False)
        w = ops.pack_weights([params], device=P.device, final_tanh=final_tanh)
        fP, logJ = ops.plaq_coupling_fwd(P, w, mu, off, act)
        ctx.save_for_backward(P, w)
        ctx.meta = (mu, off, act, any(p.requires_grad for p in params), ops.arch_of(w), len(params))
        return fP, logJ

    @staticmethod
    def backward(ctx, gfP, glogJ):
        P, w = ctx.saved_tensors
        mu, off, act, need_gw, arch, npar = ctx.meta
        gfP = torch.zeros_like(P) if gfP is None else gfP.contiguous()
        glogJ = torch.zeros(P.shape[0], dtype=P.dtype, device=P.device) if glogJ is None else glogJ.contiguous()
        gP, gw = ops.plaq_coupling_bwd(P, w, gfP, glogJ, mu, off, act, need_gw=need_gw, arch=arch)
        gparams = list(ops.unpack_weight_grads(gw, 1, arch=arch)[0]) if need_gw else [None] * npar
        return (gP, None, None, None, *gparams)


class NCPPlaqCouplingLayer(nn.Module):
    """Holder of the s/t conv net and stripe geometry (layers.py:324-396).  The plaquette-level
    map itself is fused into the link-level kernels used by GaugeEquivCouplingLayer."""

    def __init__(self, net: nn.Module, *, mask_shape, mask_mu: int, mask_off: int, inv_prec: float = TOL,
                 inv_max_iter: int = 1000):
        super().__init__()
        assert len(mask_shape) == 2, 'NCPPlaqCouplingLayer is implemented only in 2D'
        self.mask = make_plaq_masks(mask_shape, mask_mu, mask_off)
        self.mask_mu, self.mask_off = mask_mu, mask_off % 4
        self.net = net
        self.inv_prec, self.inv_max_iter = inv_prec, inv_max_iter

    @property
    def activation_fn(self):
        return getattr(self.net, 'activation_fn', 'silu')

    @property
    def final_tanh(self):
        return bool(getattr(self.net, 'final_tanh', False))

    def _w(self, dev):
        return ops.pack_weights([net_weights(self.net)], device=dev, final_tanh=self.final_tanh)

    def forward(self, x):
        """layers.py:348-371 on a plaquette field [B, L, L] -> (fx, logJ[B]); differentiable wrt the field and the conv
        weights (round 4: fthmc_plaq_coupling_bwd), like the reference's module under autograd."""
        assert len(x.shape) == 3, f'field should be (batch_size, *lattice_shape); got {tuple(x.shape)}'
        params = net_weights(self.net)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
            return _PlaqCouplingFn.apply(x, self.mask_mu, self.mask_off, (self.activation_fn, self.final_tanh), *params)
        return ops.plaq_coupling_fwd(x, self._w(x.device), self.mask_mu, self.mask_off, self.activation_fn)

    def reverse(self, fx, tol: float = 1e-12):
        """layers.py:373-396 -> (x, logJ[B]); per-site Newton to `tol` instead of the global bisection to
        `inv_prec` = 1e-6."""
        return ops.plaq_coupling_rev(fx.detach(), self._w(fx.device), self.mask_mu, self.mask_off,
                                     self.activation_fn, tol=tol)


class GaugeEquivCouplingLayer(nn.Module):
    """layers.py:188-210.  forward / reverse return (field, logJ[B])."""

    def __init__(self, *, lattice_shape, mask_mu, mask_off, plaq_coupling):
        super().__init__()
        self.active_mask = make_2d_link_active_stripes((len(lattice_shape),) + tuple(lattice_shape), mask_mu, mask_off)
        self.plaq_coupling = plaq_coupling
        self.mask_mu, self.mask_off = mask_mu, mask_off % 4

    @property
    def activation_fn(self):
        return getattr(self.plaq_coupling.net, 'activation_fn', 'silu')

    def forward(self, x):
        params = net_weights(self.plaq_coupling.net)
        return _CouplingFn.apply(x, self.mask_mu, self.mask_off, (self.activation_fn, self.plaq_coupling.final_tanh), *params)

    def reverse(self, fx, tol: float = 1e-12):
        w = self.plaq_coupling._w(fx.device)
        return ops.flow_layer_rev(fx.detach(), w, self.mask_mu, self.mask_off, self.activation_fn, tol=tol)


def _check_arch(hidden_sizes, kernel_size, n_mix):
    """Any net shape the reference's make_conv_net accepts runs on the HIP path: the default (hidden_sizes=[8, 8],
    kernel_size=3, n_mixture_comps=2) on the tuned kernels, anything else on the plain kernels of csrc/flow_generic.hip,
    within their limits."""
    hs = list(hidden_sizes)
    if len(hs) > 8 or any(int(h) < 1 or int(h) > 256 for h in hs) or kernel_size % 2 != 1 or not 1 <= kernel_size <= 15 \
            or not 1 <= n_mix <= 64:
        raise NotImplementedError(
            f'net shape beyond the limits of the HIP kernels (at most 8 hidden layers of 1..256 channels, odd kernel_size '
            f'<= 15, 1..64 mixture components); got {hs}, {kernel_size}, {n_mix}')


def make_u1_equiv_layers(*, n_layers, n_mixture_comps, lattice_shape, hidden_sizes, kernel_size,
                         activation_fn: str = None):
    """layers.py:399-429: layer i uses mu = i % 2, off = (i // 2) % 4."""
    _check_arch(hidden_sizes, kernel_size, n_mixture_comps)
    layers = []
    for i in range(n_layers):
        mu, off = i % 2, (i // 2) % 4
        net = make_conv_net(in_channels=2, out_channels=n_mixture_comps + 1, hidden_sizes=hidden_sizes,
                            kernel_size=kernel_size, use_final_tanh=False, activation_fn=activation_fn)
        plaq_coupling = NCPPlaqCouplingLayer(net, mask_shape=lattice_shape, mask_mu=mu, mask_off=off)
        layers.append(GaugeEquivCouplingLayer(lattice_shape=lattice_shape, mask_mu=mu, mask_off=off,
                                              plaq_coupling=plaq_coupling))
    return nn.ModuleList(layers)


def make_net_from_layers(*, lattice_shape: tuple, nets: List[nn.Module]):
    """layers.py:93-114: reuse trained (translation-equivariant) nets on another lattice size."""
    layers = []
    for i, net in enumerate(nets):
        mu, off = i % 2, (i // 2) % 4
        plaq_coupling = NCPPlaqCouplingLayer(net, mask_shape=lattice_shape, mask_mu=mu, mask_off=off)
        layers.append(GaugeEquivCouplingLayer(lattice_shape=lattice_shape, mask_mu=mu, mask_off=off,
                                              plaq_coupling=plaq_coupling))
    return nn.ModuleList(layers)


def _flow_tanh(flow: nn.ModuleList) -> bool:
    t = {bool(getattr(layer.plaq_coupling.net, 'final_tanh', False)) for layer in flow}
    if len(t) > 1:
        raise ValueError('layers of one flow must agree on use_final_tanh')
    return t.pop() if t else False


def _flow_rows(flow: nn.ModuleList):
    rows = []
    for i, layer in enumerate(flow):
        if (layer.mask_mu, layer.mask_off) != (i % 2, (i // 2) % 4):
            raise ValueError(f'layer {i}: masks (mu={layer.mask_mu}, off={layer.mask_off}) are not the reference '
                             f'schedule mu = i % 2, off = (i // 2) % 4')
        rows.append(net_weights(layer.plaq_coupling.net))
    return rows


# flat parameter buffers by the address of their storage: two ModuleLists built over the same conv nets
# (transfer_to_new_lattice reuses them) find the same buffer
_FLATS: 'weakref.WeakValueDictionary' = None


def _flat_of(rows):
    """The flat buffer the conv parameters are views of (flatten_flow), if they all still are; else None."""
    global _FLATS
    if _FLATS is None or not rows:
        return None
    p0 = rows[0][0]
    flat = _FLATS.get((p0.device, p0.untyped_storage().data_ptr()))
    if flat is None:
        return None
    base, o = flat.data_ptr(), 0
    for row in rows:
        for p in row:
            if p.data_ptr() != base + 8 * o or p.dtype != torch.float64 or not p.is_contiguous():
                return None
            o += p.numel()
    return flat if o == flat.numel() else None


def weights_generation(flat: torch.Tensor) -> int:
    """How often the flat weight buffer was written behind PyTorch's back (kernels that take raw pointers -- FlatAdam's step,
    its graph replays -- bump no tensor version): part of the key under which FieldTransformation keeps anything derived from
    the weights (the carried S_eff, the packed-weights record)."""
    return getattr(flat, '_fthmc_gen', 0)


def bump_weights_generation(flat: torch.Tensor) -> None:
    flat._fthmc_gen = getattr(flat, '_fthmc_gen', 0) + 1


def flatten_flow(flow: nn.ModuleList) -> torch.Tensor:
    """Re-home every conv parameter of the flow as a VIEW of ONE flat fp64 buffer in the order of the C ABI
    (per layer w0 b0 w1 b1 ...; include/fthmc_hip.h) and return that buffer.  Afterwards `flow_weights(flow)` is the buffer
    itself -- no packing, no copy, per call -- `load_state_dict`, optimizers and `state_dict` keys work as before (they act
    on the parameters in place), and a weight gradient written into `flow_grad_buffer(flow)` IS every parameter's `.grad`.
    Idempotent; `.to()` / `.cuda()` re-create the parameters, after which the next call flattens again."""
    import weakref
    global _FLATS
    rows = _flow_rows(flow)
    flat = _flat_of(rows)
    if flat is not None:
        return flat
    flat = ops.pack_weights(rows, device=rows[0][0].device if rows else device(), final_tanh=_flow_tanh(flow))
    o = 0
    with torch.no_grad():
        for row in rows:
            for p in row:
                n = p.numel()
                p.data = flat[o:o + n].view(p.shape)
                o += n
    if _FLATS is None:
        _FLATS = weakref.WeakValueDictionary()
    _FLATS[(flat.device, flat.untyped_storage().data_ptr())] = flat
    flow._fthmc_flat = flat                     # keeps the buffer alive as long as the flow
    return flat


def flow_grad_buffer(flow: nn.ModuleList) -> torch.Tensor:
    """Flat gradient buffer shaped like flatten_flow(flow); `attach_grads` makes every conv parameter's .grad a view of it.
    It is the head of a slightly longer allocation (`flow_grad_ext`): the training step's scalar sums ride behind the
    gradients in the same all-reduce."""
    flat = flatten_flow(flow)
    g = getattr(flow, '_fthmc_gflat', None)
    if g is None or g.numel() != flat.numel() or g.device != flat.device:
        ext = torch.zeros(flat.numel() + GRAD_EXT, dtype=flat.dtype, device=flat.device)
        g = ext[:flat.numel()]
        flow._fthmc_gext = ext
        flow._fthmc_gflat = g
        o, views = 0, []
        for row in _flow_rows(flow):
            for p in row:
                views.append(g[o:o + p.numel()].view(p.shape)); o += p.numel()
        flow._fthmc_gviews = views
    return g


GRAD_EXT = 8      # doubles behind the gradients in flow_grad_ext (slot 0: sum over this rank's chains of logq - logp)


def flow_grad_ext(flow: nn.ModuleList) -> torch.Tensor:
    """[n_params + GRAD_EXT]: the flat gradient buffer and, behind it, the scalars of a training step that are summed over the
    ranks with it (C2: ONE SUM all-reduce per step)."""
    flow_grad_buffer(flow)
    return flow._fthmc_gext


def attach_grads(flow: nn.ModuleList):
    """p.grad = its view of flow_grad_buffer(flow), for every conv parameter (optimizer.zero_grad() drops them)."""
    flow_grad_buffer(flow)
    k = 0
    for row in _flow_rows(flow):
        for p in row:
            if p.grad is not flow._fthmc_gviews[k]:
                p.grad = flow._fthmc_gviews[k]
            k += 1


def flow_weights(flow: nn.ModuleList, dev=None) -> torch.Tensor:
    """All layers' conv parameters as the flat [n_layers * params] buffer of the C ABI (955 per layer for the default net), checking
    that the layers follow the reference mask schedule.  A flow whose parameters live in one flat buffer (flatten_flow)
    hands that buffer out as it is; otherwise the parameters are packed into a new one."""
    rows = _flow_rows(flow)
    flat = _flat_of(rows)
    if flat is not None and (dev is None or torch.device(dev) == flat.device):
        return flat
    if dev is None:
        dev = rows[0][0].device if rows else device()
    return ops.pack_weights(rows, device=dev, final_tanh=_flow_tanh(flow))


def flow_activation(flow: nn.ModuleList) -> str:
    acts = {layer.activation_fn for layer in flow}
    if len(acts) > 1:
        raise ValueError(f'mixed activations {acts} in one flow')
    return acts.pop() if acts else 'silu'
