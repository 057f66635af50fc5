"""Tensor-level operators over the C ABI (one function per entry point of
include/fthmc_hip.h).  Inputs must be fp64 tensors on a HIP device; PyTorch only
provides device memory and the current stream."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ACT_CODES, MODE_LITERAL, MODE_MD, W_PER_LAYER, FthmcError, check

_WS = {}           # (device index, stream) -> workspace tensor (grown on demand)
_WS_CAPTURED = set()   # keys whose CURRENT workspace was handed out during a graph capture
_WS_RETIRED = []   # superseded workspaces a captured hipGraph may still point into, kept alive


def _dev(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f'{name}: expected a torch.Tensor')
    if not t.is_cuda:
        raise FthmcError(f'{name}: tensor lives on {t.device}; fthmc_amd runs on the MI355X only '
                         f'(no CPU fallback) -- move it with .cuda()')
    if t.dtype != torch.float64:
        raise FthmcError(f'{name}: dtype {t.dtype}; the HIP path computes in float64')
    return t.contiguous()


def _field(t, name='x'):
    t = _dev(t, name)
    if t.dim() != 4 or t.shape[1] != 2 or t.shape[2] != t.shape[3]:
        raise FthmcError(f'{name}: expected [B, 2, L, L], got {tuple(t.shape)}')
    if t.shape[2] % 4 != 0:
        raise FthmcError(f'{name}: L={t.shape[2]} must be a multiple of 4 (stripe masks have period 4)')
    return t


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


def ws_bytes(B: int, L: int, n_layers: int, arch=None) -> int:
    return int(_lib.load().fthmc_ws_bytes(_arch(arch), B, L, n_layers))


def _ws(t: torch.Tensor, B: int, L: int, nl: int, train: bool = False, arch=None):
    need = int(_lib.load().fthmc_train_ws_bytes(_arch(arch), B, L, nl)) if train else ws_bytes(B, L, nl, arch)
    # one workspace per (device, stream): chain groups running on concurrent streams must not share scratch
    key = (t.device.index, torch.cuda.current_stream(t.device).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() * 8 < need:
        if torch.cuda.is_current_stream_capturing():
            # an allocation made here would come from the graph's private pool and the pointer would be
            # baked into the graph while later eager calls keep using (or replace) the buffer
            raise FthmcError(f'the workspace of this stream would have to be {"allocated" if buf is None else "grown"} '
                             f'({need} bytes) during graph capture: run the same call once eagerly on this stream '
                             f'first (warm-up), then capture')
        if buf is not None and key in _WS_CAPTURED:
            _WS_RETIRED.append(buf)        # a graph captured on this stream replays into it: keep it alive
        # (a buffer no capture ever saw is simply dropped: the allocator orders its reuse behind this stream's work)
        _WS_CAPTURED.discard(key)
        buf = torch.empty((need + 7) // 8, dtype=torch.float64, device=t.device)
        # the head holds the weight expansions and their stamps (include/fthmc_hip.h, "Weight versions"): fresh memory may
        # carry the stamps of an earlier life of the same address
        buf[:min(buf.numel(), int(_lib.load().fthmc_ws_head_bytes()) // 8)].zero_()
        _WS[key] = buf
    if torch.cuda.is_current_stream_capturing():
        _WS_CAPTURED.add(key)
    return buf.data_ptr(), buf.numel() * 8


def release_workspaces():
    """Drop every cached workspace (current and superseded).  Only when no captured graph that used them will
    be replayed again."""
    _WS.clear()
    _WS_CAPTURED.clear()
    _WS_RETIRED.clear()


def weights_version(wkey) -> int:
    """The caller's statement of the weights' CONTENT version (`wkey`: any hashable -- FieldTransformation: parameter
    versions + the flat buffer's generation, utils/layers.py weights_generation) as the 64-bit number of the C ABI's `_v`
    entry points; None -> 0 = no statement: the call expands its weights.  The library compares it ON THE DEVICE with the
    stamps the last expansion left in the workspace (include/fthmc_hip.h, "Weight versions"): nothing is recorded here."""
    if wkey is None:
        return 0
    if isinstance(wkey, int) and 0 < wkey < (1 << 64):
        return wkey
    return (hash(('fthmc-weights', wkey)) & 0xFFFFFFFFFFFFFFFF) or 1


def _group_edges(groups, B: int):
    if isinstance(groups, (list, tuple)):
        if sum(groups) != B or min(groups) < 1:
            raise FthmcError(f'groups: sizes {tuple(groups)} do not add up to {B} chains')
        return [sum(groups[:k]) for k in range(len(groups) + 1)]
    G = max(1, min(int(groups), B))
    return [k * B // G for k in range(G + 1)]


def trajectory_workspaces(x: torch.Tensor, w, n_layers: int, groups=1, arch=None, side_streams=None, wkey=None):
    """Make sure the workspace of every stream an `ft_trajectory(x, ..., groups=groups)` call from the current stream runs on
    (the current stream and the side streams of the chain groups) exists at its final size, holding the expansion of `w`
    -> a token (the workspaces' addresses): a captured sequence stays valid while the token does."""
    x = _field(x); B, _, L, _ = x.shape
    edges = _group_edges(groups, B)
    G = len(edges) - 1
    main = torch.cuda.current_stream(x.device)
    sides = (list(side_streams)[:G - 1] if side_streams is not None else _side_streams(x.device, G - 1)) if G > 1 else []
    for st in sides:
        st.wait_stream(main)
    token = []
    for gi in list(range(1, G)) + [0]:
        st = main if gi == 0 else sides[gi - 1]
        with torch.cuda.stream(st):
            pack_workspace(x, w, n_layers, edges[gi + 1] - edges[gi], L, wkey=wkey, arch=arch)
            token.append(_WS[(x.device.index, st.cuda_stream)].data_ptr())
    for st in sides:
        main.wait_stream(st)
    return tuple(token)


def pack_workspace(t: torch.Tensor, w, n_layers: int, B: int, L: int, wkey=None, arch=None):
    """Expand `w` into the workspace of the current stream (grown to serve [B, 2, L, L] calls with n_layers layers) under the
    version `wkey` states (C ABI fthmc_pack_weights): later calls on this stream with the same `wkey` find the stamps."""
    w, ap, a = _wall(w, n_layers, arch)
    ws, nb = _ws(t, B, L, n_layers, arch=a)
    check(_lib.load().fthmc_pack_weights(_p(w), ap, n_layers, weights_version(wkey), ws, nb, _stream(t)), 'fthmc_pack_weights')


# ---------------------------------------------------------------- s/t net shape
# The shape of the s/t conv net is an ARGUMENT of every call that runs the net (C ABI: fthmc_arch_t; the library keeps no
# shape between calls).  On this side it is a tuple (hidden_sizes, kernel_size, n_mixture_comps) -- with a fourth entry True for a
# net that ends in a tanh (make_conv_net(use_final_tanh=True)) --; it comes, in this order,
# from the `arch=` argument of an op, from the tag `pack_weights` leaves on a packed weight tensor (`arch_of`), or it is
# the reference default.
DEFAULT_ARCH = ((8, 8), 3, 2)          # hidden_sizes, kernel_size, n_mixture_comps: the reference default (tuned kernels)


def norm_arch(arch) -> tuple:
    if arch is None:
        return DEFAULT_ARCH
    a = (tuple(int(h) for h in arch[0]), int(arch[1]), int(arch[2]))
    return a + (True,) if len(arch) > 3 and arch[3] else a


def arch_params(arch=DEFAULT_ARCH) -> int:
    """Doubles per layer of the canonical weight layout [w0 b0 w1 b1 ...] for a net 2 -> hidden... -> n_mix + 1."""
    hidden, k, n_mix = norm_arch(arch)[:3]
    chans = [2, *hidden, n_mix + 1]
    return sum(co * ci * k * k + co for ci, co in zip(chans[:-1], chans[1:]))


def _arch(arch):
    """-> the fthmc_arch_t* of a call (None = NULL = the default shape)"""
    arch = norm_arch(arch)
    if arch == DEFAULT_ARCH:
        return None
    hidden, k, n_mix = arch[:3]
    if len(hidden) > 8:
        raise FthmcError(f'net shape {arch}: at most 8 hidden layers')
    a = _lib.ArchT()
    a.n_hidden, a.kernel_size, a.n_mix, a.final_tanh = len(hidden), k, n_mix, int(len(arch) > 3)
    for i, h in enumerate(hidden):
        a.hidden[i] = h
    import ctypes
    return ctypes.pointer(a)


def arch_of(w, arch=None) -> tuple:
    """The net shape of a call: the explicit `arch`, else the one recorded on the packed weight tensor `w` by pack_weights
    (a plain attribute: it does not survive .to() / .clone() / arithmetic -- pass `arch=` then), else the default."""
    if arch is not None:
        return norm_arch(arch)
    return getattr(w, '_fthmc_arch', DEFAULT_ARCH)


def _tag(t, like, arch=None):
    """carry the net shape of `like` (or `arch`) over to a tensor derived from it"""
    a = arch_of(like, arch)
    if a != DEFAULT_ARCH and t is not None:
        t._fthmc_arch = a
    return t


def set_variant(v: int):
    """1: MFMA conv kernels (default), 0: VALU conv kernels."""
    check(_lib.load().fthmc_set_variant(int(v)), 'fthmc_set_variant')


def get_variant() -> int:
    return int(_lib.load().fthmc_get_variant())


def set_small_path(on: bool):
    """Lattices of L = 8, 12, 16: one fused launch per trajectory / force / action (default) or the tiled kernels."""
    check(_lib.load().fthmc_set_small_path(int(bool(on))), 'fthmc_set_small_path')


def get_small_path() -> bool:
    return bool(_lib.load().fthmc_get_small_path())


def act_code(act) -> int:
    key = act.lower() if isinstance(act, str) else act
    if key not in ACT_CODES:
        key = 'silu'                  # reference falls back to SiLU (layers.py:127-133)
    return ACT_CODES[key]


def pack_weights(nets: Sequence[Sequence[torch.Tensor]], device=None, final_tanh: bool = False) -> torch.Tensor:
    """[(w0, b0, w1, b1, ...), ...] -> flat [n_layers * params] fp64 (PyTorch order).  The conv nets may have any
    hidden sizes / odd kernel size / number of mixture components (all layers alike); the shape is recorded on the
    result (`arch_of`) and selects the kernels: the tuned ones for the reference default 2 -> 8 -> 8 -> 3, k = 3."""
    rows, arch = [], None
    for w in nets:
        w = list(w)
        shapes = [tuple(t.shape) for t in w]
        ok = len(w) >= 2 and len(w) % 2 == 0
        ci, k = 2, (shapes[0][-1] if ok and len(shapes[0]) == 4 else 0)
        for wi, bi in zip(shapes[0::2], shapes[1::2]):
            ok = ok and len(wi) == 4 and wi[1] == ci and wi[2] == wi[3] == k and bi == (wi[0],)
            ci = wi[0] if len(wi) == 4 else 0
        if not ok or k % 2 == 0 or ci < 2:
            raise FthmcError(f'unsupported s/t net {shapes}: expected Conv2d weights (c1, 2, k, k), (c1,), (c2, c1, k, k), ... '
                             f'with one odd kernel size and n_mix + 1 >= 2 output channels')
        a = (tuple(s_[0] for s_ in shapes[0:-2:2]), k, ci - 1)
        if arch is not None and a != arch:
            raise FthmcError(f'layers of one flow must share the net shape: {arch} vs {a}')
        arch = a
        rows.append(torch.cat([t.detach().reshape(-1).to(torch.float64) for t in w]))
    if arch is not None and (len(arch[0]) > 8 or max(arch[0] + (1,)) > 256 or arch[1] > 15 or arch[2] > 64):
        raise FthmcError(f'net shape {arch} beyond the limits of the HIP kernels (8 hidden layers of <= 256 channels, '
                         f'kernel_size <= 15, 64 mixture components)')
    out = torch.stack(rows).contiguous() if rows else torch.zeros(0, W_PER_LAYER, dtype=torch.float64)
    if device is not None:
        out = out.to(device)
    out = out.reshape(-1)
    if arch is not None and final_tanh:
        arch = arch + (True,)                  # a tanh behind the last conv: not visible in the weights, the caller says so
    if arch is not None and arch != DEFAULT_ARCH:
        out._fthmc_arch = arch
    return out


def unpack_weight_grads(gw: torch.Tensor, n_layers: int, arch=None):
    """flat [n_layers * params] -> list of tuples shaped like the conv parameters (w0, b0, w1, b1, ...)."""
    hidden, k, n_mix = (arch if arch is not None else arch_of(gw))[:3]
    chans = [2, *hidden, n_mix + 1]
    sizes = []
    for ci, co in zip(chans[:-1], chans[1:]):
        sizes += [(co, ci, k, k), (co,)]
    out = []
    g = gw.reshape(n_layers, -1)
    for l in range(n_layers):
        o, row = 0, []
        for s in sizes:
            n = 1
            for d in s:
                n *= d
            row.append(g[l, o:o + n].reshape(s))
            o += n
        out.append(tuple(row))
    return out


# ---------------------------------------------------------------- angle maps
def wrap(x):
    x = _dev(x, 'x'); out = torch.empty_like(x)
    check(_lib.load().fthmc_wrap(_p(x), _p(out), x.numel(), _stream(x)), 'fthmc_wrap')
    return out


def regularize(x):
    x = _dev(x, 'x'); out = torch.empty_like(x)
    check(_lib.load().fthmc_regularize(_p(x), _p(out), x.numel(), _stream(x)), 'fthmc_regularize')
    return out


# ---------------------------------------------------------------- Wilson
def plaquettes(x):
    x = _field(x); B, _, L, _ = x.shape
    P = torch.empty(B, L, L, dtype=x.dtype, device=x.device)
    check(_lib.load().fthmc_plaquettes(_p(x), _p(P), B, L, _stream(x)), 'fthmc_plaquettes')
    return P


def _out_vec(out, key, B, like):
    """out[key] when the caller supplies it (a contiguous float64 device vector of B entries, written in place: no copy
    launches behind the kernel in a captured loop), else a fresh one"""
    t = None if out is None else out.get(key)
    if t is None:
        return torch.empty(B, dtype=like.dtype, device=like.device)
    if not t.is_cuda or t.dtype != torch.float64 or not t.is_contiguous() or t.numel() != B:
        raise FthmcError(f'out[{key!r}]: expected a contiguous float64 device vector of {B} entries')
    return t


def wilson_action_charge(x, beta: float, out=None):
    """-> (S, Q, plaq) per chain; out: optional dict with any of 'S', 'Q', 'plaq' to write into"""
    x = _field(x); B, _, L, _ = x.shape
    S, Q, plaq = (_out_vec(out, k, B, x) for k in ('S', 'Q', 'plaq'))
    check(_lib.load().fthmc_wilson_action_charge(_p(x), B, L, float(beta), _p(S), _p(Q), _p(plaq), _stream(x)),
          'fthmc_wilson_action_charge')
    return S, Q, plaq


def wilson_force(x, beta: float):
    x = _field(x); B, _, L, _ = x.shape
    F = torch.empty_like(x)
    check(_lib.load().fthmc_wilson_force(_p(x), B, L, float(beta), _p(F), _stream(x)), 'fthmc_wilson_force')
    return F


def kinetic(v):
    v = _field(v, 'v'); B, _, L, _ = v.shape
    K = torch.empty(B, dtype=v.dtype, device=v.device)
    check(_lib.load().fthmc_kinetic(_p(v), B, L, _p(K), _stream(v)), 'fthmc_kinetic')
    return K


def stats_accumulate(acc, plaq, Q, qold, dH, vec):
    """vec[8] += sums over the chains of (1, acc, plaq, Q, Q^2, |Q - qold|, dH, exp(-dH)); qold <- Q (in place)."""
    B = acc.numel()
    for t, n in ((acc, B), (plaq, B), (Q, B), (qold, B), (dH, B), (vec, 8)):
        if not t.is_cuda or t.dtype != torch.float64 or not t.is_contiguous() or t.numel() != n:
            raise FthmcError('stats_accumulate: contiguous float64 device tensors of B (vec: 8) elements expected')
    check(_lib.load().fthmc_stats_accumulate(_p(acc), _p(plaq), _p(Q), _p(qold), _p(dH), B, _p(vec), _stream(acc)),
          'fthmc_stats_accumulate')


def random_momenta(seeds, shape, need_u=True, out_v=None, out_u=None):
    """v ~ N(0,1) of `shape` = (B, 2, L, L) and u ~ U[0,1) [B] from per-chain int64 seeds
    (written into out_v / out_u when given: contiguous float64 device tensors of those shapes)."""
    if not seeds.is_cuda or seeds.dtype != torch.int64:
        raise FthmcError('seeds: expected an int64 tensor on the HIP device')
    seeds = seeds.contiguous()
    B = int(shape[0]); n = 1
    for d in shape[1:]:
        n *= int(d)
    if seeds.numel() != B:
        raise FthmcError(f'seeds: expected {B}, got {seeds.numel()}')
    for t, want in ((out_v, B * n), (out_u, B)):
        if t is not None and (not t.is_cuda or t.dtype != torch.float64 or not t.is_contiguous() or t.numel() != want):
            raise FthmcError('random_momenta: out tensors must be contiguous float64 device tensors of the result shapes')
    v = out_v if out_v is not None else torch.empty(tuple(shape), dtype=torch.float64, device=seeds.device)
    u = out_u if out_u is not None else (torch.empty(B, dtype=torch.float64, device=seeds.device) if need_u else None)
    check(_lib.load().fthmc_random_momenta(_p(seeds), B, n, _p(v), _p(u), _stream(v)), 'fthmc_random_momenta')
    return v, u


def chain_seeds(seed: int, lo: int, B: int, traj: int = 0, counter: Optional[torch.Tensor] = None, advance: bool = False,
                out: Optional[torch.Tensor] = None, device=None):
    """Per-chain int64 seeds of trajectory / step `traj (+ counter[0])` for the global chain ids lo .. lo + B - 1, formed on
    the device (C ABI fthmc_chain_seeds): the same numbers as parallel.chain_seeds.  `counter`: a device int64 scalar added
    to `traj`; advance=True adds 1 to it behind the read, so that a captured launch draws fresh seeds at every replay."""
    if out is None:
        out = torch.empty(B, dtype=torch.int64, device=device if device is not None else counter.device)
    if not out.is_cuda or out.dtype != torch.int64 or not out.is_contiguous() or out.numel() != B:
        raise FthmcError(f'chain_seeds: out must be a contiguous int64 device tensor of {B} entries')
    if counter is not None and (not counter.is_cuda or counter.dtype != torch.int64 or counter.numel() != 1):
        raise FthmcError('chain_seeds: counter must be one int64 on the HIP device')
    check(_lib.load().fthmc_chain_seeds(int(seed), int(lo), int(B), int(traj), _p(counter), int(bool(advance)), _p(out),
                                        _stream(out)), 'fthmc_chain_seeds')
    return out


def random_uniform(seeds, shape, lo: float, hi: float, out=None):
    """U[lo, hi) of `shape` = (B, ...) from per-chain int64 seeds: the prior draw of a training step on the device."""
    if not seeds.is_cuda or seeds.dtype != torch.int64:
        raise FthmcError('seeds: expected an int64 tensor on the HIP device')
    seeds = seeds.contiguous()
    B = int(shape[0]); n = 1
    for d in shape[1:]:
        n *= int(d)
    if seeds.numel() != B:
        raise FthmcError(f'seeds: expected {B}, got {seeds.numel()}')
    if out is None:
        out = torch.empty(tuple(shape), dtype=torch.float64, device=seeds.device)
    elif not out.is_cuda or out.dtype != torch.float64 or not out.is_contiguous() or out.numel() != B * n:
        raise FthmcError('random_uniform: out must be a contiguous float64 device tensor of the result shape')
    check(_lib.load().fthmc_random_uniform(_p(seeds), B, n, float(lo), float(hi), _p(out), _stream(out)), 'fthmc_random_uniform')
    return out


def train_metrics(xi, x, logq, logp, beta: float, dkl_factor: float = 1.0, out=None):
    """-> row [2 + 5 B] = (loss_dkl, ess, logp[B], logq[B], q[B], dq[B], plaq[B]) of one rank's training batch, on the
    device (C ABI fthmc_train_metrics): everything train_step reports, in one buffer."""
    xi = _field(xi, 'xi'); x = _field(x); B, _, L, _ = x.shape
    logq = _dev(logq, 'logq').reshape(-1); logp = _dev(logp, 'logp').reshape(-1)
    if logq.numel() != B or logp.numel() != B or xi.shape != x.shape:
        raise FthmcError(f'train_metrics: expected logq, logp of {B} entries and xi shaped like x')
    if out is None:
        out = torch.empty(2 + 5 * B, dtype=torch.float64, device=x.device)
    elif not out.is_cuda or out.dtype != torch.float64 or not out.is_contiguous() or out.numel() != 2 + 5 * B:
        raise FthmcError(f'train_metrics: out must be a contiguous float64 device tensor of {2 + 5 * B} entries')
    ws, nb = _ws(x, B, L, 0)
    check(_lib.load().fthmc_train_metrics(_p(xi), _p(x), _p(logq), _p(logp), B, L, float(beta), float(dkl_factor),
                                          _p(out), ws, nb, _stream(x)), 'fthmc_train_metrics')
    return out


def adam_step(w, gw, exp_avg, exp_avg_sq, hyper, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
              decoupled: bool = False):
    """One Adam (AdamW: decoupled=True) step on flat fp64 device buffers, in place, ONE launch (C ABI fthmc_adam_step).
    `hyper` = [steps taken, learning rate, 0] on the device: the launch reads both from there and moves the count on."""
    n = w.numel()
    for t, want in ((w, n), (gw, n), (exp_avg, n), (exp_avg_sq, n), (hyper, 3)):
        if not t.is_cuda or t.dtype != torch.float64 or not t.is_contiguous() or t.numel() != want:
            raise FthmcError('adam_step: contiguous float64 device tensors expected (w, gw, exp_avg, exp_avg_sq alike; hyper: 3)')
    check(_lib.load().fthmc_adam_step(_p(w), _p(gw), _p(exp_avg), _p(exp_avg_sq), _p(hyper), n, float(betas[0]), float(betas[1]),
                                      float(eps), float(weight_decay), int(bool(decoupled)), _stream(w)), 'fthmc_adam_step')


def split_metrics(row, B: int) -> dict:
    """views of a train_metrics row (device or host tensor / array) under train_step's keys"""
    r = row[2:].reshape(5, B)
    return {'loss_dkl': row[0], 'ess': row[1], 'logp': r[0], 'logq': r[1], 'q': r[2], 'dq': r[3], 'plaq': r[4]}


def leapfrog(x, p, beta: float, dt: float, nstep: int):
    x = _field(x); p = _field(p, 'p'); B, _, L, _ = x.shape
    xo, po = torch.empty_like(x), torch.empty_like(p)
    ws, nb = _ws(x, B, L, 0)
    check(_lib.load().fthmc_leapfrog(_p(x), _p(p), B, L, float(beta), float(dt), int(nstep), _p(xo), _p(po),
                                     ws, nb, _stream(x)), 'fthmc_leapfrog')
    return xo, po


def hmc_trajectory(x, v, u, beta: float, dt: float, nstep: int, out=None):
    """-> dict(x_new, dH, acc, H0, H1), per chain; out: optional dict with any of these to write into (x_new must not be x)."""
    x = _field(x); v = _field(v, 'v'); u = _dev(u, 'u').reshape(-1); B, _, L, _ = x.shape
    if u.numel() != B:
        raise FthmcError(f'u: expected {B} uniforms, got {u.numel()}')
    xn = None if out is None else out.get('x_new')
    if xn is None:
        xn = torch.empty_like(x)
    elif xn.shape != x.shape or xn.dtype != x.dtype or not xn.is_cuda or not xn.is_contiguous() or xn.data_ptr() == x.data_ptr():
        raise FthmcError('out[\'x_new\']: expected a contiguous device tensor of the shape of x that is not x')
    dH, acc, H0, H1 = (_out_vec(out, k, B, x) for k in ('dH', 'acc', 'H0', 'H1'))
    ws, nb = _ws(x, B, L, 0)
    check(_lib.load().fthmc_hmc_trajectory(_p(x), _p(v), _p(u), B, L, float(beta), float(dt), int(nstep),
                                           _p(xn), _p(dH), _p(acc), _p(H0), _p(H1), ws, nb, _stream(x)),
          'fthmc_hmc_trajectory')
    return {'x_new': xn, 'dH': dH, 'acc': acc, 'H0': H0, 'H1': H1}


# ---------------------------------------------------------------- coupling layer
def _w1(w, x, arch=None):
    """-> (one layer's weights, flat and tagged; the call's fthmc_arch_t*; the shape)"""
    a = arch_of(w, arch)
    npl = arch_params(a)
    w = _tag(_dev(w, 'w').reshape(-1), w, a)
    if w.numel() != npl:
        raise FthmcError(f'w: expected {npl} doubles for one layer of net shape {a}, got {w.numel()}')
    return w, _arch(a), a


def flow_layer_fwd(x, w, mu: int, off: int, act='silu', arch=None):
    x = _field(x); w, ap, a = _w1(w, x, arch); B, _, L, _ = x.shape
    y = torch.empty_like(x); logJ = torch.empty(B, dtype=x.dtype, device=x.device)
    ws, nb = _ws(x, B, L, 1, arch=a)
    check(_lib.load().fthmc_flow_layer_fwd(_p(x), _p(w), ap, B, L, int(mu), int(off), act_code(act), _p(y), _p(logJ),
                                           ws, nb, _stream(x)), 'fthmc_flow_layer_fwd')
    return y, logJ


def flow_layer_bwd(x, w, gy, glogJ, mu: int, off: int, act='silu', need_gw=False, arch=None):
    x = _field(x); w, ap, a = _w1(w, x, arch); gy = _field(gy, 'gy'); glogJ = _dev(glogJ, 'glogJ').reshape(-1)
    B, _, L, _ = x.shape
    gx = torch.empty_like(x)
    gw = _tag(torch.empty(w.numel(), dtype=x.dtype, device=x.device), w) if need_gw else None
    ws, nb = _ws(x, B, L, 1, train=need_gw, arch=a)
    check(_lib.load().fthmc_flow_layer_bwd(_p(x), _p(w), ap, _p(gy), _p(glogJ), B, L, int(mu), int(off), act_code(act),
                                           _p(gx), _p(gw), ws, nb, _stream(x)), 'fthmc_flow_layer_bwd')
    return gx, gw


def flow_layer_fwd_stash(x, w, mu: int, off: int, act='silu', arch=None):
    """-> (y, logJ, stash): the layer forward that keeps its activations for `flow_layer_bwd_stash` (stash is None where
    the kernels have none -- the VALU variant: use flow_layer_bwd then)."""
    x = _field(x); w, ap, a = _w1(w, x, arch); B, _, L, _ = x.shape
    nbytes = int(_lib.load().fthmc_layer_stash_bytes(ap, B, L))
    if nbytes == 0:
        return (*flow_layer_fwd(x, w, mu, off, act, arch=a), None)
    y = torch.empty_like(x); logJ = torch.empty(B, dtype=x.dtype, device=x.device)
    stash = torch.empty(nbytes // 8, dtype=torch.float64, device=x.device)
    ws, nb = _ws(x, B, L, 1, arch=a)
    check(_lib.load().fthmc_flow_layer_fwd_stash(_p(x), _p(w), ap, B, L, int(mu), int(off), act_code(act), _p(y), _p(logJ),
                                                 _p(stash), ws, nb, _stream(x)), 'fthmc_flow_layer_fwd_stash')
    return y, logJ, stash


def flow_layer_bwd_stash(stash, shape, w, gy, glogJ, mu: int, off: int, act='silu', need_gw=False, arch=None):
    """VJP of the layer from the stash of `flow_layer_fwd_stash` (`shape` = that call's x.shape): nothing recomputed."""
    w_, ap, a = _w1(w, gy, arch); gy = _field(gy, 'gy'); glogJ = _dev(glogJ, 'glogJ').reshape(-1)
    B, _, L, _ = shape
    gx = torch.empty_like(gy)
    gw = _tag(torch.empty(w_.numel(), dtype=gy.dtype, device=gy.device), w_) if need_gw else None
    ws, nb = _ws(gy, B, L, 1, train=need_gw, arch=a)
    check(_lib.load().fthmc_flow_layer_bwd_stash(_p(stash), _p(w_), ap, _p(gy), _p(glogJ), B, L, int(mu), int(off), act_code(act),
                                                 _p(gx), _p(gw), ws, nb, _stream(gy)), 'fthmc_flow_layer_bwd_stash')
    return gx, gw


def flow_layer_rev(y, w, mu: int, off: int, act='silu', tol: float = 1e-12, arch=None):
    y = _field(y, 'y'); w, ap, a = _w1(w, y, arch); B, _, L, _ = y.shape
    x = torch.empty_like(y); logJ = torch.empty(B, dtype=y.dtype, device=y.device)
    ws, nb = _ws(y, B, L, 1, arch=a)
    check(_lib.load().fthmc_flow_layer_rev(_p(y), _p(w), ap, B, L, int(mu), int(off), act_code(act), float(tol),
                                           _p(x), _p(logJ), ws, nb, _stream(y)), 'fthmc_flow_layer_rev')
    return x, logJ


def _plaq_field(P, name='P'):
    P = _dev(P, name)
    if P.dim() != 3 or P.shape[1] != P.shape[2] or P.shape[1] % 4 != 0:
        raise FthmcError(f'{name}: expected a plaquette field [B, L, L] with L % 4 == 0, got {tuple(P.shape)}')
    return P


def plaq_coupling_fwd(P, w, mu: int, off: int, act='silu', arch=None):
    """NCPPlaqCouplingLayer.forward on a plaquette field [B, L, L] -> (fP, logJ[B])."""
    P = _plaq_field(P); B, L, _ = P.shape
    w, ap, a = _w1(w, P, arch)
    fP = torch.empty_like(P); logJ = torch.empty(B, dtype=P.dtype, device=P.device)
    ws, nb = _ws(P, B, L, 1, arch=a)
    check(_lib.load().fthmc_plaq_coupling_fwd(_p(P), _p(w), ap, B, L, int(mu), int(off), act_code(act), _p(fP), _p(logJ),
                                              ws, nb, _stream(P)), 'fthmc_plaq_coupling_fwd')
    return fP, logJ


def plaq_coupling_bwd(P, w, gfP, glogJ, mu: int, off: int, act='silu', need_gw=False, arch=None):
    """VJP of plaq_coupling_fwd: -> (gP, gw or None) for the upstream gradients gfP [B, L, L] and glogJ [B]."""
    P = _plaq_field(P); gfP = _plaq_field(gfP, 'gfP'); B, L, _ = P.shape
    glogJ = _dev(glogJ, 'glogJ').reshape(-1)
    if gfP.shape != P.shape or glogJ.numel() != B:
        raise FthmcError('plaq_coupling_bwd: gfP must be shaped like P, glogJ [B]')
    w, ap, a = _w1(w, P, arch)
    gP = torch.empty_like(P)
    gw = _tag(torch.empty(w.numel(), dtype=P.dtype, device=P.device), w) if need_gw else None
    ws, nb = _ws(P, B, L, 1, train=need_gw, arch=a)
    check(_lib.load().fthmc_plaq_coupling_bwd(_p(P), _p(w), ap, _p(gfP), _p(glogJ), B, L, int(mu), int(off), act_code(act),
                                              _p(gP), _p(gw), ws, nb, _stream(P)), 'fthmc_plaq_coupling_bwd')
    return gP, gw


def plaq_coupling_rev(fP, w, mu: int, off: int, act='silu', tol: float = 1e-12, arch=None):
    """NCPPlaqCouplingLayer.reverse on a plaquette field [B, L, L] -> (P, logJ[B])."""
    fP = _plaq_field(fP, 'fP'); B, L, _ = fP.shape
    w, ap, a = _w1(w, fP, arch)
    P = torch.empty_like(fP); logJ = torch.empty(B, dtype=fP.dtype, device=fP.device)
    ws, nb = _ws(fP, B, L, 1, arch=a)
    check(_lib.load().fthmc_plaq_coupling_rev(_p(fP), _p(w), ap, B, L, int(mu), int(off), act_code(act), float(tol),
                                              _p(P), _p(logJ), ws, nb, _stream(fP)), 'fthmc_plaq_coupling_rev')
    return P, logJ


# ---------------------------------------------------------------- whole flow
def _wall(w, n_layers, arch=None):
    """-> (all layers' weights, flat and tagged; the call's fthmc_arch_t*; the shape); (None, NULL, default) without layers"""
    if not n_layers:
        return None, None, DEFAULT_ARCH
    a = arch_of(w, arch)
    npl = arch_params(a)
    w = _tag(_dev(w, 'w').reshape(-1), w, a)
    if w.numel() != n_layers * npl:
        raise FthmcError(f'w: expected {n_layers}*{npl} doubles for net shape {a}, got {w.numel()}')
    return w, _arch(a), a


def flow_forward(x, w, n_layers: int, act='silu', arch=None, wkey=None):
    x = _field(x); B, _, L, _ = x.shape
    w, ap, a = _wall(w, n_layers, arch)
    y = torch.empty_like(x); ld = torch.empty(B, dtype=x.dtype, device=x.device)
    ws, nb = _ws(x, B, L, n_layers, arch=a)
    check(_lib.load().fthmc_flow_forward_v(_p(x), _p(w), ap, n_layers, B, L, act_code(act), _p(y), _p(ld), ws, nb,
                                         _stream(x), weights_version(wkey)), 'fthmc_flow_forward')
    return y, ld


def flow_reverse(y, w, n_layers: int, act='silu', tol: float = 1e-12, arch=None, wkey=None):
    y = _field(y, 'y'); B, _, L, _ = y.shape
    w, ap, a = _wall(w, n_layers, arch)
    x = torch.empty_like(y); ld = torch.empty(B, dtype=y.dtype, device=y.device)
    ws, nb = _ws(y, B, L, n_layers, arch=a)
    check(_lib.load().fthmc_flow_reverse_v(_p(y), _p(w), ap, n_layers, B, L, act_code(act), float(tol), _p(x), _p(ld),
                                         ws, nb, _stream(y), weights_version(wkey)), 'fthmc_flow_reverse')
    return x, ld


def ft_action(x, w, n_layers: int, beta: float, act='silu', arch=None, wkey=None):
    """-> (S_eff, logdet, plaq, Q) each [B]."""
    x = _field(x); B, _, L, _ = x.shape
    w, ap, a = _wall(w, n_layers, arch)
    S, ld, plaq, Q = (torch.empty(B, dtype=x.dtype, device=x.device) for _ in range(4))
    ws, nb = _ws(x, B, L, n_layers, arch=a)
    check(_lib.load().fthmc_ft_action_v(_p(x), _p(w), ap, n_layers, B, L, act_code(act), float(beta), _p(S), _p(ld),
                                      _p(plaq), _p(Q), ws, nb, _stream(x), weights_version(wkey)), 'fthmc_ft_action')
    return S, ld, plaq, Q


def ft_force(x, w, n_layers: int, beta: float, act='silu', arch=None, wkey=None):
    x = _field(x); B, _, L, _ = x.shape
    w, ap, a = _wall(w, n_layers, arch)
    F = torch.empty_like(x)
    ws, nb = _ws(x, B, L, n_layers, arch=a)
    check(_lib.load().fthmc_ft_force_v(_p(x), _p(w), ap, n_layers, B, L, act_code(act), float(beta), _p(F), ws, nb,
                                     _stream(x), weights_version(wkey)), 'fthmc_ft_force')
    return F


def ft_leapfrog(x, v, w, n_layers: int, beta: float, dt: float, nstep: int, act='silu', arch=None, wkey=None):
    x = _field(x); v = _field(v, 'v'); B, _, L, _ = x.shape
    w, ap, a = _wall(w, n_layers, arch)
    xo, vo = torch.empty_like(x), torch.empty_like(v)
    ws, nb = _ws(x, B, L, n_layers, arch=a)
    check(_lib.load().fthmc_ft_leapfrog_v(_p(x), _p(v), _p(w), ap, n_layers, B, L, act_code(act), float(beta), float(dt),
                                        int(nstep), _p(xo), _p(vo), ws, nb, _stream(x), weights_version(wkey)), 'fthmc_ft_leapfrog')
    return xo, vo


_SIDE_STREAMS: dict = {}


def default_groups(B: int, L: int) -> int:
    """Two chain groups once a launch is large enough to amortise the extra launches (measured: +10 % at
    B=128, L=64; -50 % at B=32, L=16 where the sequence is launch-bound)."""
    if L <= 16 and get_small_path() and get_variant() == 1:
        return 1              # small lattices: a sweep is ONE launch with a workgroup per chain, there is no tail to fill
    return 2 if B >= 16 and B * L * L >= (1 << 17) else 1


def default_train_groups(B: int, L: int) -> int:
    """Chain groups of a training gradient.  On the shapes the fused training backward serves (csrc/flow_bwd_train.hip: L a power
    of two >= 32) ONE: that kernel holds a CU by itself (141 KB of LDS, 241 VGPRs), a second stream finds no room beside it and
    only halves the walks (measured at the config-5 shard: 7.11 ms on one stream, 7.20-7.25 on two; level since the layers'
    partials are reduced in one go); elsewhere as the sampler."""
    if L >= 32 and (L & (L - 1)) == 0 and get_variant() == 1:
        return 1
    return default_groups(B, L)


def _side_streams(device, n: int):
    key = device.index
    pool = _SIDE_STREAMS.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def ft_trajectory(x, v, u, w, n_layers: int, beta: float, dt: float, nstep: int, act='silu', mode='md',
                  out: Optional[dict] = None, state_in: Optional[torch.Tensor] = None, groups: int = 1, arch=None, wkey=None,
                  side_streams=None):
    """One ftHMC trajectory per chain -> dict(x_new, dH, acc, H0, H1, plaq, Q, state).

    `out` may carry preallocated output tensors (same keys) so that a caller can replay the call
    inside a captured graph.  `state` is [3, B] = (S_eff, plaq, Q) of x_new; fed back as `state_in`
    of the next trajectory of the same chains (x = x_new) it saves that call's H0 flow sweep.

    groups > 1 splits the chains into that many contiguous groups (a list gives the group sizes) whose trajectories
    run on concurrent streams (forked from and joined back into the current stream, each with its own workspace):
    chains are independent, and one group's kernels fill the CUs that the other's leave idle in every kernel
    tail and dispatch gap.  Results do not depend on `groups`."""
    x = _field(x); v = _field(v, 'v'); u = _dev(u, 'u').reshape(-1); B, _, L, _ = x.shape
    if u.numel() != B:
        raise FthmcError(f'u: expected {B} uniforms, got {u.numel()}')
    w, ap, a = _wall(w, n_layers, arch)
    if out is None:
        out = {'x_new': torch.empty_like(x)}
        for k in ('dH', 'acc', 'H0', 'H1', 'plaq', 'Q'):
            out[k] = torch.empty(B, dtype=x.dtype, device=x.device)
    if 'state' not in out:
        out['state'] = torch.empty(3, B, dtype=x.dtype, device=x.device)
    edges = _group_edges(groups, B)                                     # an int, or explicit group sizes (chains per group)
    G = len(edges) - 1
    if G > 1:
        arch_ = a
        if state_in is not None:
            state_in = _dev(state_in, 'state_in')
            if state_in.numel() != 3 * B:
                raise FthmcError(f'state_in: expected [3, {B}]')
            state_in = state_in.reshape(3, B)
        main = torch.cuda.current_stream(x.device)
        sides = list(side_streams)[:G - 1] if side_streams is not None else _side_streams(x.device, G - 1)
        if len(sides) != G - 1:
            raise FthmcError(f'side_streams: {G - 1} streams needed for {G} chain groups')
        parts = []
        for st in sides:                                                # fork before anything of this call is on `main`
            st.wait_stream(main)
        for gi in list(range(1, G)) + [0]:
            a, b_ = edges[gi], edges[gi + 1]
            st = main if gi == 0 else sides[gi - 1]
            with torch.cuda.stream(st):
                og = {k: t[a:b_] for k, t in out.items() if k != 'state'}
                sg = state_in[:, a:b_].contiguous() if state_in is not None else None
                ft_trajectory(x[a:b_], v[a:b_], u[a:b_], w, n_layers, beta, dt, nstep, act, mode, og, sg, arch=arch_, wkey=wkey)   # one group: no side streams
                out['state'][:, a:b_].copy_(og['state'])
                parts.append(og)                                        # keep the group's temporaries alive until the join
        for st in sides:
            main.wait_stream(st)
        return out
    if state_in is not None:
        state_in = _dev(state_in, 'state_in')
        if state_in.numel() != 3 * B:
            raise FthmcError(f'state_in: expected [3, {B}]')
    m = {'md': MODE_MD, 'literal': MODE_LITERAL, 'reference_literal': MODE_LITERAL}[mode]
    ws, nb = _ws(x, B, L, n_layers, arch=a)
    check(_lib.load().fthmc_ft_trajectory_v(_p(x), _p(v), _p(u), _p(w), ap, n_layers, B, L, act_code(act), float(beta),
                                          float(dt), int(nstep), m, _p(out['x_new']), _p(out['dH']), _p(out['acc']),
                                          _p(out['H0']), _p(out['H1']), _p(out['plaq']), _p(out['Q']),
                                          _p(state_in), _p(out['state']), ws, nb,
                                          _stream(x), weights_version(wkey)), 'fthmc_ft_trajectory')
    return out


def train_grad(xi, w, n_layers: int, beta: float, act='silu', need_gw=True, groups: int = 1, _out=None, out_gw=None, arch=None):
    """-> dict(x, logq, logp, gw): pieces of train.train_step for a fixed prior draw; gw = gradient of
    mean_b (logq - logp) wrt the packed weights (written into `out_gw` when given: a contiguous float64 device tensor
    of w.numel() entries, e.g. the flat gradient buffer the conv parameters' .grad are views of).

    groups > 1: the chains are split into contiguous groups that run on concurrent streams (as in
    ft_trajectory); the groups' weight gradients are combined with weights B_g / B."""
    xi = _field(xi, 'xi'); B, _, L, _ = xi.shape
    w, ap, arch_ = _wall(w, n_layers, arch)
    G = max(1, min(int(groups), B))
    if out_gw is not None and (not out_gw.is_cuda or out_gw.dtype != torch.float64 or not out_gw.is_contiguous()
                               or out_gw.numel() != w.numel()):
        raise FthmcError(f'train_grad: out_gw must be a contiguous float64 device tensor of {w.numel()} entries')
    if _out is None:
        x = torch.empty_like(xi)
        logq, logp = (torch.empty(B, dtype=xi.dtype, device=xi.device) for _ in range(2))
    else:
        x, logq, logp = _out
    if G > 1:
        gws = torch.empty(G, w.numel(), dtype=xi.dtype, device=xi.device) if need_gw else None
        main = torch.cuda.current_stream(xi.device)
        sides = _side_streams(xi.device, G - 1)
        for st in sides:
            st.wait_stream(main)
        for gi in list(range(1, G)) + [0]:
            a, b_ = gi * B // G, (gi + 1) * B // G
            with torch.cuda.stream(main if gi == 0 else sides[gi - 1]):
                r = train_grad(xi[a:b_], w, n_layers, beta, act, need_gw, 1, (x[a:b_], logq[a:b_], logp[a:b_]), arch=arch_)
                if need_gw:
                    torch.mul(r['gw'], (b_ - a) / B, out=gws[gi])
        for st in sides:
            main.wait_stream(st)
        if need_gw and out_gw is not None:
            torch.sum(gws, 0, out=out_gw)
        return {'x': x, 'logq': logq, 'logp': logp,
                'gw': _tag(out_gw if out_gw is not None else gws.sum(0), w) if need_gw else None}
    gw = _tag(out_gw.reshape(-1) if out_gw is not None else torch.empty(w.numel(), dtype=xi.dtype, device=xi.device), w) \
        if need_gw else None
    ws, nb = _ws(xi, B, L, n_layers, train=True, arch=arch_)
    check(_lib.load().fthmc_train_grad(_p(xi), _p(w), ap, n_layers, B, L, act_code(act), float(beta), _p(x), _p(logq),
                                       _p(logp), _p(gw), ws, nb, _stream(xi)), 'fthmc_train_grad')
    return {'x': x, 'logq': logq, 'logp': logp, 'gw': gw}


def time_kernel(kind: str, x, w=None, mu=0, off=0, act='silu', beta=1.0, reps=20) -> float:
    """Average milliseconds per launch of one kernel ('flow_fwd', 'flow_bwd', 'leap_step', 'hmc_trajectory'),
    measured with HIP events on the current stream (synchronises)."""
    import ctypes
    x = _field(x); B, _, L, _ = x.shape
    k = {'flow_fwd': 0, 'flow_bwd': 1, 'leap_step': 2, 'hmc_trajectory': 3}[kind]
    wp = _p(_w1(w, x)[0]) if k < 2 else None
    ms = ctypes.c_double(0.0)
    ws, nb = _ws(x, B, L, 1)
    check(_lib.load().fthmc_time_kernel(k, _p(x), wp, None, B, L, int(mu), int(off), act_code(act), float(beta),
                                        int(reps), ctypes.byref(ms), ws, nb, _stream(x)), 'fthmc_time_kernel')
    return ms.value


def profile_stages(kind: str, x, w, mu=0, off=0, act='silu', beta=1.0):
    """Mean cycles per stage of one MFMA coupling-layer kernel launch ('flow_fwd' | 'flow_bwd' | 'flow_bwd_train' | 'flow_wgrad')."""
    import ctypes
    x = _field(x); B, _, L, _ = x.shape
    k = {'flow_fwd': 0, 'flow_bwd': 1, 'flow_bwd_train': 2, 'flow_wgrad': 3}[kind]
    buf = (ctypes.c_double * 16)()
    ws, nb = _ws(x, B, L, 1, train=(k >= 2))
    check(_lib.load().fthmc_profile_stages(k, _p(x), _p(_w1(w, x)[0]), None, B, L, int(mu), int(off), act_code(act),
                                           float(beta), buf, ws, nb, _stream(x)), 'fthmc_profile_stages')
    return list(buf)


def small_profile(x, v, u, w, n_layers: int, beta: float, dt: float, nstep: int, act='silu'):
    """Mean cycles per stage (32 slots, include/fthmc_hip.h) of one trajectory on the small-lattice fused path."""
    import ctypes
    x = _field(x); v = _field(v, 'v'); u = _dev(u, 'u').reshape(-1); B, _, L, _ = x.shape
    buf = (ctypes.c_double * 32)()
    ws, nb = _ws(x, B, L, n_layers)
    check(_lib.load().fthmc_small_profile(_p(x), _p(v), _p(u), _p(_wall(w, n_layers)[0]), None, n_layers, B, L, act_code(act), float(beta),
                                          float(dt), int(nstep), buf, ws, nb, _stream(x)), 'fthmc_small_profile')
    return list(buf)


def time_small(x, v, u, w, n_layers: int, beta: float, dt: float, nstep: int, act='silu', reps=20) -> float:
    """Average milliseconds per launch of the small-lattice fused trajectory kernel (HIP events on the current stream)."""
    import ctypes
    x = _field(x); v = _field(v, 'v'); u = _dev(u, 'u').reshape(-1); B, _, L, _ = x.shape
    ms = ctypes.c_double(0.0)
    ws, nb = _ws(x, B, L, n_layers)
    check(_lib.load().fthmc_time_small(_p(x), _p(v), _p(u), _p(_wall(w, n_layers)[0]), None, n_layers, B, L, act_code(act), float(beta),
                                       float(dt), int(nstep), int(reps), ctypes.byref(ms), ws, nb, _stream(x)), 'fthmc_time_small')
    return ms.value
