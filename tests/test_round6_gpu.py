"""GPU tests added in round 6: the training backward that computes its weight gradients itself (csrc/flow_bwd_train.hip: one
kernel per layer walks (chain, tile) items with the gradients' accumulators in registers) against the oracle's autograd,
on every stripe direction / offset, walks of one and of several items per workgroup, chains that do not fill a walk."""
import math

import numpy as np
import pytest
import torch

from conftest import ROOT  # noqa: F401

pytestmark = pytest.mark.gpu

ops = None
R = None


@pytest.fixture(scope='module', autouse=True)
def _mods():
    global ops, R
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    from fthmc_amd import ops as _ops
    from oracle import ref_cpu as _R
    ops, R = _ops, _R
    ops.set_variant(1)


def close(a, b, rtol=0.0, atol=0.0):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize('B,L,nl,beta', [(1, 32, 8, 2.0),      # 4 items: every workgroup walks one tile; all eight (mu, off)
                                         (3, 64, 8, 6.0),      # 48 items
                                         (20, 64, 2, 4.0),     # 320 items > 256 workgroups: walks of two items, the last round ragged
                                         (2, 128, 3, 3.0)])    # 128 items, tiles far from the lattice edge
def test_fused_training_backward_vs_oracle(B, L, nl, beta):
    """fthmc_train_grad on the tiled-exactly shapes (L a power of two >= 32: csrc/flow_bwd_train.hip) = loss.backward() of
    train_step (fthmc/train.py:191-210) through the oracle's autograd: loss pieces to 1e-11, every weight gradient to 1e-8
    relative (fp64; sums over up to 3e5 sites in another order); bit-identical on repetition (fixed summation order)."""
    gen = torch.Generator().manual_seed(600 + L + B)
    flow = R.default_flow(nl, gen)
    xi = (torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi
    nref = min(B, 3)                                            # the oracle's autograd is slow: the first chains, each on its own
    w = ops.pack_weights(flow, device='cuda')
    r = ops.train_grad(xi.cuda(), w, nl, beta, groups=1)
    r2 = ops.train_grad(xi.cuda(), w, nl, beta, groups=1)
    assert torch.equal(r['gw'], r2['gw']) and torch.equal(r['logq'], r2['logq'])
    per_chain = []
    for c in range(nref):
        out, grads = R.train_grads(xi[c:c + 1], flow, beta)
        close(r['logq'][c:c + 1], out['logq'], rtol=1e-11); close(r['logp'][c:c + 1], out['logp'], rtol=1e-11)
        rc = ops.train_grad(xi[c:c + 1].cuda(), w, nl, beta, groups=1)
        gws = ops.unpack_weight_grads(rc['gw'], nl)
        for li in range(nl):
            for pi in range(6):
                scale = float(grads[li][pi].abs().max())
                close(gws[li][pi], grads[li][pi], rtol=1e-8, atol=1e-10 * max(scale, 1.0))
        per_chain.append(rc['gw'])
    # the batch's gradient is the mean of its chains' (each chain's own call is checked above or is one more walk of the same kernel)
    for c in range(nref, B):
        per_chain.append(ops.train_grad(xi[c:c + 1].cuda(), w, nl, beta, groups=1)['gw'])
    close(r['gw'], sum(per_chain) / B, rtol=1e-9, atol=1e-10)


def test_fused_training_backward_leaves_the_force_path_alone():
    """the plaquette-gradient field the fused kernel hands down the sweep is the force path's: ft_force (k_flow_bwd_gather, no
    weight gradients) and the x-gradient implied by train_grad agree through the loss pieces; chain groups give the same gradient"""
    gen = torch.Generator().manual_seed(66)
    B, L, nl, beta = 4, 64, 4, 5.0
    flow = R.default_flow(nl, gen)
    w = ops.pack_weights(flow, device='cuda')
    xi = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    r1 = ops.train_grad(xi, w, nl, beta, groups=1)
    r2 = ops.train_grad(xi, w, nl, beta, groups=2)
    assert torch.equal(r1['logq'], r2['logq']) and torch.equal(r1['logp'], r2['logp'])
    close(r1['gw'], r2['gw'], rtol=1e-10, atol=1e-12)
    # directional derivative of the loss in weight space
    dw = torch.randn(w.numel(), generator=gen, dtype=torch.float64).cuda()
    dw = dw / dw.norm()

    def loss(wv):
        t = ops.train_grad(xi, wv, nl, beta, need_gw=False)
        return float((t['logq'] - t['logp']).mean())
    eps = 1e-6
    fd = (loss(w + eps * dw) - loss(w - eps * dw)) / (2 * eps)
    an = float((r1['gw'] * dw).sum())
    assert abs(an - fd) < 1e-5 * max(1.0, abs(an)), (an, fd)


@pytest.mark.parametrize('B,L', [(5, 128), (32, 256), (127, 128)])
def test_wave_split_action_sums_are_bit_identical(B, L):
    """few chains of a large lattice (B < 128, L >= 128) with a workspace at hand: the action / charge sums run one WAVE per
    workgroup (csrc/wilson.hip k_action_charge_waves) -- the same threads' sums added in the same order as the one-workgroup-
    per-chain kernel, which fthmc_wilson_action_charge (no workspace) still launches: every output bit-equal; the sum over a
    sweep's log J partials with one wave per layer (k_sum_parts) against the layers' own log J, added in order"""
    gen = torch.Generator().manual_seed(7 + B)
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    beta = 3.0
    S0, Q0, p0 = ops.wilson_action_charge(x, beta)
    S1, ld, p1, Q1 = ops.ft_action(x, torch.zeros(0, dtype=torch.float64, device='cuda'), 0, beta)
    assert torch.equal(S0, S1) and torch.equal(Q0, Q1) and torch.equal(p0, p1) and float(ld.abs().max()) == 0.0
    if B <= 32:
        nl = 3
        flow = R.default_flow(nl, gen)
        w = ops.pack_weights(flow, device='cuda')
        S, ld, _, _ = ops.ft_action(x, w, nl, beta)
        y, tot = x, torch.zeros(B, dtype=torch.float64, device='cuda')
        for l in range(nl):
            y, lj = ops.flow_layer_fwd(y, w[l * 955:(l + 1) * 955], l % 2, (l // 2) % 4)
            tot = tot + lj
        assert torch.equal(ld, tot)


def test_non_temporal_stash_instances_agree_with_the_cached_ones():
    """a layer's stash of 128 MB or more is stored past the caches (csrc/flow_fwd.hip launch_fwd: the SWEEP = 5 / 6 instances of
    the forward) -- the same kernel with another store instruction: a batch big enough to take them gives, chain by chain, what
    the chains give alone (small stash: the cached instances), and the first chain agrees with the oracle"""
    gen = torch.Generator().manual_seed(61)
    L, nl, beta = 256, 2, 3.0
    flow = R.default_flow(nl, gen)
    w = ops.pack_weights(flow, device='cuda')
    # training sweep: 8 chains x 65536 sites x 280 bytes = 147 MB per layer
    B = 8
    xi = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    r = ops.train_grad(xi, w, nl, beta, groups=1)
    alone = [ops.train_grad(xi[c:c + 1], w, nl, beta, groups=1) for c in range(B)]
    assert torch.equal(r['logq'], torch.cat([a['logq'] for a in alone])) and torch.equal(r['logp'], torch.cat([a['logp'] for a in alone]))
    close(r['gw'], sum(a['gw'] for a in alone) / B, rtol=1e-9, atol=1e-11)
    out, grads = R.train_grads(xi[:1].cpu(), flow, beta)
    close(alone[0]['logq'], out['logq'], rtol=1e-11)
    gws = ops.unpack_weight_grads(alone[0]['gw'], nl)
    for li in range(nl):
        for pi in range(6):
            scale = float(grads[li][pi].abs().max())
            close(gws[li][pi], grads[li][pi], rtol=1e-8, atol=1e-10 * max(scale, 1.0))
    # force sweep: 16 chains x 65536 sites x 152 bytes = 159 MB per layer
    B = 16
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    F = ops.ft_force(x, w, nl, beta)
    for c in (0, 7, 15):
        assert torch.equal(F[c:c + 1], ops.ft_force(x[c:c + 1], w, nl, beta))
    # ... and by POSITION in the sweep (csrc/api.hip sweep_forward: FlowLayerArgs::stash_far): a chain group of the headline shape
    # (64 chains of L = 64, 8 layers: 38 MiB of stash per layer) stores the stash of every layer but the last two past the caches
    L, nl, B = 64, 8, 64
    flow = R.default_flow(nl, gen)
    w = ops.pack_weights(flow, device='cuda')
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    F = ops.ft_force(x, w, nl, 6.0)
    for c in (0, 31, 63):
        assert torch.equal(F[c:c + 1], ops.ft_force(x[c:c + 1], w, nl, 6.0))
    close(F[:1], R.ft_force(x[:1].cpu(), flow, 6.0), rtol=1e-9, atol=1e-9)


def test_capture_holds_the_garbage_collector_off():
    """graph_loop.capture: no automatic collection starts in the capturing thread (a finaliser that calls into the HIP runtime --
    an older graph's, a stream's -- would end the process there); the collector's state comes back, also after an error"""
    import gc
    from fthmc_amd.graph_loop import capture
    st, t = torch.cuda.Stream(), torch.zeros(4, dtype=torch.float64, device='cuda')
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        t.add_(1.0)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        assert gc.isenabled()
        with capture(g, st):
            assert not gc.isenabled()
            t.add_(1.0)
        assert gc.isenabled()
        g.replay(); g.replay()
        st.synchronize()
        assert float(t[0]) == 3.0
        with pytest.raises(ZeroDivisionError):
            with capture(torch.cuda.CUDAGraph(), st):
                1 / 0
        assert gc.isenabled() and not torch.cuda.is_current_stream_capturing()
