"""GPU tests added in round 4: the device-bound training loop (flat parameter buffer, captured step, metrics kernel,
Philox prior draw) against the step-by-step route and the oracle; the single-rank RCCL rehearsal (FTHMC_FORCE_PG=1) of
bench.py and train_step; the delta-Q^2 tooling on HIP-produced histories; the full config-5 batch."""
import json
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_flow, load_golden

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


def H(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def close(a, b, rtol=1e-10, atol=1e-10):
    np.testing.assert_allclose(H(a), H(b), rtol=rtol, atol=atol)


# ---------------------------------------------------------------- flat parameter buffer
def test_flat_parameter_buffer_keeps_the_reference_module_layout():
    """flatten_flow re-homes the conv parameters as views of one buffer in ABI order: state_dict keys and values,
    load_state_dict, optimizers and transfer_to_new_lattice keep working, flow_weights hands the buffer out without a copy."""
    from fthmc_amd import train as T
    from fthmc_amd.config import TrainConfig
    from fthmc_amd.utils import layers as LY
    tc = TrainConfig(L=8, beta=2.0, n_layers=4, batch_size=8)
    torch.manual_seed(5)
    model = T.get_model(tc)
    before = {k: v.clone() for k, v in model.layers.state_dict().items()}
    packed = LY.flow_weights(model.layers)                       # not flattened yet: a packed copy
    flat = LY.flatten_flow(model.layers)
    assert torch.equal(flat, packed) and flat.numel() == 4 * 955
    assert LY.flatten_flow(model.layers) is flat                 # idempotent
    assert LY.flow_weights(model.layers).data_ptr() == flat.data_ptr()          # no copy any more
    after = model.layers.state_dict()
    assert list(after) == list(before) and all(torch.equal(after[k], before[k]) for k in before)
    assert '0.plaq_coupling.net.0.weight' in after and after['3.plaq_coupling.net.4.bias'].shape == (3,)
    # in-place updates through the parameters are updates of the buffer, and the other way round
    p = model.layers[1].plaq_coupling.net[2].weight
    with torch.no_grad():
        p.mul_(2.0)
    assert torch.equal(LY.flow_weights(model.layers)[955 + 152:955 + 152 + 576].view(8, 8, 3, 3), p)
    model.layers.load_state_dict(before)                         # copies in place: still views
    assert LY.flatten_flow(model.layers) is flat and torch.equal(flat, packed)
    # gradient buffer: every .grad is a view of it
    g = LY.flow_grad_buffer(model.layers)
    LY.attach_grads(model.layers)
    g.fill_(3.0)
    assert all(float(q.grad.min()) == 3.0 for q in model.layers.parameters())
    # a second ModuleList over the same nets (transfer) shares the buffer
    big = T.transfer_to_new_lattice(16, model.layers)
    assert LY.flow_weights(big.layers).data_ptr() == flat.data_ptr()
    # .to() re-creates the parameters: the next call flattens again
    model.layers.to(torch.float64)
    assert LY.flow_weights(model.layers).numel() == flat.numel()


# ---------------------------------------------------------------- device pieces of a training step
def test_prior_draw_and_metrics_kernels():
    from fthmc_amd import parallel
    B, L, beta = 37, 12, 3.0
    seeds = parallel.chain_seeds(7, 100, 100 + B, 3).cuda()
    xi = ops.random_uniform(seeds, (B, 2, L, L), -math.pi, math.pi)
    assert xi.shape == (B, 2, L, L) and float(xi.min()) >= -math.pi and float(xi.max()) < math.pi
    assert abs(float(xi.mean())) < 0.1 and abs(float(xi.var()) - math.pi ** 2 / 3) < 0.2
    # a chain's draw depends on its seed only: any sub-batch reproduces its chains bit for bit
    assert torch.equal(ops.random_uniform(seeds[5:9], (4, 2, L, L), -math.pi, math.pi), xi[5:9])
    assert not torch.equal(xi[0], xi[1])
    gen = torch.Generator().manual_seed(3)
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    logq = torch.randn(B, generator=gen, dtype=torch.float64).cuda() * 3 - 500
    logp = torch.randn(B, generator=gen, dtype=torch.float64).cuda() * 3 + 200
    row = ops.train_metrics(xi, x, logq, logp, beta, dkl_factor=0.7)
    m = ops.split_metrics(row, B)
    logw = logp - logq
    ess = torch.exp(2 * torch.logsumexp(logw, 0) - torch.logsumexp(2 * logw, 0)) / B     # calc_ess, distributions.py:27-37
    close(m['loss_dkl'], 0.7 * (logq - logp).mean(), rtol=1e-13)
    close(m['ess'], ess, rtol=1e-11)
    q, qi = ops.wilson_action_charge(x, beta)[1], ops.wilson_action_charge(xi, beta)[1]
    assert torch.equal(m['logp'], logp) and torch.equal(m['logq'], logq) and torch.equal(m['q'], q)
    close(m['dq'], (q - qi).abs(), atol=0); close(m['plaq'], logp / (beta * L * L), rtol=1e-15)


@pytest.mark.parametrize('decoupled,wd', [(False, 0.0), (False, 1e-2), (True, 1e-2)])
def test_flat_adam_equals_torch_adam(decoupled, wd):
    """FlatAdam (ONE launch on the flat buffers, step count and rate on the device) against torch.optim.Adam / AdamW on the
    same gradients, with a learning-rate change on the way, and its state_dict loaded into the torch optimizer and back."""
    from fthmc_amd import train as T
    from fthmc_amd.config import TrainConfig
    from fthmc_amd.utils import layers as LY
    tc = TrainConfig(L=8, beta=2.0, n_layers=3, batch_size=4)
    torch.manual_seed(2)
    a, b = T.get_model(tc), T.get_model(tc)
    b.layers.load_state_dict(a.layers.state_dict())
    oa = T.FlatAdam(a.layers, lr=3e-3, weight_decay=wd, decoupled=decoupled)
    ob = (torch.optim.AdamW if decoupled else torch.optim.Adam)(b.layers.parameters(), lr=3e-3, weight_decay=wd, foreach=False)
    ga = LY.flow_grad_buffer(a.layers)
    gen = torch.Generator().manual_seed(9)
    for k in range(7):
        g = torch.randn(ga.numel(), generator=gen, dtype=torch.float64).cuda() * (1.0 + k)
        ga.copy_(g)
        o = 0
        for p in b.layers.parameters():
            p.grad = g[o:o + p.numel()].view(p.shape).clone(); o += p.numel()
        if k == 4:
            for opt in (oa, ob):
                opt.param_groups[0]['lr'] = 1e-3             # what a scheduler does
        oa.step(); ob.step()
        close(LY.flow_weights(a.layers), LY.flow_weights(b.layers), rtol=1e-13, atol=1e-15)
    import copy
    sa, sb = copy.deepcopy(oa.state_dict()), copy.deepcopy(ob.state_dict())      # state_dict() hands out references, not copies
    assert sa['param_groups'][0]['lr'] == 1e-3 and len(sa['state']) == len(sb['state']) == 18
    for i in sa['state']:
        close(sa['state'][i]['exp_avg'], sb['state'][i]['exp_avg'], rtol=1e-13, atol=1e-18)
        close(sa['state'][i]['exp_avg_sq'], sb['state'][i]['exp_avg_sq'], rtol=1e-13, atol=1e-18)
        assert float(sa['state'][i]['step']) == float(sb['state'][i]['step']) == 7.0
    # either state loads into the other kind and the two keep walking together
    oa.load_state_dict(sb); ob.load_state_dict(sa)
    g = torch.randn(ga.numel(), generator=gen, dtype=torch.float64).cuda()
    ga.copy_(g); o = 0
    for p in b.layers.parameters():
        p.grad = g[o:o + p.numel()].view(p.shape).clone(); o += p.numel()
    oa.step(); ob.step()
    close(LY.flow_weights(a.layers), LY.flow_weights(b.layers), rtol=1e-13, atol=1e-15)


@pytest.mark.parametrize('L,B,nl', [(8, 16, 4), (16, 8, 2), (20, 4, 3)])
def test_graph_trainer_equals_step_by_step(L, B, nl):
    """GraphTrainer (captured step, Philox prior, flat gradient buffer, FlatAdam: one launch per step) walks through the same
    weights and metrics as train_step called step by step on the same draws with the reference's optimizer
    (optim.Adam, train.py:297), and its eager mode equals its captured mode bit for bit."""
    from fthmc_amd import parallel, train as T
    from fthmc_amd.config import TrainConfig
    from fthmc_amd.utils import layers as LY
    from fthmc_amd.utils import qed_helpers as qed
    tc = TrainConfig(L=L, beta=2.5, n_layers=nl, batch_size=B, base_lr=2e-3, print_freq=0)
    torch.manual_seed(21)
    m0 = T.get_model(tc)
    init = {k: v.clone() for k, v in m0.layers.state_dict().items()}
    nsteps, seed = 5, 99
    runs = {}
    for mode in ('graph', 'eager'):
        model = T.get_model(tc); model.layers.load_state_dict(init)
        tr = T.GraphTrainer(model, tc, T.make_optimizer(model, tc), B, seed=seed, use_graph=(mode == 'graph'))
        for _ in range(nsteps):
            tr.step()
        hist_ = tr.history()                                     # synchronises the trainer's stream
        runs[mode] = (LY.flow_weights(model.layers).clone(), hist_)
    assert torch.equal(runs['graph'][0], runs['eager'][0])
    for k in T.METRIC_KEYS:
        assert all(np.array_equal(a, b) for a, b in zip(runs['graph'][1][k], runs['eager'][1][k])), k
    # step by step through the reference-shaped API with a plain Adam on the same draws
    model = T.get_model(tc); model.layers.load_state_dict(init)
    opt = torch.optim.Adam(model.layers.parameters(), lr=tc.base_lr)
    act = qed.BatchAction(tc.beta)
    hist = []
    for k in range(nsteps):
        xi = ops.random_uniform(parallel.chain_seeds(seed, 0, B, k).cuda(), (B, 2, L, L), -math.pi, math.pi)
        hist.append(T.train_step(model, tc, act, opt, B, xi=xi))
    close(LY.flow_weights(model.layers), runs['graph'][0], rtol=1e-9, atol=1e-12)
    gh = runs['graph'][1]
    assert len(gh['loss_dkl']) == nsteps and gh['logp'][0].shape == (B,) and gh['ess'][0].shape == ()
    for k in range(nsteps):
        for key in ('loss_dkl', 'ess', 'logp', 'logq', 'plaq'):
            close(hist[k][key], gh[key][k], rtol=1e-8, atol=1e-9)
        close(hist[k]['q'], gh['q'][k], atol=1e-8); close(hist[k]['dq'], gh['dq'][k], atol=1e-8)


def test_train_step_metrics_match_the_autograd_route():
    """train_step(fused=True) (flat gradient buffer, metrics kernel) against train_step(fused=False) (layer-wise autograd,
    torch formulas) on one draw: loss, ESS, per-chain arrays and every parameter gradient."""
    from fthmc_amd import train as T
    from fthmc_amd.config import TrainConfig
    from fthmc_amd.utils import qed_helpers as qed
    tc = TrainConfig(L=12, beta=3.0, n_layers=3, batch_size=6, base_lr=0.0, print_freq=0)
    torch.manual_seed(8)
    model = T.get_model(tc)
    xi = model.prior.sample_n(6)
    act = qed.BatchAction(tc.beta)
    outs, grads = [], []
    for fused in (True, False):
        opt = torch.optim.SGD(model.layers.parameters(), lr=0.0)
        outs.append(T.train_step(model, tc, act, opt, 6, xi=xi, fused=fused, dkl_factor=1.3))
        grads.append([p.grad.clone() for p in model.layers.parameters()])
    for k in T.METRIC_KEYS:
        close(outs[0][k], outs[1][k], rtol=1e-9, atol=1e-9)
    gmax = max(float(g.abs().max()) for g in grads[1])
    for a, b in zip(*grads):
        close(a, b, rtol=1e-8, atol=1e-11 * max(gmax, 1.0))


# ---------------------------------------------------------------- small lattices: the training sweep in one launch
@pytest.mark.parametrize('L,nl,B,beta,act', [(8, 4, 5, 2.0, 'silu'), (12, 3, 3, 3.0, 'silu'), (16, 8, 9, 4.0, 'silu'), (16, 2, 2, 2.0, 'relu'),
                                             (8, 1, 1, 1.0, 'leaky_relu')])
def test_small_lattice_training_sweep(L, nl, B, beta, act):
    """fthmc_train_grad at L = 8, 12, 16: ONE launch runs the forward sweep (stash, h1, h2, log J), the loss pieces and the
    backward sweep of every chain (flow_small.hip, training sweep), k_flow_wgrad turns the pre-activation gradients it leaves
    into weight gradients.  Against the tiled path (small path off) and against the oracle's autograd."""
    gen = torch.Generator().manual_seed(900 + L + nl)
    flow = R.default_flow(nl, gen)
    w = ops.pack_weights(flow, device='cuda')
    xi = (torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi
    assert ops.get_small_path()
    r = ops.train_grad(xi.cuda(), w, nl, beta, act)
    ops.set_small_path(False)
    try:
        rt = ops.train_grad(xi.cuda(), w, nl, beta, act)
    finally:
        ops.set_small_path(True)
    d = (H(r['x']) - H(rt['x']) + np.pi) % (2 * np.pi) - np.pi
    assert np.abs(d).max() < 1e-11
    close(r['logq'], rt['logq'], rtol=1e-12); close(r['logp'], rt['logp'], rtol=1e-12)
    gmax = float(rt['gw'].abs().max())
    close(r['gw'], rt['gw'], rtol=1e-9, atol=1e-12 * max(gmax, 1.0))
    out, grads = R.train_grads(xi, flow, beta, act)
    close(r['logq'], out['logq'], rtol=1e-11); close(r['logp'], out['logp'], rtol=1e-11)
    gws = ops.unpack_weight_grads(r['gw'], nl)
    for li in range(nl):
        for pi in range(6):
            close(gws[li][pi], grads[li][pi], rtol=1e-8, atol=1e-11 * max(gmax, 1.0))
    assert torch.equal(ops.train_grad(xi.cuda(), w, nl, beta, act)['gw'], r['gw'])          # deterministic
    # the chains of a batch do not see each other: chain 0 alone
    r1 = ops.train_grad(xi[:1].cuda(), w, nl, beta, act, need_gw=False)
    close(r1['logq'], r['logq'][:1], rtol=1e-13); close(r1['logp'], r['logp'][:1], rtol=1e-13)


# ---------------------------------------------------------------- the net shape is an argument of the call
def test_two_net_shapes_on_two_threads_and_streams():
    """Two flows of different s/t net shapes driven from two Python threads on two streams at the same time (ctypes drops
    the GIL inside the C entry points): the shape travels with each call (fthmc_arch_t), the library keeps none, so neither
    thread can see the other's -- every result of every repetition equals the oracle's."""
    import threading
    gen = torch.Generator().manual_seed(77)
    B, L, nl, beta = 4, 16, 3, 2.0
    jobs = []
    for hidden, k, n_mix in (((8, 8), 3, 2), ((4, 6, 5), 5, 1)):
        flow = R.default_flow(nl, gen, hidden=hidden, n_mix=n_mix, k=k)
        x = (torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi
        jobs.append(dict(w=ops.pack_weights(flow, device='cuda'), x=x.cuda(), F=R.ft_force(x, flow, beta),
                         S=R.ft_action(x, flow, beta).detach(), stream=torch.cuda.Stream(), out=[], err=[]))
    torch.cuda.synchronize()
    gate = threading.Barrier(2)

    def run(j):
        try:
            with torch.cuda.stream(j['stream']):
                gate.wait()
                for _ in range(25):
                    j['out'].append((ops.ft_force(j['x'], j['w'], nl, beta), ops.ft_action(j['x'], j['w'], nl, beta)[0]))
                j['stream'].synchronize()
        except Exception as e:                                   # surfaces in the main thread below
            j['err'].append(e)
    threads = [threading.Thread(target=run, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for j in jobs:
        assert not j['err'], j['err']
        assert len(j['out']) == 25
        for F, S in j['out']:
            close(F, j['F'], rtol=1e-9, atol=1e-10); close(S, j['S'], rtol=1e-11, atol=1e-10)


@pytest.mark.parametrize('hidden,k,n_mix,L,nl', [((8, 8), 3, 2, 8, 3), ((6,), 3, 3, 12, 2)])
def test_final_tanh_option(hidden, k, n_mix, L, nl):
    """make_conv_net(use_final_tanh=True) (layers.py:144,163-164; never built by the reference, :419): the tanh behind the last
    conv through the C ABI (`fthmc_arch_t.final_tanh`, plain kernels) against the oracle -- forward, log det, reverse, force,
    training gradient -- and through the reference-shaped layer with autograd."""
    gen = torch.Generator().manual_seed(300 + L)
    B, beta = 3, 2.0
    flow = R.default_flow(nl, gen, hidden=hidden, n_mix=n_mix, k=k)
    flow = [tuple(t * 2.0 for t in lw) for lw in flow]                       # large enough for the tanh to bend
    w = ops.pack_weights(flow, device='cuda', final_tanh=True)
    assert ops.arch_of(w) == (tuple(hidden), k, n_mix, True)
    x = (torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi
    y, ld = ops.flow_forward(x.cuda(), w, nl)
    yc, ldc = R.flow_forward(x, flow, 'silu+tanh')
    d = (H(y) - H(yc) + np.pi) % (2 * np.pi) - np.pi
    assert np.abs(d).max() < 1e-11
    close(ld, ldc, rtol=1e-11, atol=1e-11)
    y0, _ = ops.flow_forward(x.cuda(), ops.pack_weights(flow, device='cuda'), nl)
    assert float((y - y0).abs().max()) > 1e-3                                # it is a different map
    xb, ldb = ops.flow_reverse(y, w, nl)
    d = (H(xb) - H(x) + np.pi) % (2 * np.pi) - np.pi
    assert np.abs(d).max() < 1e-9 and float((ldb + ld).abs().max()) < 1e-8
    close(ops.ft_force(x.cuda(), w, nl, beta), R.ft_force(x, flow, beta, 'silu+tanh'), rtol=1e-9, atol=1e-10)
    out, grads = R.train_grads(x, flow, beta, 'silu+tanh')
    r = ops.train_grad(x.cuda(), w, nl, beta)
    close(r['logq'], out['logq'], rtol=1e-11); close(r['logp'], out['logp'], rtol=1e-11)
    gws = ops.unpack_weight_grads(r['gw'], nl)
    gmax = max(float(g.abs().max()) for lg in grads for g in lg)
    for li in range(nl):
        for pi in range(len(flow[0])):
            close(gws[li][pi], grads[li][pi], rtol=1e-8, atol=1e-11 * max(gmax, 1.0))
    # the reference-shaped modules: a coupling layer built on a net with the tanh, forward and autograd
    from fthmc_amd.utils import layers as LY
    net = LY.make_conv_net(hidden_sizes=list(hidden), kernel_size=k, in_channels=2, out_channels=n_mix + 1, use_final_tanh=True)
    assert isinstance(net[-1], torch.nn.Tanh) and net.final_tanh
    with torch.no_grad():
        for p_, t_ in zip(LY.net_weights(net), flow[0]):
            p_.copy_(t_.cuda())
    layer = LY.GaugeEquivCouplingLayer(lattice_shape=(L, L), mask_mu=0, mask_off=0,
                                       plaq_coupling=LY.NCPPlaqCouplingLayer(net, mask_shape=(L, L), mask_mu=0, mask_off=0))
    xg = x.cuda().requires_grad_(True)
    yl, lj = layer(xg)
    xr_ = x.clone().requires_grad_(True)
    wr_ = [t.clone().requires_grad_(True) for t in flow[0]]
    ylc, ljc = R.layer_forward(xr_, wr_, 0, 0, 'silu+tanh')
    close(lj, ljc.detach(), rtol=1e-11, atol=1e-11)
    c = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64)
    ((yl * c.cuda()).sum() + lj.sum()).backward()
    ((ylc * c).sum() + ljc.sum()).backward()
    close(xg.grad, xr_.grad, rtol=1e-9, atol=1e-10 * float(xr_.grad.abs().max()))
    for p_, t_ in zip(LY.net_weights(net), wr_):
        close(p_.grad, t_.grad, rtol=1e-8, atol=1e-10 * max(1.0, float(t_.grad.abs().max())))
    xi_, _ = layer.reverse(yl.detach())
    d = (H(xi_) - H(x) + np.pi) % (2 * np.pi) - np.pi
    assert np.abs(d).max() < 1e-9


@pytest.mark.parametrize('hidden,k,n_mix,L,B', [((8, 8), 3, 2, 8, 3), ((8, 8), 3, 2, 40, 2), ((4, 6), 5, 3, 12, 2)])
def test_plaquette_level_autograd(hidden, k, n_mix, L, B):
    """NCPPlaqCouplingLayer.forward on a plaquette field under autograd (layers.py:348-371): fthmc_plaq_coupling_bwd -- the
    link-level backward kernels with the upstream gradient dressed as a link gradient -- against the oracle's autograd, wrt the
    plaquette field and wrt every conv weight, all (mu, off) classes, tuned and plain kernels, through `ops` and the module."""
    from fthmc_amd.utils import layers as LY
    gen = torch.Generator().manual_seed(77 + L + k)
    flow = R.default_flow(1, gen, hidden=hidden, n_mix=n_mix, k=k)[0]
    w = ops.pack_weights([flow], device='cuda')
    for mu, off in ((0, 0), (1, 2), (0, 3), (1, 1)):
        P = (torch.rand(B, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi
        c = torch.randn(B, L, L, generator=gen, dtype=torch.float64); dl = torch.randn(B, generator=gen, dtype=torch.float64)
        Pr = P.clone().requires_grad_(True)
        wr = [t.clone().requires_grad_(True) for t in flow]
        fPc, ljc = R.plaq_coupling_forward(Pr, wr, mu, off)
        ((fPc * c).sum() + (ljc * dl).sum()).backward()
        gP, gw = ops.plaq_coupling_bwd(P.cuda(), w, c.cuda(), dl.cuda(), mu, off, need_gw=True)
        close(gP, Pr.grad, rtol=1e-9, atol=1e-10 * max(1.0, float(Pr.grad.abs().max())))
        for g_, t_ in zip(ops.unpack_weight_grads(gw, 1)[0], wr):
            close(g_, t_.grad, rtol=1e-8, atol=1e-10 * max(1.0, float(t_.grad.abs().max())))
        assert torch.equal(ops.plaq_coupling_bwd(P.cuda(), w, c.cuda(), dl.cuda(), mu, off)[0], gP)
    # the module: P.requires_grad no longer raises
    net = LY.make_conv_net(hidden_sizes=list(hidden), kernel_size=k, in_channels=2, out_channels=n_mix + 1)
    with torch.no_grad():
        for p_, t_ in zip(LY.net_weights(net), flow):
            p_.copy_(t_.cuda())
    layer = LY.NCPPlaqCouplingLayer(net, mask_shape=(L, L), mask_mu=mu, mask_off=off)
    Pg = P.cuda().requires_grad_(True)
    fP, lj = layer(Pg)
    ((fP * c.cuda()).sum() + (lj * dl.cuda()).sum()).backward()
    close(Pg.grad, Pr.grad, rtol=1e-9, atol=1e-10 * max(1.0, float(Pr.grad.abs().max())))
    for p_, t_ in zip(LY.net_weights(net), wr):
        close(p_.grad, t_.grad, rtol=1e-8, atol=1e-10 * max(1.0, float(t_.grad.abs().max())))
    with torch.no_grad():
        fP2, _ = layer(P.cuda())
    assert torch.equal(fP2, fP.detach())


def test_kernel_wider_than_the_lattice_is_refused():
    """A circular pad wider than the lattice (kernel_size // 2 > L) is refused before any launch, as torch's circular Conv2d
    refuses it (the plain kernels fold an index once)."""
    from fthmc_amd._lib import FthmcError
    gen = torch.Generator().manual_seed(4)
    flow = R.default_flow(1, gen, hidden=(4,), n_mix=2, k=11)
    w = ops.pack_weights(flow, device='cuda')
    x = torch.zeros(2, 2, 4, 4, dtype=torch.float64, device='cuda')
    for call in (lambda: ops.flow_forward(x, w, 1), lambda: ops.ft_force(x, w, 1, 1.0), lambda: ops.flow_layer_fwd(x, w, 0, 0)):
        with pytest.raises(FthmcError, match='unsupported'):
            call()
    ok = ops.pack_weights(R.default_flow(1, gen, hidden=(4,), n_mix=2, k=9), device='cuda')      # 9 // 2 = 4 <= L
    assert torch.isfinite(ops.flow_forward(x, ok, 1)[0]).all()


# ---------------------------------------------------------------- row-strip stencil kernels of the flowed step
@pytest.mark.parametrize('B,L,nl', [(3, 64, 2), (2, 128, 1)])
def test_row_strip_seed_and_kick_kernels(B, L, nl):
    """k_gp_rows / k_kick_rows (16-byte accesses, two sites per thread; serve L % 64 == 0) == k_force<2> / k_kick_from_gp bit
    for bit, through the flowed force and the flowed leapfrog (seed of the backward sweep, kick + drift), and the plain Wilson
    part of the force == the oracle."""
    gen = torch.Generator().manual_seed(640 + L)
    beta, dt, nstep = 5.0, 0.1, 2
    w = ops.pack_weights(R.default_flow(nl, gen), device='cuda')
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    v = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64).cuda()

    def run():
        return (ops.ft_force(x, w, nl, beta), *ops.ft_leapfrog(x, v, w, nl, beta, dt, nstep), ops.ft_force(x, None, 0, beta))
    a = run()
    os.environ['FTHMC_LEAP_ROWS'] = '0'
    try:
        ops.set_variant(1)
        b = run()
    finally:
        os.environ['FTHMC_LEAP_ROWS'] = '1'
        ops.set_variant(1)
    for ta, tb in zip(a, b):
        assert torch.equal(ta, tb)
    close(a[3], R.wilson_force_analytic(x.cpu(), beta), rtol=1e-12, atol=1e-12)     # zero layers: seed + stencil adjoint alone


# ---------------------------------------------------------------- single-rank RCCL rehearsal
def _bench_child(tmp_path, tag, env_extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='2', **env_extra)
    env.pop('FTHMC_DIST_BACKEND', None)
    dump = str(tmp_path / tag)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', '2', '--steps', '3', '--warmup', '1',
                        '--no-cpu-baseline', '--regions', '1', '--dump', dump], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
    return line, dict(np.load(dump + '.1.0.npz'))


def test_single_rank_nccl_group_rehearses_the_multi_gpu_path(tmp_path):
    """FTHMC_FORCE_PG=1: bench.py creates a ONE-rank `nccl` (= RCCL) process group in a fresh process and takes every
    collective branch of the 8-GPU run -- communicator creation before the capture, the asynchronous C1 all-reduce after
    every graph replay, barrier(device_ids=...), the MAX-reduced region time -- and ends in the same chains, bit for bit,
    as the run without a group."""
    plain, d0 = _bench_child(tmp_path, 'plain', {})
    forced, d1 = _bench_child(tmp_path, 'forced', {'FTHMC_FORCE_PG': '1'})
    assert plain['config']['process_group'] is None and forced['config']['process_group'] == 'nccl'
    assert forced['n_gpus'] == 1 and forced['config']['launch'] == 'hipGraph replay'
    for k in ('x0', 'x', 'dH', 'acc', 'Q', 'plaq'):
        assert np.array_equal(d0[k], d1[k]), k
    assert plain['acceptance'] == forced['acceptance'] and plain['plaq'] == forced['plaq']


_TRAIN_WORKER = r'''
import os, sys, json, hashlib, math
sys.path.insert(0, os.environ['FTHMC_ROOT'])
import torch
from fthmc_amd import ops, parallel, train as T
from fthmc_amd.config import TrainConfig
from fthmc_amd.utils import layers as LY, qed_helpers as qed
parallel.init()
tc = TrainConfig(L=16, beta=4.0, n_layers=4, batch_size=32, base_lr=1e-3, print_freq=0)
torch.manual_seed(17)
model = T.get_model(tc)
opt = torch.optim.Adam(model.layers.parameters(), lr=tc.base_lr)
act = qed.BatchAction(tc.beta)
out = None
for k in range(3):
    xi = ops.random_uniform(parallel.chain_seeds(5, 0, 32, k).cuda(), (32, 2, 16, 16), -math.pi, math.pi)
    out = T.train_step(model, tc, act, opt, 32, xi=xi, fused=True)
# and the loop object: with a group its captured step carries the C2 collectives
tr = T.GraphTrainer(model, tc, T.make_optimizer(model, tc), 32, seed=5)
for _ in range(4):
    tr.step()
m = tr.metrics()
w = LY.flow_weights(model.layers).cpu().numpy()
print(json.dumps({'group': parallel.have_group(), 'backend': torch.distributed.get_backend() if parallel.have_group() else None,
                  'captured': tr.graph is not None, 'w': hashlib.sha256(w.tobytes()).hexdigest(),
                  'loss': float(out['loss_dkl']), 'ess': float(out['ess']), 'loss2': float(m['loss_dkl'])}))
if parallel.have_group():
    torch.distributed.destroy_process_group()
'''


def test_single_rank_nccl_group_train_step():
    """train_step(fused=True) and GraphTrainer under a one-rank `nccl` group: the C2 branches (gradient all-reduce, global
    loss mean, MAX + SUM all-reduces of the ESS logsumexp) run over RCCL -- in GraphTrainer as part of the CAPTURED step, replayed
    -- and leave the same weights, bit for bit, as no group."""
    res = {}
    for tag, extra in (('plain', {}), ('forced', {'FTHMC_FORCE_PG': '1'})):
        env = dict(os.environ, FTHMC_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='2', **extra)
        env.pop('FTHMC_DIST_BACKEND', None)
        p = subprocess.run([sys.executable, '-c', _TRAIN_WORKER], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        res[tag] = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
    assert res['plain']['group'] is False and res['plain']['captured'] is True
    assert res['forced']['group'] is True and res['forced']['backend'] == 'nccl' and res['forced']['captured'] is True
    assert res['plain']['w'] == res['forced']['w']
    # with a group the loss mean and the ESS go through torch sums + all-reduces instead of the metrics kernel: rounding only
    for k in ('loss', 'loss2', 'ess'):
        assert abs(res['plain'][k] - res['forced'][k]) <= 1e-12 * max(1.0, abs(res['plain'][k])), k


# ---------------------------------------------------------------- f4: delta-Q^2 tooling on HIP-produced histories
def test_observables_tooling_on_hip_histories():
    """64 physical-field ftHMC trajectories of 8 chains at L = 8 through qed_helpers.ft_hmc (what ft_run loops over: inverse
    sweep -> trajectory -> forward sweep, ipynb/ft_hmc.py:420-435) on the HIP path, and the same chains through the oracle
    with the same momenta and accept draws: equal accept and charge histories, and equal rows out of the delta-Q^2-vs-lag /
    block-error tooling (ipynb/ft_hmc.py:16-53,168-176) in the reference's literal mode, chain by chain and for the ensemble."""
    from fthmc_amd.config import Param
    from fthmc_amd.utils import layers as LY, observables as OB, qed_helpers as qed
    gen = torch.Generator().manual_seed(64)
    B, L, nl, beta, tau, nstep, ntraj = 8, 8, 4, 2.0, 1.0, 10, 64
    wts = R.default_flow(nl, gen)
    flow = LY.make_u1_equiv_layers(n_layers=nl, n_mixture_comps=2, lattice_shape=(L, L), hidden_sizes=[8, 8], kernel_size=3)
    names = ('net.0.weight', 'net.0.bias', 'net.2.weight', 'net.2.bias', 'net.4.weight', 'net.4.bias')
    flow.load_state_dict({f'{li}.plaq_coupling.{n}': wts[li][pi].cuda() for li in range(nl) for pi, n in enumerate(names)})
    param = Param(beta=beta, L=L, tau=tau, nstep=nstep)
    x0 = (torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi
    vs = torch.randn(ntraj, B, 2, L, L, generator=gen, dtype=torch.float64)
    us = torch.rand(ntraj, B, generator=gen, dtype=torch.float64)
    # HIP: one configuration at a time, as the notebook runs it
    fields = [x0[b:b + 1].cuda() for b in range(B)]
    qh = np.zeros((ntraj + 1, B)); acc_h = np.zeros((ntraj, B), dtype=bool)
    qh[0] = np.asarray(ops.wilson_action_charge(x0.cuda(), beta)[1].cpu())
    for k in range(ntraj):
        vk, uk = vs[k].cuda(), us[k].cuda()
        for b in range(B):
            _, _, acc, fields[b] = qed.ft_hmc(param, flow, fields[b], v=vk[b:b + 1], u=uk[b])
            acc_h[k, b] = bool(acc)
        qh[k + 1] = np.asarray(ops.wilson_action_charge(torch.cat(fields), beta)[1].cpu())
    # oracle: the chains as a batch of independent systems, same draws, inverse by bisection down to fp resolution
    xc = x0.clone()
    qc = np.zeros((ntraj + 1, B)); acc_c = np.zeros((ntraj, B), dtype=bool)
    qc[0] = R.charge(xc).numpy()
    for k in range(ntraj):
        with torch.no_grad():
            xi = R.flow_reverse(xc, wts, tol=1e-13)[0]
        _, _, acc, newx, _, _ = R.ft_hmc(xi, vs[k], us[k], wts, beta, tau / nstep, nstep, mode='md')
        with torch.no_grad():
            xc = R.flow_forward(newx, wts)[0]
        acc_c[k] = acc.numpy(); qc[k + 1] = R.charge(xc).numpy()
    assert np.abs(qh - np.round(qh)).max() < 1e-8
    qh, qc = np.round(qh), np.round(qc)
    assert np.array_equal(acc_h, acc_c) and 0.2 < acc_h.mean() <= 1.0
    assert np.array_equal(qh, qc) and np.abs(np.diff(qh, axis=0)).max() > 0        # the charge does move
    for q_h, q_c in [(qh[:, b], qc[:, b]) for b in range(B)] + [(qh, qc)]:
        rows_h = OB.change_sqr_vs_dt(q_h, dt_range=8, reference_literal=True)
        rows_c = OB.change_sqr_vs_dt(q_c, dt_range=8, reference_literal=True)
        np.testing.assert_array_equal(np.asarray(rows_h), np.asarray(rows_c))
    assert np.isfinite(np.asarray(OB.change_sqr_vs_dt(qh, dt_range=8))[:, 1]).all()


# ---------------------------------------------------------------- BASELINE configs[4]: the FULL batch on one GPU
def test_config5_full_batch_on_one_gpu():
    """configs[4] = 256 chains of L=256, beta=7, 16 layers (32 per GPU on 8 GPUs).  All 256 fit one MI355X (force workspace
    ~50 GB, training ~90 GB of the 288): size-independent properties on the full batch -- integer Q, forward o reverse = id,
    force = finite-difference gradient of S_eff, training gradient = finite-difference derivative of the loss -- and chains
    0 / 131 / 255 bit-equal to the same chain inside its 32-chain shard.  Workspaces are released afterwards."""
    gen = torch.Generator().manual_seed(256)
    B, L, nl, beta = 256, 256, 16, 7.0
    flow = R.default_flow(nl, gen)
    w = ops.pack_weights(flow, device='cuda')
    try:
        x = ops.random_uniform(torch.arange(B, dtype=torch.int64, device='cuda') + 77, (B, 2, L, L), -math.pi, math.pi)
        y, ld = ops.flow_forward(x, w, nl)
        Q = ops.wilson_action_charge(y, beta)[1]
        assert float((Q - Q.round()).abs().max()) < 1e-6
        xb, ldb = ops.flow_reverse(y, w, nl, tol=1e-13)
        d = (xb - x + math.pi) % (2 * math.pi) - math.pi
        assert float(d.abs().max()) < 1e-8 and float((ldb + ld).abs().max()) < 1e-5
        del xb, ldb, d
        F = ops.ft_force(x, w, nl, beta)
        dirn = ops.random_momenta(torch.arange(B, dtype=torch.int64, device='cuda') + 5, (B, 2, L, L), need_u=False)[0]
        eps = 1e-5
        fd = (ops.ft_action(x + eps * dirn, w, nl, beta)[0] - ops.ft_action(x - eps * dirn, w, nl, beta)[0]) / (2 * eps)
        close((F * dirn).flatten(1).sum(1), fd, rtol=5e-5, atol=2e-3)
        # shards: chains 0, 131, 255 inside their 32-chain blocks
        for c in (0, 131, 255):
            lo = c // 32 * 32
            Fs = ops.ft_force(x[lo:lo + 32].contiguous(), w, nl, beta)
            assert torch.equal(Fs[c - lo], F[c])
            ys = ops.flow_forward(x[lo:lo + 32].contiguous(), w, nl)[0]
            assert torch.equal(ys[c - lo], y[c])
        del F, dirn, fd, y
        # training gradient on the full batch: directional finite difference of the loss in weight space
        r = ops.train_grad(x, w, nl, beta, groups=1)
        loss = lambda ww: float((lambda t: (t['logq'] - t['logp']).mean())(ops.train_grad(x, ww, nl, beta, need_gw=False)))
        dw = torch.randn(w.numel(), generator=gen, dtype=torch.float64).cuda()
        dw = dw / dw.norm()
        epsw = 1e-6
        fdw = (loss(w + epsw * dw) - loss(w - epsw * dw)) / (2 * epsw)
        an = float((r['gw'] * dw).sum())
        assert abs(an - fdw) < 1e-4 * max(1.0, abs(an)), (an, fdw)
        rs = ops.train_grad(x[128:160].contiguous(), w, nl, beta, groups=1, need_gw=False)
        assert torch.equal(rs['logq'][3], r['logq'][131]) and torch.equal(rs['logp'][3], r['logp'][131])
    finally:
        ops.release_workspaces()
        torch.cuda.empty_cache()


def test_plain_hmc_outputs_in_place():
    """ops.hmc_trajectory / ops.wilson_action_charge write into caller-supplied tensors (`out=`: the bench loop's captured sequence
    carries no copy launches behind them) and return the same numbers as the allocating form; x_new must not alias x."""
    import math
    from fthmc_amd import ops
    from fthmc_amd.ops import FthmcError
    gen = torch.Generator().manual_seed(5)
    B, L = 3, 8
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    v = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64).cuda()
    u = torch.rand(B, generator=gen, dtype=torch.float64).cuda()
    ref = ops.hmc_trajectory(x, v, u, 2.0, 0.1, 10)
    out = {'x_new': torch.empty_like(x)}
    for k in ('dH', 'acc', 'H0', 'H1', 'plaq', 'Q'):
        out[k] = torch.full((B,), float('nan'), dtype=torch.float64, device='cuda')
    r = ops.hmc_trajectory(x, v, u, 2.0, 0.1, 10, out=out)
    for k in ('x_new', 'dH', 'acc', 'H0', 'H1'):
        assert r[k].data_ptr() == out[k].data_ptr() and torch.equal(out[k], ref[k]), k
    S, Q, plaq = ops.wilson_action_charge(out['x_new'], 2.0)
    S2, Q2, plaq2 = ops.wilson_action_charge(out['x_new'], 2.0, out=out)
    assert Q2.data_ptr() == out['Q'].data_ptr() and plaq2.data_ptr() == out['plaq'].data_ptr()
    assert torch.equal(S, S2) and torch.equal(Q, out['Q']) and torch.equal(plaq, out['plaq'])
    with pytest.raises(FthmcError):
        ops.hmc_trajectory(x, v, u, 2.0, 0.1, 10, out={'x_new': x})
    with pytest.raises(FthmcError):
        ops.hmc_trajectory(x, v, u, 2.0, 0.1, 10, out={'dH': torch.empty(B + 1, dtype=torch.float64, device='cuda')})


@pytest.mark.parametrize('L,nl,B', [(8, 2, 6), (32, 2, 4)])
def test_reference_shaped_run_carries_state_and_observables(L, nl, B):
    """FieldTransformation.run(batch=True) (ft_hmc.py:272-346): the trajectory hands (S_eff, plaq, Q) of the accepted field to the
    next one and its plaq / Q to the history -- no H0 sweep, no extra flow sweep for the metrics -- and the history equals the one of
    the loop that recomputes both, as the reference does (ft_hmc.py:205, 266-270), on the same momenta and uniforms."""
    from fthmc_amd import train as T
    from fthmc_amd.config import TrainConfig, lfConfig
    from fthmc_amd.ft_hmc import FieldTransformation
    cfg = TrainConfig(L=L, beta=2.0, n_layers=nl, batch_size=B, print_freq=0)
    torch.manual_seed(11)
    model = T.get_model(cfg)
    x0 = (0.3 * (2 * torch.rand(B, 2, L, L, dtype=torch.float64) - 1)).cuda()
    n = 4
    ft = FieldTransformation(flow=model.layers, config=cfg, lfconfig=lfConfig(tau=1.0, nstep=8))
    torch.manual_seed(5); torch.cuda.manual_seed(5)
    h = ft.run(x0.clone(), nprint=0, num_trajs=n, batch=True)
    assert ft._carry is not None and ft._carry[0] is ft.x_last
    ft2 = FieldTransformation(flow=model.layers, config=cfg, lfconfig=lfConfig(tau=1.0, nstep=8))
    torch.manual_seed(5); torch.cuda.manual_seed(5)
    x = x0.clone()
    qold = ft2.lattice_metrics(ft2.flow_forward(x)[0], torch.zeros(B, dtype=torch.float64, device='cuda'))['q']
    for i in range(n):
        x, m = ft2._batch_hmc(x.clone(), step=i)                         # a copy: nothing is carried over
        lm = ft2.lattice_metrics(ft2.flow_forward(x)[0], qold)
        assert torch.equal(m['acc'], h['acc'][i])
        assert torch.allclose(m['dh'], h['dh'][i], rtol=0, atol=1e-9)
        assert torch.allclose(lm['plaq'], h['plaq'][i], rtol=1e-13, atol=0)
        assert torch.allclose(lm['q'], h['q'][i], rtol=0, atol=1e-11)
        assert torch.allclose(lm['dq'], h['dq'][i], rtol=0, atol=1e-11)
        qold = lm['q']
    assert torch.equal(x, ft.x_last)


def test_reference_shaped_single_chain_run_carries_state():
    """FieldTransformation.run() on the reference's [1, 2, L, L] field: same carry as the batch loop; the history equals the loop
    that recomputes H0 and flows the field again for its metrics."""
    from fthmc_amd import train as T
    from fthmc_amd.config import TrainConfig, lfConfig
    from fthmc_amd.ft_hmc import FieldTransformation
    L, nl, n = 8, 2, 5
    cfg = TrainConfig(L=L, beta=2.0, n_layers=nl, batch_size=1, print_freq=0)
    torch.manual_seed(13)
    model = T.get_model(cfg)
    x0 = (0.3 * (2 * torch.rand(1, 2, L, L, dtype=torch.float64) - 1)).cuda()
    ft = FieldTransformation(flow=model.layers, config=cfg, lfconfig=lfConfig(tau=1.0, nstep=8))
    torch.manual_seed(6); torch.cuda.manual_seed(6)
    h = ft.run(x0.clone(), nprint=0, num_trajs=n)
    ft2 = FieldTransformation(flow=model.layers, config=cfg, lfconfig=lfConfig(tau=1.0, nstep=8))
    torch.manual_seed(6); torch.cuda.manual_seed(6)
    x = x0.clone()
    for i in range(n):
        x, m = ft2.hmc(x.clone(), step=i)
        lm = ft2.lattice_metrics(ft2.flow_forward(x)[0], torch.zeros(1, dtype=torch.float64, device='cuda'))
        assert bool(m['acc']) == bool(h['acc'][i])
        assert torch.allclose(m['dh'], h['dh'][i], rtol=0, atol=1e-9)
        assert torch.allclose(lm['plaq'], h['plaq'][i], rtol=1e-13, atol=0) and torch.allclose(lm['q'], h['q'][i], rtol=0, atol=1e-11)
    assert torch.equal(x, ft.x_last)


def test_headline_size_sampler_reproduces_the_exact_plaquette():
    """L = 64 (the tiled MFMA kernels, two chain groups), beta = 6: plain HMC and ftHMC with an untrained 2-layer flow from the
    near-cold start; <cos P> of the (flowed) field against I1/I0 and <exp(-dH)> against 1, errors over independent chains
    (tools/operating_point.py is the long version: profiles/r04_headline_size_plaquette.json).  Seeds are fixed: deterministic."""
    import math
    from fthmc_amd import ops, parallel, train as T
    from fthmc_amd.config import PLAQ_EXACT, TrainConfig
    from fthmc_amd.utils.layers import net_weights
    L, beta, B, nl, therm, ntraj, nstep = 64, 6.0, 128, 2, 250, 250, 20
    dev = torch.device('cuda', 0)
    torch.manual_seed(3)
    model = T.get_model(TrainConfig(L=L, beta=beta, n_layers=nl, batch_size=B, print_freq=0))
    w = ops.pack_weights([net_weights(l.plaq_coupling.net) for l in model.layers], device=dev)
    for flowed in (False, True):
        g0, _ = ops.random_momenta(parallel.chain_seeds(77, 0, B, 0).to(dev), (B, 2, L, L), need_u=False)
        x = (0.1 * torch.erf(g0 / math.sqrt(2.0))).contiguous()
        plaq_s = torch.zeros(B, dtype=torch.float64, device=dev); emdh_s = torch.zeros_like(plaq_s); acc_s = torch.zeros_like(plaq_s)
        for it in range(therm + ntraj):
            v, u = ops.random_momenta(parallel.chain_seeds(78, 0, B, it).to(dev), (B, 2, L, L))
            if flowed:
                r = ops.ft_trajectory(x, v, u, w, nl, beta, 1.0 / nstep, nstep, mode='md', groups=2)
                plaq = r['plaq']
            else:
                r = ops.hmc_trajectory(x, v, u, beta, 1.0 / nstep, nstep)
                plaq = ops.wilson_action_charge(r['x_new'], beta)[2]
            x = r['x_new']
            if it >= therm:
                plaq_s += plaq; emdh_s += torch.exp(-r['dH']); acc_s += r['acc']
        p = (plaq_s / ntraj).cpu().numpy(); e = (emdh_s / ntraj).cpu().numpy()
        perr, eerr = p.std(ddof=1) / math.sqrt(B), e.std(ddof=1) / math.sqrt(B)
        assert float(acc_s.mean()) / ntraj > 0.6
        assert perr < 5e-5 and abs(p.mean() - PLAQ_EXACT[beta]) < 4 * perr, (flowed, p.mean(), perr)
        assert abs(e.mean() - 1.0) < 4 * eerr, (flowed, e.mean(), eerr)
