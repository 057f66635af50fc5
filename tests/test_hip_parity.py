"""GPU parity: HIP path (through the C ABI) vs golden vectors from the reference
and vs the CPU oracle on seeded inputs.  Tolerances: fp64, 1e-6 relative is the
north-star bar; the per-function checks below are far tighter."""
import math

import numpy as np
import pytest
import torch

from conftest import golden_flow, load_golden

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


@pytest.fixture(params=[(1, 1), (0, 1), (1, 0)], ids=['mfma', 'valu', 'mfma-tiled'], autouse=True)
def variant(request, _mods):
    """Every parity test runs against both conv-kernel variants (C ABI: fthmc_set_variant), and the MFMA variant also
    with the small-lattice fused path switched off (fthmc_set_small_path: L = 8, 12, 16 then take the tiled kernels)."""
    ops.set_variant(request.param[0])
    ops.set_small_path(bool(request.param[1]))
    yield request.param[0]
    ops.set_variant(1)
    ops.set_small_path(True)


def D(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64).copy()).cuda()


def H(t):
    return t.detach().cpu().numpy()


def close(a, b, rtol=1e-10, atol=1e-10):
    a = H(a) if torch.is_tensor(a) else np.asarray(a)
    b = H(b) if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def angle_close(a, b, atol=1e-9):
    a = H(a) if torch.is_tensor(a) else np.asarray(a)
    b = H(b) if torch.is_tensor(b) else np.asarray(b)
    d = (a - b + np.pi) % (2 * np.pi) - np.pi
    assert np.max(np.abs(d)) < atol, np.max(np.abs(d))


def W(flow):
    return ops.pack_weights(flow, device='cuda')


# ---------------------------------------------------------------- Wilson part
@pytest.mark.parametrize('L', [8, 12, 16])
def test_observables_golden(L):
    g = load_golden(f'obs_L{L}')
    x, beta = D(g['x']), float(g['beta'])
    S, Q, plaq = ops.wilson_action_charge(x, beta)
    close(ops.plaquettes(x), g['plaqs'], rtol=0, atol=1e-13)
    close(S, g['S'], rtol=1e-12); close(Q, g['Q'], atol=1e-9); close(plaq, g['plaq'], rtol=1e-12)
    close(ops.regularize(x), g['regularize'], atol=1e-12)
    close(ops.wrap(x), g['wrap'], rtol=0, atol=1e-15)


def test_known_answer():
    g = load_golden('known_answer')
    x, beta = D(g['x']), float(g['beta'])
    S, Q, plaq = ops.wilson_action_charge(x, beta)
    close(S, g['S'], rtol=1e-13); close(Q, g['Q']); close(plaq, g['plaq'], rtol=1e-13)
    close(ops.wilson_force(x, beta), g['F'], atol=1e-13)
    flow = golden_flow(g)
    y0, lj0 = ops.flow_layer_fwd(x, W(flow[:1]), 0, 0)
    close(y0, g['y0'], atol=1e-12); close(lj0, g['logJ0'], atol=1e-12)
    y1, lj1 = ops.flow_layer_fwd(y0, W(flow[1:2]), 1, 0)
    close(y1, g['y1'], atol=1e-12); close(lj1, g['logJ1'], atol=1e-12)
    Se, ld, _, _ = ops.ft_action(x, W(flow), 2, beta)
    close(Se, g['S_eff'], rtol=1e-12)
    close(ops.ft_force(x, W(flow), 2, beta), g['ft_force'], atol=1e-11)


@pytest.mark.parametrize('name', ['hmc_L8_n10', 'hmc_L8_n1', 'hmc_L16_n5', 'hmc_zero_L8'])
def test_plain_hmc_golden(name):
    g = load_golden(name)
    x, beta, dt, nstep = D(g['x']), float(g['beta']), float(g['dt']), int(g['nstep'])
    if 'force' in g:
        close(ops.wilson_force(x, beta), g['force'], atol=1e-13)
        x_, p_ = ops.leapfrog(x, D(g['p']), beta, dt, nstep)
        close(x_, g['lf_x'], atol=1e-11); close(p_, g['lf_p'], atol=1e-11)
    r = ops.hmc_trajectory(x, D(g['v']), D(g['u']).reshape(1), beta, dt, nstep)
    close(r['dH'], np.atleast_1d(g['dH']), rtol=1e-8, atol=1e-10)
    assert bool(r['acc'][0] > 0.5) == bool(g['acc'])
    close(r['x_new'], g['newx'], atol=1e-10)


def test_plain_hmc_batch_vs_oracle():
    gen = torch.Generator().manual_seed(5)
    B, L, beta, dt, nstep = 6, 24, 3.0, 0.1, 7
    x = (torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi
    v = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64)
    u = torch.rand(B, generator=gen, dtype=torch.float64)
    dH, _, acc, newx = R.hmc(x, v, u, beta, dt, nstep, joint=False)
    r = ops.hmc_trajectory(x.cuda(), v.cuda(), u.cuda(), beta, dt, nstep)
    close(r['dH'], dH, rtol=1e-8, atol=1e-9)
    assert np.array_equal(H(r['acc']) > 0.5, acc.numpy())
    close(r['x_new'], newx, atol=1e-9)


# ---------------------------------------------------------------- coupling layers
@pytest.mark.parametrize('name', ['layers_L8_silu', 'layers_L12_silu', 'layers_L8_relu', 'layers_L8_leaky_relu'])
def test_layers_golden(name):
    g = load_golden(name)
    act = str(g['act'])
    flow = golden_flow(g)
    for li, w in enumerate(flow):
        mu, off = li % 2, (li // 2) % 4
        wl = W([w])
        x = D(g[f'x{li}'])
        y, lj = ops.flow_layer_fwd(x, wl, mu, off, act)
        close(y, g[f'y{li}'], atol=1e-12); close(lj, g[f'logJ{li}'], atol=1e-12)
        gx, gw = ops.flow_layer_bwd(x, wl, D(g[f'c{li}']), D(g[f'd{li}']), mu, off, act, need_gw=True)
        close(gx, g[f'gx{li}'], rtol=1e-9, atol=1e-11)
        ref_gw = np.concatenate([g[f'gw{li}_{pi}'].reshape(-1) for pi in range(6)])
        close(gw, ref_gw, rtol=1e-9, atol=1e-11)
        xr, ljr = ops.flow_layer_rev(D(g[f'y{li}']), wl, mu, off, act, tol=1e-13)
        angle_close(xr, g[f'x{li}'], atol=1e-9)             # exact inverse, not the 1e-6 bisection
        close(ljr, -g[f'logJ{li}'], atol=1e-8)
        angle_close(xr, g[f'rev_x{li}'], atol=5e-6)         # reference bisection tolerance
        close(ljr, g[f'rev_logJ{li}'], atol=5e-5)


@pytest.mark.parametrize('name', ['ft_L8_n2', 'ft_L8_n8', 'ft_L16_n4', 'ft_L8_n16'])
def test_ft_action_force_golden(name):
    g = load_golden(name)
    x, beta, flow = D(g['x']), float(g['beta']), golden_flow(g)
    nl = len(flow); w = W(flow)
    y, ld = ops.flow_forward(x, w, nl)
    close(y, g['y'], atol=1e-11); close(ld, g['logdet'], rtol=1e-11, atol=1e-11)
    Se, ld2, plaq, Q = ops.ft_action(x, w, nl, beta)
    close(Se, g['S_eff'], rtol=1e-11, atol=1e-11); close(Q, g['Q'], atol=1e-8)
    close(ops.ft_force(x, w, nl, beta), g['ft_force'], rtol=1e-8, atol=1e-10)
    xb, ldb = ops.flow_reverse(D(g['y']), w, nl, tol=1e-13)
    angle_close(xb, g['x'], atol=1e-8)
    close(ldb, -g['logdet'], atol=1e-7)


@pytest.mark.parametrize('name', ['traj_md_L8', 'traj_md_L16', 'traj_md_config2'])
def test_traj_md_golden(name):
    g = load_golden(name)
    flow = golden_flow(g); nl = len(flow)
    r = ops.ft_trajectory(D(g['x']), D(g['v']), D(g['u']), W(flow), nl, float(g['beta']), float(g['dt']),
                          int(g['nstep']), mode='md')
    close(r['H0'], g['H0'], rtol=1e-10); close(r['H1'], g['H1'], rtol=1e-7)
    close(r['dH'], g['dH'], rtol=1e-6, atol=1e-6)
    assert np.array_equal(H(r['acc']) > 0.5, g['acc'])
    angle_close(r['x_new'], g['newx'], atol=1e-6)
    close(r['plaq'], g['plaq'], rtol=1e-6); close(r['Q'], g['Q'], atol=1e-6)
    xo, vo = ops.ft_leapfrog(D(g['x']), D(g['v']), W(flow), nl, float(g['beta']), float(g['dt']), int(g['nstep']))
    close(xo, g['lf_x'], rtol=1e-7, atol=1e-7); close(vo, g['lf_p'], rtol=1e-7, atol=1e-7)


def test_chained_trajectories_match_stateless():
    """state_out -> state_in (skips the H0 flow sweep) must not change anything."""
    gen = torch.Generator().manual_seed(77)
    B, L, nl, beta, dt, nstep = 6, 16, 4, 4.0, 0.05, 4
    flow = R.default_flow(nl, gen)
    w = W(flow)
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    xa, xb, state = x.clone(), x.clone(), None
    for t in range(3):
        v = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64).cuda()
        u = torch.rand(B, generator=gen, dtype=torch.float64).cuda()
        ra = ops.ft_trajectory(xa, v, u, w, nl, beta, dt, nstep)
        rb = ops.ft_trajectory(xb, v, u, w, nl, beta, dt, nstep, state_in=state)
        for k in ('x_new', 'dH', 'acc', 'H0', 'H1', 'plaq', 'Q', 'state'):
            assert torch.equal(ra[k], rb[k]), k
        Se, _, plaq, Q = ops.ft_action(ra['x_new'], w, nl, beta)
        close(ra['state'][0], Se, rtol=1e-13); close(ra['state'][1], plaq, rtol=1e-13); close(ra['state'][2], Q, atol=1e-9)
        xa, xb, state = ra['x_new'].clone(), rb['x_new'].clone(), rb['state'].clone()
    assert float(ra['acc'].sum()) > 0          # some chains moved, so the carried state was exercised


def test_traj_literal_golden():
    g = load_golden('traj_literal_L8')
    flow = golden_flow(g); nl = len(flow)
    r = ops.ft_trajectory(D(g['x']), D(g['v']), D(g['u']).reshape(1), W(flow), nl, float(g['beta']),
                          float(g['dt']), int(g['nstep']), mode='literal')
    close(r['dH'], np.atleast_1d(g['dH']), rtol=1e-8, atol=1e-9)
    assert bool(r['acc'][0] > 0.5) == bool(g['acc'])
    close(r['x_new'], g['newx'], atol=1e-11)
    close(r['plaq'], np.atleast_1d(g['plaq']), rtol=1e-9); close(r['Q'], np.atleast_1d(g['Q']), atol=1e-8)


@pytest.mark.parametrize('name', ['train_L8', 'train_L16'])
def test_train_grad_golden(name):
    g = load_golden(name)
    flow = golden_flow(g); nl = len(flow)
    r = ops.train_grad(D(g['xi']), W(flow), nl, float(g['beta']))
    close(r['logp'], g['logp'], rtol=1e-11); close(r['logq'], g['logq'], rtol=1e-11)
    loss = (r['logq'] - r['logp']).mean()
    close(loss, g['loss_dkl'], rtol=1e-11)
    gws = ops.unpack_weight_grads(r['gw'], nl)
    for li in range(nl):
        for pi in range(6):
            close(gws[li][pi], g[f'gw{li}_{pi}'], rtol=1e-8, atol=1e-12)


# ---------------------------------------------------------------- oracle at bench-like sizes
@pytest.mark.parametrize('B,L,nl,beta', [(3, 64, 8, 6.0), (2, 32, 5, 5.0), (1, 20, 3, 2.0), (2, 12, 4, 3.0),
                                        (9, 8, 8, 2.0), (1, 24, 2, 4.0), (2, 40, 3, 2.5)])
def test_ft_vs_oracle_random(B, L, nl, beta):
    gen = torch.Generator().manual_seed(1331 + L)
    flow = R.default_flow(nl, gen)
    x = (torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi
    w = W(flow)
    y, ld = R.flow_forward(x, flow)
    yg, ldg = ops.flow_forward(x.cuda(), w, nl)
    close(yg, y, atol=1e-11); close(ldg, ld, rtol=1e-11, atol=1e-11)
    Se = R.ft_action(x, flow, beta)
    Seg, _, plaq, Q = ops.ft_action(x.cuda(), w, nl, beta)
    close(Seg, Se, rtol=1e-11)
    close(plaq, R.plaq_mean(y, beta), rtol=1e-11); close(Q, R.charge(y), atol=1e-8)
    F = R.ft_force(x, flow, beta)
    close(ops.ft_force(x.cuda(), w, nl, beta), F, rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize('B,L,nl,beta', [(3, 20, 3, 2.0), (2, 24, 2, 3.0), (5, 12, 4, 2.0), (2, 32, 3, 4.0), (1, 40, 2, 2.5)])
def test_train_grad_vs_oracle_random(B, L, nl, beta):
    """Reverse-KL gradient (train.py:162-228) on lattices that do not divide into tiles: ragged 8 x 16 and
    16 x 16 tiles, windows that wrap onto themselves, chains that do not fill a block group."""
    gen = torch.Generator().manual_seed(77 + L)
    flow = R.default_flow(nl, gen)
    xi = (torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi
    out, grads = R.train_grads(xi, flow, beta)
    r = ops.train_grad(xi.cuda(), W(flow), nl, beta)
    close(r['logq'], out['logq'], rtol=1e-11); close(r['logp'], out['logp'], rtol=1e-11)
    angle_close(r['x'], out['x'], atol=1e-10)
    gws = ops.unpack_weight_grads(r['gw'], nl)
    for li in range(nl):
        for pi in range(6):
            close(gws[li][pi], grads[li][pi], rtol=1e-8, atol=1e-11)


@pytest.mark.parametrize('B,L,nl,act,scale,beta', [(2, 4, 8, 'silu', 1.0, 2.0), (3, 16, 4, 'relu', 1.0, 3.0),
                                                  (2, 28, 3, 'leaky_relu', 1.0, 2.5), (2, 16, 4, 'silu', 4.0, 4.0),
                                                  (1, 36, 2, 'relu', 3.0, 2.0), (10, 8, 5, 'leaky_relu', 2.0, 2.0)])
def test_activations_and_extremes_vs_oracle(B, L, nl, act, scale, beta):
    """Every activation on lattices of every kind (the smallest one, ragged tiles, a partial block group), weights
    scaled up until the sigmoids saturate and the transform's slopes get steep, plaquettes pinned near +-pi:
    forward, log det, force and the training gradient against the oracle."""
    gen = torch.Generator().manual_seed(4242 + L + nl)
    flow = [tuple(t * scale for t in lw) for lw in R.default_flow(nl, gen)]
    x = (torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi
    x[0, 0, 0, :] = math.pi - 1e-9                       # plaquettes at the branch cut of the wrap / of tan(P/2)
    x[0, 1, :, 0] = -math.pi + 1e-9
    w = W(flow)
    y, ld = R.flow_forward(x, flow, act)
    yg, ldg = ops.flow_forward(x.cuda(), w, nl, act)
    angle_close(yg, y, atol=1e-9); close(ldg, ld, rtol=1e-9, atol=1e-9)
    F = R.ft_force(x, flow, beta, act)
    Fg = ops.ft_force(x.cuda(), w, nl, beta, act)
    close(Fg, F, rtol=1e-7, atol=1e-7 * float(F.abs().max()))
    out, grads = R.train_grads(x, flow, beta, act)
    r = ops.train_grad(x.cuda(), w, nl, beta, act)
    close(r['logq'], out['logq'], rtol=1e-9); close(r['logp'], out['logp'], rtol=1e-9)
    gws = ops.unpack_weight_grads(r['gw'], nl)
    gmax = max(float(g.abs().max()) for lg in grads for g in lg)
    for li in range(nl):
        for pi in range(6):
            close(gws[li][pi], grads[li][pi], rtol=1e-7, atol=1e-9 * max(gmax, 1.0))


def test_config5_shape_properties():
    """BASELINE config 5 shard shape (L=256, 16 layers; 2 chains here): size-independent properties of the
    force and training paths."""
    gen = torch.Generator().manual_seed(4242)
    B, L, nl, beta = 2, 256, 16, 7.0
    flow = R.default_flow(nl, gen)
    w = W(flow)
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    y, ld = ops.flow_forward(x, w, nl)
    xb, ldb = ops.flow_reverse(y, w, nl, tol=1e-13)
    angle_close(xb, x, atol=1e-8); close(ldb, -ld, atol=1e-6)
    Q = ops.wilson_action_charge(y, beta)[1]
    assert float((Q - Q.round()).abs().max()) < 1e-7
    F = ops.ft_force(x, w, nl, beta)
    d = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64).cuda()
    eps = 1e-5
    fd = (ops.ft_action(x + eps * d, w, nl, beta)[0] - ops.ft_action(x - eps * d, w, nl, beta)[0]) / (2 * eps)
    close((F * d).flatten(1).sum(1), fd, rtol=5e-5, atol=1e-3)
    assert torch.equal(F, ops.ft_force(x, w, nl, beta))
    # training gradient: directional derivative of the reverse-KL loss along a random weight direction
    r = ops.train_grad(x, w, nl, beta)
    dw = torch.randn(w.numel(), generator=gen, dtype=torch.float64).cuda() * 1e-2
    flat = w.reshape(-1)

    def loss(wv):
        t = ops.train_grad(x, wv.reshape(w.shape), nl, beta, need_gw=False)
        return (t['logq'] - t['logp']).mean()
    eps = 1e-4
    fdw = (loss(flat + eps * dw) - loss(flat - eps * dw)) / (2 * eps)
    close((r['gw'] * dw).sum(), fdw, rtol=1e-5, atol=1e-6)


def test_full_size_properties():
    """BASELINE config 3 shape (B=128, L=64, 8 layers): size-independent properties."""
    gen = torch.Generator().manual_seed(1331)
    B, L, nl, beta = 128, 64, 8, 6.0
    flow = R.default_flow(nl, gen)
    w = W(flow)
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    y, ld = ops.flow_forward(x, w, nl)
    # topological charge is an integer and the flow keeps links in [-pi, pi)
    S, Q, plaq = ops.wilson_action_charge(y, beta)
    assert float((Q - Q.round()).abs().max()) < 1e-8
    # forward o reverse == identity, logdets cancel
    xb, ldb = ops.flow_reverse(y, w, nl, tol=1e-13)
    angle_close(xb, x, atol=1e-8)
    close(ldb, -ld, atol=1e-6)
    # gauge invariance of S_eff (layers.py:177-185): x_mu += a - roll(a, -1, mu+1)
    a = (torch.rand(B, L, L, generator=gen, dtype=torch.float64) * 2 * math.pi).cuda()
    xg = x.clone()
    xg[:, 0] += a - torch.roll(a, -1, 1)
    xg[:, 1] += a - torch.roll(a, -1, 2)
    Se0 = ops.ft_action(x, w, nl, beta)[0]
    Se1 = ops.ft_action(xg, w, nl, beta)[0]
    close(Se1, Se0, rtol=1e-10)
    # force is the gradient of S_eff: directional finite difference
    F = ops.ft_force(x, w, nl, beta)
    d = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64).cuda()
    eps = 1e-5
    Sp = ops.ft_action(x + eps * d, w, nl, beta)[0]
    Sm = ops.ft_action(x - eps * d, w, nl, beta)[0]
    fd = (Sp - Sm) / (2 * eps)
    an = (F * d).flatten(1).sum(1)
    close(an, fd, rtol=2e-5, atol=1e-4)
    # leapfrog reversibility: flip momenta and integrate back
    v = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64).cuda()
    x1, v1 = ops.ft_leapfrog(x, v, w, nl, beta, 0.05, 4)
    x2, v2 = ops.ft_leapfrog(x1, -v1, w, nl, beta, 0.05, 4)
    close(x2, x, atol=1e-8); close(-v2, v, atol=1e-8)
    # determinism: same inputs, same bits
    F2 = ops.ft_force(x, w, nl, beta)
    assert torch.equal(F, F2)


def test_errors_are_loud():
    from fthmc_amd._lib import FthmcError
    x = torch.zeros(1, 2, 8, 8, dtype=torch.float64)
    with pytest.raises(FthmcError):
        ops.wilson_force(x, 1.0)                       # CPU tensor: no fallback
    with pytest.raises(FthmcError):
        ops.wilson_force(x.cuda().float(), 1.0)        # fp32
    with pytest.raises(FthmcError):
        ops.wilson_force(torch.zeros(1, 2, 6, 6, dtype=torch.float64).cuda(), 1.0)   # L % 4 != 0
