"""GPU tests of the reference-shaped Python API (fthmc_amd.ft_hmc / hmc / train / utils.*)
against golden vectors from the reference and against the oracle."""
import math

import numpy as np
import pytest
import torch

from conftest import golden_flow, load_golden

pytestmark = pytest.mark.gpu


def D(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64).copy()).cuda()


def H(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def close(a, b, rtol=1e-9, atol=1e-9):
    np.testing.assert_allclose(H(a), H(b), rtol=rtol, atol=atol)


def build_layers(g, L, act='silu'):
    """nn layers with the golden weights loaded through the reference state_dict key layout."""
    from fthmc_amd.utils import layers as Lyr
    nl = int(g['n_layers'])
    flow = Lyr.make_u1_equiv_layers(n_layers=nl, n_mixture_comps=2, lattice_shape=(L, L), hidden_sizes=[8, 8],
                                    kernel_size=3, activation_fn=act)
    names = ['net.0.weight', 'net.0.bias', 'net.2.weight', 'net.2.bias', 'net.4.weight', 'net.4.bias']
    sd = {f'{li}.plaq_coupling.{n}': D(g[f'w{li}_{pi}']) for li in range(nl) for pi, n in enumerate(names)}
    flow.load_state_dict(sd)                      # same keys as the reference's checkpoints
    return flow


def test_qed_helpers_plain():
    from fthmc_amd.config import Param
    from fthmc_amd.utils import qed_helpers as qed
    g = load_golden('obs_L16')
    x, beta = D(g['x']), float(g['beta'])
    close(qed.BatchAction(beta)(x), g['S'], rtol=1e-12)
    close(qed.batch_plaqs(x), g['plaqs'], atol=1e-13)
    close(qed.batch_charges(x), g['Q'], atol=1e-9)
    close(qed.batch_charges(plaqs=qed.torch_wrap(qed.batch_plaqs(x))), g['Q'], atol=1e-9)
    close(qed.topo_charge(x), g['topo'], atol=1e-9)
    close(qed.regularize(x), g['regularize'], atol=1e-12)
    g = load_golden('hmc_L8_n10')
    param = Param(beta=float(g['beta']), L=8, tau=float(g['dt']) * int(g['nstep']), nstep=int(g['nstep']))
    x = D(g['x'])
    close(qed.force(param, x), g['force'], atol=1e-13)
    x_, p_ = qed.leapfrog(param, x, D(g['p']), verbose=False)
    close(x_, g['lf_x'], atol=1e-11); close(p_, g['lf_p'], atol=1e-11)
    dH, e, acc, newx = qed.hmc(param, x, verbose=False, v=D(g['v']), u=D(g['u']))
    close(dH, g['dH'], rtol=1e-8, atol=1e-10); close(e, g['exp_mdH'], rtol=1e-8)
    assert bool(acc) == bool(g['acc'])
    close(newx, g['newx'], atol=1e-10)
    # joint (one system) semantics for a batch: oracle
    from oracle import ref_cpu as R
    gen = torch.Generator().manual_seed(3)
    xb = (torch.rand(3, 2, 8, 8, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi
    vb = torch.randn(3, 2, 8, 8, generator=gen, dtype=torch.float64)
    ub = torch.rand([], generator=gen, dtype=torch.float64)
    dH_c, _, acc_c, newx_c = R.hmc(xb, vb, ub, param.beta, param.dt, param.nstep, joint=True)
    dH_g, _, acc_g, newx_g = qed.hmc(param, xb.cuda(), verbose=False, v=vb.cuda(), u=ub.cuda())
    close(dH_g, dH_c, rtol=1e-8, atol=1e-9); assert bool(acc_g) == bool(acc_c); close(newx_g, newx_c, atol=1e-9)


def test_layers_module_api():
    from fthmc_amd.utils import layers as Lyr
    from fthmc_amd.utils import qed_helpers as qed
    from fthmc_amd.config import Param
    g = load_golden('ft_L8_n8')
    flow = build_layers(g, 8)
    assert sorted(flow.state_dict().keys())[0] == '0.plaq_coupling.net.0.bias'
    x, beta = D(g['x']), float(g['beta'])
    param = Param(beta=beta, L=8)
    # layer by layer through autograd: forward values and the force (= ft_force by autograd)
    xg = x.clone().requires_grad_(True)
    y, logdet = xg, 0.
    for layer in flow:
        y, lj = layer.forward(y)
        logdet = logdet + lj
    close(y, g['y'], atol=1e-11); close(logdet, g['logdet'], atol=1e-11)
    s = (qed.BatchAction(beta)(y.detach()) - logdet.detach())
    close(s, g['S_eff'], rtol=1e-11)
    close(qed.ft_action(param, flow, x), g['S_eff'], rtol=1e-11)
    close(qed.ft_force(param, flow, x), g['ft_force'], rtol=1e-8, atol=1e-10)
    close(qed.ft_flow(flow, x), g['y'], atol=1e-11)
    xb = qed.ft_flow_inv(flow, D(g['y']))
    d = (H(xb) - g['x'] + np.pi) % (2 * np.pi) - np.pi
    assert np.abs(d).max() < 1e-8
    # reverse of a single layer + get_nets / transfer to a larger lattice
    xr, ljr = flow[3].reverse(flow[3].forward(x)[0].detach())
    assert np.abs((H(xr) - H(x) + np.pi) % (2 * np.pi) - np.pi).max() < 1e-9
    nets = Lyr.get_nets(flow)
    big = Lyr.make_net_from_layers(lattice_shape=(16, 16), nets=nets)
    xb16 = (torch.rand(2, 2, 16, 16, dtype=torch.float64, device='cuda') * 2 - 1) * math.pi
    yb, ljb = big[0].forward(xb16)
    assert yb.shape == xb16.shape and ljb.shape == (2,)


def test_autograd_layer_vjp_and_wgrad():
    g = load_golden('layers_L8_silu')
    flow = build_layers(g, 8)
    for li, layer in enumerate(flow):
        x = D(g[f'x{li}']).requires_grad_(True)
        y, lj = layer(x)
        obj = (D(g[f'c{li}']) * y).sum() + (D(g[f'd{li}']) * lj).sum()
        params = [p for p in layer.parameters()]
        grads = torch.autograd.grad(obj, [x] + params)
        close(grads[0], g[f'gx{li}'], rtol=1e-9, atol=1e-11)
        for pi, gw in enumerate(grads[1:]):
            close(gw, g[f'gw{li}_{pi}'], rtol=1e-9, atol=1e-11)


def test_field_transformation_hmc_literal_and_md():
    from fthmc_amd.config import TrainConfig, lfConfig
    from fthmc_amd.ft_hmc import FieldTransformation
    g = load_golden('traj_literal_L8')
    flow = build_layers(g, 8)
    cfg = TrainConfig(L=8, beta=float(g['beta']), n_layers=len(flow))
    ft = FieldTransformation(flow, cfg, lfConfig(tau=float(g['dt']) * 10, nstep=10), leapfrog_mode='reference_literal')
    xnew, m = ft.hmc(D(g['x']), v=D(g['v']), u=D(g['u']))
    close(m['dh'], g['dH'], rtol=1e-8, atol=1e-9)
    assert bool(m['acc']) == bool(g['acc'])
    close(xnew, g['newx'], atol=1e-11)
    yphys, _ = ft.flow_forward(xnew)
    lm = ft.lattice_metrics(yphys, torch.zeros(1, dtype=torch.float64, device='cuda'))
    close(lm['plaq'], g['plaq'], rtol=1e-9); close(lm['q'], g['Q'], atol=1e-8)
    # intended integrator, batch of chains
    g = load_golden('traj_md_L16')
    flow = build_layers(g, 16)
    cfg = TrainConfig(L=16, beta=float(g['beta']), n_layers=len(flow))
    ft = FieldTransformation(flow, cfg, lfConfig(tau=float(g['dt']) * int(g['nstep']), nstep=int(g['nstep'])))
    xnew, m = ft._batch_hmc(D(g['x']), v=D(g['v']), u=D(g['u']))
    close(m['dh'], g['dH'], rtol=1e-6, atol=1e-6)
    assert np.array_equal(H(m['acc']) > 0.5, g['acc'])
    close(ft.action(D(g['x'])) + 0.5 * (D(g['v']) ** 2).flatten(1).sum(1), g['H0'], rtol=1e-10)
    close(ft.force(D(g['x'])).shape, D(g['x']).shape)
    x_, v_ = ft.leapfrog(D(g['x']), D(g['v']))
    close(x_, g['lf_x'], rtol=1e-7, atol=1e-7)
    # short run loops work and keep their metric keys
    hist = ft.run(x=D(g['x'])[:1], num_trajs=3, nprint=0)
    assert set(hist) >= {'dt', 'acc', 'dh', 'plaq', 'q', 'dq'} and len(hist['acc']) == 3
    hist = ft.run(x=D(g['x']), num_trajs=2, nprint=0, batch=True)
    assert hist['acc'][0].shape == (4,)


@pytest.mark.parametrize('name,L', [('train_L8', 8), ('train_L16', 16)])
@pytest.mark.parametrize('fused', [True, False])
def test_train_step_matches_reference(name, L, fused):
    from fthmc_amd.config import FlowModel, TrainConfig
    from fthmc_amd.train import train_step
    from fthmc_amd.utils import qed_helpers as qed
    from fthmc_amd.utils.distributions import MultivariateUniform
    g = load_golden(name)
    flow = build_layers(g, L)
    beta = float(g['beta'])
    B = g['xi'].shape[0]
    cfg = TrainConfig(L=L, beta=beta, n_layers=len(flow), batch_size=B, base_lr=float(g['lr']))
    prior = MultivariateUniform(-math.pi * torch.ones(2, L, L, dtype=torch.float64, device='cuda'),
                                math.pi * torch.ones(L, L, dtype=torch.float64, device='cuda'))
    model = FlowModel(prior=prior, layers=flow)
    opt = torch.optim.Adam(flow.parameters(), lr=cfg.base_lr)
    m = train_step(model, cfg, qed.BatchAction(beta), opt, B, xi=D(g['xi']), fused=fused)
    close(m['loss_dkl'], g['loss_dkl'], rtol=1e-10); close(m['ess'], g['ess'], rtol=1e-8)
    close(m['logp'], g['logp'], rtol=1e-10); close(m['logq'], g['logq'], rtol=1e-10)
    close(m['q'], g['q'], atol=1e-8); close(m['dq'], g['dq'], atol=1e-8); close(m['plaq'], g['plaq'], rtol=1e-10)
    names = ['net.0.weight', 'net.0.bias', 'net.2.weight', 'net.2.bias', 'net.4.weight', 'net.4.bias']
    sd = flow.state_dict()
    for li in range(len(flow)):
        for pi, n in enumerate(names):
            p = dict(flow.named_parameters())[f'{li}.plaq_coupling.{n}']
            close(p.grad, g[f'gw{li}_{pi}'], rtol=1e-8, atol=1e-12)
            # Adam's first step is lr * sign(g) up to eps: compare where the gradient is not tiny
            ref, got = g[f'w_after{li}_{pi}'], H(sd[f'{li}.plaq_coupling.{n}'])
            mask = np.abs(g[f'gw{li}_{pi}']) > 1e-6
            np.testing.assert_allclose(got[mask], ref[mask], rtol=0, atol=1e-9)


def test_train_loop_improves_and_transfers():
    from fthmc_amd.config import TrainConfig
    from fthmc_amd.train import train, transfer_to_new_lattice
    torch.manual_seed(5)
    cfg = TrainConfig(L=8, beta=1.0, n_layers=4, batch_size=64, n_era=1, n_epoch=25, base_lr=5e-3, print_freq=0)
    out = train(cfg, verbose=False)
    loss = np.array([float(v) for v in out['history']['loss_dkl']])
    assert loss[-5:].mean() < loss[:5].mean()
    model16 = transfer_to_new_lattice(16, out['model'].layers)
    xi = model16.prior.sample_n(4)
    y, lj = model16.layers[0].forward(xi)
    assert y.shape == (4, 2, 16, 16)


def test_run_hmc_statistics():
    """config 1: L=8, beta=2.0 plain HMC, 1 chain.  <plaq> must approach I1/I0 (PLAQ_EXACT)."""
    from fthmc_amd.config import PLAQ_EXACT, Param
    from fthmc_amd.hmc import run_hmc
    torch.manual_seed(1331)
    param = Param(beta=2.0, L=8, tau=1.0, nstep=10, ntraj=400, nrun=1, nprint=0)
    fields, hist = run_hmc(param)
    h = hist[0]
    plaq = np.array([float(p) for p in h['plaq']])[100:]
    acc = np.array([float(a) for a in h['acc']])
    assert acc.mean() > 0.6
    assert abs(plaq.mean() - PLAQ_EXACT[2.0]) < 0.03
    assert all(abs(float(q) - round(float(q))) < 1e-6 for q in h['q'])


def test_independence_sampler():
    """samplers.make_mcmc_ensemble: serial accept chain over GPU-generated proposals."""
    from fthmc_amd.config import TrainConfig
    from fthmc_amd.train import get_model
    from fthmc_amd.utils import qed_helpers as qed
    from fthmc_amd.utils.samplers import generate_ensemble, make_mcmc_ensemble
    torch.manual_seed(3)
    cfg = TrainConfig(L=8, beta=1.0, n_layers=4, batch_size=16)
    model = get_model(cfg)
    action = qed.BatchAction(cfg.beta)
    h = make_mcmc_ensemble(model, action, 16, 80, keep_x=True)
    assert h['acc'][0] == 1.0 and 0.0 < h['acc'].mean() <= 1.0
    assert all(len(h[k]) == 80 for k in ('q', 'dqsq', 'logq', 'logp', 'acc'))
    assert np.abs(h['q'] - np.round(h['q'])).max() < 1e-6
    # the recorded logp / q belong to the configuration the chain sits on
    close(-action(h['x']), h['logp'], rtol=1e-12)
    close(qed.batch_charges(h['x']), h['q'], atol=1e-8)
    # rejected steps repeat the previous state
    rej = np.where(h['acc'] == 0.0)[0]
    assert all(h['logp'][i] == h['logp'][i - 1] and h['dqsq'][i] == 0.0 for i in rej)
    out = generate_ensemble(model, action, ensemble_size=64, batch_size=16, nboot=10, binsize=8)
    assert np.isfinite(out['suscept_mean']) and out['suscept_err'] >= 0


def test_chain_groups_do_not_change_results():
    """ops.ft_trajectory(groups=G): chains split over concurrent streams give the same per-chain results."""
    from fthmc_amd import ops
    from oracle import ref_cpu as R
    gen = torch.Generator().manual_seed(99)
    B, L, nl, beta = 10, 16, 4, 3.0
    flow = R.default_flow(nl, gen)
    w = ops.pack_weights(flow, device='cuda')
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    v = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64).cuda()
    u = torch.rand(B, generator=gen, dtype=torch.float64).cuda()
    ref = ops.ft_trajectory(x, v, u, w, nl, beta, 0.1, 5)
    for G in (2, 3):
        r = ops.ft_trajectory(x, v, u, w, nl, beta, 0.1, 5, groups=G)
        torch.cuda.synchronize()
        for k in ('x_new', 'dH', 'acc', 'H0', 'H1', 'plaq', 'Q', 'state'):
            assert torch.equal(r[k], ref[k]), (G, k)
        # chained through the carried state
        r2 = ops.ft_trajectory(r['x_new'], v, u, w, nl, beta, 0.1, 5, state_in=r['state'], groups=G)
        ref2 = ops.ft_trajectory(ref['x_new'], v, u, w, nl, beta, 0.1, 5, state_in=ref['state'])
        torch.cuda.synchronize()
        assert torch.equal(r2['dH'], ref2['dH']) and torch.equal(r2['x_new'], ref2['x_new'])


def test_train_grad_chain_groups():
    """ops.train_grad(groups=G): same loss pieces, weight gradient equal up to summation order."""
    from fthmc_amd import ops
    from oracle import ref_cpu as R
    gen = torch.Generator().manual_seed(5)
    B, L, nl, beta = 10, 16, 3, 2.0
    w = ops.pack_weights(R.default_flow(nl, gen), device='cuda')
    xi = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    ref = ops.train_grad(xi, w, nl, beta)
    for G in (2, 3):
        r = ops.train_grad(xi, w, nl, beta, groups=G)
        torch.cuda.synchronize()
        for k in ('x', 'logq', 'logp'):
            assert torch.equal(r[k], ref[k]), (G, k)
        close(r['gw'], ref['gw'], rtol=1e-12, atol=1e-14)


def test_batch_hmc_and_train_step_take_the_grouped_path():
    """Shapes large enough for ops.default_groups() = 2: the Python API's batched trajectory and fused training
    step give the results of the ungrouped ops."""
    from fthmc_amd import ops
    from fthmc_amd.config import TrainConfig, lfConfig
    from fthmc_amd.ft_hmc import FieldTransformation
    from fthmc_amd.utils import layers as Lyr
    B, L, nl = 32, 64, 2
    assert ops.default_groups(B, L) == 2 and ops.default_groups(4, 16) == 1
    torch.manual_seed(11)
    flow = Lyr.make_u1_equiv_layers(n_layers=nl, n_mixture_comps=2, lattice_shape=(L, L), hidden_sizes=[8, 8],
                                    kernel_size=3, activation_fn='silu').cuda().double()
    cfg = TrainConfig(L=L, beta=3.0, n_layers=nl, batch_size=B)
    ft = FieldTransformation(flow, cfg, lfConfig(tau=0.3, nstep=3))
    gen = torch.Generator().manual_seed(3)
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    v = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64).cuda()
    u = torch.rand(B, generator=gen, dtype=torch.float64).cuda()
    xnew, m = ft._batch_hmc(x, v=v, u=u)
    ref = ops.ft_trajectory(x, v, u, ft.weights(x.device), nl, cfg.beta, ft.dt, ft.nstep, groups=1)
    torch.cuda.synchronize()
    assert torch.equal(m['dh'], ref['dH']) and torch.equal(xnew, ref['x_new'])
    w = Lyr.flow_weights(flow, x.device)
    g2, g1 = ops.train_grad(x, w, nl, cfg.beta, groups=2), ops.train_grad(x, w, nl, cfg.beta, groups=1)
    torch.cuda.synchronize()
    assert torch.equal(g2['logq'], g1['logq'])
    close(g2['gw'], g1['gw'], rtol=1e-12, atol=1e-14)
