"""Pin oracle/ref_cpu.py against golden vectors produced by the real reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import golden_flow, load_golden
from oracle import ref_cpu as R

T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64).copy())


def _np(a):
    return a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)


def close(a, b, rtol=1e-12, atol=1e-12):
    np.testing.assert_allclose(_np(a), _np(b), rtol=rtol, atol=atol)


def angle_close(a, b, atol=1e-10):
    d = _np(a) - _np(b)
    d = (d + np.pi) % (2 * np.pi) - np.pi
    assert np.max(np.abs(d)) < atol, np.max(np.abs(d))


def test_known_answer_survey_values():
    g = load_golden('known_answer')
    # values quoted in SURVEY.md 8c (computed from the unmodified reference)
    close(g['S'], [-10.20200089828726, -5.485545465718607])
    close(g['Q'], [-3.0, -2.0])
    close(g['logJ0'], [-0.6108947523052225, -1.2325090461928587])
    close(g['S_eff'], [-14.791670146352471, -12.147020252280395])
    close(np.linalg.norm(g['ft_force']), 28.884497191072708)
    x, beta = T(g['x']), float(g['beta'])
    flow = golden_flow(g)
    close(R.action(x, beta), g['S'])
    close(R.charge(x), g['Q'])
    close(R.plaq_mean(x, beta), g['plaq'])
    close(R.wilson_force(x, beta), g['F'])
    close(R.wilson_force_analytic(x, beta), g['F'])
    y0, lj0 = R.layer_forward(x, flow[0], 0, 0)
    y1, lj1 = R.layer_forward(y0, flow[1], 1, 0)
    close(y0, g['y0']); close(lj0, g['logJ0']); close(y1, g['y1']); close(lj1, g['logJ1'])
    close(R.ft_action(x, flow, beta), g['S_eff'])
    close(R.ft_force(x, flow, beta), g['ft_force'])


@pytest.mark.parametrize('L', [8, 12, 16])
def test_observables(L):
    g = load_golden(f'obs_L{L}')
    x, beta = T(g['x']), float(g['beta'])
    close(R.plaq(x), g['plaqs'])
    close(R.action(x, beta), g['S'])
    close(R.charge(x), g['Q'], atol=1e-10)
    close(R.charge(x), g['topo'], atol=1e-10)
    close(R.plaq_mean(x, beta), g['plaq'])
    close(R.regularize(x), g['regularize'])
    close(R.wrap(x), g['wrap'])
    close(R.wrap(x), g['layers_mod'])


@pytest.mark.parametrize('name', ['hmc_L8_n10', 'hmc_L8_n1', 'hmc_L16_n5'])
def test_plain_hmc(name):
    g = load_golden(name)
    x, p, beta, dt, nstep = T(g['x']), T(g['p']), float(g['beta']), float(g['dt']), int(g['nstep'])
    close(R.wilson_force(x, beta), g['force'])
    close(R.wilson_force_analytic(x, beta), g['force'])
    x_, p_ = R.leapfrog(x, p, lambda y: R.wilson_force(y, beta), dt, nstep)
    close(x_, g['lf_x'], rtol=1e-10, atol=1e-10); close(p_, g['lf_p'], rtol=1e-10, atol=1e-10)
    dH, e, acc, newx = R.hmc(x, T(g['v']), T(g['u']), beta, dt, nstep, joint=True)
    close(dH, g['dH'], rtol=1e-9, atol=1e-9)
    close(e, g['exp_mdH'], rtol=1e-9, atol=1e-9)
    assert bool(acc) == bool(g['acc'])
    close(newx, g['newx'], rtol=1e-9, atol=1e-9)
    # per-chain mode coincides with joint mode for B=1
    dH2, _, acc2, newx2 = R.hmc(x, T(g['v']), T(g['u']).reshape(1), beta, dt, nstep, joint=False)
    close(dH2[0], g['dH'], rtol=1e-9, atol=1e-9)
    close(newx2, g['newx'], rtol=1e-9, atol=1e-9)


def test_plain_hmc_zero_start():
    g = load_golden('hmc_zero_L8')
    dH, e, acc, newx = R.hmc(T(g['x']), T(g['v']), T(g['u']), float(g['beta']),
                             float(g['dt']), int(g['nstep']), joint=True)
    close(dH, g['dH'], rtol=1e-9, atol=1e-9)
    assert bool(acc) == bool(g['acc'])
    close(newx, g['newx'], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize('name', ['layers_L8_silu', 'layers_L12_silu', 'layers_L8_relu', 'layers_L8_leaky_relu'])
def test_single_layers_fwd_vjp_wgrad_reverse(name):
    g = load_golden(name)
    act = str(g['act'])
    flow = golden_flow(g)
    for li, w in enumerate(flow):
        mu, off = R.layer_mu_off(li)
        x = T(g[f'x{li}']).requires_grad_(True)
        leaves = [t.clone().requires_grad_(True) for t in w]
        y, lj = R.layer_forward(x, leaves, mu, off, act)
        close(y, g[f'y{li}'].astype(np.float64)); close(lj, g[f'logJ{li}'])
        obj = (T(g[f'c{li}']) * y).sum() + (T(g[f'd{li}']) * lj).sum()
        grads = torch.autograd.grad(obj, [x] + leaves)
        close(grads[0], g[f'gx{li}'], rtol=1e-10, atol=1e-11)
        for pi, gw in enumerate(grads[1:]):
            close(gw, g[f'gw{li}_{pi}'], rtol=1e-10, atol=1e-11)
        with torch.no_grad():
            xr, ljr = R.layer_reverse(T(g[f'y{li}']), w, mu, off, act)
        close(xr, g[f'rev_x{li}'], rtol=1e-9, atol=1e-9)
        close(ljr, g[f'rev_logJ{li}'], rtol=1e-9, atol=1e-9)
        angle_close(xr, g[f'x{li}'], atol=5e-6)       # bisection tolerance


@pytest.mark.parametrize('name', ['ft_L8_n2', 'ft_L8_n8', 'ft_L16_n4', 'ft_L8_n16'])
def test_ft_action_force(name):
    g = load_golden(name)
    x, beta, flow = T(g['x']), float(g['beta']), golden_flow(g)
    y, logdet = R.flow_forward(x, flow)
    close(y, g['y']); close(logdet, g['logdet'], rtol=1e-11)
    close(R.ft_action(x, flow, beta), g['S_eff'], rtol=1e-11)
    close(R.ft_force(x, flow, beta), g['ft_force'], rtol=1e-9, atol=1e-10)
    close(R.charge(y), g['Q'], atol=1e-9)
    if x.shape[0] <= 2:
        with torch.no_grad():
            xb, ldb = R.flow_reverse(T(g['y']), flow)
        close(xb, g['rev_x'], rtol=1e-8, atol=1e-8)
        close(ldb, g['rev_logdet'], rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize('name', ['traj_md_L8', 'traj_md_L16', 'traj_md_config2'])
def test_traj_md(name):
    g = load_golden(name)
    flow = golden_flow(g)
    beta = float(g['beta'])
    dH, e, acc, newx, h0, h1 = R.ft_hmc(T(g['x']), T(g['v']), T(g['u']), flow, beta,
                                        float(g['dt']), int(g['nstep']), mode='md')
    close(h0, g['H0'], rtol=1e-10); close(h1, g['H1'], rtol=1e-8)
    close(dH, g['dH'], rtol=1e-6, atol=1e-7)
    assert np.array_equal(np.asarray(acc), g['acc'])
    close(newx, g['newx'], rtol=1e-7, atol=1e-7)
    y, _ = R.flow_forward(newx, flow)
    close(R.plaq_mean(y, beta), g['plaq'], rtol=1e-8)
    close(R.charge(y), g['Q'], atol=1e-8)


def test_traj_literal():
    g = load_golden('traj_literal_L8')
    flow = golden_flow(g)
    dH, e, acc, newx, h0, h1 = R.ft_hmc(T(g['x']), T(g['v']), T(g['u']), flow, float(g['beta']),
                                        float(g['dt']), int(g['nstep']), mode='literal', joint=True)
    close(dH, g['dH'], rtol=1e-9, atol=1e-9)
    assert bool(acc) == bool(g['acc'])
    close(newx, g['newx'])


@pytest.mark.parametrize('name', ['train_L8', 'train_L16'])
def test_train_step(name):
    g = load_golden(name)
    flow = golden_flow(g)
    out, grads = R.train_grads(T(g['xi']), flow, float(g['beta']))
    close(out['loss_dkl'], g['loss_dkl'], rtol=1e-11)
    close(out['ess'], g['ess'], rtol=1e-9)
    close(out['logp'], g['logp'], rtol=1e-11); close(out['logq'], g['logq'], rtol=1e-11)
    close(out['q'], g['q'], atol=1e-9); close(out['plaq'], g['plaq'], rtol=1e-11)
    close(torch.sqrt((out['q'] - out['qi']) ** 2), g['dq'], atol=1e-9)
    for li, gl in enumerate(grads):
        for pi, gw in enumerate(gl):
            close(gw, g[f'gw{li}_{pi}'], rtol=1e-9, atol=1e-12)


def test_masks():
    g = load_golden('masks')
    for L in (8, 12):
        for mu in (0, 1):
            for off in range(4):
                mA, mF, mP, mL = R.stripe_masks(L, mu, off)
                k = f'L{L}_mu{mu}_off{off}_'
                assert np.array_equal(mA.numpy(), g[k + 'active'])
                assert np.array_equal(mF.numpy(), g[k + 'frozen'])
                assert np.array_equal(mP.numpy(), g[k + 'passive'])
                assert np.array_equal(mL.numpy(), g[k + 'link'])


@pytest.mark.parametrize('L', [8, 16])
@pytest.mark.parametrize('tag,tol', [('ref', 1e-6), ('tight', 1e-14)])
def test_physical_field_fthmc(L, tag, tol):
    """ipynb/ft_hmc.py:420-435 composed from the packaged reference functions: the oracle reproduces the
    reference's own inverse (same bisection, same stop rule), so both tolerances agree tightly."""
    g = load_golden(f'fthmc_phys_L{L}_{tag}')
    flow = golden_flow(g)
    dH, e, acc, newfield, x = R.ft_hmc_phys(T(g['field']), T(g['v']), T(g['u']), flow, float(g['beta']), float(g['dt']),
                                            int(g['nstep']), tol=tol)
    angle_close(x, g['x_inv'], atol=1e-12)
    close(dH, g['dH'], rtol=1e-8, atol=1e-9)
    close(e, g['exp_mdH'], rtol=1e-8)
    assert bool(acc) == bool(g['acc'])
    angle_close(newfield, g['newfield'], atol=1e-9)
    close(R.plaq_mean(newfield, float(g['beta'])), g['plaq'], rtol=1e-10)
    close(R.charge(newfield), g['Q'], atol=1e-9)


def test_independence_sampler_chain():
    """samplers.make_mcmc_ensemble (samplers.py:182-259) replayed on its recorded proposals and uniforms.
    The reference's generator proposes the PRIOR draws (see make_golden.py section 9): proposals =
    (xi, logq, -S(xi)); its histories are float32 (torch.Tensor(v))."""
    g = load_golden('sampler_L8')
    beta, flow = float(g['beta']), golden_flow(g)
    xi = T(g['xi'])
    xf, logdet = R.flow_forward(xi, flow)
    angle_close(xf, g['xflow'], atol=1e-12)
    close(R.prior_log_prob(xi) - logdet, g['logq'], rtol=1e-12)
    close(-R.action(xi, beta), g['logp_xi'], rtol=1e-12)
    close(-R.action(xf, beta), g['logp_flow'], rtol=1e-12)
    h = R.mcmc_chain(list(zip(xi, T(g['logq']), T(g['logp_xi']))), list(g['u']))
    assert np.array_equal(np.array(h['acc']), g['hist_acc'])
    assert 0 < np.sum(g['hist_acc']) < len(g['hist_acc'])            # the fixture holds accepts and rejects
    for k in ('q', 'dqsq', 'logq', 'logp'):
        close(h[k], g['hist_' + k], rtol=2e-7, atol=1e-6)            # float32 histories


def test_observables_tooling_golden(tmp_path):
    """delta-Q^2 versus lag, block means and their error against the outputs of the reference's own
    statistics helpers (ipynb/ft_hmc.py:16-53, 168-176; fixture written by make_golden.py section 10)."""
    from fthmc_amd.utils import observables as O
    g = load_golden('observables')
    assert int(g['n_block']) == O.N_BLOCK
    for tag in ('long', 'short', 'tiny'):
        q, want = g[f'q_{tag}'], g[f'rows_{tag}']
        rows = np.array(O.change_sqr_vs_dt(q, 10, reference_literal=True))
        np.testing.assert_allclose(rows, want, rtol=1e-13, atol=1e-15, equal_nan=True)
        np.testing.assert_allclose(O.block_means(q), g[f'blocks_{tag}'], rtol=1e-14)
        # the default error is the standard error of the block means; the reference's is var / sqrt(n - 1)
        std = np.array(O.change_sqr_vs_dt(q, 10))
        np.testing.assert_allclose(std[:, 1], want[:, 1], rtol=1e-13, atol=1e-15, equal_nan=True)
        ok = np.isfinite(want[:, 2]) & (want[:, 2] > 0)
        nblk = np.minimum(len(q) - np.arange(1, 11), O.N_BLOCK)[ok]          # block means behind each row
        np.testing.assert_allclose(std[ok, 2] ** 2, want[ok, 2] / np.sqrt(nblk - 1), rtol=1e-12)
        if tag != 'tiny':
            fn = tmp_path / f'dq2_{tag}.txt'
            O.save_topo_change_sqr(str(fn), q, reference_literal=True)
            np.testing.assert_allclose(np.loadtxt(fn, ndmin=2), g[f'file_{tag}'], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(O.sub_avg(g['q_short']), g['sub_avg'], rtol=1e-14, atol=1e-14)
    np.testing.assert_allclose(O.sigma(g['q_short'], reference_literal=True), float(g['sigma_short']), rtol=1e-13)


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_net_shapes_other_than_the_default(tag):
    """hidden_sizes / kernel_size / n_mixture_comps the reference accepts besides its default (layers.py:138-167, 399-429):
    the oracle on the reference's own outputs (make_golden.py section 12)."""
    g = load_golden(f'netshape_{tag}')
    flow, beta, act = golden_flow(g), float(g['beta']), str(g['act'])
    assert [w.shape[0] for w in flow[0][0:-2:2]] == list(g['hidden']) and flow[0][-1].shape[0] == int(g['n_mix']) + 1
    x = T(g['x'])
    y, ld = R.flow_forward(x, flow, act)
    close(y, g['y'], rtol=1e-11, atol=1e-11); close(ld, g['logdet'], rtol=1e-11, atol=1e-11)
    close(R.ft_action(x, flow, beta, act), g['S_eff'], rtol=1e-11, atol=1e-11)
    close(R.ft_force(x, flow, beta, act), g['ft_force'], rtol=1e-9, atol=1e-10)
    tflow = []
    li = 0
    while f'tw{li}_0' in g:
        row, pi = [], 0
        while f'tw{li}_{pi}' in g:
            row.append(T(g[f'tw{li}_{pi}'])); pi += 1
        tflow.append(tuple(row)); li += 1
    out, grads = R.train_grads(T(g['xi']), tflow, beta, act)
    close(out['loss_dkl'], g['loss_dkl'], rtol=1e-10); close(out['logq'], g['logq'], rtol=1e-10); close(out['logp'], g['logp'], rtol=1e-10)
    for li, row in enumerate(grads):
        for pi, gr in enumerate(row):
            close(gr, g[f'tgw{li}_{pi}'], rtol=1e-8, atol=1e-11)
