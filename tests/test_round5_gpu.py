"""GPU tests added in round 5: the reference-shaped drivers as captured loops (FieldTransformation.run, qed_helpers.ft_run)
against their eager loops, the carried state keyed by the weights' content, per-chain seeds formed on the device, the
weight versions of the C ABI (round 6: checked on the device)."""
import math

import numpy as np
import pytest
import torch

from conftest import ROOT  # noqa: F401
from fthmc_amd.graph_loop import capture       # torch.cuda.graph with the garbage collector held off (see there)

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


def _model(L, nl, B, beta=2.0, seed=11):
    from fthmc_amd import train as T
    from fthmc_amd.config import TrainConfig
    cfg = TrainConfig(L=L, beta=beta, n_layers=nl, batch_size=B, print_freq=0)
    torch.manual_seed(seed)
    return cfg, T.get_model(cfg)


# ---------------------------------------------------------------- seeds on the device
@pytest.mark.parametrize('seed,lo,B,traj', [(1331, 0, 128, 0), (7, 96, 32, 41), (2 ** 40 + 3, 1000, 5, 2 ** 33)])
def test_chain_seeds_on_the_device_equal_the_host_helper(seed, lo, B, traj):
    """fthmc_chain_seeds = parallel.chain_seeds (SplitMix64 of (seed, global chain id, trajectory)), also through a device
    counter that a captured launch advances."""
    from fthmc_amd import parallel
    want = parallel.chain_seeds(seed, lo, lo + B, traj)
    got = ops.chain_seeds(seed, lo, B, traj=traj, device='cuda')
    assert torch.equal(got.cpu(), want)
    counter = torch.tensor([traj], dtype=torch.int64, device='cuda')
    out = torch.empty(B, dtype=torch.int64, device='cuda')
    for k in range(3):
        ops.chain_seeds(seed, lo, B, counter=counter, advance=True, out=out)
        assert torch.equal(out.cpu(), parallel.chain_seeds(seed, lo, lo + B, traj + k)), k
    assert int(counter) == traj + 3


# ---------------------------------------------------------------- the captured run loop
@pytest.mark.parametrize('L,nl,B', [(8, 2, 4), (16, 4, 8), (32, 2, 16)])
def test_captured_run_equals_the_eager_loop(L, nl, B):
    """FieldTransformation.run(batch=True): the captured loop (one graph launch per trajectory, momenta and uniforms from
    torch's generator inside the graph, state carried in place, history read back lazily) returns the history and the
    final field of the eager loop, bit for bit (fthmc/ft_hmc.py:272-346)."""
    from fthmc_amd.config import lfConfig
    from fthmc_amd.ft_hmc import FieldTransformation, LazyHistory
    cfg, model = _model(L, nl, B)
    x0 = (0.3 * (2 * torch.rand(B, 2, L, L, dtype=torch.float64) - 1)).cuda()
    n = 6
    out = {}
    for mode in ('graph', 'eager'):
        ft = FieldTransformation(flow=model.layers, config=cfg, lfconfig=lfConfig(tau=1.0, nstep=6))
        torch.manual_seed(5); torch.cuda.manual_seed(5)
        h = ft.run(x0.clone(), nprint=0, num_trajs=n, batch=True, use_graph=(mode == 'graph'))
        assert isinstance(h, LazyHistory) == (mode == 'graph')
        assert (ft._loop is not None and ft._loop['loop'].captured) == (mode == 'graph')
        # a second run from the field the first one returned: the carried state is taken over (no H0 sweep) in both modes
        h2 = ft.run(ft.x_last, nprint=0, num_trajs=3, batch=True, use_graph=(mode == 'graph'))
        out[mode] = (h, h2, ft.x_last.clone())
    for a, b in zip(out['graph'][:2], out['eager'][:2]):
        assert set(a.keys()) == set(b.keys()) == {'traj', 'dt', 'acc', 'dh', 'exp_mdh', 'plaq', 'q', 'dq'}
        for k in ('acc', 'dh', 'exp_mdh', 'plaq', 'q', 'dq'):
            assert len(a[k]) == len(b[k])
            for i, (ta, tb) in enumerate(zip(a[k], b[k])):
                assert torch.equal(ta, tb), (k, i)
        assert a['traj'] == b['traj']
    assert torch.equal(out['graph'][2], out['eager'][2])


def test_captured_single_chain_run_equals_the_eager_loop():
    """the reference's own shape: one [1, 2, L, L] system, one accept (ft_hmc.py:190-224)"""
    from fthmc_amd.config import lfConfig
    from fthmc_amd.ft_hmc import FieldTransformation
    L, nl, n = 8, 2, 7
    cfg, model = _model(L, nl, 1, seed=13)
    x0 = (0.3 * (2 * torch.rand(1, 2, L, L, dtype=torch.float64) - 1)).cuda()
    out = {}
    for mode in ('graph', 'eager'):
        ft = FieldTransformation(flow=model.layers, config=cfg, lfconfig=lfConfig(tau=1.0, nstep=8))
        torch.manual_seed(6); torch.cuda.manual_seed(6)
        out[mode] = (ft.run(x0.clone(), nprint=0, num_trajs=n, use_graph=(mode == 'graph')), ft.x_last.clone())
    a, b = out['graph'][0], out['eager'][0]
    for k in ('dh', 'plaq', 'q', 'dq'):
        for ta, tb in zip(a[k], b[k]):
            assert torch.equal(ta, tb), k
    assert [bool(t) for t in a['acc']] == [bool(t) for t in b['acc']]
    assert torch.equal(out['graph'][1], out['eager'][1])


def test_literal_reference_modes_stay_on_the_eager_loop():
    """energy_mode='reference_literal' (calc_energy's batch-wide kinetic term) has no fused trajectory: run() keeps
    the step-by-step loop and returns a plain dict"""
    from fthmc_amd.config import lfConfig
    from fthmc_amd.ft_hmc import FieldTransformation, LazyHistory
    cfg, model = _model(8, 2, 3)
    ft = FieldTransformation(flow=model.layers, config=cfg, lfconfig=lfConfig(tau=0.5, nstep=4), energy_mode='reference_literal')
    x0 = (0.3 * (2 * torch.rand(3, 2, 8, 8, dtype=torch.float64) - 1)).cuda()
    h = ft.run(x0, nprint=0, num_trajs=2, batch=True)
    assert not isinstance(h, LazyHistory) and len(h['acc']) == 2 and '_plaq' not in h and '_q' not in h


# ---------------------------------------------------------------- carry keyed by the weights' content
@pytest.mark.parametrize('use_graph', [True, False])
def test_carried_state_is_dropped_when_the_weights_change(use_graph):
    """run, a FlatAdam step (the kernel writes the flat buffer through raw pointers: no tensor version moves and
    flow_weights() hands out the same object), run(ft.x_last): the second run must take H0 from the NEW weights -- equal to the
    loop that recomputes everything -- and the same after load_state_dict and after a change of beta."""
    from fthmc_amd import train as T
    from fthmc_amd.config import lfConfig
    from fthmc_amd.ft_hmc import FieldTransformation
    from fthmc_amd.utils import layers as LY
    L, nl, B = 8, 2, 6
    cfg, model = _model(L, nl, B)
    opt = T.make_optimizer(model, cfg)                                   # FlatAdam: flattens the flow
    assert isinstance(opt, T.FlatAdam)
    ft = FieldTransformation(flow=model.layers, config=cfg, lfconfig=lfConfig(tau=1.0, nstep=6))
    x0 = (0.3 * (2 * torch.rand(B, 2, L, L, dtype=torch.float64) - 1)).cuda()
    torch.manual_seed(3); torch.cuda.manual_seed(3)
    ft.run(x0.clone(), nprint=0, num_trajs=3, batch=True, use_graph=use_graph)
    saved = {k: v.clone() for k, v in model.layers.state_dict().items()}

    def change_by_optimizer():
        w_before = LY.flow_weights(model.layers)
        LY.flow_grad_buffer(model.layers).normal_()
        opt.param_groups[0]['lr'] = 0.05
        opt.step()
        assert LY.flow_weights(model.layers) is w_before                 # the same object: identity says nothing

    def change_by_load():
        model.layers.load_state_dict(saved)

    def change_beta():
        cfg.beta = 2.7
        ft._denom = cfg.beta * cfg.volume

    for change in (change_by_optimizer, change_by_load, change_beta):
        x = ft.x_last
        change()
        torch.manual_seed(4); torch.cuda.manual_seed(4)
        h = ft.run(x, nprint=0, num_trajs=2, batch=True, use_graph=use_graph)
        # the stateless loop under the new weights on the same draws
        ft2 = FieldTransformation(flow=model.layers, config=cfg, lfconfig=lfConfig(tau=1.0, nstep=6))
        torch.manual_seed(4); torch.cuda.manual_seed(4)
        xx = x.clone()
        for i in range(2):
            xx, m = ft2._batch_hmc(xx.clone(), step=i)
            assert torch.equal(m['acc'], h['acc'][i]), change.__name__
            assert torch.equal(m['dh'], h['dh'][i]), change.__name__
        assert torch.equal(xx, ft.x_last), change.__name__


def test_weights_cache_follows_a_replaced_net():
    """FieldTransformation.weights() keeps the parameter list of the flow it saw; a conv net swapped inside the same
    ModuleList (transfer-style reuse) must be picked up"""
    from fthmc_amd.config import lfConfig
    from fthmc_amd.ft_hmc import FieldTransformation
    cfg, model = _model(8, 2, 2)
    _, other = _model(8, 2, 2, seed=99)
    ft = FieldTransformation(flow=model.layers, config=cfg, lfconfig=lfConfig(tau=1.0, nstep=4))
    x = (0.3 * (2 * torch.rand(2, 2, 8, 8, dtype=torch.float64) - 1)).cuda()
    s0 = ft.action(x).clone()
    model.layers[1].plaq_coupling.net = other.layers[1].plaq_coupling.net
    s1 = ft.action(x)
    ft_fresh = FieldTransformation(flow=model.layers, config=cfg, lfconfig=lfConfig(tau=1.0, nstep=4))
    assert torch.equal(s1, ft_fresh.action(x)) and not torch.equal(s0, s1)


# ---------------------------------------------------------------- weight versions (C ABI `_v` entry points)
def test_weight_versions_are_checked_on_the_device():
    """ops with a caller-stated weight version (`wkey` -> the `_v` entry points of the C ABI) expand the weights once per version
    into the stream's workspace; the library compares the version with the stamps its last expansion left there ON THE DEVICE:
    a wrong or stale version, other weights under the same version, a call without a version in between all end in an
    expansion and in the numbers of calls that state nothing.  That an equal version really skips the expansion is shown by
    wiping the expansion behind its stamps."""
    gen = torch.Generator().manual_seed(3)
    L, nl, B, beta = 16, 4, 5, 3.0
    wa = ops.pack_weights(R.default_flow(nl, gen), device='cuda')
    wb = ops.pack_weights(R.default_flow(nl, gen), device='cuda')
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    ref_a = ops.ft_action(x, wa, nl, beta)[0].clone()
    ref_b = ops.ft_action(x, wb, nl, beta)[0].clone()
    Fa = ops.ft_force(x, wa, nl, beta).clone()
    assert torch.equal(ops.ft_action(x, wa, nl, beta, wkey='v1')[0], ref_a)      # expands, stamps
    assert torch.equal(ops.ft_action(x, wa, nl, beta, wkey='v1')[0], ref_a)      # stamps found
    assert torch.equal(ops.ft_force(x, wa, nl, beta, wkey='v1'), Fa)
    # the expansion of layer 1 wiped behind its stamps (the head = 64 layer regions, the last 8 doubles of a region are its stamps: csrc/kernels.h)
    ws = ops._WS[(x.device.index, torch.cuda.current_stream().cuda_stream)]
    from fthmc_amd import _lib
    W = int(_lib.load().fthmc_ws_head_bytes()) // (64 * 8)
    assert W >= 8768 and int(_lib.load().fthmc_ws_head_bytes()) == 64 * W * 8
    ws[W:2 * W - 8].zero_()
    assert not torch.equal(ops.ft_action(x, wa, nl, beta, wkey='v1')[0], ref_a)  # same version: nothing was expanded (the proof)
    assert torch.equal(ops.ft_action(x, wa, nl, beta, wkey='other')[0], ref_a)   # a WRONG version: expanded, same numbers
    assert torch.equal(ops.ft_action(x, wa, nl, beta, wkey='other')[0], ref_a)
    assert torch.equal(ops.ft_action(x, wb, nl, beta, wkey='other')[0], ref_b)   # other weights under the same version: expanded
    assert torch.equal(ops.ft_action(x, wa, nl, beta, wkey='other')[0], ref_a)
    ops.flow_layer_fwd(x, wb[:955].contiguous(), 0, 0)                           # a call without a version overwrites layer 0 and clears its stamps
    assert torch.equal(ops.ft_action(x, wa, nl, beta, wkey='other')[0], ref_a)
    with pytest.raises(ops.FthmcError):                                          # a call that fails early leaves nothing behind (there is no state to leave)
        ops.ft_force(x[:, :, :6, :6].contiguous(), wb, nl, beta, wkey='other')
    assert torch.equal(ops.ft_action(x, wb, nl, beta, wkey='other')[0], ref_b)
    wa.mul_(1.01)                                                                # new content, new version
    ref_a2 = ops.ft_action(x, wa, nl, beta)[0].clone()
    assert torch.equal(ops.ft_action(x, wa, nl, beta, wkey='v2')[0], ref_a2) and not torch.equal(ref_a2, ref_a)
    # a graph captured under one version keeps producing the right numbers when other weights passed through its workspace
    # in between (the advisor's sequence: eager w1 -> replay with w2 -> eager w1)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        out_b = ops.ft_action(x, wb, nl, beta, wkey='g')[0]                      # warm-up: workspace of this stream
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with capture(g, st):
            out_b = ops.ft_action(x, wb, nl, beta, wkey='g')[0]
        assert torch.equal(ops.ft_action(x, wa, nl, beta, wkey='v2')[0], ref_a2)
        g.replay()
        assert torch.equal(out_b, ref_b)
        assert torch.equal(ops.ft_action(x, wa, nl, beta, wkey='v2')[0], ref_a2)
        g.replay()
        assert torch.equal(out_b, ref_b)
    torch.cuda.current_stream().wait_stream(st)


# ---------------------------------------------------------------- physical-field driver
def test_captured_ft_run_equals_the_eager_loop(tmp_path):
    """qed_helpers.ft_run (ipynb/ft_hmc.py:437-487: inverse flow, trajectory, forward flow, observables per trajectory of ONE
    configuration) as a captured loop against its eager loop: same histories, same final field, same log lines."""
    from fthmc_amd.config import Param
    from fthmc_amd.utils import qed_helpers as qed
    L, nl = 8, 2
    _, model = _model(L, nl, 1, seed=21)
    param = Param(beta=2.0, L=L, tau=1.0, nstep=6, ntraj=5, nrun=2)
    f0 = (0.4 * (2 * torch.rand(2, L, L, dtype=torch.float64) - 1)).cuda()
    res = {}
    for mode in ('graph', 'eager'):
        torch.manual_seed(8); torch.cuda.manual_seed(8)
        log = tmp_path / f'{mode}.log'
        field, hist = qed.ft_run(param, model.layers, f0.clone(), logfile=str(log), use_graph=(mode == 'graph'))
        res[mode] = (field.clone(), {k: list(v) for k, v in hist.items()}, log.read_text())
    assert torch.equal(res['graph'][0], res['eager'][0])
    for k in ('dH', 'exp_mdH', 'acc', 'plaq', 'topo'):
        assert res['graph'][1][k] == res['eager'][1][k], k
    assert res['graph'][2] == res['eager'][2]


# ---------------------------------------------------------------- the reference's GradScaler option
def test_train_step_with_a_grad_scaler_equals_the_plain_step():
    """train_step(scaler=GradScaler()) (train.py:206-209, 321-324): scale(loss).backward(), scaler.step, scaler.update on the
    autograd route leave the weights of the plain step (fp64: the scale is a power of two, scaling and unscaling are exact) and
    the same metrics; train(use_scaler=True) runs."""
    from fthmc_amd import train as T
    from fthmc_amd.config import TrainConfig
    from fthmc_amd.utils import layers as LY
    from fthmc_amd.utils import qed_helpers as qed
    tc = TrainConfig(L=8, beta=2.0, n_layers=2, batch_size=6, base_lr=1e-3, print_freq=0, n_era=1, n_epoch=3)
    torch.manual_seed(4)
    m0 = T.get_model(tc)
    init = {k: v.clone() for k, v in m0.layers.state_dict().items()}
    xi = m0.prior.sample_n(6)
    act = qed.BatchAction(tc.beta)
    res = {}
    for tag in ('plain', 'scaler'):
        model = T.get_model(tc); model.layers.load_state_dict(init)
        opt = torch.optim.Adam(model.layers.parameters(), lr=tc.base_lr)
        sc = torch.amp.GradScaler('cuda') if tag == 'scaler' else None
        met = [T.train_step(model, tc, act, opt, 6, xi=xi, scaler=sc, fused=False) for _ in range(3)]
        res[tag] = (LY.flow_weights(model.layers).clone(), met)
    assert torch.equal(res['plain'][0], res['scaler'][0])
    for a, b in zip(res['plain'][1], res['scaler'][1]):
        for k in T.METRIC_KEYS:
            np.testing.assert_array_equal(a[k], b[k])
    out = T.train(tc, use_scaler=True, verbose=False)
    assert len(out['history']['loss_dkl']) == 3 and np.isfinite(out['history']['loss_dkl'][-1])


def test_two_captured_loops_with_different_flows_do_not_disturb_each_other():
    """Two FieldTransformations over different flows, their captured runs interleaved (and an eager batch call in between on
    the shared stream pool): every run equals the eager loop of its own flow -- the replays carry no weight expansion, so each
    loop owns the streams (and with them the workspaces) it replays into, and re-establishes them at every run."""
    import pickle
    from fthmc_amd.config import lfConfig
    from fthmc_amd.ft_hmc import FieldTransformation
    L, nl, B = 32, 2, 16                                                  # two chain groups: the side-stream path
    cfg, ma = _model(L, nl, B, seed=31)
    _, mb = _model(L, nl, B, seed=32)
    x0 = (0.3 * (2 * torch.rand(B, 2, L, L, dtype=torch.float64) - 1)).cuda()
    lf = lfConfig(tau=1.0, nstep=5)
    want = {}
    for tag, m in (('a', ma), ('b', mb)):
        ft = FieldTransformation(flow=m.layers, config=cfg, lfconfig=lf)
        torch.manual_seed(7); torch.cuda.manual_seed(7)
        h1 = ft.run(x0.clone(), nprint=0, num_trajs=3, batch=True, use_graph=False)
        h2 = ft.run(ft.x_last, nprint=0, num_trajs=3, batch=True, use_graph=False)
        want[tag] = (torch.stack(h1['dh'] + h2['dh']), ft.x_last.clone())
    fa = FieldTransformation(flow=ma.layers, config=cfg, lfconfig=lf)
    fb = FieldTransformation(flow=mb.layers, config=cfg, lfconfig=lf)
    torch.manual_seed(7); torch.cuda.manual_seed(7)
    ga = torch.cuda.get_rng_state()
    ha1 = fa.run(x0.clone(), nprint=0, num_trajs=3, batch=True)
    sa = torch.cuda.get_rng_state()
    torch.cuda.set_rng_state(ga)
    hb1 = fb.run(x0.clone(), nprint=0, num_trajs=3, batch=True)
    sb = torch.cuda.get_rng_state()
    fb._batch_hmc(x0.clone())                                             # an eager two-group call on the shared side-stream pool
    torch.cuda.set_rng_state(sa)
    ha2 = fa.run(fa.x_last, nprint=0, num_trajs=3, batch=True)
    torch.cuda.set_rng_state(sb)
    hb2 = fb.run(fb.x_last, nprint=0, num_trajs=3, batch=True)
    assert torch.equal(torch.stack(ha1['dh'] + ha2['dh']), want['a'][0]) and torch.equal(fa.x_last, want['a'][1])
    assert torch.equal(torch.stack(hb1['dh'] + hb2['dh']), want['b'][0]) and torch.equal(fb.x_last, want['b'][1])
    plain = pickle.loads(pickle.dumps(ha1))                               # a LazyHistory travels as the dict it stands for
    assert type(plain) is dict and torch.equal(torch.stack(plain['dh']), torch.stack(ha1['dh']))


# ---------------------------------------------------------------- more of the headline size against the oracle
def test_headline_size_two_chain_groups_against_the_oracle():
    """BASELINE configs[2] at its own lattice size and depth (L = 64, beta = 6, 8 layers), 24 chains = two chain groups on two
    streams: S_eff, log det J, plaquette, Q, the force and ONE WHOLE 10-step trajectory (H0, H1, dH, accepts, end field) against
    the oracle on all 24 chains (the round-4 suite compared 3 chains at this size; bench.py compares 128 in every run)."""
    B, L, nl, beta, dt, nstep = 24, 64, 8, 6.0, 0.1, 10
    gen = torch.Generator().manual_seed(4242)
    flow = R.default_flow(nl, gen)
    w = ops.pack_weights(flow, device='cuda')
    x = 0.35 * (torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1)        # near-cold, as the bench's chains
    v = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64)
    u = torch.rand(B, generator=gen, dtype=torch.float64)
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    y, ld = R.flow_forward(x, flow)
    Sg, ldg, pg, qg = ops.ft_action(x.cuda(), w, nl, beta)
    np.testing.assert_allclose(H(ldg), H(ld), rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(H(Sg), H(R.ft_action(x, flow, beta)), rtol=1e-12)
    np.testing.assert_allclose(H(pg), H(R.plaq_mean(y, beta)), rtol=1e-12)
    np.testing.assert_allclose(H(qg), H(R.charge(y)), atol=1e-8)
    F = R.ft_force(x, flow, beta)
    np.testing.assert_allclose(H(ops.ft_force(x.cuda(), w, nl, beta)), H(F), rtol=1e-8, atol=1e-9 * float(F.abs().max()))
    dH, _, acc, newx, h0, h1 = R.ft_hmc(x, v, u, flow, beta, dt, nstep, mode='md')
    r = ops.ft_trajectory(x.cuda(), v.cuda(), u.cuda(), w, nl, beta, dt, nstep, mode='md', groups=2)
    np.testing.assert_allclose(H(r['H0']), H(h0), rtol=1e-12)
    np.testing.assert_allclose(H(r['H1']), H(h1), rtol=1e-9)                     # ten MD steps amplify rounding
    np.testing.assert_allclose(H(r['dH']), H(dH), rtol=0, atol=1e-6 * float(h1.abs().max()))
    border = (u - torch.exp(-dH)).abs() < 1e-9
    assert bool((((r['acc'].cpu() > 0.5) == acc) | border).all())
    d = (r['x_new'].cpu() - newx + math.pi) % (2 * math.pi) - math.pi
    assert float(d.abs().max()) < 1e-6
