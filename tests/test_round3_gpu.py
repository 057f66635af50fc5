"""GPU tests added in round 3: BASELINE configs[3] (1024 chains) on one GPU through size-independent properties, a
multi-rank rehearsal of bench.py whose chains equal the one-rank run bit for bit, the reference's `fthmc.*` import path
on goldens, aliased (in-place) layer calls, and the small-lattice fused force path against goldens and the tiled path."""
import json
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_flow, load_golden
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


def D(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64).copy()).cuda()


def H(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def close(a, b, rtol=1e-10, atol=1e-10):
    np.testing.assert_allclose(H(a), H(b), rtol=rtol, atol=atol)


def angle_close(a, b, atol=1e-9):
    d = (H(a) - H(b) + np.pi) % (2 * np.pi) - np.pi
    assert np.max(np.abs(d)) < atol, np.max(np.abs(d))


# ---------------------------------------------------------------- BASELINE configs[3]: 1024 chains
def test_config4_1024_chains_on_one_gpu():
    """configs[3] = 1024 chains of L=64, beta=6, 8 layers (128 per GPU on 8 GPUs).  All of them fit one MI355X
    (workspace 6.3 GB): size-independent properties on the full batch, and any chain of the 1024 equals the same
    chain run alone and run inside its 128-chain shard, bit for bit (what sharding over ranks relies on)."""
    gen = torch.Generator().manual_seed(1024)
    B, L, nl, beta, dt, nstep = 1024, 64, 8, 6.0, 0.1, 10
    flow = R.default_flow(nl, gen)
    w = ops.pack_weights(flow, device='cuda')
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    y, ld = ops.flow_forward(x, w, nl)
    S, Q, plaq = ops.wilson_action_charge(y, beta)
    assert float((Q - Q.round()).abs().max()) < 1e-8                       # integer topological charge
    xb, ldb = ops.flow_reverse(y, w, nl, tol=1e-13)
    angle_close(xb, x, atol=1e-9); close(ldb, -ld, atol=1e-6)             # forward o reverse = id
    F = ops.ft_force(x, w, nl, beta)
    assert torch.equal(F, ops.ft_force(x, w, nl, beta))                    # deterministic
    d = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64).cuda()
    eps = 1e-5
    fd = (ops.ft_action(x + eps * d, w, nl, beta)[0] - ops.ft_action(x - eps * d, w, nl, beta)[0]) / (2 * eps)
    close((F * d).flatten(1).sum(1), fd, rtol=2e-5, atol=5e-3)             # force = gradient of S_eff
    # gauge invariance of the effective action: x_mu(n) -> x_mu(n) + a(n) - a(n + mu)
    a = (torch.rand(B, L, L, generator=gen, dtype=torch.float64) * 2 * math.pi).cuda()
    xg = torch.stack([x[:, 0] + a - torch.roll(a, -1, 1), x[:, 1] + a - torch.roll(a, -1, 2)], 1).contiguous()
    close(ops.ft_action(xg, w, nl, beta)[0], ops.ft_action(x, w, nl, beta)[0], rtol=1e-10, atol=1e-7)
    # whole trajectories: chain k of the 1024 == chain k alone == chain k inside its 128-chain shard (rank k // 128)
    v = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64).cuda()
    u = torch.rand(B, generator=gen, dtype=torch.float64).cuda()
    x_cold = (0.2 * x).contiguous()
    r = ops.ft_trajectory(x_cold, v, u, w, nl, beta, dt, nstep, groups=2)
    assert float(r['dH'].abs().max()) < 1e3 and bool(torch.isfinite(r['x_new']).all())
    for k in (0, 517, 1023):
        r1 = ops.ft_trajectory(x_cold[k:k + 1].contiguous(), v[k:k + 1].contiguous(), u[k:k + 1].contiguous(), w, nl, beta, dt, nstep)
        for key in ('x_new', 'dH', 'acc', 'H0', 'H1', 'plaq', 'Q'):
            assert torch.equal(r1[key][0], r[key][k]), (k, key)
    lo = 512
    rs = ops.ft_trajectory(x_cold[lo:lo + 128].contiguous(), v[lo:lo + 128].contiguous(), u[lo:lo + 128].contiguous(),
                           w, nl, beta, dt, nstep, groups=2)
    for key in ('x_new', 'dH', 'acc', 'H0', 'H1', 'plaq', 'Q'):
        assert torch.equal(rs[key], r[key][lo:lo + 128]), key
    # leapfrog reversibility on the full batch
    xo, vo = ops.ft_leapfrog(x_cold, v, w, nl, beta, dt, nstep)
    xr, vr = ops.ft_leapfrog(xo, -vo, w, nl, beta, dt, nstep)
    angle_close(xr, x_cold, atol=1e-8); close(-vr, v, rtol=1e-8, atol=1e-8)
    ops.release_workspaces()


# ---------------------------------------------------------------- bench.py over several ranks == one rank, per chain
def _bench(tmp, tag, gpus, batch, extra=()):
    env = dict(os.environ, FTHMC_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='2')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    dump = str(tmp / tag)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(gpus), '--config', '2', '--batch', str(batch),
                        '--steps', '2', '--warmup', '1', '--thermalize', '3', '--regions', '1', '--no-cpu-baseline', '--dump', dump,
                        *extra], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
    parts = [dict(np.load(f'{dump}.{gpus}.{r}.npz')) for r in range(gpus)]
    return line, parts


def test_bench_multi_rank_rehearsal_equals_one_rank(tmp_path):
    """`bench.py --gpus 4 --config 2 --batch 4` (four ranks sharing this GPU, collectives over gloo: the code path of the
    driver's 8-GPU run up to the backend; at most 6 processes may use a GPU box's card, so 4 ranks here and 8 ranks in
    the CPU test tests/test_host_logic.py::test_eight_rank_gloo_equals_single_process) reports n_gpus 4 and 16 chains,
    and every chain ends where it ends in the one-rank run of the same 16 chains: start field, thermalisation, momenta
    and accept draws are keyed by the global chain id."""
    one, p1 = _bench(tmp_path, 'one', 1, 16)
    four, p4 = _bench(tmp_path, 'four', 4, 4)
    assert four['n_gpus'] == 4 and four['config']['chains_total'] == 16 and four['config']['chains_per_gpu'] == 4
    assert one['n_gpus'] == 1 and one['config']['chains_total'] == 16
    assert [(int(p['lo']), int(p['hi'])) for p in p4] == [(0, 4), (4, 8), (8, 12), (12, 16)]
    for key in ('x0', 'x', 'dH', 'acc', 'Q', 'plaq'):
        got = np.concatenate([p[key] for p in p4])
        assert np.array_equal(got, p1[0][key]), key
    # the global statistics (C1 all-reduce) agree with the one-rank run
    assert abs(four['acceptance'] - one['acceptance']) < 1e-12 and abs(four['plaq'] - one['plaq']) < 1e-9
    assert four['regions']['n'] == 1 and one['value'] > 0 and four['value'] > 0
    # strong scaling keeps the total
    strong, _ = _bench(tmp_path, 'strong', 2, 16, ('--scaling', 'strong'))
    assert strong['n_gpus'] == 2 and strong['config']['chains_total'] == 16 and strong['config']['chains_per_gpu'] == 8


def test_bench_times_several_regions_when_one_is_short(tmp_path):
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', '1', '--steps', '5', '--warmup', '2',
                        '--thermalize', '2', '--no-cpu-baseline', '--min-seconds', '0.02'], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
    reg = line['regions']
    # at least five regions when one is short, and as many as it takes to time --min-seconds of GPU work (default 6 s)
    assert reg['n'] >= 5 and len(reg['seconds']) == reg['n'] and line['steps'] == 5 and sum(reg['seconds']) >= 0.02
    med = sorted(reg['seconds'])[reg['n'] // 2]
    assert abs(line['ms_per_step'] - med / 5 * 1e3) < 1e-2 * line['ms_per_step']
    assert abs(line['value'] - line['config']['chains_total'] * 10 * 5 / med) < 1e-2 * line['value']


# ---------------------------------------------------------------- the reference's import path
def test_reference_style_caller_through_fthmc_alias():
    """What a user of nftqcd/fthmc writes (fthmc/main.py:73-104, fthmc/train.py:162-228), imports unchanged."""
    from fthmc.config import FlowModel, TrainConfig, lfConfig
    from fthmc.ft_hmc import FieldTransformation
    from fthmc.train import train_step
    import fthmc.utils.layers as layers
    import fthmc.utils.qed_helpers as qed
    from fthmc.utils.distributions import MultivariateUniform

    def build(g, L):
        nl = int(g['n_layers'])
        flow = layers.make_u1_equiv_layers(n_layers=nl, n_mixture_comps=2, lattice_shape=(L, L), hidden_sizes=[8, 8],
                                           kernel_size=3, activation_fn='silu')
        names = ['net.0.weight', 'net.0.bias', 'net.2.weight', 'net.2.bias', 'net.4.weight', 'net.4.bias']
        flow.load_state_dict({f'{li}.plaq_coupling.{n}': D(g[f'w{li}_{pi}']) for li in range(nl) for pi, n in enumerate(names)})
        return flow
    # run_fthmc's body: FieldTransformation(flow=..., config=..., lfconfig=...) then trajectories
    g = load_golden('traj_md_L16')
    flow = build(g, 16)
    config = TrainConfig(L=16, beta=float(g['beta']), n_layers=len(flow))
    lfconfig = lfConfig(tau=float(g['dt']) * int(g['nstep']), nstep=int(g['nstep']))
    ft = FieldTransformation(flow=flow, config=config, lfconfig=lfconfig)
    xnew, m = ft._batch_hmc(D(g['x']), v=D(g['v']), u=D(g['u']))
    close(m['dh'], g['dH'], rtol=1e-6, atol=1e-6)
    assert np.array_equal(H(m['acc']) > 0.5, g['acc'])
    close(qed.ft_action(config, flow, D(g['x'])) + 0.5 * (D(g['v']) ** 2).flatten(1).sum(1), g['H0'], rtol=1e-10)
    # train_step on the golden prior draw
    g = load_golden('train_L8')
    flow = build(g, 8)
    B = g['xi'].shape[0]
    cfg = TrainConfig(L=8, beta=float(g['beta']), n_layers=len(flow), batch_size=B, base_lr=float(g['lr']))
    prior = MultivariateUniform(-math.pi * torch.ones(2, 8, 8, dtype=torch.float64, device='cuda'),
                                math.pi * torch.ones(8, 8, dtype=torch.float64, device='cuda'))
    model = FlowModel(prior=prior, layers=flow)
    opt = torch.optim.Adam(flow.parameters(), lr=cfg.base_lr)
    met = train_step(model, cfg, qed.BatchAction(float(g['beta'])), opt, B, xi=D(g['xi']))
    close(met['loss_dkl'], g['loss_dkl'], rtol=1e-10); close(met['ess'], g['ess'], rtol=1e-8)


# ---------------------------------------------------------------- aliased layer calls
@pytest.mark.parametrize('variant', [1, 0])
def test_layer_calls_with_output_aliasing_input(variant):
    """include/fthmc_hip.h: y may alias x in fthmc_flow_layer_fwd / _rev and in fthmc_flow_reverse -- lattices of several
    tiles per chain (halos cross tile borders), both kernel variants, all (mu, off), against the out-of-place call."""
    from fthmc_amd import _lib
    lib = _lib.load()
    ops.set_variant(variant)
    try:
        gen = torch.Generator().manual_seed(33)
        B, L = 3, 48
        flow = R.default_flow(8, gen)
        w = ops.pack_weights(flow, device='cuda')
        x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
        wsb = torch.empty(ops.ws_bytes(B, L, 8) // 8 + 1, dtype=torch.float64, device='cuda')
        st = torch.cuda.current_stream().cuda_stream
        for li in range(8):
            mu, off = li % 2, (li // 2) % 4
            wl = w[li * 955:(li + 1) * 955].contiguous()
            y, lj = ops.flow_layer_fwd(x, wl, mu, off)
            xa = x.clone(); lja = torch.empty_like(lj)
            assert lib.fthmc_flow_layer_fwd(xa.data_ptr(), wl.data_ptr(), None, B, L, mu, off, 0, xa.data_ptr(), lja.data_ptr(),
                                            wsb.data_ptr(), wsb.numel() * 8, st) == 0
            torch.cuda.synchronize()
            assert torch.equal(xa, y) and torch.equal(lja, lj), (mu, off)
            xr, ljr = ops.flow_layer_rev(y, wl, mu, off)
            ya = y.clone(); ljb = torch.empty_like(lj)
            assert lib.fthmc_flow_layer_rev(ya.data_ptr(), wl.data_ptr(), None, B, L, mu, off, 0, 1e-12, ya.data_ptr(), ljb.data_ptr(),
                                            wsb.data_ptr(), wsb.numel() * 8, st) == 0
            torch.cuda.synchronize()
            assert torch.equal(ya, xr) and torch.equal(ljb, ljr), (mu, off)
            angle_close(xr, x, atol=1e-9)
        # the sweep (last layer out of place, the rest in place) == explicit out-of-place layer calls
        yfull, _ = ops.flow_forward(x, w, 8)
        xs, lds = ops.flow_reverse(yfull, w, 8)
        cur, tot = yfull, torch.zeros(B, dtype=torch.float64, device='cuda')
        for li in reversed(range(8)):
            cur, lj = ops.flow_layer_rev(cur, w[li * 955:(li + 1) * 955].contiguous(), li % 2, (li // 2) % 4)
            tot = tot + lj
        assert torch.equal(xs, cur)
        close(lds, tot, rtol=1e-13, atol=1e-12)
    finally:
        ops.set_variant(1)


# ---------------------------------------------------------------- small-lattice fused path (csrc/flow_small.hip)
def _both_paths(fn):
    ops.set_small_path(True)
    try:
        a = fn()
        ops.set_small_path(False)
        b = fn()
    finally:
        ops.set_small_path(True)
    return a, b


@pytest.mark.parametrize('L,nl,B,beta,act', [(16, 4, 32, 4.0, 'silu'), (8, 2, 3, 2.0, 'silu'), (12, 8, 5, 3.0, 'silu'),
                                             (16, 16, 2, 4.0, 'relu'), (8, 8, 1, 2.0, 'leaky_relu'), (12, 1, 7, 4.0, 'silu')])
def test_small_lattice_path_equals_tiled_path(L, nl, B, beta, act):
    """L = 8, 12, 16 run by default as ONE launch per trajectory / force / action (one workgroup per chain, periodic planes
    in LDS).  Same results as the tiled kernels (one launch per layer) on the same inputs: every entry point that takes
    the path, to fp64 round-off (1e-11; trajectories 1e-9 after 6 MD steps)."""
    import bench
    gen = torch.Generator().manual_seed(100 + L + nl)
    flow = bench.make_flow(gen, nl)
    w = ops.pack_weights(flow, device='cuda')
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    v = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64).cuda()
    u = torch.rand(B, generator=gen, dtype=torch.float64).cuda()
    (ya, lda), (yb, ldb) = _both_paths(lambda: ops.flow_forward(x, w, nl, act))
    angle_close(ya, yb, atol=1e-11); close(lda, ldb, rtol=1e-12, atol=1e-10)
    (Sa, la, pa, qa), (Sb, lb, pb, qb) = _both_paths(lambda: ops.ft_action(x, w, nl, beta, act))
    close(Sa, Sb, rtol=1e-12, atol=1e-10); close(la, lb, rtol=1e-12, atol=1e-10); close(pa, pb, rtol=1e-12); close(qa, qb, atol=1e-10)
    Fa, Fb = _both_paths(lambda: ops.ft_force(x, w, nl, beta, act))
    close(Fa, Fb, rtol=1e-10, atol=1e-11 * float(Fb.abs().max()))
    assert torch.equal(Fa, ops.ft_force(x, w, nl, beta, act))                       # deterministic
    xs = (0.3 * x).contiguous()
    (xa, va), (xb_, vb) = _both_paths(lambda: ops.ft_leapfrog(xs, v, w, nl, beta, 0.05, 4, act))
    angle_close(xa, xb_, atol=1e-10); close(va, vb, rtol=1e-9, atol=1e-9)
    ra, rb = _both_paths(lambda: ops.ft_trajectory(xs, v, u, w, nl, beta, 0.05, 6, act))
    close(ra['H0'], rb['H0'], rtol=1e-12); close(ra['H1'], rb['H1'], rtol=1e-10); close(ra['dH'], rb['dH'], atol=1e-8)
    assert torch.equal(ra['acc'], rb['acc'])
    angle_close(ra['x_new'], rb['x_new'], atol=1e-9); close(ra['state'], rb['state'], rtol=1e-10, atol=1e-9)
    close(ra['plaq'], rb['plaq'], rtol=1e-10); close(ra['Q'], rb['Q'], atol=1e-9)
    # chained (state_in from the previous launch) == stateless, bit for bit; a chain alone == the chain in its batch
    rc = ops.ft_trajectory(ra['x_new'].clone(), v, u, w, nl, beta, 0.05, 6, act, state_in=ra['state'].clone())
    rd = ops.ft_trajectory(ra['x_new'].clone(), v, u, w, nl, beta, 0.05, 6, act)
    for k in ('x_new', 'dH', 'H0', 'H1', 'acc', 'state', 'plaq', 'Q'):
        assert torch.equal(rc[k], rd[k]), k
    k = B - 1
    r1 = ops.ft_trajectory(xs[k:k + 1].contiguous(), v[k:k + 1].contiguous(), u[k:k + 1].contiguous(), w, nl, beta, 0.05, 6, act)
    for key in ('x_new', 'dH', 'H0', 'H1', 'acc'):
        assert torch.equal(r1[key][0], ra[key][k]), key


def test_small_lattice_path_against_the_oracle_and_in_a_graph():
    """Config-2 shape against the oracle (not only against the other kernels), and the one-launch trajectory captured
    into a hipGraph and replayed."""
    gen = torch.Generator().manual_seed(2)
    B, L, nl, beta, dt, nstep = 6, 16, 4, 4.0, 0.1, 10
    flow = R.default_flow(nl, gen)
    w = ops.pack_weights(flow, device='cuda')
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * 0.4).contiguous()
    v = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64)
    u = torch.rand(B, generator=gen, dtype=torch.float64)
    assert ops.get_small_path()
    r = ops.ft_trajectory(x.cuda(), v.cuda(), u.cuda(), w, nl, beta, dt, nstep)
    dH, _, acc, newx, h0, h1 = R.ft_hmc(x, v, u, flow, beta, dt, nstep, mode='md')
    close(r['H0'], h0, rtol=1e-11); close(r['H1'], h1, rtol=1e-8); close(r['dH'], dH, rtol=1e-6, atol=1e-7)
    assert np.array_equal(H(r['acc']) > 0.5, H(acc))
    angle_close(r['x_new'], newx, atol=1e-7)
    close(ops.ft_force(x.cuda(), w, nl, beta), R.ft_force(x, flow, beta), rtol=1e-9, atol=1e-9)
    # graph capture: workspace warmed by the eager call above
    xd, vd, ud = x.cuda(), v.cuda(), u.cuda()
    out = {k: torch.empty_like(t) for k, t in r.items()}
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ops.ft_trajectory(xd, vd, ud, w, nl, beta, dt, nstep, out=out)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with capture(g, st):
            ops.ft_trajectory(xd, vd, ud, w, nl, beta, dt, nstep, out=out)
        for t in out.values():
            t.zero_()
        g.replay()
        st.synchronize()
    for k in ('x_new', 'dH', 'H0', 'H1', 'acc', 'plaq', 'Q'):
        assert torch.equal(out[k], r[k]), k


# ---------------------------------------------------------------- row-strip leapfrog kernel (L % 64 == 0)
@pytest.mark.parametrize('B,L', [(3, 64), (2, 128), (1, 192)])
def test_row_strip_leapfrog_kernel(B, L):
    """k_leap_rows (16-byte accesses, two sites per thread; serves L % 64 == 0) == the 16 x 16-tile kernel bit for bit
    (same arithmetic per site), == the oracle's leapfrog to round-off, and reversible."""
    gen = torch.Generator().manual_seed(64 + L)
    beta, dt, nstep = 3.0, 0.1, 5
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi)
    p = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64)
    xa, pa = ops.leapfrog(x.cuda(), p.cuda(), beta, dt, nstep)
    os.environ['FTHMC_LEAP_ROWS'] = '0'
    try:
        ops.set_variant(1)
        xb, pb = ops.leapfrog(x.cuda(), p.cuda(), beta, dt, nstep)
    finally:
        os.environ['FTHMC_LEAP_ROWS'] = '1'
        ops.set_variant(1)
    assert torch.equal(xa, xb) and torch.equal(pa, pb)
    xr, pr = R.leapfrog(x, p, lambda y: R.wilson_force_analytic(y, beta), dt, nstep)
    close(xa, xr, rtol=1e-11, atol=1e-11); close(pa, pr, rtol=1e-11, atol=1e-11)
    x2, p2 = ops.leapfrog(xa, -pa, beta, dt, nstep)
    close(x2, x, rtol=1e-10, atol=1e-10); close(-p2, p, rtol=1e-10, atol=1e-10)


# ---------------------------------------------------------------- any s/t net shape (csrc/flow_generic.hip)
@pytest.mark.parametrize('hidden,k,n_mix,L,nl,act', [((8, 8), 3, 3, 8, 4, 'silu'), ((4, 6, 5), 5, 1, 12, 3, 'silu'), ((16,), 3, 2, 16, 2, 'relu'),
                                                     ((), 3, 2, 8, 8, 'silu'), ((8, 8), 1, 2, 8, 2, 'leaky_relu'), ((12, 12), 3, 4, 20, 2, 'silu')])
def test_generic_net_shapes_against_the_oracle(hidden, k, n_mix, L, nl, act):
    """hidden_sizes / kernel_size / n_mixture_comps other than the reference default (fthmc/utils/layers.py:138-167, 399-429
    accept any): every flowed entry point against the oracle -- layer forward / VJP / weight gradient / reverse, the sweeps,
    S_eff, ft_force, an MD trajectory, the training gradient -- and the default shape still takes the tuned kernels."""
    gen = torch.Generator().manual_seed(500 + L + k + n_mix + len(hidden))
    B, beta = 3, 2.5
    flow = R.default_flow(nl, gen, hidden=hidden, n_mix=n_mix, k=k)
    w = ops.pack_weights(flow, device='cuda')
    assert ops.arch_of(w) == (tuple(hidden), k, n_mix) and w.numel() == nl * ops.arch_params(ops.arch_of(w))
    x = (torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi
    xd = x.cuda()
    # one layer: forward, VJP, weight gradient, reverse -- all (mu, off) of the first layers
    for li in range(min(nl, 3)):
        mu, off = li % 2, (li // 2) % 4
        wl = ops.pack_weights([flow[li]], device='cuda')
        y, lj = ops.flow_layer_fwd(xd, wl, mu, off, act)
        xr_ = x.clone().requires_grad_(True)
        wr_ = [t.clone().requires_grad_(True) for t in flow[li]]
        yc, ljc = R.layer_forward(xr_, wr_, mu, off, act)
        angle_close(y, yc.detach(), atol=1e-11); close(lj, ljc.detach(), rtol=1e-11, atol=1e-11)
        c = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64); dlog = torch.randn(B, generator=gen, dtype=torch.float64)
        ((yc * c).sum() + (ljc * dlog).sum()).backward()
        gx, gw = ops.flow_layer_bwd(xd, wl, c.cuda(), dlog.cuda(), mu, off, act, need_gw=True)
        close(gx, xr_.grad, rtol=1e-9, atol=1e-10 * float(xr_.grad.abs().max()))
        for g_, t_ in zip(ops.unpack_weight_grads(gw, 1)[0], wr_):
            close(g_, t_.grad, rtol=1e-9, atol=1e-10 * max(1.0, float(t_.grad.abs().max())))
        xb_, ljb = ops.flow_layer_rev(y, wl, mu, off, act)
        angle_close(xb_, xd, atol=1e-9); close(ljb, -lj, atol=1e-8)
    # the flow
    y, ld = ops.flow_forward(xd, w, nl, act)
    yc, ldc = R.flow_forward(x, flow, act)
    angle_close(y, yc, atol=1e-10); close(ld, ldc, rtol=1e-10, atol=1e-10)
    xb_, ldb = ops.flow_reverse(y, w, nl, act)
    angle_close(xb_, xd, atol=1e-8); close(ldb, -ld, atol=1e-7)
    S, ld2, plq, Q = ops.ft_action(xd, w, nl, beta, act)
    close(S, R.ft_action(x, flow, beta, act), rtol=1e-11, atol=1e-10); close(plq, R.plaq_mean(yc, beta), rtol=1e-10); close(Q, R.charge(yc), atol=1e-9)
    F = ops.ft_force(xd, w, nl, beta, act)
    Fc = R.ft_force(x, flow, beta, act)
    close(F, Fc, rtol=1e-9, atol=1e-10 * float(Fc.abs().max()))
    assert torch.equal(F, ops.ft_force(xd, w, nl, beta, act))
    v = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64); u = torch.rand(B, generator=gen, dtype=torch.float64)
    xs = 0.3 * x
    r = ops.ft_trajectory(xs.cuda(), v.cuda(), u.cuda(), w, nl, beta, 0.05, 4, act)
    dH, _, acc, newx, h0, h1 = R.ft_hmc(xs, v, u, flow, beta, 0.05, 4, act=act, mode='md')
    close(r['H0'], h0, rtol=1e-11); close(r['H1'], h1, rtol=1e-9); close(r['dH'], dH, rtol=1e-6, atol=1e-7)
    assert np.array_equal(H(r['acc']) > 0.5, H(acc))
    angle_close(r['x_new'], newx, atol=1e-8)
    outc, gc = R.train_grads(x, flow, beta, act)
    tr = ops.train_grad(xd, w, nl, beta, act)
    close(tr['logq'], outc['logq'], rtol=1e-10); close(tr['logp'], outc['logp'], rtol=1e-10)
    for li, row in enumerate(ops.unpack_weight_grads(tr['gw'], nl)):
        for g_, t_ in zip(row, gc[li]):
            close(g_, t_, rtol=1e-8, atol=1e-10 * max(1.0, float(t_.abs().max())))
    # plaquette-level map
    P = (torch.rand(B, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi
    fP, lj = ops.plaq_coupling_fwd(P.cuda(), ops.pack_weights([flow[0]], device='cuda'), 0, 0, act)
    fPc, ljc = R.plaq_coupling_forward(P, flow[0], 0, 0, act)
    angle_close(fP, fPc, atol=1e-11); close(lj, ljc, rtol=1e-11, atol=1e-11)
    Pb, _ = ops.plaq_coupling_rev(fP, ops.pack_weights([flow[0]], device='cuda'), 0, 0, act)
    angle_close(Pb, P, atol=1e-9)
    # a default-shaped flow right behind it: the shape is an argument of each call, the library remembers nothing
    fd = R.default_flow(2, gen)
    wd = ops.pack_weights(fd, device='cuda')
    xq = (torch.rand(2, 2, 8, 8, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi
    close(ops.ft_force(xq.cuda(), wd, 2, beta), R.ft_force(xq, fd, beta), rtol=1e-10, atol=1e-10)
    # the shape given explicitly to a PLAIN tensor (a clone drops pack_weights' tag) == the tagged call
    wp = w.clone()
    assert ops.arch_of(wp) == ops.DEFAULT_ARCH
    assert torch.equal(ops.ft_force(x.cuda(), wp, nl, beta, act, arch=(hidden, k, n_mix)), ops.ft_force(x.cuda(), w, nl, beta, act))


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_net_shapes_golden_through_the_reference_api(tag):
    """The same through the reference-shaped Python API on goldens of the reference itself: make_u1_equiv_layers with
    other hidden_sizes / kernel_size / n_mixture_comps, qed.ft_flow / ft_action / ft_force, train_step."""
    from fthmc.config import FlowModel, Param, TrainConfig
    from fthmc.train import get_model, train_step
    import fthmc.utils.layers as layers
    import fthmc.utils.qed_helpers as qed
    g = load_golden(f'netshape_{tag}')
    hidden, k, n_mix, act, beta = [int(h) for h in g['hidden']], int(g['kernel_size']), int(g['n_mix']), str(g['act']), float(g['beta'])
    L, nl = g['x'].shape[-1], int(g['n_layers'])

    def load(flow, prefix):
        sd = {}
        for li in range(nl):
            for pi in range(2 * (len(hidden) + 1)):
                sd[f'{li}.plaq_coupling.net.{2 * (pi // 2)}.{"weight" if pi % 2 == 0 else "bias"}'] = D(g[f'{prefix}{li}_{pi}'])
        flow.load_state_dict(sd)                              # the reference's state_dict keys: Conv2d at even indices
    flow = layers.make_u1_equiv_layers(n_layers=nl, n_mixture_comps=n_mix, lattice_shape=(L, L), hidden_sizes=hidden,
                                       kernel_size=k, activation_fn=act)
    load(flow, 'w')
    param = Param(beta=beta, L=L)
    x = D(g['x'])
    angle_close(qed.ft_flow(flow, x), g['y'], atol=1e-10)
    close(qed.ft_action(param, flow, x), g['S_eff'], rtol=1e-11, atol=1e-10)
    close(qed.ft_force(param, flow, x), g['ft_force'], rtol=1e-8, atol=1e-10)
    xb = qed.ft_flow_inv(flow, D(g['y']))
    angle_close(xb, g['x'], atol=1e-8)
    B = g['xi'].shape[0]
    tc = TrainConfig(L=L, beta=beta, n_layers=nl, batch_size=B, base_lr=1e-3, hidden_sizes=hidden, kernel_size=k, n_s_nets=n_mix,
                     activation_fn=act)
    model = get_model(tc)
    load(model.layers, 'tw')
    for fused in (True, False):
        opt = torch.optim.SGD(model.layers.parameters(), lr=0.0)
        opt.zero_grad(set_to_none=True)
        met = train_step(model, tc, qed.BatchAction(beta), opt, B, xi=D(g['xi']), fused=fused)
        close(met['loss_dkl'], g['loss_dkl'], rtol=1e-10); close(met['ess'], g['ess'], rtol=1e-8)
        for li in range(nl):
            for pi, p in enumerate(model.layers[li].parameters()):
                close(p.grad, g[f'tgw{li}_{pi}'], rtol=1e-8, atol=1e-11)


# ---------------------------------------------------------------- autograd route: the backward reads the forward's stash
@pytest.mark.parametrize('hidden,k,n_mix', [((8, 8), 3, 2), ((6,), 3, 3)])
def test_layer_backward_from_the_forwards_stash(hidden, k, n_mix):
    """fthmc_flow_layer_fwd_stash + fthmc_flow_layer_bwd_stash (what the autograd bridge of GaugeEquivCouplingLayer uses:
    the layer is not run a second time inside its backward) == fthmc_flow_layer_bwd bit for bit; the VALU variant, which
    has no stash, falls back to it."""
    gen = torch.Generator().manual_seed(91)
    B, L = 3, 24
    flow = R.default_flow(4, gen, hidden=hidden, n_mix=n_mix, k=k)
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    gy = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64).cuda()
    gl = torch.randn(B, generator=gen, dtype=torch.float64).cuda()
    for li in range(4):
        mu, off = li % 2, (li // 2) % 4
        w = ops.pack_weights([flow[li]], device='cuda')
        y0, lj0 = ops.flow_layer_fwd(x, w, mu, off)
        y, lj, stash = ops.flow_layer_fwd_stash(x, w, mu, off)
        assert stash is not None and torch.equal(y, y0) and torch.equal(lj, lj0)
        for need_gw in (False, True):
            gx0, gw0 = ops.flow_layer_bwd(x, w, gy, gl, mu, off, need_gw=need_gw)
            gx, gw = ops.flow_layer_bwd_stash(stash, x.shape, w, gy, gl, mu, off, need_gw=need_gw)
            assert torch.equal(gx, gx0) and (not need_gw or torch.equal(gw, gw0))
    if tuple(hidden) == (8, 8):
        ops.set_variant(0)
        try:
            w = ops.pack_weights([flow[0]], device='cuda')
            assert ops.flow_layer_fwd_stash(x, w, 0, 0)[2] is None
        finally:
            ops.set_variant(1)
    # through autograd: a two-layer composition, gradients wrt x and wrt the weights against the oracle
    from fthmc_amd.utils import layers as Lyr
    nets = Lyr.make_u1_equiv_layers(n_layers=2, n_mixture_comps=n_mix, lattice_shape=(L, L), hidden_sizes=list(hidden),
                                    kernel_size=k, activation_fn='silu')
    names = [f'net.{2 * i}.{n}' for i in range(len(hidden) + 1) for n in ('weight', 'bias')]
    nets.load_state_dict({f'{li}.plaq_coupling.{n}': flow[li][pi].cuda() for li in range(2) for pi, n in enumerate(names)})
    xr = x.clone().requires_grad_(True)
    y1, l1 = nets[0](xr); y2, l2 = nets[1](y1)
    loss = (y2 * gy).sum() + ((l1 + l2) * gl).sum()
    loss.backward()
    xc = x.cpu().clone().requires_grad_(True)
    wc = [[t.clone().requires_grad_(True) for t in flow[li]] for li in range(2)]
    a1, b1 = R.layer_forward(xc, wc[0], 0, 0); a2, b2 = R.layer_forward(a1, wc[1], 1, 0)
    ((a2 * gy.cpu()).sum() + ((b1 + b2) * gl.cpu()).sum()).backward()
    close(xr.grad, xc.grad, rtol=1e-9, atol=1e-10 * float(xc.grad.abs().max()))
    for li in range(2):
        for p, t in zip(nets[li].parameters(), wc[li]):
            close(p.grad, t.grad, rtol=1e-8, atol=1e-10 * max(1.0, float(t.grad.abs().max())))


def test_device_side_run_statistics():
    """RunStats.add_device (one launch of fthmc_stats_accumulate) == RunStats.add (the stacked torch reduction), and qold moves on."""
    from fthmc_amd import parallel as P
    gen = torch.Generator().manual_seed(17)
    B = 37
    a, b = P.RunStats.zeros('cuda'), P.RunStats.zeros('cuda')
    qold = torch.randint(-3, 4, (B,), generator=gen).double().cuda()
    qa = qold.clone()
    for _ in range(3):
        acc = (torch.rand(B, generator=gen) < 0.5).double().cuda()
        plaq = torch.rand(B, generator=gen, dtype=torch.float64).cuda()
        q = torch.randint(-3, 4, (B,), generator=gen).double().cuda()
        dh = torch.randn(B, generator=gen, dtype=torch.float64).cuda()
        a.add(acc, plaq, q, q - qa, dh); qa = q.clone()
        b.add_device(acc, plaq, q, qold, dh)
        assert torch.equal(qold, q)
    close(b.vec, a.vec, rtol=1e-13, atol=1e-12)
    assert float(b.vec[0]) == 3 * B


# ---------------------------------------------------------------- the sampler at an operating point with a TRAINED flow
def test_sampler_is_exact_with_a_trained_flow():
    """bench.py times a random-init flow (the workload prescribes it).  Here the flow is trained with the reference-shaped
    loop (train -> train_step -> fthmc_train_grad) until the MD visibly changes, and ftHMC must still be an exact sampler:
    <exp(-dH)> = 1 and <cos P> of the flowed field = I1(beta) / I0(beta) (config.PLAQ_EXACT; finite-volume correction
    (I1/I0)^64 ~ 1e-10), errors from the 256 independent chains.  The trained weights (larger than any initialisation)
    also go through force and action against the oracle."""
    from fthmc_amd import parallel
    from fthmc_amd.config import PLAQ_EXACT, TrainConfig
    from fthmc_amd.train import get_model, train
    from fthmc_amd.utils.layers import net_weights
    L, beta, nl, B, nstep, therm, ntraj = 8, 2.0, 8, 256, 40, 60, 60
    torch.manual_seed(1331)
    cfg = TrainConfig(L=L, beta=beta, n_layers=nl, batch_size=512, n_era=1, n_epoch=600, base_lr=1e-3, print_freq=0)
    model = get_model(cfg)
    w0 = ops.pack_weights([net_weights(l.plaq_coupling.net) for l in model.layers], device='cuda')
    hist = train(cfg, model=model, verbose=False, save=False)['history']
    ess = [float(e) for e in hist['ess']]
    assert np.mean(ess[-30:]) > 2 * np.mean(ess[:30]), (np.mean(ess[:30]), np.mean(ess[-30:]))     # it did train
    layers = [net_weights(l.plaq_coupling.net) for l in model.layers]
    w = ops.pack_weights(layers, device='cuda')
    assert float((w - w0).abs().max()) > 0.05

    # trained weights against the oracle
    flow_cpu = [[p.detach().cpu() for p in lw] for lw in layers]
    xs = ((torch.rand(2, 2, L, L, generator=torch.Generator().manual_seed(4), dtype=torch.float64) * 2 - 1) * math.pi)
    S = ops.ft_action(xs.cuda(), w, nl, beta)[0]
    close(S, R.ft_action(xs, flow_cpu, beta), rtol=1e-10)
    close(ops.ft_force(xs.cuda(), w, nl, beta), R.ft_force(xs, flow_cpu, beta), rtol=1e-8, atol=1e-8)

    def run(w_, nst):
        g0, _ = ops.random_momenta(parallel.chain_seeds(7, 0, B, 0).cuda(), (B, 2, L, L), need_u=False)
        x = (0.1 * torch.erf(g0 / math.sqrt(2.0))).contiguous()
        acc = torch.zeros(B, dtype=torch.float64, device='cuda'); em = torch.zeros_like(acc); pl = torch.zeros_like(acc)
        for it in range(therm + ntraj):
            v, u = ops.random_momenta(parallel.chain_seeds(11, 0, B, it).cuda(), (B, 2, L, L))
            r = ops.ft_trajectory(x, v, u, w_, nl, beta, 1.0 / nst, nst)
            x = r['x_new']
            if it >= therm:
                acc += r['acc']; em += torch.exp(-r['dH']); pl += r['plaq']
        stat = lambda t: (float((t / ntraj).mean()), float((t / ntraj).std() / math.sqrt(B)))
        return stat(acc), stat(em), stat(pl)
    a0, _, _ = run(w0, 10)
    a1, _, _ = run(w, 10)
    assert a0[0] > 0.9 and a1[0] < 0.5, (a0, a1)          # the trained map stiffens the MD at the untrained step size ...
    acc, em, pl = run(w, nstep)
    assert acc[0] > 0.6, acc                               # ... and a finer step restores the acceptance
    assert abs(em[0] - 1.0) < 5 * em[1] and em[1] < 0.1, em
    assert abs(pl[0] - PLAQ_EXACT[beta]) < 5 * pl[1] and pl[1] < 2e-3, (pl, PLAQ_EXACT[beta])
