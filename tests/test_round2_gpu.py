"""GPU tests added in round 2: physical-field ftHMC wrapper and independence sampler against reference goldens,
plaquette-level coupling map, the BASELINE config-5 shard at its real size, workspace / graph-capture safety,
out-of-place inverse sweep, sharded chains and sharded training across two processes on one GPU."""
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


def build_layers(g, L, act='silu'):
    from fthmc_amd.utils import layers as Lyr
    nl = int(g['n_layers'])
    flow = Lyr.make_u1_equiv_layers(n_layers=nl, n_mixture_comps=2, lattice_shape=(L, L), hidden_sizes=[8, 8],
                                    kernel_size=3, activation_fn=act)
    names = ['net.0.weight', 'net.0.bias', 'net.2.weight', 'net.2.bias', 'net.4.weight', 'net.4.bias']
    flow.load_state_dict({f'{li}.plaq_coupling.{n}': D(g[f'w{li}_{pi}']) for li in range(nl) for pi, n in enumerate(names)})
    return flow


# ---------------------------------------------------------------- f1: ft_hmc on the physical field
@pytest.mark.parametrize('L', [8, 16])
def test_physical_field_fthmc_golden(L):
    """qed_helpers.ft_hmc (ipynb/ft_hmc.py:420-435) against the reference run with captured v, u.
    'tight' golden (reference bisection down to fp resolution): the HIP inverse (Newton to 1e-12) agrees to
    rounding amplified by the trajectory; 'ref' golden (the reference's 1e-6 inverse tolerance): agreement
    within what that tolerance allows."""
    from fthmc_amd.config import Param
    from fthmc_amd.utils import qed_helpers as qed
    for tag, tol_dH, tol_f in (('tight', 1e-7, 1e-7), ('ref', 5e-3, 5e-5)):
        g = load_golden(f'fthmc_phys_L{L}_{tag}')
        flow = build_layers(g, L)
        param = Param(beta=float(g['beta']), L=L, tau=float(g['dt']) * int(g['nstep']), nstep=int(g['nstep']))
        field = D(g['field'])
        angle_close(qed.ft_flow_inv(flow, field), g['x_inv'], atol=1e-9 if tag == 'tight' else 2e-6)
        dH, e, acc, newfield = qed.ft_hmc(param, flow, field, v=D(g['v']), u=D(g['u']))
        g_dH, g_e, g_u = (float(np.asarray(g[k]).reshape(-1)[0]) for k in ('dH', 'exp_mdH', 'u'))
        assert abs(dH - g_dH) < tol_dH * max(1.0, abs(g_dH)), (tag, dH, g_dH)
        assert abs(e - g_e) < 10 * tol_dH * max(1.0, g_e)
        if abs(g_u - g_e) > 1e-2:
            assert bool(acc) == bool(g['acc'])
            angle_close(newfield, g['newfield'], atol=tol_f)
        assert newfield.shape == field.shape
    # a configuration [2, L, L] (what ft_run passes around) and the run loop
    param.nrun, param.ntraj = 1, 3
    f1, hist = qed.ft_run(param, flow, field[0])
    assert f1.shape == field[0].shape and len(hist['dH']) == 3 and all(np.isfinite(hist['plaq']))
    assert all(abs(q - round(q)) < 1e-8 for q in hist['topo'])
    big = qed.flow_resize(flow, (2 * L, 2 * L))
    assert len(big) == len(flow) and big[0].plaq_coupling.net is flow[0].plaq_coupling.net


# ---------------------------------------------------------------- f3: independence sampler
def test_independence_sampler_golden():
    """utils.samplers.make_mcmc_ensemble against the recorded reference chain (samplers.py:182-259): same
    proposals + same uniforms -> identical accept sequence and histories; and the GPU proposal generator
    reproduces the reference's proposals (flowed sample, logq, both logp flavours)."""
    from fthmc_amd.utils import qed_helpers as qed
    from fthmc_amd.utils import samplers as S
    from fthmc_amd.utils.distributions import MultivariateUniform
    g = load_golden('sampler_L8')
    L, beta, bs = 8, float(g['beta']), int(g['batch_size'])
    layers = build_layers(g, L)
    xi = D(g['xi'])
    n = xi.shape[0]
    action = qed.BatchAction(beta)

    class FixedPrior(MultivariateUniform):
        """the reference's prior, replaying the recorded draws"""
        def __init__(self):
            super().__init__(-math.pi * torch.ones(2, L, L, dtype=torch.float64, device='cuda'),
                             math.pi * torch.ones(L, L, dtype=torch.float64, device='cuda'))
            self.k = 0

        def sample_n(self, b):
            out = xi[self.k:self.k + b]
            self.k += b
            return out
    for literal, kx, kp in ((True, 'xi', 'logp_xi'), (False, 'xflow', 'logp_flow')):
        model = {'layers': layers, 'prior': FixedPrior()}
        props = list(S.serial_sample_generator(model, action, bs, n, reference_literal=literal))
        angle_close(torch.stack([p[0] for p in props]), g[kx], atol=1e-10)
        close(torch.stack([p[1] for p in props]), g['logq'], rtol=1e-11)
        close(torch.stack([p[2] for p in props]), g[kp], rtol=1e-11)
    model = {'layers': layers, 'prior': FixedPrior()}
    h = S.make_mcmc_ensemble(model, action, bs, n, uniforms=list(g['u']), reference_literal=True)
    assert np.array_equal(h['acc'], g['hist_acc'])
    for k in ('q', 'dqsq', 'logq', 'logp'):
        close(h[k], g['hist_' + k], rtol=2e-7, atol=1e-6)          # the reference's histories are float32


# ---------------------------------------------------------------- plaquette-level coupling map
@pytest.mark.parametrize('L,B', [(8, 2), (20, 3), (64, 4)])
def test_plaq_coupling_layer(L, B):
    """NCPPlaqCouplingLayer.forward / .reverse (layers.py:348-396) on plaquette fields vs the oracle."""
    from fthmc_amd.utils import layers as Lyr
    gen = torch.Generator().manual_seed(90 + L)
    flow = R.default_flow(8, gen)
    P = (torch.rand(B, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi
    for li, wts in enumerate(flow):
        mu, off = li % 2, (li // 2) % 4
        w = ops.pack_weights([wts], device='cuda')
        fP, lj = ops.plaq_coupling_fwd(P.cuda(), w, mu, off)
        fPc, ljc = R.plaq_coupling_forward(P, wts, mu, off)
        angle_close(fP, fPc, atol=1e-11); close(lj, ljc, rtol=1e-11, atol=1e-11)
        Pb, ljb = ops.plaq_coupling_rev(fP, w, mu, off, tol=1e-13)
        angle_close(Pb, P, atol=1e-9); close(ljb, -lj, atol=1e-8)
        Pc, ljc2 = R.plaq_coupling_reverse(fPc, wts, mu, off, tol=1e-13)
        close(ljb, ljc2, atol=1e-7)
    # through the layer classes
    layers = Lyr.make_u1_equiv_layers(n_layers=2, n_mixture_comps=2, lattice_shape=(L, L), hidden_sizes=[8, 8], kernel_size=3)
    pc = layers[1].plaq_coupling
    fx, lj = pc.forward(P.cuda())
    xb, ljb = pc.reverse(fx)
    angle_close(xb, P, atol=1e-9); close(ljb, -lj, atol=1e-8)
    ops.set_variant(0)
    try:
        with pytest.raises(Exception, match='unsupported'):
            ops.plaq_coupling_fwd(P.cuda(), w, 0, 0)
    finally:
        ops.set_variant(1)


# ---------------------------------------------------------------- inverse sweep runs out of place
def test_flow_reverse_in_place_and_out_of_place_agree():
    gen = torch.Generator().manual_seed(12)
    B, L, nl = 5, 48, 8                                   # 9 tiles per chain: halos cross tile borders
    flow = R.default_flow(nl, gen)
    w = ops.pack_weights(flow, device='cuda')
    y = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    x1, ld1 = ops.flow_reverse(y, w, nl)
    # same call with the output aliasing the input (C ABI allows x == y)
    from fthmc_amd import _lib
    y2 = y.clone(); ld2 = torch.empty_like(ld1)
    wsb = torch.empty(ops.ws_bytes(B, L, nl) // 8 + 1, dtype=torch.float64, device='cuda')
    rc = _lib.load().fthmc_flow_reverse(y2.data_ptr(), w.data_ptr(), None, nl, B, L, 0, 1e-12, y2.data_ptr(), ld2.data_ptr(),
                                        wsb.data_ptr(), wsb.numel() * 8, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(y2, x1) and torch.equal(ld2, ld1)
    yb, _ = ops.flow_forward(x1, w, nl)
    angle_close(yb, y, atol=1e-9)


# ---------------------------------------------------------------- workspace vs captured graphs
def test_workspace_survives_growth_after_capture():
    """A graph captured on a stream keeps replaying correctly after a larger call on the same stream replaced
    that stream's workspace; and growing a workspace during capture is refused."""
    from fthmc_amd._lib import FthmcError
    gen = torch.Generator().manual_seed(3)
    nl, beta = 2, 2.0
    flow = R.default_flow(nl, gen)
    w = ops.pack_weights(flow, device='cuda')
    xs = ((torch.rand(2, 2, 16, 16, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    xl = ((torch.rand(8, 2, 32, 32, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        want = ops.ft_force(xs, w, nl, beta).clone()            # warm-up: allocates this stream's workspace
        st.synchronize()
        out = torch.empty_like(xs)
        graph = torch.cuda.CUDAGraph()
        with capture(graph, st):
            out.copy_(ops.ft_force(xs, w, nl, beta))
        with pytest.raises(FthmcError, match='graph capture'):
            g2 = torch.cuda.CUDAGraph()
            with capture(g2, st):
                ops.ft_force(xl, w, nl, beta)                     # would have to grow the workspace while capturing
        big = ops.ft_force(xl, w, nl, beta)                       # eager: the workspace is replaced, the old one retired
        big2 = ops.ft_force(xl, w, nl, beta)
        out.zero_()
        graph.replay()
        st.synchronize()
        assert torch.equal(out, want)
        assert torch.equal(big, big2)
    close(big, R.ft_force(xl.cpu(), flow, beta), rtol=1e-8, atol=1e-9)


# ---------------------------------------------------------------- BASELINE config 5 shard at its real size
def test_config5_shard_properties_and_oracle():
    """Config 5 per-GPU shard (B=32, L=256, 16 layers, beta=7): size-independent properties at the full shard,
    and the oracle on one of its chains for ft_force and the training gradient."""
    gen = torch.Generator().manual_seed(555)
    B, L, nl, beta = 32, 256, 16, 7.0
    flow = R.default_flow(nl, gen)
    w = ops.pack_weights(flow, device='cuda')
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    y, ld = ops.flow_forward(x, w, nl)
    Q = ops.wilson_action_charge(y, beta)[1]
    assert float((Q - Q.round()).abs().max()) < 1e-7
    xb, ldb = ops.flow_reverse(y, w, nl, tol=1e-13)
    angle_close(xb, x, atol=1e-8); close(ldb, -ld, atol=1e-5)
    F = ops.ft_force(x, w, nl, beta)
    assert torch.equal(F, ops.ft_force(x, w, nl, beta))          # deterministic
    d = torch.randn(B, 2, L, L, generator=gen, dtype=torch.float64).cuda()
    eps = 1e-5
    fd = (ops.ft_action(x + eps * d, w, nl, beta)[0] - ops.ft_action(x - eps * d, w, nl, beta)[0]) / (2 * eps)
    close((F * d).flatten(1).sum(1), fd, rtol=5e-5, atol=1e-2)
    # a chain's results do not depend on what else is in the batch
    F1 = ops.ft_force(x[5:6].contiguous(), w, nl, beta)
    assert torch.equal(F1[0], F[5])
    # oracle on chain 5: effective action, force, training gradient
    xc = x[5:6].cpu()
    Sg, ldg, pg, qg = ops.ft_action(x[5:6].contiguous(), w, nl, beta)
    yc, ldc = R.flow_forward(xc, flow)
    close(Sg, R.action(yc, beta) - ldc, rtol=1e-10)
    close(ldg, ldc, rtol=1e-9, atol=1e-7); close(pg, R.plaq_mean(yc, beta), rtol=1e-10); close(qg, R.charge(yc), atol=1e-7)
    close(F1, R.ft_force(xc, flow, beta), rtol=1e-7, atol=1e-8)
    outc, gc = R.train_grads(xc, flow, beta)
    r = ops.train_grad(x[5:6].contiguous(), w, nl, beta)
    close(r['logq'], outc['logq'], rtol=1e-10); close(r['logp'], outc['logp'], rtol=1e-10)
    gws = ops.unpack_weight_grads(r['gw'], nl)
    for li in range(nl):
        for pi in range(6):
            scale = float(gc[li][pi].abs().max())
            close(gws[li][pi], gc[li][pi], rtol=1e-7, atol=1e-9 * max(scale, 1.0))
    # a second chain of the shard against the oracle (effective action and force), from the other end of the batch
    xc2 = x[31:32].cpu()
    yc2, ldc2 = R.flow_forward(xc2, flow)
    close(ops.ft_action(x[31:32].contiguous(), w, nl, beta)[0], R.action(yc2, beta) - ldc2, rtol=1e-10)
    close(F[31:32], R.ft_force(xc2, flow, beta), rtol=1e-7, atol=1e-8)
    # full-shard training gradient = mean of per-chain gradients (chains 0..3 checked against their own calls)
    rb = ops.train_grad(x[:4].contiguous(), w, nl, beta)
    acc = sum(ops.train_grad(x[i:i + 1].contiguous(), w, nl, beta)['gw'] for i in range(4)) / 4
    close(rb['gw'], acc, rtol=1e-9, atol=1e-9)


# ---------------------------------------------------------------- two processes on one GPU
_WORKER = r'''
import os, sys, math
import numpy as np, torch
sys.path.insert(0, os.environ["FT_ROOT"])
from fthmc_amd import ops, parallel as P
import bench
rank, world, local = P.init()                       # FTHMC_DIST_BACKEND=gloo: both ranks share GPU 0
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
B, L, nl, beta, dt, nstep = 8, 16, 4, 4.0, 0.1, 5
gen = torch.Generator().manual_seed(5)
flow = bench.make_flow(gen, nl)
w = ops.pack_weights(flow, device=dev)
x_all = (torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi
xi_all = (torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi
lo, hi = P.shard_range(B, rank, world)
x = x_all[lo:hi].to(dev)
stats = P.RunStats.zeros(dev)
S, _, p, q = ops.ft_action(x, w, nl, beta)
state = torch.stack([S, p, q]).contiguous(); qold = q.clone()
dHs = []
for traj in range(3):
    seeds = P.chain_seeds(77, lo, hi, traj).to(dev)
    v, u = ops.random_momenta(seeds, x.shape)
    r = ops.ft_trajectory(x, v, u, w, nl, beta, dt, nstep, state_in=state)
    x = r["x_new"].clone(); state = r["state"].clone(); dHs.append(r["dH"].clone())
    stats.add(r["acc"], r["plaq"], r["Q"], r["Q"] - qold, r["dH"]); qold = r["Q"].clone()
    h = stats.reduce(async_op=world > 1)
    if h is not None: h.wait()
m = stats.means()
# training: train_step(fused=True) on this rank's share of a fixed prior draw (lr = 0: weights stay put)
from fthmc_amd import train as T
from fthmc_amd.config import TrainConfig
from fthmc_amd.utils import layers as Lyr, qed_helpers as qed
tc = TrainConfig(L=L, beta=beta, n_layers=nl, batch_size=hi - lo)
model = T.get_model(tc)
names = ["net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.weight", "net.4.bias"]
model.layers.load_state_dict({f"{li}.plaq_coupling.{n}": flow[li][pi].to(dev) for li in range(nl) for pi, n in enumerate(names)})
opt = torch.optim.SGD(model.layers.parameters(), lr=0.0)
met = T.train_step(model, tc, qed.BatchAction(beta), opt, hi - lo, xi=xi_all[lo:hi].to(dev), fused=True)
grads = torch.cat([p_.grad.reshape(-1) for p_ in model.layers.parameters()])
# the training LOOP object: every rank draws the prior of its own global chain ids on the device and steps the same weights
model2 = T.get_model(tc)
model2.layers.load_state_dict({f"{li}.plaq_coupling.{n}": flow[li][pi].to(dev) for li in range(nl) for pi, n in enumerate(names)})
tc2 = TrainConfig(L=L, beta=beta, n_layers=nl, batch_size=hi - lo, base_lr=1e-3)
tr = T.GraphTrainer(model2, tc2, T.make_optimizer(model2, tc2), hi - lo, seed=9)
for _ in range(3):
    tr.step()
mt = tr.metrics()
np.savez(os.environ["FT_OUT"] + f".{world}.{rank}.npz", lo=lo, hi=hi, x=x.cpu().numpy(), dH=torch.stack(dHs).cpu().numpy(),
         means=np.array([m[k] for k in sorted(m)]), grads=grads.cpu().numpy(), loss=met["loss_dkl"], ess=met["ess"],
         logq=met["logq"], logp=met["logp"], w_loop=Lyr.flow_weights(model2.layers).cpu().numpy(), loop_captured=tr.captured,
         loop_loss=mt["loss_dkl"], loop_ess=mt["ess"])
if world > 1:
    torch.distributed.destroy_process_group()
'''


def _run(world, out, extra_env=None):
    env = dict(os.environ, FT_ROOT=ROOT, FT_OUT=out, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29700 + world + os.getpid() % 200),
               FTHMC_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='2')
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, '-c', _WORKER], env=dict(env, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]
    return [dict(np.load(f'{out}.{world}.{r}.npz')) for r in range(world)]


def test_two_process_sharding_equals_one_process(tmp_path):
    """Chains [0, B/2) + [B/2, B) in two fresh processes (HIP trajectories, per-chain Philox seeds, C1 all-reduce over
    gloo) == all B chains in one process, bit for bit; train_step(fused=True) on two half-batches (C2: gradient
    all-reduce, global loss mean and ESS logsumexp) == one process on the full batch."""
    one = _run(1, str(tmp_path / 'a'))[0]
    two = _run(2, str(tmp_path / 'b'))
    x2 = np.concatenate([t['x'] for t in two]); dH2 = np.concatenate([t['dH'] for t in two], axis=1)
    assert [int(t['lo']) for t in two] == [0, 4] and int(two[1]['hi']) == 8
    assert np.array_equal(x2, one['x']) and np.array_equal(dH2, one['dH'])
    for t in two:                                                  # every rank holds the global statistics
        np.testing.assert_allclose(t['means'], one['means'], rtol=1e-13, atol=1e-13)
    assert one['means'][sorted(['n', 'acc', 'plaq', 'q', 'q2', 'absdq', 'dh', 'exp_mdh', 'chi_q']).index('n')] == 24.0
    # training
    np.testing.assert_allclose(np.concatenate([t['logq'] for t in two]), one['logq'], rtol=1e-13)
    np.testing.assert_allclose(np.concatenate([t['logp'] for t in two]), one['logp'], rtol=1e-13)
    for t in two:
        np.testing.assert_allclose(t['loss'], one['loss'], rtol=1e-12)
        np.testing.assert_allclose(t['ess'], one['ess'], rtol=1e-12)
        scale = np.abs(one['grads']).max()
        np.testing.assert_allclose(t['grads'], one['grads'], rtol=1e-10, atol=1e-12 * scale)
    assert np.abs(one['grads']).max() > 0
    # GraphTrainer: captured without a group; over gloo (host-side collectives cannot be captured) the eager sequence -- the same
    # weights, loss and ESS on both ranks as the one process on the full batch (the draws are keyed by the global chain id)
    assert bool(one['loop_captured']) and not any(bool(t['loop_captured']) for t in two)
    for t in two:
        np.testing.assert_allclose(t['w_loop'], one['w_loop'], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(t['loop_loss'], one['loop_loss'], rtol=1e-10)
        np.testing.assert_allclose(t['loop_ess'], one['loop_ess'], rtol=1e-9)


def test_bench_self_launches_its_ranks(tmp_path):
    """`python bench.py --gpus 2` without torchrun starts two ranks itself (gloo rehearsal on this one GPU) and
    reports n_gpus 2; --scaling strong splits a fixed total."""
    env = dict(os.environ, FTHMC_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    for scaling, total in (('weak', 32), ('strong', 16)):
        p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--config', '2', '--batch', '16',
                            '--scaling', scaling, '--steps', '2', '--warmup', '1', '--thermalize', '2', '--no-cpu-baseline'],
                           env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        line = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
        assert line['n_gpus'] == 2 and line['config']['chains_total'] == total and line['scaling'] == scaling
        assert line['value'] > 0 and line['roofline']['bound'] == 'mfma'


# ---------------------------------------------------------------- training loop: checkpoints, scheduler, transfer
def test_train_checkpoint_round_trip_and_scheduler(tmp_path):
    """train(save=True) writes one reference-format .tar per era (io.py:114-172) holding numpy histories;
    restore_model_from_checkpoint (train.py:77-92) reads it back exactly; ReduceLROnPlateau (train.py:314-317)
    steps on the loss; transfer_to_new_lattice reuses the nets on 2L with the same per-site results."""
    from fthmc_amd import train as T
    from fthmc_amd.config import SchedulerConfig, TrainConfig
    tc = TrainConfig(L=8, beta=2.0, n_layers=4, batch_size=16, base_lr=5e-3, n_era=2, n_epoch=3, print_freq=0)
    tc.update_logdirs(str(tmp_path / 'run'))
    torch.manual_seed(11)
    sc = SchedulerConfig(factor=0.5, patience=0, threshold=1e9, threshold_mode='abs', min_lr=1e-6)   # every step "plateaus"
    out = T.train(tc, scheduler_config=sc, save=True, verbose=False)
    assert len(out['ckpt_files']) == 2 and os.path.basename(out['ckpt_files'][-1]) == 'ckpt-era1-epoch3.tar'
    assert out['optimizer'].param_groups[0]['lr'] < tc.base_lr                    # the scheduler stepped
    assert len(out['history']['loss_dkl']) == 6 and isinstance(out['history']['logp'][0], np.ndarray)
    raw = torch.load(out['ckpt_files'][-1], weights_only=False)
    assert set(raw) == {'era', 'epoch', 'model_state_dict', 'optimizer_state_dict', 'history'}
    assert '0.plaq_coupling.net.0.weight' in raw['model_state_dict'] and len(raw['history']['ess']) == 6
    back = T.restore_model_from_checkpoint(out['ckpt_files'][-1], tc)
    for (k, a), (k2, b) in zip(out['model'].layers.state_dict().items(), back['model'].layers.state_dict().items()):
        assert k == k2 and torch.equal(a, b)
    # Adam state (restore builds AdamW like the reference does: same state tensors)
    sa, sb = out['optimizer'].state_dict()['state'], back['optimizer'].state_dict()['state']
    assert len(sa) == len(sb) == 24 and all(torch.equal(sa[i]['exp_avg'], sb[i]['exp_avg']) for i in sa)
    # transfer: on a 16 x 16 lattice made of four copies of an 8 x 8 field the (translation-equivariant, periodic)
    # flow gives four copies of the 8 x 8 result and four times the log-det
    from fthmc_amd.utils import qed_helpers as qed
    big = T.transfer_to_new_lattice(16, out['model'].layers)
    x8 = out['model'].prior.sample_n(3)
    y8, ld8 = ops.flow_forward(x8, qed.flow_weights(out['model'].layers, x8.device), 4)
    x16 = x8.repeat(1, 1, 2, 2)
    y16, ld16 = ops.flow_forward(x16, qed.flow_weights(big.layers, x8.device), 4)
    angle_close(y16, y8.repeat(1, 1, 2, 2), atol=1e-10); close(ld16, 4 * ld8, rtol=1e-10)
    assert big.prior.sample_n(2).shape == (2, 2, 16, 16)
