"""GPU tests of the `torch.ops.fthmc_hip.*` operator boundary (fthmc_amd/torch_ops.py) against the
reference's golden vectors: values, autograd through the registered formulas, schema checks."""
import numpy as np
import pytest
import torch

from conftest import golden_flow, load_golden

pytestmark = pytest.mark.gpu


def D(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64).copy()).cuda()


def close(a, b, rtol=1e-9, atol=1e-10):
    np.testing.assert_allclose(a.detach().cpu().numpy() if torch.is_tensor(a) else a, np.asarray(b), rtol=rtol, atol=atol)


def packed(g):
    from fthmc_amd import ops
    return ops.pack_weights(golden_flow(g), device='cuda')


@pytest.fixture(scope='module')
def T():
    import fthmc_amd.torch_ops as t
    return t


def test_wilson_ops_match_known_answers(T):
    g = load_golden('known_answer')
    x, beta = D(g['x']), float(g['beta'])
    S, Q, plaq = torch.ops.fthmc_hip.wilson_action_charge(x, beta)
    close(S, g['S']); close(Q, g['Q'], atol=1e-12); close(plaq, g['plaq'])
    close(torch.ops.fthmc_hip.wilson_force(x, beta), g['F'])
    # autograd through the registered formula: d(sum S)/dx is the force, plaq = -S / (beta L^2)
    xr = x.clone().requires_grad_(True)
    S, _, plaq = torch.ops.fthmc_hip.wilson_action_charge(xr, beta)
    (S.sum() + 3.0 * plaq.sum()).backward()
    close(xr.grad, np.asarray(g['F']) * (1.0 - 3.0 / (beta * 64)), rtol=1e-9, atol=1e-11)


def test_hmc_trajectory_op(T):
    g = load_golden('hmc_L8_n10')
    xn, dH, acc = torch.ops.fthmc_hip.hmc_trajectory(D(g['x']), D(g['v']), D(g['u']), float(g['beta']),
                                                     float(g['dt']), int(g['nstep']))
    close(dH, g['dH'], rtol=1e-8, atol=1e-9); close(acc, g['acc'], atol=0); close(xn, g['newx'], atol=1e-9)


@pytest.mark.parametrize('name,act', [('layers_L8_silu', 0), ('layers_L8_relu', 1), ('layers_L8_leaky_relu', 2)])
def test_layer_ops_values_and_autograd(T, name, act):
    g = load_golden(name)
    flow = golden_flow(g)
    from fthmc_amd import ops
    for li in range(int(g['n_layers'])):
        mu, off = li % 2, (li // 2) % 4
        w = ops.pack_weights([flow[li]], device='cuda').reshape(-1)
        x = D(g[f'x{li}'])
        y, lj = torch.ops.fthmc_hip.flow_layer_fwd(x, w, mu, off, 2, act)
        close(y, g[f'y{li}']); close(lj, g[f'logJ{li}'])
        gy, gl = D(g[f'c{li}']), D(g[f'd{li}'])
        close(torch.ops.fthmc_hip.flow_layer_bwd_x(x, gy, gl, w, mu, off, 2, act), g[f'gx{li}'], atol=1e-11)
        gw = torch.ops.fthmc_hip.flow_layer_bwd_w(x, gy, gl, w, mu, off, 2, act)
        ref = np.concatenate([np.asarray(g[f'gw{li}_{pi}']).reshape(-1) for pi in range(6)])
        close(gw, ref, atol=1e-11)
        # the same numbers through torch autograd on the operator
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y, lj = torch.ops.fthmc_hip.flow_layer_fwd(xr, wr, mu, off, 2, act)
        ((gy * y).sum() + (gl * lj).sum()).backward()
        close(xr.grad, g[f'gx{li}'], atol=1e-11); close(wr.grad, ref, atol=1e-11)
        # inverse operator: reverse(forward(x)) = x up to the root-finder tolerance, logJ_rev = -logJ
        xb, ljb = torch.ops.fthmc_hip.flow_layer_rev(y.detach(), w, mu, off, 2, act, 1e-13)
        d = (xb - x + np.pi) % (2 * np.pi) - np.pi
        assert float(d.abs().max()) < 1e-9
        close(ljb, -np.asarray(g[f'logJ{li}']), rtol=1e-8, atol=1e-9)


def test_ft_action_force_and_trajectory_ops(T):
    g = load_golden('known_answer')
    w = packed(g).reshape(-1)
    S, ld, F = torch.ops.fthmc_hip.ft_action_force(D(g['x']), w, int(g['n_layers']), float(g['beta']), 0)
    close(S, g['S_eff']); close(F, g['ft_force'], atol=1e-11)
    close(ld, np.asarray(g['logJ0']) + np.asarray(g['logJ1']))
    for name, mode in (('traj_md_L8', 0), ('traj_literal_L8', 1)):
        t = load_golden(name)
        xn, dH, acc, plaq, Q = torch.ops.fthmc_hip.fthmc_trajectory(
            D(t['x']), D(t['v']), D(t['u']), packed(t).reshape(-1), int(t['n_layers']), float(t['beta']),
            float(t['dt']), int(t['nstep']), mode, 0)
        close(dH, t['dH'], rtol=1e-7, atol=1e-8); close(acc, t['acc'], atol=0)
        close(xn, t['newx'], atol=1e-8); close(plaq, t['plaq'], rtol=1e-8); close(Q, t['Q'], atol=1e-9)


def test_train_grad_op_matches_reference(T):
    g = load_golden('train_L8')
    nl = int(g['n_layers'])
    x, logq, logp, gw = torch.ops.fthmc_hip.train_grad(D(g['xi']), packed(g).reshape(-1), nl, float(g['beta']), 0)
    loss = (logq - logp).mean()
    close(loss, g['loss_dkl'], rtol=1e-9)
    from fthmc_amd import ops
    close(logq, g['logq'], rtol=1e-11); close(logp, g['logp'], rtol=1e-11)
    grads = ops.unpack_weight_grads(gw, nl)
    for li in range(nl):
        for pi in range(6):
            close(grads[li][pi], g[f'gw{li}_{pi}'], rtol=1e-8, atol=1e-12)


def test_ops_reject_cpu_tensors_and_bad_codes(T):
    x = torch.zeros(1, 2, 8, 8, dtype=torch.float64)
    with pytest.raises(NotImplementedError):
        torch.ops.fthmc_hip.wilson_force(x, 1.0)
    xc = x.cuda(); w = torch.zeros(955, dtype=torch.float64, device='cuda')
    with pytest.raises(RuntimeError):                                  # (FthmcError from the Python registration, c10::Error from the compiled one)
        torch.ops.fthmc_hip.flow_layer_fwd(xc, w, 0, 0, 3, 0)          # three mixture components need 1028 weights
    with pytest.raises(RuntimeError):
        torch.ops.fthmc_hip.flow_layer_fwd(xc, w, 0, 0, 2, 7)          # unknown activation
    with pytest.raises(RuntimeError):
        torch.ops.fthmc_hip.flow_layer_fwd(xc.float(), w, 0, 0, 2, 0)  # the HIP path computes in float64


def test_opcheck_schema_and_fake(T):
    g = load_golden('layers_L8_silu')
    from fthmc_amd import ops
    w = ops.pack_weights([golden_flow(g)[0]], device='cuda').reshape(-1)
    x = D(g['x0'])
    torch.library.opcheck(torch.ops.fthmc_hip.flow_layer_fwd.default, (x, w, 0, 0, 2, 0),
                          test_utils=('test_schema', 'test_faketensor'))
    torch.library.opcheck(torch.ops.fthmc_hip.wilson_force.default, (x, 2.0),
                          test_utils=('test_schema', 'test_faketensor'))


_OPS_WORKER = r'''
import os, sys, json, hashlib, math
sys.path.insert(0, os.environ['FTHMC_ROOT']); sys.path.insert(0, os.path.join(os.environ['FTHMC_ROOT'], 'tests'))
import numpy as np, torch
import fthmc_amd.torch_ops as T
from fthmc_amd import ops
from oracle import ref_cpu as R
gen = torch.Generator().manual_seed(5)
out = {'backend': T.BACKEND}
def h(*ts): return hashlib.sha256(b''.join(t.detach().cpu().numpy().tobytes() for t in ts)).hexdigest()
for tag, hidden, k, n_mix in (('default', None, 3, 2), ('generic', [4, 6, 5], 5, 1)):
    flow = R.default_flow(2, gen, hidden=tuple(hidden) if hidden else (8, 8), n_mix=n_mix, k=k)
    w = torch.cat([t.reshape(-1) for lw in flow for t in lw]).cuda()
    x = ((torch.rand(3, 2, 12, 12, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    v = torch.randn(3, 2, 12, 12, generator=gen, dtype=torch.float64).cuda(); u = torch.rand(3, generator=gen, dtype=torch.float64).cuda()
    npl = w.numel() // 2
    y, lj = torch.ops.fthmc_hip.flow_layer_fwd(x, w[:npl], 1, 2, n_mix, 0, hidden, k)
    gx, gw = torch.ops.fthmc_hip.flow_layer_bwd(x, v, u, w[:npl], 1, 2, n_mix, 0, hidden, k)
    S, ld, F = torch.ops.fthmc_hip.ft_action_force(x, w, 2, 2.0, 0, n_mix, hidden, k)
    tr = torch.ops.fthmc_hip.fthmc_trajectory(x, v, u, w, 2, 2.0, 0.1, 3, 0, 0, n_mix, hidden, k)
    tg = torch.ops.fthmc_hip.train_grad(x, w, 2, 2.0, 0, n_mix, hidden, k)
    xr, wr = x.clone().requires_grad_(True), w[:npl].clone().requires_grad_(True)
    yy, ll = torch.ops.fthmc_hip.flow_layer_fwd(xr, wr, 0, 1, n_mix, 0, hidden, k)
    ((yy * v).sum() + (ll * u).sum()).backward()
    out[tag] = h(y, lj, gx, gw, S, ld, F, *tr, *tg, xr.grad, wr.grad)
    Fc = R.ft_force(x.cpu(), flow, 2.0)
    out[tag + '_force_err'] = float((F.cpu() - Fc).abs().max())
print(json.dumps(out))
'''


def test_compiled_and_python_registrations_agree():
    """torch.ops.fthmc_hip.* from the compiled TORCH_LIBRARY (libfthmc_torch.so, csrc/torch_library.cpp) and from the Python
    registration over ctypes (FTHMC_TORCH_OPS=python): same schemas, bit-identical results -- layer forward / backward, S_eff and
    force, a trajectory, the training gradient, autograd through the layer operator -- for the default net shape and for one
    given through the schema's `hidden` / `kernel_size` arguments; both against the oracle's force."""
    import json, os, subprocess, sys
    from conftest import ROOT
    import fthmc_amd.torch_ops as T
    assert os.path.exists(os.path.join(ROOT, 'fthmc_amd', 'libfthmc_torch.so')), 'make -C fthmc_amd/csrc builds it'
    assert T.BACKEND == 'compiled'
    res = {}
    for mode in ('compiled', 'python'):
        env = dict(os.environ, FTHMC_ROOT=ROOT, FTHMC_TORCH_OPS=mode)
        p = subprocess.run([sys.executable, '-c', _OPS_WORKER], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        res[mode] = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
    assert res['compiled']['backend'] == 'compiled' and res['python']['backend'] == 'python'
    for tag in ('default', 'generic'):
        assert res['compiled'][tag] == res['python'][tag], tag
        assert res['compiled'][tag + '_force_err'] < 1e-10
