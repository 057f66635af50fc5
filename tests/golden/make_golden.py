#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (/root/reference).

Runs only in the build container (the reference never travels to the GPU box).
Nothing from the reference is copied: the fixtures hold inputs (fields, momenta,
uniforms, conv weights) and the outputs the reference's own functions return.

    cd /root/repo && python tests/golden/make_golden.py

The reference needs three import-time stubs here: tensorboard's
SummaryWriter, IPython.display, and `np.float`.  fp64 is selected *after*
importing fthmc.config because that module resets the default dtype (Q1).
"""
import os
import sys
import tempfile
import types
import math

import numpy as np

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))

np.float = float  # removed in NumPy 2; the reference's plot helpers still use it


def _stub_modules():
    tb = types.ModuleType('torch.utils.tensorboard')
    tbw = types.ModuleType('torch.utils.tensorboard.writer')

    class SummaryWriter:  # no-op
        def __init__(self, *a, **k): pass
        def add_scalar(self, *a, **k): pass
        def add_histogram(self, *a, **k): pass
        def close(self): pass
    tb.SummaryWriter = SummaryWriter
    tbw.SummaryWriter = SummaryWriter
    tb.writer = tbw
    sys.modules['torch.utils.tensorboard'] = tb
    sys.modules['torch.utils.tensorboard.writer'] = tbw
    ip = types.ModuleType('IPython')
    ipd = types.ModuleType('IPython.display')

    class DisplayHandle:
        def __init__(self, *a, **k): pass
        def update(self, *a, **k): pass
    ipd.DisplayHandle = DisplayHandle
    ipd.display = lambda *a, **k: None
    ip.display = ipd
    ip.get_ipython = lambda: None
    sys.modules['IPython'] = ip
    sys.modules['IPython.display'] = ipd


def main():
    # `--only a,b`: write only the fixtures whose name starts with one of the prefixes (the others are
    # still computed, so every section sees the RNG state it always saw)
    only = None
    if '--only' in sys.argv:
        only = tuple(sys.argv[sys.argv.index('--only') + 1].split(','))
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import torch
    _stub_modules()
    work = tempfile.mkdtemp(prefix='fthmc_golden_')
    os.chdir(work)                      # TrainConfig(debug=True) writes ./debug

    import fthmc.config as cfg          # resets default dtype to fp32 (Q1)
    torch.set_default_dtype(torch.float64)
    import fthmc.utils.qed_helpers as qed
    import fthmc.utils.layers as layers
    import fthmc.utils.distributions as distributions
    import fthmc.utils.samplers as samplers
    import fthmc.ft_hmc as ft_hmc
    import fthmc.train as train
    cfg.DTYPE = torch.float64
    ft_hmc.DTYPE = torch.float64
    samplers.DTYPE = torch.float64

    def npy(t):
        return t.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(t) else np.asarray(t)

    def make_flow(n_layers, L, seed, act='silu'):
        torch.manual_seed(seed)
        return layers.make_u1_equiv_layers(
            n_layers=n_layers, n_mixture_comps=2, lattice_shape=(L, L),
            hidden_sizes=[8, 8], kernel_size=3, activation_fn=act)

    def flow_arrays(flow):
        d = {'n_layers': np.int64(len(flow))}
        for li, layer in enumerate(flow):
            for pi, p in enumerate(layer.parameters()):
                d[f'w{li}_{pi}'] = npy(p)
        return d

    def save(name, **arrs):
        if only is not None and not name.startswith(only):
            return
        path = os.path.join(OUT, name + '.npz')
        np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
        print(f'{name}: {os.path.getsize(path)/1024:.1f} KiB')

    # ---- 0. RNG-free known answer ---------------------------
    B, L, beta = 2, 8, 2.0
    b_, m_, i_, j_ = np.meshgrid(np.arange(B), np.arange(2), np.arange(L), np.arange(L), indexing='ij')
    x = torch.tensor(2.5 * np.sin(0.7 * i_ + 1.3 * j_ ** 2 + 2.1 * m_ + 0.9 * b_))
    flow = make_flow(2, L, 0)
    with torch.no_grad():
        for li, layer in enumerate(flow):
            for pi, p in enumerate(layer.parameters()):
                n = p.numel()
                p.copy_((0.3 * torch.sin(1.0 + 0.37 * torch.arange(n, dtype=torch.float64) + li + 0.5 * pi)).reshape(p.shape))
    param = cfg.Param(beta=beta, L=L)
    S = qed.BatchAction(beta)(x)
    y0, lj0 = flow[0].forward(x)
    y1, lj1 = flow[1].forward(y0)
    save('known_answer', x=npy(x), beta=beta, S=npy(S), Q=npy(qed.batch_charges(x)),
         plaq=npy(-S / (beta * L * L)), F=npy(qed.force(param, x.clone())),
         y0=npy(y0), logJ0=npy(lj0), y1=npy(y1), logJ1=npy(lj1),
         S_eff=npy(qed.ft_action(param, flow, x)), ft_force=npy(qed.ft_force(param, flow, x.clone())),
         **flow_arrays(flow))

    # ---- 1. observables -------------------------------------------------
    torch.manual_seed(101)
    for (B, L, beta) in [(3, 8, 2.0), (2, 12, 3.5), (4, 16, 4.0)]:
        x = (torch.rand(B, 2, L, L) * 2 - 1) * 3 * math.pi     # beyond [-pi,pi) on purpose
        S = qed.BatchAction(beta)(x)
        save(f'obs_L{L}', x=npy(x), beta=beta, S=npy(S), plaqs=npy(qed.batch_plaqs(x)),
             Q=npy(qed.batch_charges(x)), plaq=npy(-S / (beta * L * L)),
             topo=npy(qed.topo_charge(x)), regularize=npy(qed.regularize(x)),
             wrap=npy(qed.torch_wrap(x)), layers_mod=npy(layers.torch_mod(x)))

    # ---- 2. plain force / leapfrog / hmc --------------------------------
    for (L, beta, tau, nstep, seed) in [(8, 2.0, 1.0, 10, 1331), (8, 2.0, 0.3, 1, 7), (16, 4.0, 0.5, 5, 11)]:
        param = cfg.Param(beta=beta, L=L, tau=tau, nstep=nstep)
        torch.manual_seed(seed)
        x = torch.empty(1, 2, L, L).uniform_(-math.pi, math.pi)
        p = torch.randn_like(x)
        Fx = qed.force(param, x.clone())
        x_, p_ = qed.leapfrog(param, x.clone(), p.clone(), verbose=False)
        # hmc with captured v, u: re-seed, call, then replay the draws
        torch.manual_seed(seed + 1)
        dH, exp_mdH, acc, newx = qed.hmc(param, x.clone(), verbose=False)
        torch.manual_seed(seed + 1)
        v = torch.randn_like(x)
        u = torch.rand([], dtype=torch.float64)
        save(f'hmc_L{L}_n{nstep}', x=npy(x), p=npy(p), beta=beta, dt=param.dt, nstep=nstep,
             force=npy(Fx), lf_x=npy(x_), lf_p=npy(p_),
             v=npy(v), u=npy(u), dH=npy(dH), exp_mdH=npy(exp_mdH), acc=np.bool_(bool(acc)), newx=npy(newx))
    # zero start (reference default randinit=False)
    param = cfg.Param(beta=2.0, L=8, tau=1.0, nstep=10)
    x = param.initializer().to(torch.float64)
    torch.manual_seed(1331)
    dH, exp_mdH, acc, newx = qed.hmc(param, x.clone(), verbose=False)
    torch.manual_seed(1331)
    v = torch.randn_like(x)
    u = torch.rand([], dtype=torch.float64)
    save('hmc_zero_L8', x=npy(x), beta=2.0, dt=param.dt, nstep=10, v=npy(v), u=npy(u),
         dH=npy(dH), exp_mdH=npy(exp_mdH), acc=np.bool_(bool(acc)), newx=npy(newx))

    # ---- 3. single layers: all 8 (mu, off) combos, fwd + VJP + wgrad ----
    for (B, L, act) in [(2, 8, 'silu'), (2, 12, 'silu'), (1, 8, 'relu'), (1, 8, 'leaky_relu')]:
        flow = make_flow(8, L, 500 + L)
        if act != 'silu':
            flow = make_flow(8, L, 500 + L, act)
        torch.manual_seed(77 + L)
        d = {'beta': 0.0, 'act': act, **flow_arrays(flow)}
        for li, layer in enumerate(flow):
            x = torch.empty(B, 2, L, L).uniform_(-math.pi, math.pi).requires_grad_(True)
            c = torch.randn(B, 2, L, L)
            dd = torch.randn(B)
            y, lj = layer.forward(x)
            obj = (c * y).sum() + (dd * lj).sum()
            params = list(layer.parameters())
            grads = torch.autograd.grad(obj, [x] + params)
            d.update({f'x{li}': npy(x), f'c{li}': npy(c), f'd{li}': npy(dd), f'y{li}': npy(y),
                      f'logJ{li}': npy(lj), f'gx{li}': npy(grads[0])})
            for pi, g in enumerate(grads[1:]):
                d[f'gw{li}_{pi}'] = npy(g)
            with torch.no_grad():
                xr, ljr = layer.reverse(y.detach())
            d.update({f'rev_x{li}': npy(xr), f'rev_logJ{li}': npy(ljr)})
        save(f'layers_L{L}_{act}', **d)

    # ---- 4. ft_action / ft_force / round trip ---------------------------
    for (B, L, beta, nl) in [(2, 8, 2.0, 2), (2, 8, 2.0, 8), (32, 16, 4.0, 4), (1, 8, 2.0, 16)]:
        flow = make_flow(nl, L, 900 + nl)
        param = cfg.Param(beta=beta, L=L)
        torch.manual_seed(31 + nl)
        x = torch.empty(B, 2, L, L).uniform_(-math.pi, math.pi)
        ft = ft_hmc.FieldTransformation(flow, cfg.TrainConfig(L=L, beta=beta, debug=True, n_layers=nl),
                                        cfg.lfConfig(tau=1.0, nstep=10))
        with torch.no_grad():
            y, logdet = ft.flow_forward(x)
            xb, logdet_b = ft.flow_backward(y)
            S_eff = qed.ft_action(param, flow, x)
        Ff = qed.ft_force(param, flow, x.clone())
        save(f'ft_L{L}_n{nl}', x=npy(x), beta=beta, y=npy(y), logdet=npy(logdet), S_eff=npy(S_eff),
             ft_force=npy(Ff), rev_x=npy(xb), rev_logdet=npy(logdet_b),
             Q=npy(qed.batch_charges(y)), **flow_arrays(flow))

    # ---- 5. trajectories -------------------------------------------------
    # (A) intended ftHMC: qed.leapfrog structure with ft_force / ft_action
    #     (semantics of ipynb/ft_hmc.py:394-435 without the flow-inverse wrapper)
    for (B, L, beta, nl, tau, nstep) in [(1, 8, 2.0, 4, 1.0, 10), (4, 16, 4.0, 4, 1.0, 10)]:
        flow = make_flow(nl, L, 1200 + L)
        param = cfg.Param(beta=beta, L=L, tau=tau, nstep=nstep)
        torch.manual_seed(1331)
        x = torch.empty(B, 2, L, L).uniform_(-math.pi, math.pi)
        v = torch.randn_like(x)
        u = torch.rand(B, dtype=torch.float64)
        dt = param.dt
        with torch.no_grad():
            h0 = qed.ft_action(param, flow, x) + 0.5 * (v * v).flatten(1).sum(1)
        x_ = x + 0.5 * dt * v
        p_ = v + (-dt) * qed.ft_force(param, flow, x_)
        for _ in range(nstep - 1):
            x_ = x_ + dt * p_
            p_ = p_ + (-dt) * qed.ft_force(param, flow, x_)
        x_ = x_ + 0.5 * dt * p_
        xr = qed.regularize(x_)
        with torch.no_grad():
            h1 = qed.ft_action(param, flow, xr) + 0.5 * (p_ * p_).flatten(1).sum(1)
            dH = h1 - h0
            acc = u < torch.exp(-dH)
            newx = torch.where(acc[:, None, None, None], xr, x)
            yphys = qed.ft_flow(flow, newx)
        save(f'traj_md_L{L}', x=npy(x), v=npy(v), u=npy(u), beta=beta, dt=dt, nstep=nstep,
             lf_x=npy(x_), lf_p=npy(p_), H0=npy(h0), H1=npy(h1), dH=npy(dH), acc=npy(acc).astype(bool),
             newx=npy(newx), plaq=npy(-qed.BatchAction(beta)(yphys) / (beta * L * L)),
             Q=npy(qed.batch_charges(yphys)), **flow_arrays(flow))
    # (B) literal FieldTransformation.hmc (B=1 only)
    L, beta, nl = 8, 2.0, 4
    flow = make_flow(nl, L, 1300)
    ft = ft_hmc.FieldTransformation(flow, cfg.TrainConfig(L=L, beta=beta, debug=True, n_layers=nl),
                                    cfg.lfConfig(tau=1.0, nstep=10))
    torch.manual_seed(4242)
    x = torch.empty(1, 2, L, L).uniform_(0, 2 * math.pi)
    torch.manual_seed(99)
    xnew, metrics = ft.hmc(x.clone())
    torch.manual_seed(99)
    v = torch.randn_like(x)
    u = torch.rand([], dtype=torch.float64)
    with torch.no_grad():
        yphys, _ = ft.flow_forward(xnew)
        lm = ft.lattice_metrics(yphys, torch.zeros(1))
    save('traj_literal_L8', x=npy(x), v=npy(v), u=npy(u), beta=beta, dt=ft.dt, nstep=10,
         newx=npy(xnew), dH=npy(metrics['dh']), acc=np.bool_(bool(metrics['acc'])),
         plaq=npy(lm['plaq']), Q=npy(lm['q']), **flow_arrays(flow))

    # ---- 6. one training step -------------------------------------------
    for (B, L, beta, nl) in [(4, 8, 2.0, 4), (8, 16, 4.0, 2)]:
        tc = cfg.TrainConfig(L=L, beta=beta, debug=True, n_layers=nl, batch_size=B, base_lr=1e-3)
        torch.manual_seed(2024 + L)
        model = train.get_model(tc)
        w_before = flow_arrays(model.layers)
        optimizer = torch.optim.Adam(model.layers.parameters(), lr=tc.base_lr)
        xi = model.prior.sample_n(B)
        metrics = train.train_step(model, tc, qed.BatchAction(beta), optimizer, B, xi=xi.clone())
        d = {'xi': npy(xi), 'beta': beta, 'lr': tc.base_lr}
        d.update(w_before)
        for li, layer in enumerate(model.layers):
            for pi, p in enumerate(layer.parameters()):
                d[f'gw{li}_{pi}'] = npy(p.grad)
                d[f'w_after{li}_{pi}'] = npy(p)
        for k in ('ess', 'logp', 'logq', 'loss_dkl', 'q', 'dq', 'plaq'):
            d[k] = np.asarray(metrics[k], dtype=np.float64)
        save(f'train_L{L}', **d)

    # ---- 7. masks ---------------------------------------------------------
    d = {}
    for L in (8, 12):
        for mu in (0, 1):
            for off in range(4):
                pm = layers.make_plaq_masks((L, L), mu, off)
                d[f'L{L}_mu{mu}_off{off}_active'] = npy(pm['active'])
                d[f'L{L}_mu{mu}_off{off}_frozen'] = npy(pm['frozen'])
                d[f'L{L}_mu{mu}_off{off}_passive'] = npy(pm['passive'])
                d[f'L{L}_mu{mu}_off{off}_link'] = npy(layers.make_2d_link_active_stripes((2, L, L), mu, off))
    save('masks', **d)
    d = {f'{b}': v for b, v in cfg.PLAQ_EXACT.items()}
    save('plaq_exact', betas=np.array(list(cfg.PLAQ_EXACT.keys())), values=np.array(list(cfg.PLAQ_EXACT.values())))

    # ---- 8. physical-field ftHMC (ipynb/ft_hmc.py:420-435: ft_flow_inv -> trajectory -> ft_flow), composed
    #         from the packaged functions (the notebook module runs an experiment at import: not imported).
    #         'ref': the reference's own inverse tolerance (bisection to a global 1e-6);
    #         'tight': the same layers with inv_prec = 1e-14 (bisection down to floating-point resolution)
    for (L, beta, nl, tau, nstep, seed) in [(8, 2.0, 4, 1.0, 10, 1500), (16, 4.0, 4, 0.5, 5, 1516)]:
        for tag, inv_prec in (('ref', None), ('tight', 1e-14)):
            flow = make_flow(nl, L, seed)
            if inv_prec is not None:
                for layer in flow:
                    layer.plaq_coupling.inv_prec = inv_prec
            param = cfg.Param(beta=beta, L=L, tau=tau, nstep=nstep)
            torch.manual_seed(seed + 1)
            field = torch.empty(1, 2, L, L).uniform_(-math.pi, math.pi)
            with torch.no_grad():
                x = qed.ft_flow_inv(flow, field)
            torch.manual_seed(seed + 2)
            p = torch.randn_like(x)
            u = torch.rand([], dtype=torch.float64)                 # drawn after the leapfrog in the notebook; it
            dt = param.dt                                            # consumes no random numbers in between
            with torch.no_grad():
                act0 = qed.ft_action(param, flow, x) + 0.5 * torch.sum(p * p)
            x_ = x + 0.5 * dt * p
            p_ = p + (-dt) * qed.ft_force(param, flow, x_)
            for _ in range(nstep - 1):
                x_ = x_ + dt * p_
                p_ = p_ + (-dt) * qed.ft_force(param, flow, x_)
            x_ = x_ + 0.5 * dt * p_
            xr = qed.regularize(x_)
            with torch.no_grad():
                act = qed.ft_action(param, flow, xr) + 0.5 * torch.sum(p_ * p_)
                dH = act - act0
                exp_mdH = torch.exp(-dH)
                acc = u < exp_mdH
                newx = xr if bool(acc) else x
                newfield = qed.ft_flow(flow, newx)
                prop_field = qed.ft_flow(flow, xr)
            S = qed.BatchAction(beta)(newfield)
            save(f'fthmc_phys_L{L}_{tag}', field=npy(field), x_inv=npy(x), v=npy(p), u=npy(u), beta=beta, dt=dt,
                 nstep=nstep, dH=npy(dH), exp_mdH=npy(exp_mdH), acc=np.bool_(bool(acc)), newx=npy(newx),
                 newfield=npy(newfield), prop_field=npy(prop_field), plaq=npy(-S / (beta * L * L)),
                 Q=npy(qed.batch_charges(newfield)), **flow_arrays(flow))

    # ---- 9. flow-proposal independence Metropolis (samplers.make_mcmc_ensemble, samplers.py:182-259) with the
    #         proposals and the uniforms of the accept chain recorded.  NOTE the reference's generator
    #         (samplers.py:123-137) unpacks `_, x, logq = apply_flow_to_prior(...)` although that function
    #         returns (x, xi, logq): its proposals are the PRIOR draws xi with logp = -S(xi).  Both are kept.
    for (L, beta, nl, bs, nsamp, seed) in [(8, 2.0, 4, 8, 40, 1600)]:
        tc = cfg.TrainConfig(L=L, beta=beta, debug=True, n_layers=nl, batch_size=bs)
        torch.manual_seed(seed)
        model = train.get_model(tc)
        action = qed.BatchAction(beta)
        rec = {'xi': [], 'xflow': [], 'logq': [], 'u': []}
        real_apply, real_rand = samplers.apply_flow_to_prior, torch.rand

        def rec_apply(prior, layers_, *, batch_size, xi=None):
            x_, xi_, logq_ = real_apply(prior, layers_, batch_size=batch_size, xi=xi)
            rec['xi'].append(npy(xi_)); rec['xflow'].append(npy(x_)); rec['logq'].append(npy(logq_))
            return x_, xi_, logq_

        def rec_rand(*a, **k):
            out = real_rand(*a, **k)
            if a == (1,) and not k:
                rec['u'].append(float(out))
            return out
        samplers.apply_flow_to_prior = rec_apply
        torch.rand = rec_rand
        try:
            torch.manual_seed(seed + 1)
            hist = samplers.make_mcmc_ensemble(model, action, bs, nsamp)
        finally:
            samplers.apply_flow_to_prior = real_apply
            torch.rand = real_rand
        xi_all = np.concatenate(rec['xi']); xf_all = np.concatenate(rec['xflow'])
        with torch.no_grad():
            logp_xi = -action(torch.from_numpy(xi_all))              # what the reference's chain uses
            logp_flow = -action(torch.from_numpy(xf_all))            # -S(F(xi)): the intended proposal weight
            q_xi = qed.batch_charges(torch.from_numpy(xi_all))
            q_flow = qed.batch_charges(torch.from_numpy(xf_all))
        assert len(rec['u']) == nsamp - 1
        save(f'sampler_L{L}', xi=xi_all[:nsamp], xflow=xf_all[:nsamp], logq=np.concatenate(rec['logq'])[:nsamp],
             logp_xi=npy(logp_xi)[:nsamp], logp_flow=npy(logp_flow)[:nsamp], q_xi=npy(q_xi)[:nsamp],
             q_flow=npy(q_flow)[:nsamp], u=np.array(rec['u']), beta=beta, batch_size=bs,
             hist_q=np.asarray(hist['q'], dtype=np.float64), hist_dqsq=np.asarray(hist['dqsq'], dtype=np.float64),
             hist_logq=np.asarray(hist['logq'], dtype=np.float64), hist_logp=np.asarray(hist['logp'], dtype=np.float64),
             hist_acc=np.asarray(hist['acc'], dtype=np.float64), **flow_arrays(model.layers))

    # ---- 10. delta-Q^2 versus MD-time lag and its blocked error (ipynb/ft_hmc.py:16-53, 168-176).  That module runs a
    #          whole experiment at import, so it is only PARSED here: the statistics helpers' function definitions (and
    #          the `n_block` constant they read) are compiled from its syntax tree into an empty namespace and called
    #          on seeded charge histories; nothing else of the file executes.
    import ast
    with open(os.path.join(REF, 'ipynb', 'ft_hmc.py')) as f:
        tree = ast.parse(f.read())
    want = {'average', 'sigma', 'sub_avg', 'block_list', 'change_sqr', 'change_sqr_vs_dt', 'save_topo_change_sqr'}
    keep = [n for n in tree.body
            if (isinstance(n, ast.FunctionDef) and n.name in want) or
               (isinstance(n, ast.Assign) and any(isinstance(t, ast.Name) and t.id == 'n_block' for t in n.targets))]
    assert {n.name for n in keep if isinstance(n, ast.FunctionDef)} == want
    ns = {'np': np, 'sys': sys, 'topo_history': []}
    exec(compile(ast.Module(body=keep, type_ignores=[]), 'ipynb/ft_hmc.py[statistics]', 'exec'), ns)
    rng = np.random.default_rng(4242)
    obs = {'n_block': np.int64(ns['n_block'])}
    for tag, nhist in (('long', 1500), ('short', 40), ('tiny', 9)):
        # integer charge history: lazy random walk, as a tunnelling topological charge looks
        q = np.cumsum(rng.integers(-1, 2, size=nhist) * (rng.random(nhist) < 0.3)).astype(np.float64)
        obs[f'q_{tag}'] = q
        # a lag the history is too short for yields the bare [lag] (change_sqr returns []): stored as NaN
        obs[f'rows_{tag}'] = np.array([r if len(r) == 3 else [r[0], np.nan, np.nan]
                                       for r in ns['change_sqr_vs_dt'](list(q), 10)], dtype=np.float64)
        obs[f'blocks_{tag}'] = np.array([np.mean(b) for b in ns['block_list'](list(q))], dtype=np.float64)
        if tag == 'tiny':
            continue                                             # the reference's writer raises IndexError on such rows
        ns['topo_history'][:] = list(q)                          # save_topo_change_sqr reads the module global
        fn = os.path.join(work, f'dq2_{tag}.txt')
        devnull, stdout = open(os.devnull, 'w'), sys.stdout
        try:
            sys.stdout = devnull                                 # it prints every row
            ns['save_topo_change_sqr'](fn)
        finally:
            sys.stdout = stdout
            devnull.close()
        obs[f'file_{tag}'] = np.loadtxt(fn, ndmin=2)
    obs['sub_avg'] = np.asarray(ns['sub_avg'](list(obs['q_short'])), dtype=np.float64)
    obs['sigma_short'] = np.float64(ns['sigma'](list(obs['q_short'])))
    save('observables', **obs)

    # ---- 11. BASELINE configs[1] at its real shape: 32 chains of L = 16, beta = 4, 4-layer flow, tau = 1, nstep = 10, one MD
    #          trajectory with captured v, u (the structure of section 5 A); a warm start so that some chains accept
    B, L, beta, nl, tau, nstep = 32, 16, 4.0, 4, 1.0, 10
    flow = make_flow(nl, L, 3100)
    param = cfg.Param(beta=beta, L=L, tau=tau, nstep=nstep)
    torch.manual_seed(3101)
    x = torch.empty(B, 2, L, L).uniform_(-math.pi, math.pi) * 0.12
    v = torch.randn_like(x)
    u = torch.rand(B, dtype=torch.float64)
    dt = param.dt
    with torch.no_grad():
        h0 = qed.ft_action(param, flow, x) + 0.5 * (v * v).flatten(1).sum(1)
    x_ = x + 0.5 * dt * v
    p_ = v + (-dt) * qed.ft_force(param, flow, x_)
    for _ in range(nstep - 1):
        x_ = x_ + dt * p_
        p_ = p_ + (-dt) * qed.ft_force(param, flow, x_)
    x_ = x_ + 0.5 * dt * p_
    xr = qed.regularize(x_)
    with torch.no_grad():
        h1 = qed.ft_action(param, flow, xr) + 0.5 * (p_ * p_).flatten(1).sum(1)
        dH = h1 - h0
        acc = u < torch.exp(-dH)
        newx = torch.where(acc[:, None, None, None], xr, x)
        yphys = qed.ft_flow(flow, newx)
    print('config-2 golden: accepted', int(acc.sum()), 'of', B, ' max |dH|', float(dH.abs().max()))
    save('traj_md_config2', x=npy(x), v=npy(v), u=npy(u), beta=beta, dt=dt, nstep=nstep,
         lf_x=npy(x_), lf_p=npy(p_), H0=npy(h0), H1=npy(h1), dH=npy(dH), acc=npy(acc).astype(bool),
         newx=npy(newx), plaq=npy(-qed.BatchAction(beta)(yphys) / (beta * L * L)),
         Q=npy(qed.batch_charges(yphys)), **flow_arrays(flow))

    # ---- 12. s/t nets other than the default 2 -> 8 -> 8 -> 3, k = 3, two components (layers.py:138-167, 399-429 accept any
    #          hidden_sizes / kernel_size / n_mixture_comps): flow forward, S_eff, ft_force, and one train_step
    for tag, (hidden, k, n_mix, L, nl, B, beta, act) in {'a': ([4, 6, 5], 5, 3, 8, 3, 2, 2.0, 'silu'),
                                                         'b': ([16], 3, 1, 12, 2, 2, 3.0, 'leaky_relu')}.items():
        torch.manual_seed(4100 + L)
        flow = layers.make_u1_equiv_layers(n_layers=nl, n_mixture_comps=n_mix, lattice_shape=(L, L), hidden_sizes=hidden,
                                           kernel_size=k, activation_fn=act)
        param = cfg.Param(beta=beta, L=L)
        x = torch.empty(B, 2, L, L).uniform_(-math.pi, math.pi)
        with torch.no_grad():
            y, logdet = qed.ft_flow(flow, x), None
            S_eff = qed.ft_action(param, flow, x)
            ld = torch.zeros(B)
            xx = x
            for layer in flow:
                xx, lj = layer.forward(xx)
                ld = ld + lj
        Ff = qed.ft_force(param, flow, x.clone())
        tc = cfg.TrainConfig(L=L, beta=beta, debug=True, n_layers=nl, batch_size=B, base_lr=1e-3, hidden_sizes=hidden,
                             kernel_size=k, n_s_nets=n_mix, activation_fn=act)
        torch.manual_seed(4200 + L)
        model = train.get_model(tc)
        w_train = {f't{kk}': v for kk, v in flow_arrays(model.layers).items()}
        optimizer = torch.optim.Adam(model.layers.parameters(), lr=tc.base_lr)
        xi = model.prior.sample_n(B)
        metrics = train.train_step(model, tc, qed.BatchAction(beta), optimizer, B, xi=xi.clone())
        d = {'x': npy(x), 'beta': beta, 'act': act, 'hidden': np.array(hidden), 'kernel_size': k, 'n_mix': n_mix,
             'y': npy(y), 'logdet': npy(ld), 'S_eff': npy(S_eff), 'ft_force': npy(Ff), 'Q': npy(qed.batch_charges(y)),
             'xi': npy(xi), 'loss_dkl': np.float64(metrics['loss_dkl']), 'ess': np.float64(metrics['ess']),
             'logq': np.asarray(metrics['logq'], dtype=np.float64), 'logp': np.asarray(metrics['logp'], dtype=np.float64)}
        d.update(flow_arrays(flow)); d.update(w_train)
        for li, layer in enumerate(model.layers):
            for pi, p_ in enumerate(layer.parameters()):
                d[f'tgw{li}_{pi}'] = npy(p_.grad)
        save(f'netshape_{tag}', **d)


if __name__ == '__main__':
    main()
