#!/usr/bin/env python3
"""The sampler at a usable operating point (GPU box): train a flow with the reference-shaped loop (fthmc.train.train ->
train_step -> fthmc_train_grad), then run ftHMC with it and check what an exact sampler must deliver whatever the flow:
<exp(-dH)> = 1 and <cos P> of the FLOWED field = I1(beta)/I0(beta) (config.PLAQ_EXACT; the finite-volume correction is
(I1/I0)^(L*L), < 1e-10 here), next to plain HMC and to the untrained flow at the same (tau, nstep).

    python3 tools/operating_point.py [--L 16] [--beta 4.0] [--layers 8] [--train-steps 1500] [--out profiles/rNN_operating_point.json]

bench.py's timed workload uses a random-init flow (the BASELINE workload prescribes it); this tool is the record that the
same kernels sample correctly and what a briefly trained flow buys at small volume.  Nothing here is timed for the headline."""
import argparse, json, math, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fthmc_amd import ops, parallel
from fthmc_amd.config import PLAQ_EXACT, TrainConfig
from fthmc_amd.train import train
from fthmc_amd.utils.layers import net_weights

ap = argparse.ArgumentParser()
ap.add_argument('--L', type=int, default=16)
ap.add_argument('--beta', type=float, default=4.0)
ap.add_argument('--layers', type=int, default=8)
ap.add_argument('--train-steps', type=int, default=1500)
ap.add_argument('--train-batch', type=int, default=512)
ap.add_argument('--lr', type=float, default=2e-3)
ap.add_argument('--chains', type=int, default=256)
ap.add_argument('--traj', type=int, default=200)
ap.add_argument('--therm', type=int, default=60)
ap.add_argument('--tau', type=float, default=1.0)
ap.add_argument('--nstep', type=int, default=10)
ap.add_argument('--nstep-trained', default='10,20,40', help='leapfrog steps per trajectory for the trained flow (same tau)')
ap.add_argument('--seed', type=int, default=1331)
ap.add_argument('--transfer-from', type=int, default=0, help='train.py:434-455 recipe: train at this lattice size first, then move the nets to --L')
ap.add_argument('--pre-steps', type=int, default=0, help='training steps at the --transfer-from size')
ap.add_argument('--anneal', default='', help='beta:steps,beta:steps,... trained in this order at the --transfer-from size before --beta (a curriculum in the coupling)')
ap.add_argument('--out', default=None)
args = ap.parse_args()
dev = torch.device('cuda', 0)
L, beta, NL, B = args.L, args.beta, args.layers, args.chains


def run(w, n_layers, label, nstep=None, prior_start=False):
    """therm + traj trajectories of B chains; statistics over the chains' means (independent chains -> honest errors)"""
    nstep = nstep or args.nstep
    dt = args.tau / nstep
    # Start, in the field the MD moves: near-cold U(-0.1, 0.1) for plain HMC and the near-identity untrained flow (as
    # bench.py: a hot start at large beta freezes defects in that take hundreds of trajectories to anneal; <exp(-dH)> != 1
    # is how an unthermalised run shows up here); a draw from the prior U(-pi, pi) for a trained flow -- that is where it
    # was trained to start from, the zero field is an atypical point of its latent space (acceptance 0 from there).
    g0, _ = ops.random_momenta(parallel.chain_seeds(args.seed + 1, 0, B, 0).to(dev), (B, 2, L, L), need_u=False)
    x = ((math.pi if prior_start else 0.1) * torch.erf(g0 / math.sqrt(2.0))).contiguous()
    acc_s = torch.zeros(B, dtype=torch.float64, device=dev); emdh_s = torch.zeros_like(acc_s)
    plaq_s = torch.zeros_like(acc_s); dq2_s = torch.zeros_like(acc_s); q2_s = torch.zeros_like(acc_s)
    qold = None
    t0 = time.perf_counter()
    for it in range(args.therm + args.traj):
        v, u = ops.random_momenta(parallel.chain_seeds(args.seed, 0, B, it).to(dev), (B, 2, L, L))
        if n_layers:
            r = ops.ft_trajectory(x, v, u, w, n_layers, beta, dt, nstep, mode='md')
            plaq, q = r['plaq'], r['Q']
        else:
            r = ops.hmc_trajectory(x, v, u, beta, dt, nstep)
            _, q, plaq = ops.wilson_action_charge(r['x_new'], beta)
        x = r['x_new']
        if it >= args.therm:
            acc_s += r['acc']; emdh_s += torch.exp(-r['dH']); plaq_s += plaq; q2_s += q * q
            dq2_s += (q - qold) ** 2
        qold = q.clone()
    torch.cuda.synchronize()
    n = args.traj

    def ms(t):                                                           # mean over chains of the chain means, its standard error
        m = (t / n).cpu().numpy()
        return float(m.mean()), float(m.std(ddof=1) / math.sqrt(len(m)))
    res = {'label': label, 'n_layers': n_layers, 'nstep': nstep, 'start': 'prior' if prior_start else 'near-cold', 'acceptance': ms(acc_s), 'exp_mdH': ms(emdh_s), 'plaq': ms(plaq_s),
           'dQ2_per_trajectory': ms(dq2_s), 'Q2': ms(q2_s), 'seconds': round(time.perf_counter() - t0, 2)}
    exact = PLAQ_EXACT[beta]
    res['plaq_minus_exact_in_sigma'] = round((res['plaq'][0] - exact) / max(res['plaq'][1], 1e-300), 2)
    res['exp_mdH_minus_1_in_sigma'] = round((res['exp_mdH'][0] - 1.0) / max(res['exp_mdH'][1], 1e-300), 2)
    print(json.dumps(res), flush=True)
    return res


torch.manual_seed(args.seed)
cfg = TrainConfig(L=L, beta=beta, n_layers=NL, batch_size=args.train_batch, n_era=1, n_epoch=args.train_steps,
                  base_lr=args.lr, print_freq=0)
t0 = time.perf_counter()
from fthmc_amd.train import get_model
pre = None
if args.transfer_from and args.pre_steps:                                # the reference's small-to-large recipe (train.py:434-455)
    from fthmc_amd.train import transfer_to_new_lattice
    cfg0 = TrainConfig(L=args.transfer_from, beta=beta, n_layers=NL, batch_size=args.train_batch, n_era=1, n_epoch=args.pre_steps,
                       base_lr=args.lr, print_freq=0)
    m0 = get_model(cfg0)
    for item in [t for t in args.anneal.split(',') if t]:                # earlier couplings first, same nets
        b_, n_ = item.split(':')
        cfga = TrainConfig(L=args.transfer_from, beta=float(b_), n_layers=NL, batch_size=args.train_batch, n_era=1, n_epoch=int(n_),
                           base_lr=args.lr, print_freq=0)
        oa = train(cfga, model=m0, verbose=False, save=False)
        m0 = oa['model']
        ea = [float(e) for e in oa['history']['ess']]
        print(json.dumps({'anneal_beta': float(b_), 'steps': int(n_), 'ess_last': round(float(np.mean(ea[-max(1, len(ea) // 20):])), 4)}), flush=True)
    o0 = train(cfg0, model=m0, verbose=False, save=False)
    e0 = [float(e) for e in o0['history']['ess']]
    k0 = max(1, len(e0) // 20)
    pre = {'L': args.transfer_from, 'steps': args.pre_steps, 'ess_first': round(float(np.mean(e0[:k0])), 4), 'ess_last': round(float(np.mean(e0[-k0:])), 4)}
    model0 = transfer_to_new_lattice(L, o0['model'].layers)
else:
    model0 = get_model(cfg)
w_init = ops.pack_weights([net_weights(l.plaq_coupling.net) for l in model0.layers], device=dev)
out = train(cfg, model=model0, verbose=False, save=False) if args.train_steps else {'model': model0, 'history': {'ess': [0.0], 'loss_dkl': [0.0]}}
torch.cuda.synchronize()
train_s = time.perf_counter() - t0
w_tr = ops.pack_weights([net_weights(l.plaq_coupling.net) for l in out['model'].layers], device=dev)
ess = [float(e) for e in out['history']['ess']]
loss = [float(e) for e in out['history']['loss_dkl']]
k = max(1, len(ess) // 20)
summary = {'L': L, 'beta': beta, 'n_layers': NL, 'plaq_exact': PLAQ_EXACT[beta], 'tau': args.tau, 'nstep': args.nstep,
           'chains': B, 'trajectories': args.traj, 'thermalisation': args.therm,
           'training': {'steps': args.train_steps, 'batch': args.train_batch, 'lr': args.lr, 'seconds': round(train_s, 1),
                        'ess_first': round(float(np.mean(ess[:k])), 4), 'ess_last': round(float(np.mean(ess[-k:])), 4),
                        'loss_dkl_first': round(float(np.mean(loss[:k])), 3), 'loss_dkl_last': round(float(np.mean(loss[-k:])), 3), 'pre_training': pre}}
print(json.dumps(summary), flush=True)
summary['runs'] = [run(None, 0, 'plain HMC'), run(w_init, NL, f'ftHMC, flow as moved from L = {args.transfer_from}' if pre else 'ftHMC, flow at its random initialisation', prior_start=bool(pre))]
for ns in ([int(t) for t in args.nstep_trained.split(',') if t] if args.train_steps or pre else []):
    summary['runs'].append(run(w_tr, NL, f'ftHMC, flow after {args.train_steps} reverse-KL steps', ns, prior_start=True))
if args.out:
    json.dump(summary, open(args.out, 'w'), indent=1)
