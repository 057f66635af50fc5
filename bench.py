#!/usr/bin/env python3
"""bench.py -- leapfrog-steps/sec of batched ftHMC chains on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2]): 2D U(1), L=64, beta=6.0, 8-layer flow
(hidden [8,8], k=3, n_mix=2, SiLU, PyTorch default init), 128 chains per GPU,
tau=1.0, nstep=10, fp64.  One bench "step" = one whole ftHMC trajectory of the
batch (momentum refresh, H0, 10 leapfrog steps = 10 force evaluations, H1,
Metropolis, observables of the accepted field).  Trajectories are chained: the
effective action of the accepted field is carried over (C ABI `state_in`), so H0
costs no second flow sweep.  Chains shard over ranks with no
data-path collective (weak scaling, 128 chains per GPU); the only exchange is the
8-double SUM all-reduce of run statistics per trajectory (RCCL), which is inside
the timed region.

Prints ONE JSON line on rank 0.  `value` = chain-leapfrog-steps per second over
all ranks (= batch x leapfrog-steps/s).  Also reports
  roofline     -- dominant kernel (coupling-layer backward) against the fp64 peak,
                  duration measured here with HIP events on the launch stream;
  cpu_baseline -- the oracle (oracle/ref_cpu.py, PyTorch CPU fp64, "port") timed on
                  this host on a bounded sample of the same workload, and the
                  parity of the HIP trajectory against it on that sample.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

L, BETA, N_LAYERS, B_PER_GPU, TAU, NSTEP, SEED = 64, 6.0, 8, 128, 1.0, 10, 1331
FP64_PEAK_TFLOPS = 78.6        # MI355X fp64 vector = matrix peak (spec); MFMA f64 measured 77.5 (tools/microbench)
CONV_FLOPS_PER_SITE = 1872     # dense 3x3 conv net 2->8->8->3, one direction


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=B_PER_GPU, help='chains per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-chains', type=int, default=128)
    ap.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying a hipGraph')
    ap.add_argument('--thermalize', type=int, default=60, help='untimed plain-HMC trajectories applied to x0')
    ap.add_argument('--groups', type=int, default=2,
                    help='split the chains of a GPU into this many groups whose trajectories run on concurrent '
                         'streams (chains are independent: one group fills the CUs the other leaves idle while a '
                         'kernel drains)')
    return ap.parse_args()


def log(msg):
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


def host_threads():
    """CPU threads this process may really use (cgroup/affinity aware), capped at 16 = one GPU's share."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, p = f.read().split()
            if q != 'max':
                n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/r01_pmc_summary.json: FETCH_SIZE + WRITE_SIZE, KiB, raw)."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'r01_pmc_summary.json')) as f:
            k = json.load(f)['kernels']
        k = next(v for n, v in k.items() if kernel in n)
        return round((k['FETCH_SIZE']['mean_per_launch'] + k['WRITE_SIZE']['mean_per_launch']) * 1024)
    except Exception:
        return None


def make_flow(gen):
    """Synthetic flow weights: PyTorch's default Conv2d init (Kaiming-uniform a = sqrt(5), i.e.
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)); SURVEY Q6: the reference's set_weights is a no-op) for the s/t net
    2 -> 8 -> 8 -> 3, k = 3, drawn from `gen`: [(w0, b0, w1, b1, w2, b2)] * N_LAYERS."""
    sizes = [2, 8, 8, 3]
    flow = []
    for _ in range(N_LAYERS):
        w = []
        for ci, co in zip(sizes[:-1], sizes[1:]):
            bound = 1.0 / math.sqrt(ci * 9)
            w.append((torch.rand(co, ci, 3, 3, generator=gen, dtype=torch.float64) * 2 - 1) * bound)
            w.append((torch.rand(co, generator=gen, dtype=torch.float64) * 2 - 1) * bound)
        flow.append(tuple(w))
    return flow


def main():
    args = parse()
    from fthmc_amd import ops, parallel
    rank, world, local = parallel.init()
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback in the product path)')
    local = local % torch.cuda.device_count()          # ranks may share a GPU in a gloo rehearsal
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    B = args.batch
    lo, hi = parallel.shard_range(B * world, rank, world)
    assert hi - lo == B
    dt = TAU / NSTEP

    # synthetic inputs, identical for the CPU and GPU paths
    gen = torch.Generator(device='cpu').manual_seed(SEED)
    flow = make_flow(gen)
    w = ops.pack_weights(flow, device=dev)
    gx = torch.Generator(device='cpu').manual_seed(SEED + 1 + rank)
    # Untimed preparation: a hot start U(-pi, pi) at beta = 6 rejects every trajectory, so the chains
    # start near-cold (|x| < 0.1) and are brought to the beta = 6 ensemble by plain Wilson HMC on
    # the HIP path; the timed ftHMC trajectories then run at a physical acceptance.
    x0 = ((torch.rand(B, 2, L, L, generator=gx, dtype=torch.float64) * 2 - 1) * 0.1)
    x = x0.to(dev)
    gt = torch.Generator(device='cpu').manual_seed(SEED + 7 + rank)
    for it in range(args.thermalize):
        vt = torch.randn(B, 2, L, L, generator=gt, dtype=torch.float64).to(dev)
        ut = torch.rand(B, generator=gt, dtype=torch.float64).to(dev)
        x = ops.hmc_trajectory(x, vt, ut, BETA, 0.05, 20)['x_new']
    x0 = x.cpu()

    stats = parallel.RunStats.zeros(dev)
    out = {'x_new': torch.empty_like(x)}
    for k in ('dH', 'acc', 'H0', 'H1', 'plaq', 'Q'):
        out[k] = torch.empty(B, dtype=torch.float64, device=dev)
    S0, _, p0, q0 = ops.ft_action(x, w, N_LAYERS, BETA)
    qold = q0.clone()
    seeds = torch.empty(B, dtype=torch.int64, device=dev)
    v = torch.empty_like(x)
    u = torch.empty(B, dtype=torch.float64, device=dev)
    stream = torch.cuda.Stream(device=dev)

    # chain groups (ops.ft_trajectory(groups=G)): contiguous blocks of this GPU's chains whose trajectories
    # run on concurrent streams, forked from / joined into `stream`
    G = max(1, min(args.groups, B))
    state = torch.stack([S0, p0, q0]).contiguous()      # (S_eff, plaq, Q) of the current x, carried along
    out['state'] = torch.empty_like(state)

    def enqueue():
        """momentum refresh + one trajectory of every chain of this GPU, forked from the current stream"""
        vv, uu = ops.random_momenta(seeds, x.shape)
        v.copy_(vv); u.copy_(uu)
        ops.ft_trajectory(x, v, u, w, N_LAYERS, BETA, dt, NSTEP, mode='md', out=out, state_in=state, groups=G)

    graph = None
    if not args.no_graph:
        # the ~200 launches of a trajectory are captured once and replayed (launch-bound otherwise)
        with torch.cuda.stream(stream):
            seeds.copy_(parallel.chain_seeds(SEED, lo, hi, 0).to(dev))
            enqueue()                       # warm allocator / workspaces before capture
            stream.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                enqueue()

    traj = [0]
    pending = [None]

    def step():
        seeds.copy_(parallel.chain_seeds(SEED, lo, hi, traj[0]).to(dev, non_blocking=True))
        if graph is not None:
            graph.replay()
        else:
            enqueue()
        x.copy_(out['x_new'])
        state.copy_(out['state'])
        dq = out['Q'] - qold
        stats.add(out['acc'], out['plaq'], out['Q'], dq, out['dH'])
        qold.copy_(out['Q'])
        if pending[0] is not None:
            pending[0].wait()
        pending[0] = stats.reduce(async_op=world > 1)
        traj[0] += 1

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    log(f'rank {rank}/{world}: setup done, graph={"yes" if graph is not None else "no"}; warmup {args.warmup}')
    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            step()
        barrier()
        log('timed region ...')
        stats.vec.zero_(); stats.glob = None
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        if pending[0] is not None:
            pending[0].wait()
        barrier()
        t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(elapsed, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(elapsed)
    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    log(f'timed region done: {elapsed:.3f} s for {args.steps} trajectories')
    chain_steps = B * world * NSTEP * args.steps
    value = chain_steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline of the dominant kernel, HIP events on this stream
    # launch shape of the timed region: one launch = one layer over one chain group (B / G chains)
    Bl = B // G if G > 1 else B
    with torch.cuda.stream(stream):
        w0 = w[:955].contiguous()
        xl = x[:Bl].contiguous()
        ms_bwd = ops.time_kernel('flow_bwd', xl, w0, mu=0, off=0, beta=BETA, reps=40)
        ms_fwd = ops.time_kernel('flow_fwd', xl, w0, mu=0, off=0, beta=BETA, reps=40)
        ms_bwd_full = ops.time_kernel('flow_bwd', x, w0, mu=0, off=0, beta=BETA, reps=40) if Bl != B else ms_bwd
        ms_fwd_full = ops.time_kernel('flow_fwd', x, w0, mu=0, off=0, beta=BETA, reps=40) if Bl != B else ms_fwd
        ms_leap = ops.time_kernel('leap_step', x, beta=BETA, reps=40)
        ms_traj = ops.time_kernel('hmc_trajectory', x, beta=BETA, reps=20)
    log(f'kernel timing ({Bl} chains per launch): bwd {ms_bwd:.4f} ms fwd {ms_fwd:.4f} ms; leap {ms_leap:.5f} ms')
    flops_launch = CONV_FLOPS_PER_SITE * L * L * Bl         # dense conv flops of one layer (fwd = dgrad), one launch
    # the dominant kernel = the one with the larger share of a trajectory: the forward kernel runs
    # N_LAYERS * (NSTEP + 1) times (force sweeps + H1), the backward kernel N_LAYERS * NSTEP times
    share = {'fwd': ms_fwd * N_LAYERS * (NSTEP + 1) * G, 'bwd': ms_bwd * N_LAYERS * NSTEP * G}
    dom = max(share, key=share.get)
    names = {'fwd': ('k_flow_fwd<16,16> (coupling-layer forward: conv net + tan-mixture transform + stash)', 'k_flow_fwd'),
             'bwd': ('k_flow_bwd_gather<16,16> (coupling-layer backward wrt x from the stash)', 'k_flow_bwd_gather')}
    ms_dom = ms_fwd if dom == 'fwd' else ms_bwd
    achieved = flops_launch / (ms_dom * 1e-3) / 1e12
    step_flops = 2 * CONV_FLOPS_PER_SITE * L * L * N_LAYERS * B    # fwd + dgrad, per batched leapfrog step
    roofline = {
        'bound': 'mfma', 'kernel': names[dom][0],
        'achieved': round(achieved, 3), 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
        'frac': round(achieved / FP64_PEAK_TFLOPS, 4), 'traffic': pmc_traffic(names[dom][1]),
        'avg_launch_ms': round(ms_dom, 4),
        'algorithmic_flops_per_launch': flops_launch,
        'chains_per_launch': Bl,
        'kernel_ms_per_trajectory': {k: round(v, 3) for k, v in share.items()},
        'full_batch_exclusive': {'chains_per_launch': B, 'fwd_kernel_ms': round(ms_fwd_full, 4), 'bwd_kernel_ms': round(ms_bwd_full, 4),
                                 'fwd_frac': round(CONV_FLOPS_PER_SITE * L * L * B / (ms_fwd_full * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4),
                                 'bwd_frac': round(CONV_FLOPS_PER_SITE * L * L * B / (ms_bwd_full * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4)},
        'fwd_kernel_ms': round(ms_fwd, 4), 'bwd_kernel_ms': round(ms_bwd, 4),
        'bwd_kernel': {'kernel': names['bwd'][0], 'achieved': round(flops_launch / (ms_bwd * 1e-3) / 1e12, 3),
                       'frac': round(flops_launch / (ms_bwd * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4),
                       'traffic': pmc_traffic(names['bwd'][1])},
        'whole_step_tflops': round(step_flops * (NSTEP * args.steps) / elapsed / 1e12, 3),
        'stencil': {'kernel': 'k_force<1> (fused plain-HMC leapfrog step, one launch per step)',
                    'avg_launch_ms': round(ms_leap, 5),
                    'achieved_GBps': round(64.0 * L * L * B / (ms_leap * 1e-3) / 1e9, 1), 'peak_GBps': 8000.0,
                    'persistent': {'kernel': 'k_hmc_trajectory (10 plain-HMC steps + H0/H1 + accept in one launch, '
                                             'links in LDS, momenta in registers)',
                                   'avg_launch_ms': round(ms_traj, 5), 'steps_per_launch': 10,
                                   'algorithmic_GBps': round(64.0 * L * L * B * 10 / (ms_traj * 1e-3) / 1e9, 1),
                                   'note': 'no HBM traffic between steps: the per-step HBM model is an upper bound'}},
    }

    # ---- CPU baseline (oracle = "port") on a bounded sample + parity of the HIP path on it
    cpu = None
    if not args.no_cpu_baseline and world == 1:          # rank 0 at N = 1 only
        from oracle import ref_cpu as R
        nb = min(args.cpu_chains, B)
        torch.set_num_threads(host_threads())
        log(f'cpu baseline: {nb} chains on {torch.get_num_threads()} threads ...')
        xs = x0[:nb].clone()
        gs = torch.Generator(device='cpu').manual_seed(SEED + 99)
        vs = torch.randn(nb, 2, L, L, generator=gs, dtype=torch.float64)
        us = torch.rand(nb, generator=gs, dtype=torch.float64)
        tc0 = time.perf_counter()
        dH_c, _, acc_c, newx_c, h0_c, h1_c = R.ft_hmc(xs, vs, us, flow, BETA, dt, NSTEP, mode='md')
        tc = time.perf_counter() - tc0
        log(f'cpu baseline done in {tc:.1f} s')
        r = ops.ft_trajectory(xs.to(dev), vs.to(dev), us.to(dev), w, N_LAYERS, BETA, dt, NSTEP, mode='md')
        rel = lambda a, b: float(((a.cpu() - b).abs() / b.abs().clamp_min(1e-300)).max())
        border = (us - torch.exp(-dH_c)).abs() < 1e-9
        acc_ok = bool(((r['acc'].cpu() > 0.5) == acc_c)[~border].all())
        cpu = {
            'value': round(nb * NSTEP / tc, 3), 'unit': 'chain-leapfrog-steps/s',
            'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'one trajectory ({NSTEP} leapfrog steps + H0/H1) of {nb} of the {B} chains, '
                      f'oracle/ref_cpu.py (PyTorch CPU fp64 autograd), {tc:.1f} s',
            'parity': {'H0_rel': rel(r['H0'], h0_c), 'H1_rel': rel(r['H1'], h1_c),
                       'dH_abs': float((r['dH'].cpu() - dH_c).abs().max()), 'accept_equal': acc_ok,
                       'tolerance': 1e-6},
        }

    m = stats.means()
    line = {
        'metric': 'leapfrog-steps/sec (batched chains)', 'value': round(value, 2),
        'unit': 'chain-leapfrog-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': '2D U(1) L=64 beta=6.0 8-layer flow ftHMC, 128 chains per GPU, tau=1.0 nstep=10 '
                               '(BASELINE.json configs[2]; configs[3] = the same per GPU on 8 GPUs)',
                   'chains_per_gpu': B, 'chains_total': B * world, 'L': L, 'beta': BETA, 'n_layers': N_LAYERS,
                   'nstep': NSTEP, 'tau': TAU, 'parallelism': f'chains sharded x{world}',
                   'launch': 'eager' if graph is None else 'hipGraph replay', 'chain_groups': G},
        'batched_leapfrog_steps_per_s': round(NSTEP * args.steps / elapsed, 3),
        'acceptance': round(m['acc'], 4), 'plaq': round(m['plaq'], 6),
        'roofline': roofline, 'cpu_baseline': cpu,
    }
    print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
