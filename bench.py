#!/usr/bin/env python3
"""bench.py -- leapfrog-steps/sec of batched ftHMC chains on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config {1,2,3,5}] [--scaling {weak,strong}]

Default workload = BASELINE.json configs[2] (`--config 3`): 2D U(1), L=64, beta=6.0, 8-layer flow
(hidden [8,8], k=3, n_mix=2, SiLU, PyTorch default init), 128 chains per GPU, tau=1.0, nstep=10,
fp64.  configs[3] is the same per GPU on 8 GPUs (`--gpus 8`).  The other configs are extra legs:
  --config 1  L=8,  beta=2.0, plain HMC (no flow), 1 chain            (the reference's CPU-runnable case)
  --config 2  L=16, beta=4.0, 4-layer flow, 32 chains
  --config 5  L=256, beta=7.0, 16-layer flow, 32 chains per GPU (the per-GPU shard of configs[4]) and
              `train_step` steps/s of the same flow (fthmc/train.py:162-228)

One bench "step" = one whole trajectory of the batch (momentum refresh, H0, nstep leapfrog steps =
nstep force evaluations, H1, Metropolis, observables of the accepted field).  Trajectories are chained:
the effective action of the accepted field is carried over (C ABI `state_in`), so H0 costs no second
flow sweep.  Chains shard over ranks with no data-path collective; the only exchange is the 8-double
SUM all-reduce of run statistics per trajectory (RCCL), inside the timed region.  `--scaling weak`
(default) keeps the chains per GPU fixed, `--scaling strong` keeps the total (128 at config 3).

With `--gpus N > 1` and no torchrun environment the script starts its own N ranks (one process per GPU,
`python -m torch.distributed.run`) BEFORE anything touches the GPU and relays rank 0's line.

Prints ONE JSON line on rank 0.  `value` = chain-leapfrog-steps per second over all ranks.  Also:
  roofline     -- the kernel with the largest share of a trajectory (coupling-layer forward for the
                  flowed configs: fp64 matrix/vector peak; the fused trajectory kernel for plain HMC:
                  HBM), duration measured here with HIP events on the launch stream;
  cpu_baseline -- the oracle (oracle/ref_cpu.py, PyTorch CPU fp64, "port") timed on this host on a
                  bounded sample of the same workload, on all host threads and on one thread, and the
                  parity of the HIP trajectory against it on that sample.  A parity failure makes the
                  exit code non-zero.
"""
import argparse
import json
import math
import os
import re
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 1331
TAU, NSTEP = 1.0, 10
CONFIGS = {
    1: dict(L=8, beta=2.0, n_layers=0, B=1, train=False,
            label='2D U(1) L=8 beta=2.0 plain HMC (no flow), 1 chain, tau=1.0 nstep=10 (BASELINE.json configs[0])'),
    2: dict(L=16, beta=4.0, n_layers=4, B=32, train=False,
            label='2D U(1) L=16 beta=4.0 4-layer flow ftHMC, 32 chains, tau=1.0 nstep=10 (BASELINE.json configs[1])'),
    3: dict(L=64, beta=6.0, n_layers=8, B=128, train=False,
            label='2D U(1) L=64 beta=6.0 8-layer flow ftHMC, 128 chains per GPU, tau=1.0 nstep=10 '
                  '(BASELINE.json configs[2]; configs[3] = the same per GPU on 8 GPUs)'),
    5: dict(L=256, beta=7.0, n_layers=16, B=32, train=True,
            label='2D U(1) L=256 beta=7.0 16-layer flow ftHMC + train_step, 32 chains per GPU, tau=1.0 nstep=10 '
                  '(per-GPU shard of BASELINE.json configs[4])'),
}
FP64_PEAK_TFLOPS = 78.6        # MI355X fp64 vector = matrix peak (spec); MFMA f64 measured 77.5 (tools/microbench)
HBM_PEAK_GBPS = 8000.0         # spec; 6.29 TB/s measured copy (MI355X_MICROARCH.md)
CONV_FLOPS_PER_SITE = 1872     # dense 3x3 conv net 2->8->8->3, one direction
TRAIN_FLOPS_PER_SITE = 5616    # forward + dgrad + wgrad
PARITY_TOL = 1e-6              # north_star: 1e-6 relative, fp64


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                    help='weak: --batch chains per GPU; strong: --batch chains in total, split over the GPUs')
    ap.add_argument('--batch', type=int, default=None, help='chains per GPU (weak) / in total (strong)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-chains', type=int, default=None)
    ap.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying a hipGraph')
    ap.add_argument('--thermalize', type=int, default=60, help='untimed plain-HMC trajectories applied to x0')
    ap.add_argument('--groups', type=int, default=None,
                    help='split the chains of a GPU into this many groups whose trajectories run on concurrent '
                         'streams (chains are independent: one group fills the CUs the other leaves idle while a '
                         'kernel drains); default: 2 for large launches, else 1')
    ap.add_argument('--group-sizes', type=str, default=None, help='explicit chain-group sizes, e.g. 96,32 (experiments)')
    ap.add_argument('--regions', type=int, default=None,
                    help='timed regions of --steps trajectories each (value = the median region); default: 1, or 5 when '
                         'the first region is shorter than 1 s')
    ap.add_argument('--min-seconds', type=float, default=6.0,
                    help='keep timing regions of exactly --steps trajectories until this many seconds of timed GPU work have '
                         'accumulated (the value is the median region; a monitor that samples the card every few seconds sees it busy)')
    ap.add_argument('--dump', type=str, default=None,
                    help='write every rank\'s per-chain end state (global chain ids, field, last dH / acc / Q) to '
                         'DUMP.<world>.<rank>.npz (sharding tests)')
    return ap.parse_args()


def log(msg):
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


def launch_command(n, argv, port):
    """The command the parent starts for `--gpus n` without a torchrun environment."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
            '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def visible_gpus():
    """GPUs this process would see, counted WITHOUT any HIP / torch.cuda call (the parent of a self-launched job
    must not initialise the GPU): the *_VISIBLE_DEVICES lists, else the GPU nodes of the KFD topology in sysfs."""
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(',') if t.strip() != ''])
    n = 0
    base = '/sys/class/kfd/kfd/topology/nodes'
    try:
        for node in os.listdir(base):
            with open(os.path.join(base, node, 'properties')) as f:
                props = dict(line.split(None, 1) for line in f if ' ' in line)
            if int(props.get('simd_count', '0')) > 0:           # CPU nodes report 0 SIMDs
                n += 1
    except (OSError, ValueError):
        return None
    return n


def self_launch(args):
    """--gpus N > 1 outside torchrun: start N fresh ranks as a CHILD job and relay its output.  This process never
    touches the GPU (devices are counted from the environment / sysfs, visible_gpus()) and never replaces itself
    with another program (no exec: forbidden on the GPU pool once a runtime is up; tests/test_host_logic.py
    asserts this file has none)."""
    env = dict(os.environ)
    have = visible_gpus()
    if have is None:
        have = args.gpus                      # unknown topology: trust the caller, RCCL reports a shortfall itself
    if have < args.gpus and 'FTHMC_DIST_BACKEND' not in env:
        # rehearsal on a smaller box: ranks share GPUs, which RCCL refuses -> gloo for the 8-double all-reduce
        log(f'{args.gpus} ranks on {have} GPU(s): ranks share devices, collectives over gloo (rehearsal, not a scaling number)')
        env['FTHMC_DIST_BACKEND'] = 'gloo'
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    port = 29500 + os.getpid() % 2000
    cmd = launch_command(args.gpus, sys.argv[1:], port)
    log('starting ' + ' '.join(cmd))
    p = subprocess.run(cmd, env=env)
    return p.returncode


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


def csrc_sha16():
    """Fingerprint of the kernel sources on disk (tools/csrc_sha.py: fthmc_amd/csrc/*.hip, *.h in name order)."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from csrc_sha import csrc_sha16 as f
    return f(ROOT)


def lib_sha16():
    """Fingerprint of the sources the LOADED library was built from (built in by csrc/Makefile, reported by fthmc_version());
    differs from csrc_sha16() for a library that was not rebuilt after an edit, or an A/B build loaded through FTHMC_LIB."""
    from fthmc_amd import _lib
    v = _lib.load().fthmc_version().decode()
    return v.split(' src ')[-1] if ' src ' in v else None


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary of this command
    (profiles/rNN_pmc_summary.json: FETCH_SIZE + WRITE_SIZE, KiB, raw) and which summary that was.  Counters cannot
    be read from inside the timed process (rocprofv3 must be the parent and serialises the launches), so the value is
    the committed one; `stale` says whether the kernel sources have changed since it was taken."""
    prof = os.path.join(ROOT, 'profiles')
    try:
        names = sorted(f for f in os.listdir(prof) if f.endswith('_pmc_summary.json'))
        with open(os.path.join(prof, names[-1])) as f:
            d = json.load(f)
        k = next(v for n, v in d['kernels'].items() if kernel in n)
        return (round((k['FETCH_SIZE']['mean_per_launch'] + k['WRITE_SIZE']['mean_per_launch']) * 1024),
                {'file': 'profiles/' + names[-1], 'commit': d.get('commit'), 'launch_shape': d.get('launch_shape'),
                 'csrc_sha16': d.get('csrc_sha16'), 'library_sha16': lib_sha16(),
                 # stale: the summary was taken on other kernels than the ones the loaded library was built from
                 'stale': d.get('csrc_sha16') != lib_sha16(), 'library_matches_sources': lib_sha16() == csrc_sha16()})
    except Exception:
        return None, None


def rocprof_launch(kernel):
    """Average launch duration (ms) of `kernel` in the newest committed `rocprofv3 --kernel-trace --stats` summary of this
    command (profiles/rNN_kernel_stats.csv: in situ, both chain groups' streams running) and which summary that was: the
    figure roofline.frac_rocprof is reproducible from.  The tiled coupling kernels are instantiated per stripe direction:
    the call-weighted average over the instances of the timed region (forward sweeps: not the REV = true instances)."""
    import csv
    prof = os.path.join(ROOT, 'profiles')
    try:
        names = sorted(f for f in os.listdir(prof) if re.fullmatch(r'r\d+_kernel_stats\.csv', f))
        calls = tot = 0
        with open(os.path.join(prof, names[-1])) as f:
            for row in csv.DictReader(f):
                m = re.search(re.escape(kernel) + r'<([^>]*)>', row['Name'])
                if not m:
                    continue
                targs = [t.strip() for t in m.group(1).split(',')]
                # k_flow_fwd<TR, TC, FASTW, REV, MU, EXACT, SILU, SWEEP> / k_flow_bwd_gather<TR, TC, FASTW, MU, EXACT, SWEEP>: the
                # force-sweep instances (SWEEP = 1: what the HIP-event figure of this run times; for the forward also 5, the same kernel
                # with non-temporal stash stores, which a sweep's early layers run), never the inverse map (REV)
                if kernel == 'k_flow_fwd' and (targs[3] != 'false' or (len(targs) >= 8 and targs[7] not in ('1', '5'))):
                    continue
                if kernel == 'k_flow_bwd_gather' and len(targs) >= 6 and targs[5] != '1':
                    continue
                calls += int(row['Calls']); tot += int(row['TotalDurationNs'])
        meta = {}
        try:
            with open(os.path.join(prof, names[-1].replace('.csv', '.meta.json'))) as f:
                meta = json.load(f)
        except Exception:
            pass
        if not calls:
            return None, None
        return tot / calls * 1e-6, {'file': 'profiles/' + names[-1], 'launches': calls, 'csrc_sha16': meta.get('csrc_sha16'),
                                    'command': meta.get('command'), 'library_sha16': lib_sha16(),
                                    'stale': meta.get('csrc_sha16') != lib_sha16()}
    except Exception:
        return None, None


def make_flow(gen, n_layers):
    """Synthetic flow weights: PyTorch's default Conv2d init (Kaiming-uniform a = sqrt(5), i.e.
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)); SURVEY Q6: the reference's set_weights is a no-op) for the s/t net
    2 -> 8 -> 8 -> 3, k = 3, drawn from `gen`: [(w0, b0, w1, b1, w2, b2)] * n_layers."""
    sizes = [2, 8, 8, 3]
    flow = []
    for _ in range(n_layers):
        w = []
        for ci, co in zip(sizes[:-1], sizes[1:]):
            bound = 1.0 / math.sqrt(ci * 9)
            w.append((torch.rand(co, ci, 3, 3, generator=gen, dtype=torch.float64) * 2 - 1) * bound)
            w.append((torch.rand(co, generator=gen, dtype=torch.float64) * 2 - 1) * bound)
        flow.append(tuple(w))
    return flow


def cpu_leg(R, cfg, flow, xs, vs, us, dt, nstep, threads):
    """One oracle trajectory of the sample chains on `threads` host threads -> (seconds, results)."""
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    if cfg['n_layers'] == 0:
        dH, _, acc, newx = R.hmc(xs, vs, us, cfg['beta'], dt, nstep)
        out = {'dH': dH, 'acc': acc, 'newx': newx}
    else:
        dH, _, acc, newx, h0, h1 = R.ft_hmc(xs, vs, us, flow, cfg['beta'], dt, nstep, mode='md')
        out = {'dH': dH, 'acc': acc, 'newx': newx, 'H0': h0, 'H1': h1}
    return time.perf_counter() - t0, out


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))
    from fthmc_amd import graph_loop, ops, parallel
    rank, world, local = parallel.init()
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: start with `python bench.py --gpus {args.gpus}` '
                         f'(self-launching) or torchrun --nproc-per-node {args.gpus}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback in the product path)')
    local = local % torch.cuda.device_count()          # ranks may share a GPU in a gloo rehearsal
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    # a process group exists for world > 1 -- and for ONE rank under FTHMC_FORCE_PG=1 (parallel.force_group: the RCCL
    # code path of the 8-GPU run rehearsed on a single GPU); every collective below follows the group, not the world size
    grouped = parallel.have_group()
    nccl = grouped and torch.distributed.get_backend() == 'nccl'
    if grouped:
        # first collective NOW: RCCL builds its communicator (and fails, if it is going to) before any graph is
        # captured or any timing starts, on the device this rank will compute on
        t_ = torch.ones(1, dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t_)
        torch.cuda.synchronize()
        if int(t_.item()) != world:
            raise SystemExit(f'rank {rank}: warm-up all-reduce returned {t_.item()} for {world} ranks')
    cfg = CONFIGS[args.config]
    L, BETA, N_LAYERS = cfg['L'], cfg['beta'], cfg['n_layers']
    batch = args.batch if args.batch is not None else cfg['B']
    if args.scaling == 'strong':
        if batch % world:
            raise SystemExit(f'--scaling strong: {batch} chains do not split over {world} GPUs')
        B_total = batch
    else:
        B_total = batch * world
    lo, hi = parallel.shard_range(B_total, rank, world)
    B = hi - lo
    dt = TAU / NSTEP
    flowed = N_LAYERS > 0

    # synthetic inputs, identical for the CPU and GPU paths
    gen = torch.Generator(device='cpu').manual_seed(SEED)
    flow = make_flow(gen, N_LAYERS)
    w = ops.pack_weights(flow, device=dev) if flowed else None
    WKEY = ('bench', time.time_ns()) if flowed else None     # the weights' content version: fixed for the run
    # Untimed preparation: a hot start U(-pi, pi) at large beta rejects every trajectory, so the chains start
    # near-cold (|x| < 0.1) and are brought to the Wilson ensemble at this beta by plain HMC on the HIP path.
    # The flow is untrained (random init, as the workload prescribes), so the timed ftHMC trajectories accept
    # rarely (5.7 % at config 3): the number measured is throughput of the MD path, not a tuned sampler.
    # Every draw is keyed by the GLOBAL chain id (Philox streams, parallel.chain_seeds), never by the rank: a chain
    # starts from and runs through the same numbers on 1 or 8 GPUs.
    g0, _ = ops.random_momenta(parallel.chain_seeds(SEED + 1, lo, hi, 0).to(dev), (B, 2, L, L), need_u=False)
    x = (0.1 * torch.erf(g0 / math.sqrt(2.0))).contiguous()             # U(-0.1, 0.1) from the chain's normal draws
    for it in range(args.thermalize):
        vt, ut = ops.random_momenta(parallel.chain_seeds(SEED + 7, lo, hi, it).to(dev), (B, 2, L, L))
        x = ops.hmc_trajectory(x, vt, ut, BETA, 0.05, 20)['x_new']
    x0 = x.cpu()

    stats = parallel.RunStats.zeros(dev)
    out = {'x_new': x if flowed else torch.empty_like(x)}
    for k in ('dH', 'acc', 'H0', 'H1', 'plaq', 'Q'):
        out[k] = torch.empty(B, dtype=torch.float64, device=dev)
    if flowed:
        S0, _, p0, q0 = ops.ft_action(x, w, N_LAYERS, BETA)
    else:
        S0, q0, p0 = ops.wilson_action_charge(x, BETA)
    qold = q0.clone()
    seeds = torch.empty(B, dtype=torch.int64, device=dev)
    traj_dev = torch.zeros(1, dtype=torch.int64, device=dev)          # trajectories done, on the device: keys every trajectory's draws
    v = torch.empty_like(x)
    u = torch.empty(B, dtype=torch.float64, device=dev)
    stream = torch.cuda.Stream(device=dev)

    # chain groups (ops.ft_trajectory(groups=G)): contiguous blocks of this GPU's chains whose trajectories
    # run on concurrent streams, forked from / joined into `stream`
    G = args.groups if args.groups is not None else ops.default_groups(B, L)
    G = max(1, min(G, B)) if flowed else 1
    Gsplit = [int(t) for t in args.group_sizes.split(',')] if (args.group_sizes and flowed) else G
    if isinstance(Gsplit, list):
        G = len(Gsplit)
    state = torch.stack([S0, p0, q0]).contiguous()      # (S_eff, plaq, Q) of the current x, carried along
    out['state'] = state

    def enqueue(stateless=False):
        """momentum refresh + one trajectory of every chain of this GPU, forked from the current stream.
        stateless: H0 is recomputed by a flow sweep at the start of the trajectory, as the reference does (ft_hmc.py:205),
        instead of being carried over from the previous trajectory's H1 (bit-identical numbers either way)."""
        # the trajectory's per-chain seeds from (SEED, global chain id, trajectory counter) on the device; the counter moves on inside
        # the launch, so a replay costs the host no copy (= parallel.chain_seeds(SEED, lo, hi, trajectory))
        ops.chain_seeds(SEED, lo, B, counter=traj_dev, advance=True, out=seeds)
        ops.random_momenta(seeds, x.shape, out_v=v, out_u=u)
        if flowed:
            # in place: the accepted field replaces x, its (S_eff, plaq, Q) replace the carried state (the C ABI allows
            # x_new == x and state_out == state_in: both are read before they are written)
            ops.ft_trajectory(x, v, u, w, N_LAYERS, BETA, dt, NSTEP, mode='md', out=out, state_in=None if stateless else state,
                              groups=Gsplit, wkey=WKEY)
        else:
            ops.hmc_trajectory(x, v, u, BETA, dt, NSTEP, out=out)          # results written in place: no copy launches behind it
            ops.wilson_action_charge(out['x_new'], BETA, out=out)
        # the chain state moves on and the run statistics accumulate inside the same (captured) sequence
        if not flowed:
            x.copy_(out['x_new'])
        stats.add_device(out['acc'], out['plaq'], out['Q'], qold, out['dH'])      # one launch; also qold <- Q

    graph = graph_sl = None
    if not args.no_graph:
        # the launches of a trajectory (~200 with the flow) are captured once and replayed (launch-bound otherwise)
        with torch.cuda.stream(stream):
            enqueue()                       # warm allocator / workspaces before capture (trajectory 0's draws)
            stream.synchronize()
            graph = torch.cuda.CUDAGraph()
            # The weights do not change while sampling: every trajectory call states their content version (wkey -> the C ABI's
            # `_v` entry points), the library expands them once and finds its stamps on the device from then on.
            # thread_local: the process group's watchdog thread may poll its events while this thread captures
            with graph_loop.capture(graph, stream):
                enqueue()
            if flowed:                      # the stateless variant of the same trajectory, for the side figure below
                graph_sl = torch.cuda.CUDAGraph()
                with graph_loop.capture(graph_sl, stream):
                    enqueue(stateless=True)

    traj = [0]
    pending = [None]

    def step(stateless=False):
        if graph is not None:
            (graph_sl if stateless else graph).replay()
        else:
            enqueue(stateless)
        if pending[0] is not None:
            pending[0].wait()
        pending[0] = stats.reduce(async_op=grouped)
        traj[0] += 1

    def barrier():
        if grouped:
            torch.distributed.barrier(device_ids=[local]) if nccl else torch.distributed.barrier()
        torch.cuda.synchronize()

    log(f'rank {rank}/{world}: config {args.config}, {B} chains here / {B_total} in total, groups {G}, '
        f'graph={"yes" if graph is not None else "no"}; warmup {args.warmup}')
    def region(stateless=False):
        """EXACTLY --steps trajectories between two (barrier + device synchronize) brackets; seconds, MAX over ranks"""
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(stateless)
        if pending[0] is not None:
            pending[0].wait()
            pending[0] = None
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if grouped:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t)

    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            step()
        barrier()
        log('timed region ...')
        stats.vec.zero_(); stats.glob = None
        times = [region()]
        # a region shorter than a second is at the mercy of the clock ramp and of one late host wake-up: time more
        # regions of the same --steps trajectories and report the median one (every rank sees the same MAX-reduced
        # time, so all ranks agree on the count)
        # ... and keep going until --min-seconds of timed GPU work have accumulated, so that the record of a run shows the
        # card busy whatever --steps was (20 trajectories are 0.12 s)
        nmin = args.regions if args.regions else (5 if times[0] < 1.0 else 1)
        while len(times) < nmin or (not args.regions and sum(times) < args.min_seconds and len(times) < 400):
            times.append(region())
        # side figure: the same trajectories with H0 recomputed (one more flow sweep per trajectory), one region
        elapsed_sl = region(stateless=True) if flowed else None
    elapsed = sorted(times)[len(times) // 2]
    if args.dump:
        import numpy as np
        np.savez(f'{args.dump}.{world}.{rank}.npz', lo=lo, hi=hi, x=x.cpu().numpy(), dH=out['dH'].cpu().numpy(),
                 acc=out['acc'].cpu().numpy(), Q=out['Q'].cpu().numpy(), plaq=out['plaq'].cpu().numpy(), x0=x0.numpy())

    # ---- training leg (config 5): train_step's compute = ops.train_grad on a fixed prior draw, all ranks,
    #      gradients all-reduced (C2) like train.train_step does
    train = None
    if cfg['train']:
        gxi, _ = ops.random_momenta(parallel.chain_seeds(SEED + 31, lo, hi, 0).to(dev), (B, 2, L, L), need_u=False)
        xi = (math.pi * torch.erf(gxi / math.sqrt(2.0))).contiguous()          # prior draw U(-pi, pi), keyed by chain id
        Gt = ops.default_train_groups(B, L)

        def tstep():
            r = ops.train_grad(xi, w, N_LAYERS, BETA, groups=Gt)
            gw = r['gw'] / world
            parallel.allreduce_grads(gw)
            return r
        with torch.cuda.stream(stream):
            for _ in range(2):
                tstep()
            barrier()
            tt0 = time.perf_counter()
            nt = max(3, min(args.steps, 10))
            for _ in range(nt):
                tstep()
            barrier()
            tt = time.perf_counter() - tt0
        tt_t = torch.tensor([tt], dtype=torch.float64, device=dev)
        if grouped:
            torch.distributed.all_reduce(tt_t, op=torch.distributed.ReduceOp.MAX)
        tt = float(tt_t)
        tflops = N_LAYERS * L * L * TRAIN_FLOPS_PER_SITE * B_total * nt / tt / 1e12

        def train_wall(Lw, Bw, nlw, betaw, steps_w):
            """wall milliseconds per WHOLE training step (prior draw on the device, fthmc_train_grad, metrics, Adam on the flat
            parameter buffer; with a process group also the C2 all-reduces) through train.GraphTrainer, MAX over ranks"""
            from fthmc_amd import train as T
            from fthmc_amd.config import TrainConfig
            tc = TrainConfig(L=Lw, beta=betaw, n_layers=nlw, batch_size=Bw, base_lr=1e-3, print_freq=0)
            torch.manual_seed(SEED)
            model = T.get_model(tc)
            tr = T.GraphTrainer(model, tc, T.make_optimizer(model, tc), Bw, seed=SEED + 31)
            for _ in range(3):
                tr.step()
            tr.synchronize(); barrier()
            t0_ = time.perf_counter()
            for _ in range(steps_w):
                tr.step()
            tr.synchronize(); barrier()
            tw = torch.tensor([time.perf_counter() - t0_], dtype=torch.float64, device=dev)
            if grouped:
                torch.distributed.all_reduce(tw, op=torch.distributed.ReduceOp.MAX)
            m_ = tr.metrics()
            captured_ = tr.captured
            del tr, model
            mode_ = ('hipGraph replay' + (' (C2 collectives captured)' if grouped else '')) if captured_ else 'eager + collectives'
            return float(tw) / steps_w * 1e3, bool(np.isfinite(m_['loss_dkl'])), mode_
        import numpy as np
        wall_ms, wall_ok, wall_mode = train_wall(L, B, N_LAYERS, BETA, nt)
        small_ms, small_ok, _ = train_wall(16, 512, 8, 4.0, 100)
        ops.release_workspaces()
        train = {'train_steps_per_s': round(nt / tt, 3), 'ms_per_train_step': round(tt / nt * 1e3, 3),
                 'wall_ms_per_train_step': round(wall_ms, 3), 'wall_over_compute': round(wall_ms / (tt / nt * 1e3), 4),
                 'wall': {'what': 'whole steps of fthmc.train (GraphTrainer: prior draw, fthmc_train_grad, metrics, FlatAdam), '
                                  'no host synchronisation inside the loop', 'launch': wall_mode, 'loss_finite': wall_ok},
                 'wall_L16_B512_8layers': {'ms_per_train_step': round(small_ms, 4), 'loss_finite': small_ok,
                                           'note': 'the size the reference trains at before transfer_to_new_lattice (train.py:434-455)'},
                 'batch_total': B_total, 'steps': nt,
                 'algorithmic_flops_per_sample_step': N_LAYERS * L * L * TRAIN_FLOPS_PER_SITE,
                 'achieved_TFLOPs': round(tflops, 3), 'frac_of_fp64_peak': round(tflops / (FP64_PEAK_TFLOPS * world), 4),
                 'note': 'ms_per_train_step / train_steps_per_s: fthmc_train_grad (forward with stash, backward with MFMA weight '
                         'gradients) + gradient all-reduce on a fixed draw; wall_ms_per_train_step: whole steps incl. the prior draw, '
                         'the metrics and the optimizer'}
    if rank != 0:
        if grouped:
            torch.distributed.destroy_process_group()
        return

    log(f'timed region(s) done: {len(times)} x {args.steps} trajectories, median {elapsed:.3f} s '
        f'(min {min(times):.3f}, max {max(times):.3f})')
    chain_steps = B_total * NSTEP * args.steps
    value = chain_steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline of the dominant kernel, HIP events on this stream
    with torch.cuda.stream(stream):
        ms_leap = ops.time_kernel('leap_step', x, beta=BETA, reps=40)
        ms_traj = ops.time_kernel('hmc_trajectory', x, beta=BETA, reps=20) if L <= 64 else None
    stencil = {'kernel': ('k_leap_rows<8> (fused plain-HMC leapfrog step: 8 rows x 64 columns per workgroup, two sites and 16-byte '
                          'accesses per thread, one launch per step)' if L % 64 == 0 else
                          'k_force<1> (fused plain-HMC leapfrog step on 16 x 16 tiles, one launch per step)'),
               'avg_launch_ms': round(ms_leap, 5),
               'achieved_GBps': round(64.0 * L * L * B / (ms_leap * 1e-3) / 1e9, 1), 'peak_GBps': HBM_PEAK_GBPS}
    if ms_traj is not None:
        stencil['persistent'] = {
            'kernel': 'k_hmc_trajectory (10 plain-HMC steps + H0/H1 + accept in one launch, links in LDS, momenta in registers)',
            'avg_launch_ms': round(ms_traj, 5), 'steps_per_launch': 10,
            'algorithmic_GBps': round(64.0 * L * L * B * 10 / (ms_traj * 1e-3) / 1e9, 1),
            'note': 'no HBM traffic between steps: the per-step HBM model is an upper bound'}
    small = flowed and L <= 16 and ops.get_small_path() and ops.get_variant() == 1
    if small:
        # small lattices: ONE launch per trajectory (csrc/flow_small.hip: a chain per workgroup, the whole MD loop on the
        # device).  Algorithmic flops of a launch = the dense conv flops of its sweeps: nstep force evaluations
        # (forward + adjoint) and two action evaluations (H0, H1; the timed region carries H0 over and runs one).
        with torch.cuda.stream(stream):
            ms_k = ops.time_small(x, v, u, w, N_LAYERS, BETA, dt, NSTEP, reps=40)
        sweeps_fwd, sweeps_bwd = NSTEP + 2, NSTEP
        flops_launch = CONV_FLOPS_PER_SITE * L * L * N_LAYERS * B * (sweeps_fwd + sweeps_bwd)
        achieved = flops_launch / (ms_k * 1e-3) / 1e12
        step_flops = 2 * CONV_FLOPS_PER_SITE * L * L * N_LAYERS * B
        roofline = {
            'bound': 'mfma', 'kernel': f'k_ft_small<{L}> (whole ftHMC trajectory in one launch: one 512-thread workgroup per '
                                       f'chain, links / activations / plaquette gradient in LDS, conv1 / conv2 / conv2^T on '
                                       f'v_mfma_f64_16x16x4_f64; {sweeps_fwd} forward and {sweeps_bwd} backward sweeps of {N_LAYERS} layers)',
            'achieved': round(achieved, 3), 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(achieved / FP64_PEAK_TFLOPS, 4),
            'traffic': None, 'avg_launch_ms': round(ms_k, 4), 'algorithmic_flops_per_launch': flops_launch,
            'chains_per_launch': B, 'workgroups_per_launch': B,
            'frac_of_occupied_cus': round(achieved / (FP64_PEAK_TFLOPS * min(B, 256) / 256), 4),
            'whole_step_tflops': round(step_flops * (NSTEP * args.steps) / elapsed / 1e12, 3),
            'whole_step_frac': round(step_flops * (NSTEP * args.steps) / elapsed / 1e12 / FP64_PEAK_TFLOPS, 4),
            'note': f'{B} chains = {B} workgroups on 256 CUs: the launch is bound by the latency of ONE chain through its '
                    f'{(sweeps_fwd + sweeps_bwd) * N_LAYERS} dependent layer passes, not by the chip; frac_of_occupied_cus '
                    'prices the kernel against the CUs it can occupy',
            'stencil': stencil,
        }
    elif flowed:
        # launch shape of the timed region: one launch = one layer over one chain group (B / G chains)
        Bl = (max(Gsplit) if isinstance(Gsplit, list) else B // G) if G > 1 else B
        with torch.cuda.stream(stream):
            w0 = w[:955].contiguous()
            xl = x[:Bl].contiguous()
            ms_bwd = ops.time_kernel('flow_bwd', xl, w0, mu=0, off=0, beta=BETA, reps=40)
            ms_fwd = ops.time_kernel('flow_fwd', xl, w0, mu=0, off=0, beta=BETA, reps=40)
            ms_bwd_full = ops.time_kernel('flow_bwd', x, w0, mu=0, off=0, beta=BETA, reps=40) if Bl != B else ms_bwd
            ms_fwd_full = ops.time_kernel('flow_fwd', x, w0, mu=0, off=0, beta=BETA, reps=40) if Bl != B else ms_fwd
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
        traffic, traffic_src = pmc_traffic(names[dom][1])
        traffic_b, _ = pmc_traffic(names['bwd'][1])
        frac = lambda ms, nb: round(CONV_FLOPS_PER_SITE * L * L * nb / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4)
        # the committed rocprofv3 summary is of the HEADLINE command: its launch times apply to this run's launch shape only there
        ms_prof, prof_src = rocprof_launch(names[dom][1]) if (args.config == 3 and Bl == 64) else (None, None)
        ms_prof_b, _ = rocprof_launch(names['bwd'][1]) if (args.config == 3 and Bl == 64) else (None, None)
        roofline = {
            'bound': 'mfma', 'kernel': names[dom][0],
            'achieved': round(achieved, 3), 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(achieved / FP64_PEAK_TFLOPS, 4), 'traffic': traffic, 'traffic_source': traffic_src,
            'avg_launch_ms': round(ms_dom, 4),
            'avg_launch_note': 'HIP events around back-to-back launches of the kernel alone, on this stream, in this run; '
                               'rocprof_avg_launch_ms: the same kernel inside the timed sequence (both chain groups\' streams running) from the '
                               'committed rocprofv3 summary -- frac_rocprof follows from it to the digit',
            'rocprof_avg_launch_ms': None if ms_prof is None else round(ms_prof, 5),
            'frac_rocprof': None if ms_prof is None else frac(ms_prof, Bl), 'rocprof_source': prof_src,
            'algorithmic_flops_per_launch': flops_launch,
            'chains_per_launch': Bl,
            'kernel_ms_per_trajectory': {k: round(v_, 3) for k, v_ in share.items()},
            'full_batch_exclusive': {'chains_per_launch': B, 'fwd_kernel_ms': round(ms_fwd_full, 4), 'bwd_kernel_ms': round(ms_bwd_full, 4),
                                     'fwd_frac': frac(ms_fwd_full, B), 'bwd_frac': frac(ms_bwd_full, B)},
            'fwd_kernel_ms': round(ms_fwd, 4), 'bwd_kernel_ms': round(ms_bwd, 4),
            'bwd_kernel': {'kernel': names['bwd'][0], 'achieved': round(flops_launch / (ms_bwd * 1e-3) / 1e12, 3),
                           'frac': frac(ms_bwd, Bl), 'traffic': traffic_b,
                           'rocprof_avg_launch_ms': None if ms_prof_b is None else round(ms_prof_b, 5),
                           'frac_rocprof': None if ms_prof_b is None else frac(ms_prof_b, Bl)},
            'whole_step_tflops': round(step_flops * (NSTEP * args.steps) / elapsed / 1e12, 3),
            'whole_step_frac': round(step_flops * (NSTEP * args.steps) / elapsed / 1e12 / FP64_PEAK_TFLOPS, 4),
            'attainable': ATTAINABLE,
            'stencil': stencil,
        }
    else:
        # plain HMC: HBM bound; the trajectory kernel moves 64 L^2 B bytes per step algorithmically
        ms_k = ms_traj if ms_traj is not None else ms_leap * NSTEP
        gbps = 64.0 * L * L * B * NSTEP / (ms_k * 1e-3) / 1e9
        roofline = {'bound': 'hbm', 'kernel': stencil.get('persistent', stencil)['kernel'],
                    'achieved': round(gbps, 2), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': round(gbps / HBM_PEAK_GBPS, 5),
                    'traffic': None, 'avg_launch_ms': round(ms_k, 5),
                    'algorithmic_bytes_per_launch': 64 * L * L * B * NSTEP, 'chains_per_launch': B,
                    'note': f'{B} chain(s) of {L}x{L}: {2 * L * L * 8 * B} B of state -- one workgroup per chain; '
                            'the launch is latency-bound, not bandwidth-bound',
                    'stencil': stencil}

    # ---- CPU baseline (oracle = "port") on a bounded sample + parity of the HIP path on it
    cpu = None
    parity_ok = True
    if not args.no_cpu_baseline and world == 1:          # rank 0 at N = 1 only
        from oracle import ref_cpu as R
        nthr = host_threads()
        # sample sizes that keep each leg at 10-30 s of CPU work
        nb = args.cpu_chains if args.cpu_chains is not None else {1: 1, 2: 32, 3: 128, 5: 1}[args.config]
        nb = min(nb, B)
        nb1 = {1: 1, 2: 8, 3: 8, 5: 1}[args.config]
        nb1 = min(nb1, nb)
        nstep1 = 2 if args.config == 5 else NSTEP          # one thread at L=256/16 layers: 2 of the 10 steps
        xs = x0[:nb].clone()
        gs = torch.Generator(device='cpu').manual_seed(SEED + 99)
        vs = torch.randn(nb, 2, L, L, generator=gs, dtype=torch.float64)
        us = torch.rand(nb, generator=gs, dtype=torch.float64)
        log(f'cpu baseline: {nb} chain(s) on {nthr} threads, then {nb1} chain(s) on 1 thread ...')

        def timed(xs_, vs_, us_, nstep_, thr):
            """median trajectory time over repeats that add up to ~5 s (at least 3 when a trajectory is short)"""
            t, o = cpu_leg(R, cfg, flow, xs_, vs_, us_, dt, nstep_, thr)
            ts = [t]
            while sum(ts) < 5.0 and len(ts) < 400 and (t < 2.0 or len(ts) < 3):
                ts.append(cpu_leg(R, cfg, flow, xs_, vs_, us_, dt, nstep_, thr)[0])
            return sorted(ts)[len(ts) // 2], o, len(ts)
        tc, oc, reps = timed(xs, vs, us, NSTEP, nthr)
        t1thr, _, reps1 = timed(xs[:nb1], vs[:nb1], us[:nb1], nstep1, 1)
        torch.set_num_threads(nthr)
        log(f'cpu baseline done: {tc:.1f} s on {nthr} threads, {t1thr:.1f} s on 1 thread')
        # the same sample through the HIP path
        xd, vd, ud = xs.to(dev), vs.to(dev), us.to(dev)
        rel = lambda a, b: float(((a.cpu() - b).abs() / b.abs().clamp_min(1e-300)).max())
        par = {'tolerance': PARITY_TOL}
        if flowed:
            r = ops.ft_trajectory(xd, vd, ud, w, N_LAYERS, BETA, dt, NSTEP, mode='md')
            y_c, ld_c = R.flow_forward(oc['newx'], flow)
            _, ld_g, _, _ = ops.ft_action(r['x_new'], w, N_LAYERS, BETA)
            par.update({'H0_rel': rel(r['H0'], oc['H0']), 'H1_rel': rel(r['H1'], oc['H1']),
                        'logdet_abs_over_volume': float((ld_g.cpu() - ld_c).abs().max()) / (L * L)})
        else:
            r = ops.hmc_trajectory(xd, vd, ud, BETA, dt, NSTEP)
            y_c = oc['newx']
            _, qn, pn = ops.wilson_action_charge(r['x_new'], BETA)
            r['plaq'], r['Q'] = pn, qn
        hscale = float(oc['H1'].abs().max()) if flowed else float(R.action(xs, BETA).abs().max() + 0.5 * (vs * vs).flatten(1).sum(1).max())
        border = (us - torch.exp(-oc['dH'])).abs() < 1e-9
        same = (r['acc'].cpu() > 0.5) == oc['acc']
        par.update({'dH_abs': float((r['dH'].cpu() - oc['dH']).abs().max()),
                    'dH_abs_over_H': float((r['dH'].cpu() - oc['dH']).abs().max()) / max(hscale, 1.0),
                    'plaq_rel': rel(r['plaq'], R.plaq_mean(y_c, BETA)),
                    'Q_abs': float((r['Q'].cpu() - R.charge(y_c)).abs().max()),
                    'accept_equal': bool(same[~border].all()), 'borderline_accepts': int(border.sum())})
        checks = [par.get('H0_rel', 0.0), par.get('H1_rel', 0.0), par['dH_abs_over_H'], par['plaq_rel'], par['Q_abs'],
                  par.get('logdet_abs_over_volume', 0.0)]
        parity_ok = all(c <= PARITY_TOL for c in checks) and par['accept_equal']
        par['ok'] = parity_ok
        cpu = {
            'value': round(nb * NSTEP / tc, 3), 'unit': 'chain-leapfrog-steps/s',
            'cores': nthr, 'kind': 'port',
            'sample': f'one trajectory ({NSTEP} leapfrog steps + H0/H1) of {nb} of the {B} chains, '
                      f'oracle/ref_cpu.py (PyTorch CPU fp64 autograd), median of {reps}: {tc:.4g} s',
            'one_thread': {'value': round(nb1 * nstep1 / t1thr, 3), 'cores': 1,
                           'sample': f'{nstep1} leapfrog steps + H0/H1 of {nb1} chain(s), median of {reps1}: {t1thr:.4g} s'},
            'parity': par,
        }

    m = stats.means()
    line = {
        'metric': 'leapfrog-steps/sec (batched chains)', 'value': round(value, 2),
        'unit': 'chain-leapfrog-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': args.scaling,
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': cfg['label'], 'baseline_config': args.config,
                   'chains_per_gpu': B, 'chains_total': B_total, 'L': L, 'beta': BETA, 'n_layers': N_LAYERS,
                   'nstep': NSTEP, 'tau': TAU, 'parallelism': f'chains sharded x{world}',
                   'process_group': (torch.distributed.get_backend() if grouped else None),
                   'launch': 'eager' if graph is None else 'hipGraph replay', 'chain_groups': G,
                   'path': 'small-lattice fused (one launch per trajectory)' if (flowed and L <= 16 and ops.get_small_path()) else
                           ('tiled, one launch per layer' if flowed else 'plain HMC, one launch per trajectory')},
        'batched_leapfrog_steps_per_s': round(NSTEP * args.steps / elapsed, 3),
        'stateless': (None if elapsed_sl is None else
                      {'ms_per_step': round(elapsed_sl / args.steps * 1e3, 4), 'value': round(chain_steps / elapsed_sl, 2),
                       'note': 'H0 recomputed by a flow sweep at the start of every trajectory, as the reference does '
                               '(ft_hmc.py:205); `value` carries S_eff of the accepted field over instead (bit-identical)'}),
        'regions': {'n': len(times), 'seconds': [round(t, 7) for t in times], 'value_from': 'median region',
                    'spread': round((max(times) - min(times)) / elapsed, 4),
                    'note': f'each region = exactly {args.steps} trajectories between barrier + synchronize brackets, '
                            'MAX over ranks; more than one region is timed when the first is shorter than 1 s'},
        'acceptance': round(m['acc'], 4), 'plaq': round(m['plaq'], 6),
        'acceptance_note': 'untrained random-init flow as the workload prescribes: throughput of the MD path, not a tuned sampler',
        'roofline': roofline, 'cpu_baseline': cpu,
    }
    if train is not None:
        line['train'] = train
    print(json.dumps(line), flush=True)
    if grouped:
        torch.distributed.destroy_process_group()
    if not parity_ok:
        log(f'PARITY FAILURE against the oracle (tolerance {PARITY_TOL}): {cpu["parity"]}')
        sys.exit(3)


# Attainable bound of the coupling-layer forward kernel at 16 x 16 tiles, from the work it cannot avoid with this
# algorithm: per workgroup 264 v_mfma_f64_16x16x4 (64 cycles each on one SIMD), 5216 sigmoids
# + 484 sincos + 64 tan-mixture transforms on the fp64 VALU, conv3 at the 64 active sites; fp64 MFMA and fp64 VALU
# share one DP pipe per SIMD (profiles/r01_microbench_fp64.txt).  Filled in by tools/attainable.py.
ATTAINABLE = None
try:
    with open(os.path.join(ROOT, 'profiles', 'attainable.json')) as _f:
        ATTAINABLE = json.load(_f)
except Exception:
    pass


if __name__ == '__main__':
    main()
