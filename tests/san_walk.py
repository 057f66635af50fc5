#!/usr/bin/env python3
"""Walk of the C ABI's HOST side under AddressSanitizer + UBSan (run by tests/test_sanitizer.py in a subprocess with the
sanitizer runtime preloaded; SURVEY 5 "race detection / sanitizers").

The library under test is the `make san` build (csrc/Makefile): host pass instrumented, every launch / copy / memset a
succeeding no-op (-DFT_DRYRUN), so each entry point runs ALL of its host code -- argument checks, workspace carving, the
sequencing of a whole trajectory or training gradient, every launcher's grid arithmetic -- on a box WITHOUT a GPU.
Device pointers are made-up addresses: nothing on the host dereferences them (a dereference is exactly what the
sanitizer would report).  Never run against the product library, never on a GPU box (refused below).

Prints one JSON line {"calls": n, "refusals": m}; any sanitizer finding aborts the process (exit code != 0).
"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

if torch.cuda.device_count() > 0:
    sys.exit('san_walk: a GPU is visible -- the walk passes made-up device pointers and must never run on a GPU box')
os.environ['FTHMC_ALLOW_DRYRUN'] = '1'
from fthmc_amd import _lib  # noqa: E402

OK, E_ARG, E_UNS, E_LAUNCH, E_WS = 0, -1, -2, -3, -4
lib = _lib.load()
assert b'DRYRUN' in lib.fthmc_version(), lib.fthmc_version()

D = 0x7E0000000000            # made-up device addresses, 1 TB apart from each other
_next = [D]


def dev(nbytes=1 << 40):
    p = _next[0]
    _next[0] += 1 << 40
    return p


WS = dev()
calls, refusals = [0], [0]


def arch(hidden=(8, 8), k=3, n_mix=2, tanh=0):
    if (tuple(hidden), k, n_mix, tanh) == ((8, 8), 3, 2, 0):
        return None
    a = _lib.ArchT()
    a.n_hidden, a.kernel_size, a.n_mix, a.final_tanh = len(hidden), k, n_mix, tanh
    for i, h in enumerate(hidden[:8]):
        a.hidden[i] = h
    return ctypes.pointer(a)


def expect(rc, want, what):
    calls[0] += 1
    if want != OK:
        refusals[0] += 1
    ok = rc in want if isinstance(want, tuple) else rc == want
    if not ok:
        sys.exit(f'san_walk: {what}: rc {rc}, expected {want} ({lib.fthmc_strerror(rc).decode()})')


def wsb(A, B, L, nl, train=False):
    n = lib.fthmc_train_ws_bytes(A, B, L, nl) if train else lib.fthmc_ws_bytes(A, B, L, nl)
    calls[0] += 1
    return int(n)


def walk_shape(A, B, L, nl, stash_ok=True):
    """every entry point once with valid arguments for (B, L, nl) -> FTHMC_OK; `stash_ok` False: the activation stash of a
    layer exceeds 2^32 doubles (32-bit plane offsets): the stash paths refuse with FTHMC_ERR_UNSUPPORTED"""
    x, v, u, w, y, o1, o2, o3, o4, st = (dev() for _ in range(10))
    n = wsb(A, B, L, max(nl, 1))
    nt = wsb(A, B, L, max(nl, 1), True)
    assert nt >= n > 0, (B, L, nl, n, nt)
    n0 = wsb(None, B, L, 0)
    tag = f'B={B} L={L} nl={nl}'
    big = OK if stash_ok else (OK, E_UNS)
    FL = big                      # any call that runs the tuned coupling kernels
    expect(lib.fthmc_wrap(x, y, B * 2 * L * L, None), OK, 'wrap')
    expect(lib.fthmc_regularize(x, y, B * 2 * L * L, None), OK, 'regularize')
    expect(lib.fthmc_plaquettes(x, y, B, L, None), OK, 'plaquettes')
    expect(lib.fthmc_wilson_action_charge(x, B, L, 2.0, o1, o2, o3, None), OK, 'action_charge')
    expect(lib.fthmc_wilson_action_charge(x, B, L, 2.0, None, None, None, None), OK, 'action_charge nulls')
    expect(lib.fthmc_wilson_force(x, B, L, 2.0, y, None), OK, 'wilson_force')
    expect(lib.fthmc_kinetic(v, B, L, o1, None), OK, 'kinetic')
    expect(lib.fthmc_stats_accumulate(o1, o2, o3, o4, o1, B, y, None), OK, 'stats')
    expect(lib.fthmc_random_momenta(x, B, 2 * L * L, v, u, None), OK, 'random_momenta')
    expect(lib.fthmc_random_uniform(x, B, 2 * L * L, -3.0, 3.0, v, None), OK, 'random_uniform')
    expect(lib.fthmc_chain_seeds(7, 0, B, 3, o1, 1, o2, None), OK, 'chain_seeds')
    expect(lib.fthmc_leapfrog(x, v, B, L, 2.0, 0.1, 10, y, o1, WS, n0, None), OK, 'leapfrog ' + tag)
    for nstep in (1, 10):
        expect(lib.fthmc_hmc_trajectory(x, v, u, B, L, 2.0, 0.1, nstep, y, o1, o2, o3, o4, WS, n0, None), OK, 'hmc_trajectory ' + tag)
    expect(lib.fthmc_train_metrics(x, y, o1, o2, B, L, 2.0, 1.0, o3, WS, n0, None), OK, 'train_metrics')
    expect(lib.fthmc_adam_step(w, y, o1, o2, o3, 955 * max(nl, 1), 0.9, 0.999, 1e-8, 0.0, 0, None), OK, 'adam')
    if nl == 0:
        expect(lib.fthmc_flow_forward(x, None, A, 0, B, L, 0, y, o1, WS, n0, None), OK, 'flow_forward nl=0')
        expect(lib.fthmc_ft_action(x, None, A, 0, B, L, 0, 2.0, o1, o2, o3, o4, WS, n0, None), OK, 'ft_action nl=0')
        return
    sb = int(lib.fthmc_layer_stash_bytes(A, B, L)); calls[0] += 1
    for mu in (0, 1):
        for off in (0, 3):
            for act in (0, 1, 2):
                expect(lib.fthmc_flow_layer_fwd(x, w, A, B, L, mu, off, act, y, o1, WS, n, None), FL, 'layer_fwd ' + tag)
                expect(lib.fthmc_flow_layer_bwd(x, w, A, o1, o2, B, L, mu, off, act, y, None, WS, n, None), big, 'layer_bwd ' + tag)
            expect(lib.fthmc_flow_layer_bwd(x, w, A, o1, o2, B, L, mu, off, 0, y, o3, WS, nt, None), big, 'layer_bwd gw ' + tag)
            expect(lib.fthmc_flow_layer_rev(x, w, A, B, L, mu, off, 0, 1e-12, y, o1, WS, n, None), FL, 'layer_rev ' + tag)
            if sb:
                expect(lib.fthmc_flow_layer_fwd_stash(x, w, A, B, L, mu, off, 0, y, o1, st, WS, n, None), big, 'fwd_stash ' + tag)
                expect(lib.fthmc_flow_layer_bwd_stash(st, w, A, o1, o2, B, L, mu, off, 0, y, None, WS, n, None), big, 'bwd_stash ' + tag)
                expect(lib.fthmc_flow_layer_bwd_stash(st, w, A, o1, o2, B, L, mu, off, 0, y, o3, WS, nt, None), big, 'bwd_stash gw ' + tag)
            pl = (OK, E_UNS)          # the plaquette-level map is served by the MFMA kernels only
            expect(lib.fthmc_plaq_coupling_fwd(x, w, A, B, L, mu, off, 0, y, o1, WS, n, None), pl, 'plaq_fwd ' + tag)
            expect(lib.fthmc_plaq_coupling_rev(x, w, A, B, L, mu, off, 0, 1e-12, y, o1, WS, n, None), pl, 'plaq_rev ' + tag)
            expect(lib.fthmc_plaq_coupling_bwd(x, w, A, o1, o2, B, L, mu, off, 0, y, o3, WS, nt, None), pl + ((E_UNS,) if not stash_ok else ()), 'plaq_bwd ' + tag)
    for ver in (0, 1, 0xFFFFFFFFFFFFFFFF):
        expect(lib.fthmc_pack_weights(w, A, nl, ver, WS, n, None), OK, 'pack_weights ' + tag)
        expect(lib.fthmc_flow_forward_v(x, w, A, nl, B, L, 0, y, o1, WS, n, None, ver), FL, 'flow_forward ' + tag)
        expect(lib.fthmc_flow_reverse_v(x, w, A, nl, B, L, 0, 1e-12, y, o1, WS, n, None, ver), FL, 'flow_reverse ' + tag)
        expect(lib.fthmc_ft_action_v(x, w, A, nl, B, L, 0, 2.0, o1, o2, o3, o4, WS, n, None, ver), FL, 'ft_action ' + tag)
        expect(lib.fthmc_ft_force_v(x, w, A, nl, B, L, 0, 2.0, y, WS, n, None, ver), big, 'ft_force ' + tag)
        expect(lib.fthmc_ft_leapfrog_v(x, v, w, A, nl, B, L, 0, 2.0, 0.1, 3, y, o1, WS, n, None, ver), big, 'ft_leapfrog ' + tag)
        for mode in (0, 1):
            for state in (None, o4):
                expect(lib.fthmc_ft_trajectory_v(x, v, u, w, A, nl, B, L, 0, 2.0, 0.1, 3, mode, y, o1, o2, None, None, o3, o4, state, st,
                                                 WS, n, None, ver), big, 'ft_trajectory ' + tag)
    expect(lib.fthmc_flow_forward(x, w, A, nl, B, L, 0, None, None, WS, n, None), FL, 'flow_forward nulls')
    expect(lib.fthmc_ft_trajectory(x, v, u, w, A, nl, B, L, 0, 2.0, 0.1, 3, 0, y, o1, o2, o3, o4, None, None, None, None, WS, n, None), big, 'ft_trajectory plain')
    expect(lib.fthmc_train_grad(x, w, A, nl, B, L, 0, 2.0, y, o1, o2, o3, WS, nt, None), big, 'train_grad ' + tag)
    expect(lib.fthmc_train_grad(x, w, A, nl, B, L, 0, 2.0, None, None, None, None, WS, nt, None), FL, 'train_grad no outputs ' + tag)


def refusals_for(A, B, L, nl):
    """FTHMC_ERR_ARG / _WS / _UNSUPPORTED of the entry points that take a workspace"""
    x, v, u, w, y, o1, o2, o3, o4 = (dev() for _ in range(9))
    n = wsb(A, B, L, nl); nt = wsb(A, B, L, nl, True)
    for Lbad in (0, 2, 6, -4):
        expect(lib.fthmc_ft_force(x, w, A, nl, B, Lbad, 0, 2.0, y, WS, n, None), E_ARG, f'ft_force L={Lbad}')
        expect(lib.fthmc_wilson_force(x, B, Lbad, 2.0, y, None), E_ARG, f'wilson_force L={Lbad}')
        expect(lib.fthmc_hmc_trajectory(x, v, u, B, Lbad, 2.0, 0.1, 3, y, o1, o2, o3, o4, WS, n, None), E_ARG, 'hmc_trajectory L')
    for Bbad in (0, -1):
        expect(lib.fthmc_ft_action(x, w, A, nl, Bbad, L, 0, 2.0, o1, o2, o3, o4, WS, n, None), E_ARG, 'ft_action B')
        expect(lib.fthmc_train_grad(x, w, A, nl, Bbad, L, 0, 2.0, y, o1, o2, o3, WS, nt, None), E_ARG, 'train_grad B')
    expect(lib.fthmc_ft_force(None, w, A, nl, B, L, 0, 2.0, y, WS, n, None), E_ARG, 'ft_force x null')
    expect(lib.fthmc_ft_force(x, None, A, nl, B, L, 0, 2.0, y, WS, n, None), E_ARG, 'ft_force w null')
    expect(lib.fthmc_ft_force(x, w, A, nl, B, L, 0, 2.0, None, WS, n, None), E_ARG, 'ft_force F null')
    expect(lib.fthmc_ft_force(x, w, A, -1, B, L, 0, 2.0, y, WS, n, None), E_ARG, 'ft_force nl < 0')
    expect(lib.fthmc_ft_force(x, w, A, nl, B, L, 7, 2.0, y, WS, n, None), E_UNS, 'ft_force act')
    expect(lib.fthmc_ft_force(x, w, A, nl, B, L, 0, 2.0, y, None, n, None), E_WS, 'ft_force ws null')
    expect(lib.fthmc_ft_force(x, w, A, nl, B, L, 0, 2.0, y, WS, n - 8, None), E_WS, 'ft_force ws short')
    expect(lib.fthmc_ft_force(x, w, A, nl, B, L, 0, 2.0, y, WS, 0, None), E_WS, 'ft_force ws 0')
    expect(lib.fthmc_ft_leapfrog(x, v, w, A, nl, B, L, 0, 2.0, 0.1, 0, y, o1, WS, n, None), E_ARG, 'ft_leapfrog nstep')
    expect(lib.fthmc_ft_trajectory(x, v, u, w, A, nl, B, L, 0, 2.0, 0.1, 3, 5, y, o1, o2, o3, o4, None, None, None, None, WS, n, None), E_UNS, 'mode')
    expect(lib.fthmc_ft_trajectory(x, v, None, w, A, nl, B, L, 0, 2.0, 0.1, 3, 0, y, o1, o2, o3, o4, None, None, None, None, WS, n, None), E_ARG, 'u null')
    expect(lib.fthmc_train_grad(x, w, A, nl, B, L, 0, 2.0, y, o1, o2, o3, WS, n - 8, None), E_WS, 'train_grad: the sampling workspace is too small')
    expect(lib.fthmc_train_grad(x, w, A, 0, B, L, 0, 2.0, y, o1, o2, o3, WS, nt, None), E_ARG, 'train_grad nl=0')
    expect(lib.fthmc_flow_layer_fwd(x, w, A, B, L, 2, 0, 0, y, o1, WS, n, None), E_ARG, 'layer_fwd mu')
    expect(lib.fthmc_flow_layer_fwd(x, w, A, B, L, 0, 4, 0, y, o1, WS, n, None), E_ARG, 'layer_fwd off')
    expect(lib.fthmc_flow_layer_bwd(x, w, A, None, o2, B, L, 0, 0, 0, y, None, WS, n, None), E_ARG, 'layer_bwd gy null')
    expect(lib.fthmc_flow_layer_bwd_stash(None, w, A, o1, o2, B, L, 0, 0, 0, y, None, WS, n, None), E_ARG, 'bwd_stash null')
    expect(lib.fthmc_pack_weights(None, A, nl, 1, WS, n, None), E_ARG, 'pack w null')
    expect(lib.fthmc_pack_weights(w, A, 0, 1, WS, n, None), E_ARG, 'pack nl=0')
    expect(lib.fthmc_pack_weights(w, A, nl, 1, WS, 1024, None), E_WS, 'pack ws')
    expect(lib.fthmc_leapfrog(x, v, B, L, 2.0, 0.1, 0, y, o1, WS, n, None), (E_ARG, OK), 'leapfrog nstep=0')
    expect(lib.fthmc_random_uniform(x, B, 16, 1.0, 1.0, v, None), E_ARG, 'uniform hi <= lo')
    expect(lib.fthmc_chain_seeds(7, 0, B, 3, None, 1, o2, None), E_ARG, 'chain_seeds advance without a counter')
    expect(lib.fthmc_time_kernel(9, x, w, A, B, L, 0, 0, 0, 2.0, 1, ctypes.byref(ctypes.c_double()), WS, n, None), E_ARG, 'time_kernel kind')
    expect(lib.fthmc_time_kernel(0, x, w, A, B, L, 0, 0, 0, 2.0, 0, ctypes.byref(ctypes.c_double()), WS, n, None), E_ARG, 'time_kernel reps')


def main():
    assert int(lib.fthmc_ws_head_bytes()) % (64 * 8) == 0 and int(lib.fthmc_ws_head_bytes()) >= 64 * 8768 * 8      # 64 layer regions
    # sizes: never negative, monotone in every argument, zero for nonsense
    for A in (None, arch((4, 6, 5), 5, 1), arch((16,), 3, 3, 1)):
        last = 0
        for B in (1, 2, 31, 128, 1024, 1 << 20):
            for L in (4, 8, 12, 16, 20, 64, 256, 1024):
                for nl in (0, 1, 8, 16, 64, 65, 200):
                    a, b_ = wsb(A, B, L, nl), wsb(A, B, L, nl, True)
                    assert 0 < a <= b_, (B, L, nl, a, b_)
                    assert wsb(A, B + 1, L, nl) >= a and wsb(A, B, L + 4, nl) >= a and wsb(A, B, L, nl + 1) >= a
                    last = a
        assert last > 0
        for bad in ((0, 8, 1), (-3, 8, 1), (2, 0, 1), (2, -8, 1), (2, 8, -1)):
            assert wsb(A, *bad) == 0 and wsb(A, *bad, True) == 0
            assert int(lib.fthmc_layer_stash_bytes(A, bad[0], bad[1])) == 0 or bad[2] < 0
    assert int(lib.fthmc_arch_params(arch((8, 8), 17, 2))) < 0 and int(lib.fthmc_arch_params(arch((300,), 3, 2))) < 0
    assert wsb(arch((8, 8), 4, 2), 2, 8, 1) == 0
    for var in (1, 0):
        expect(lib.fthmc_set_variant(var), OK, 'set_variant')
        for small in (1, 0):
            expect(lib.fthmc_set_small_path(small), OK, 'set_small_path')
            # B = 1, ragged lattices (20, 36: not a multiple of the 16 x 16 tile), the BASELINE shapes, a flow deeper than
            # the workspace head, a net of another shape
            for B, L, nl in ((1, 8, 0), (1, 8, 2), (2, 12, 3), (32, 16, 4), (3, 20, 2), (2, 36, 5), (128, 64, 8), (32, 256, 16),
                             (1, 64, 66), (1, 4, 1)):
                walk_shape(None, B, L, nl)
            walk_shape(arch((4, 6, 5), 5, 1), 3, 16, 2)
            walk_shape(arch((16,), 3, 3, 1), 2, 8, 3)
            # B = 2^20 chains: at L = 8 everything fits 32-bit plane offsets (19 * 2^20 * 64 < 2^32; training 35 * 2^26 > 2^32 is not
            # asked of this shape); at L = 64 a layer's stash is 8.2e10 doubles: the stash paths refuse, everything else runs
            walk_shape(None, 1 << 20, 8, 1, stash_ok=False)
            walk_shape(None, 1 << 20, 64, 2, stash_ok=False)
            refusals_for(None, 4, 16, 2)
            refusals_for(None, 4, 64, 3)
            refusals_for(arch((4, 6, 5), 5, 1), 4, 16, 2)
    expect(lib.fthmc_set_variant(1), OK, 'set_variant'); expect(lib.fthmc_set_small_path(1), OK, 'set_small_path')
    expect(lib.fthmc_set_variant(2), E_ARG, 'set_variant 2'); expect(lib.fthmc_set_small_path(3), E_ARG, 'set_small_path 3')
    for code in (0, -1, -2, -3, -4, -99):
        assert lib.fthmc_strerror(code)
    # the measurement hooks create events / synchronise: without a device they fail cleanly after their host-side set-up
    x, w = dev(), dev()
    out = (ctypes.c_double * 32)()
    n = wsb(None, 4, 16, 4, True)
    for kind in (0, 1, 2, 3):
        expect(lib.fthmc_profile_stages(kind, x, w, None, 4, 64, 0, 0, 0, 2.0, out, WS, wsb(None, 4, 64, 1, True), None), (E_LAUNCH, OK), 'profile_stages')
        expect(lib.fthmc_time_kernel(kind, x, w, None, 4, 64, 0, 0, 0, 2.0, 2, ctypes.byref(ctypes.c_double()), WS, wsb(None, 4, 64, 1), None),
               (E_LAUNCH, OK), 'time_kernel')
    expect(lib.fthmc_time_small(x, x, x, w, None, 4, 4, 16, 0, 2.0, 0.1, 3, 2, ctypes.byref(ctypes.c_double()), WS, n, None), (E_LAUNCH, OK), 'time_small')
    expect(lib.fthmc_small_profile(x, x, x, w, None, 4, 4, 16, 0, 2.0, 0.1, 3, out, WS, n, None), (E_LAUNCH, OK), 'small_profile')
    print(json.dumps({'calls': calls[0], 'refusals': refusals[0], 'library': lib.fthmc_version().decode()}), flush=True)


if __name__ == '__main__':
    main()
