"""ctypes binding of libfthmc_hip.so (C ABI: include/fthmc_hip.h).

The library is the product: there is NO CPU or eager-PyTorch fallback.  If the
shared object is missing or a tensor is not on a HIP device, the call raises.
"""
from __future__ import annotations

import ctypes
import os

# PyTorch-ROCm ships its own libamdhip64.so.7; importing it first makes the dynamic loader
# resolve this library's DT_NEEDED libamdhip64.so.7 to that same runtime instance, so torch's
# streams / allocations and our launches live in ONE HIP runtime (two would not share streams).
import torch  # noqa: F401  (keep before the CDLL below)
from ctypes import c_char_p, c_double, c_int, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# FTHMC_LIB: another build of the same library (A/B kernel measurements in one run on one device)
LIB_PATH = os.environ.get('FTHMC_LIB') or os.path.join(_HERE, 'libfthmc_hip.so')

W_PER_LAYER = 955
ACT_CODES = {None: 0, 'silu': 0, 'swish': 0, 'relu': 1, 'leaky_relu': 2}
MODE_MD, MODE_LITERAL = 0, 1

_D = c_void_p          # device pointer
_P = c_void_p


class ArchT(ctypes.Structure):
    """fthmc_arch_t (include/fthmc_hip.h): the s/t net's shape, an argument of every entry point that runs the net."""
    _fields_ = [('n_hidden', c_int), ('hidden', c_int * 8), ('kernel_size', c_int), ('n_mix', c_int), ('final_tanh', c_int)]


_A = ctypes.POINTER(ArchT)   # const fthmc_arch_t* (None = the default shape)

# name -> argtypes (restype is int unless listed in _RESTYPE); must match include/fthmc_hip.h
SIGNATURES = {
    'fthmc_version': [],
    'fthmc_strerror': [c_int],
    'fthmc_last_error': [],
    'fthmc_set_variant': [c_int],
    'fthmc_get_variant': [],
    'fthmc_arch_params': [_A],
    'fthmc_set_small_path': [c_int],
    'fthmc_get_small_path': [],
    'fthmc_ws_bytes': [_A, c_int, c_int, c_int],
    'fthmc_train_ws_bytes': [_A, c_int, c_int, c_int],
    'fthmc_wrap': [_D, _D, c_size_t, _P],
    'fthmc_regularize': [_D, _D, c_size_t, _P],
    'fthmc_plaquettes': [_D, _D, c_int, c_int, _P],
    'fthmc_wilson_action_charge': [_D, c_int, c_int, c_double, _D, _D, _D, _P],
    'fthmc_wilson_force': [_D, c_int, c_int, c_double, _D, _P],
    'fthmc_leapfrog': [_D, _D, c_int, c_int, c_double, c_double, c_int, _D, _D, _P, c_size_t, _P],
    'fthmc_kinetic': [_D, c_int, c_int, _D, _P],
    'fthmc_stats_accumulate': [_D, _D, _D, _D, _D, c_int, _D, _P],
    'fthmc_random_momenta': [_D, c_int, c_int, _D, _D, _P],
    'fthmc_hmc_trajectory': [_D, _D, _D, c_int, c_int, c_double, c_double, c_int, _D, _D, _D, _D, _D,
                             _P, c_size_t, _P],
    'fthmc_flow_layer_fwd': [_D, _D, _A, c_int, c_int, c_int, c_int, c_int, _D, _D, _P, c_size_t, _P],
    'fthmc_flow_layer_bwd': [_D, _D, _A, _D, _D, c_int, c_int, c_int, c_int, c_int, _D, _D, _P, c_size_t, _P],
    'fthmc_layer_stash_bytes': [_A, c_int, c_int],
    'fthmc_flow_layer_fwd_stash': [_D, _D, _A, c_int, c_int, c_int, c_int, c_int, _D, _D, _D, _P, c_size_t, _P],
    'fthmc_flow_layer_bwd_stash': [_D, _D, _A, _D, _D, c_int, c_int, c_int, c_int, c_int, _D, _D, _P, c_size_t, _P],
    'fthmc_flow_layer_rev': [_D, _D, _A, c_int, c_int, c_int, c_int, c_int, c_double, _D, _D, _P, c_size_t, _P],
    'fthmc_plaq_coupling_fwd': [_D, _D, _A, c_int, c_int, c_int, c_int, c_int, _D, _D, _P, c_size_t, _P],
    'fthmc_plaq_coupling_bwd': [_D, _D, _A, _D, _D, c_int, c_int, c_int, c_int, c_int, _D, _D, _P, c_size_t, _P],
    'fthmc_plaq_coupling_rev': [_D, _D, _A, c_int, c_int, c_int, c_int, c_int, c_double, _D, _D, _P, c_size_t, _P],
    'fthmc_flow_forward': [_D, _D, _A, c_int, c_int, c_int, c_int, _D, _D, _P, c_size_t, _P],
    'fthmc_flow_reverse': [_D, _D, _A, c_int, c_int, c_int, c_int, c_double, _D, _D, _P, c_size_t, _P],
    'fthmc_ft_action': [_D, _D, _A, c_int, c_int, c_int, c_int, c_double, _D, _D, _D, _D, _P, c_size_t, _P],
    'fthmc_ft_force': [_D, _D, _A, c_int, c_int, c_int, c_int, c_double, _D, _P, c_size_t, _P],
    'fthmc_ft_leapfrog': [_D, _D, _D, _A, c_int, c_int, c_int, c_int, c_double, c_double, c_int, _D, _D, _P, c_size_t, _P],
    'fthmc_ft_trajectory': [_D, _D, _D, _D, _A, c_int, c_int, c_int, c_int, c_double, c_double, c_int, c_int, _D, _D, _D, _D, _D, _D, _D, _D, _D, _P, c_size_t, _P],
    'fthmc_train_grad': [_D, _D, _A, c_int, c_int, c_int, c_int, c_double, _D, _D, _D, _D, _P, c_size_t, _P],
    'fthmc_random_uniform': [_D, c_int, c_int, c_double, c_double, _D, _P],
    'fthmc_chain_seeds': [ctypes.c_int64, ctypes.c_int64, c_int, ctypes.c_int64, _D, c_int, _D, _P],
    'fthmc_ws_head_bytes': [],
    'fthmc_pack_weights': [_D, _A, c_int, c_uint64, _P, c_size_t, _P],
    'fthmc_adam_step': [_D, _D, _D, _D, _D, c_size_t, c_double, c_double, c_double, c_double, c_int, _P],
    'fthmc_train_metrics': [_D, _D, _D, _D, c_int, c_int, c_double, c_double, _D, _P, c_size_t, _P],
    'fthmc_time_kernel': [c_int, _D, _D, _A, c_int, c_int, c_int, c_int, c_int, c_double, c_int, ctypes.POINTER(c_double), _P, c_size_t, _P],
    'fthmc_time_small': [_D, _D, _D, _D, _A, c_int, c_int, c_int, c_int, c_double, c_double, c_int, c_int, ctypes.POINTER(c_double), _P, c_size_t, _P],
    'fthmc_small_profile': [_D, _D, _D, _D, _A, c_int, c_int, c_int, c_int, c_double, c_double, c_int, ctypes.POINTER(c_double), _P, c_size_t, _P],
    'fthmc_profile_stages': [c_int, _D, _D, _A, c_int, c_int, c_int, c_int, c_int, c_double, ctypes.POINTER(c_double), _P, c_size_t, _P],
}
# the `_v` twins of the whole-flow entry points: the same arguments + the caller's weight version (include/fthmc_hip.h)
for _n in ('fthmc_flow_forward', 'fthmc_flow_reverse', 'fthmc_ft_action', 'fthmc_ft_force', 'fthmc_ft_leapfrog', 'fthmc_ft_trajectory'):
    SIGNATURES[_n + '_v'] = SIGNATURES[_n] + [c_uint64]
_RESTYPE = {'fthmc_ws_head_bytes': c_size_t, 'fthmc_layer_stash_bytes': c_size_t, 'fthmc_version': c_char_p, 'fthmc_last_error': c_char_p, 'fthmc_train_ws_bytes': c_size_t, 'fthmc_strerror': c_char_p, 'fthmc_ws_bytes': c_size_t}

_lib = None


class FthmcError(RuntimeError):
    pass


def load(path: str = LIB_PATH) -> ctypes.CDLL:
    """dlopen the HIP library and type every entry point; raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise FthmcError(
            f'{path} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'or `make -C fthmc_amd/csrc`.  fthmc_amd has no CPU fallback.')
    lib = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)                 # AttributeError if a symbol is missing
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, c_int)
    if b'DRYRUN' in lib.fthmc_version() and os.environ.get('FTHMC_ALLOW_DRYRUN') != '1':
        # csrc/Makefile `san`: the host-side sanitizer build launches nothing -- it must never stand in for the product
        raise FthmcError(f'{path} is the host-side sanitizer build (launches are no-ops): not a library to compute with')
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        detail = load().fthmc_last_error().decode() if rc == -3 else ''
        raise FthmcError(f'{what} failed: {load().fthmc_strerror(rc).decode()} (code {rc}) {detail}')
