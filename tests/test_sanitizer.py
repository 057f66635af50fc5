"""CPU-only: the HOST side of the C ABI under AddressSanitizer + UBSan (SURVEY 5 "race detection / sanitizers").

`make -C fthmc_amd/csrc san` builds libfthmc_hip_san.so -- the host pass of every source instrumented, the device pass as the
product builds it, launches / copies / memsets as succeeding no-ops (-DFT_DRYRUN) -- and libfthmc_torch_san.so against it;
tests/san_walk.py then takes every entry point through its argument checks, workspace carving and launch sequencing in a
subprocess that has the sanitizer runtime preloaded.  Never on a GPU box (the walk refuses; GPU sanitizers are not available
on this pool)."""
import glob
import json
import os
import shutil
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

CSRC = os.path.join(ROOT, 'fthmc_amd', 'csrc')
SAN = os.path.join(ROOT, 'fthmc_amd', 'libfthmc_hip_san.so')
SAN_TORCH = os.path.join(ROOT, 'fthmc_amd', 'libfthmc_torch_san.so')


def _runtime():
    hits = sorted(glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so'))
    return hits[-1] if hits else None


@pytest.fixture(scope='module')
def san_build():
    if torch.cuda.device_count() > 0:
        pytest.skip('the sanitizer walk passes made-up device pointers: CPU boxes only')
    if shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'):
        pytest.skip('no hipcc: the sanitizer build needs the ROCm toolchain')
    rt = _runtime()
    assert rt, 'libclang_rt.asan-x86_64.so not found under /opt/rocm/lib/llvm'
    r = subprocess.run(['make', '-C', CSRC, 'san'], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    # the build IS instrumented: it needs the sanitizer runtime's entry points
    syms = subprocess.run(['nm', '-D', '--undefined-only', SAN], capture_output=True, text=True).stdout
    assert '__asan_init' in syms and '__ubsan_handle' in syms, 'libfthmc_hip_san.so carries no sanitizer instrumentation'
    return rt


def _env(rt):
    # detect_odr_violation=0: both instrumented libraries carry their own copy of libstdc++'s header string constants
    env = dict(os.environ)
    env.update(LD_PRELOAD=rt, FTHMC_LIB=SAN, FTHMC_ALLOW_DRYRUN='1', PYTHONDONTWRITEBYTECODE='1',
               ASAN_OPTIONS='detect_leaks=0:abort_on_error=1:halt_on_error=1:detect_odr_violation=0', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    return env


def test_every_entry_point_walks_clean_under_asan_and_ubsan(san_build):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'san_walk.py')], capture_output=True, text=True,
                       env=_env(san_build), timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    assert 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error:' not in r.stderr, r.stderr[-6000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out['calls'] > 5000 and out['refusals'] > 500 and 'DRYRUN' in out['library']


def test_the_sanitizer_build_is_refused_as_a_product_library(san_build):
    """launches are no-ops in that build: fthmc_amd must never compute with it"""
    env = _env(san_build)
    env.pop('FTHMC_ALLOW_DRYRUN')
    code = 'from fthmc_amd import _lib\ntry:\n    _lib.load()\nexcept _lib.FthmcError as e:\n    print("refused:", e)\n'
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode == 0 and 'refused:' in r.stdout and 'sanitizer build' in r.stdout, (r.stdout, r.stderr[-2000:])


def test_operator_library_registers_and_refuses_cpu_tensors_under_the_sanitizers(san_build):
    """csrc/torch_library.cpp instrumented: registration of every operator schema, and the dispatcher's refusal of CPU tensors
    (the library has a device dispatch key only), with the sanitizer runtime watching"""
    code = '''
import torch
torch.ops.load_library(%r)
names = [n for n in dir(torch.ops.fthmc_hip) if not n.startswith("_")]
x = torch.zeros(2, 2, 8, 8, dtype=torch.float64)
try:
    torch.ops.fthmc_hip.wilson_force(x, 2.0)
    print("NOT refused")
except (NotImplementedError, RuntimeError) as e:
    print("refused", len(names))
''' % SAN_TORCH
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=_env(san_build), cwd=ROOT, timeout=300)
    assert r.returncode == 0 and 'refused' in r.stdout and 'NOT refused' not in r.stdout, (r.stdout, r.stderr[-4000:])
    assert 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error:' not in r.stderr, r.stderr[-4000:]
