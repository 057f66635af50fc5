#!/usr/bin/env python3
"""Fingerprint of the kernel sources (fthmc_amd/csrc/*.hip, *.h and the Makefile with its per-file compiler options, in name order): built into the library
(csrc/Makefile: -DFTHMC_SRC_SHA, reported by fthmc_version()) and stored with every counter summary
(tools/pmc_summary.py), so that bench.py can tell a summary taken on other kernels -- or a library built
from other sources than the ones on disk -- from a current one."""
import hashlib, os, sys


def csrc_sha16(root=None):
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, 'fthmc_amd', 'csrc')
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith(('.hip', '.h')) or name == 'Makefile':
            with open(os.path.join(d, name), 'rb') as f:
                h.update(name.encode() + b'\0' + f.read())
    return h.hexdigest()[:16]


if __name__ == '__main__':
    print(csrc_sha16(sys.argv[1] if len(sys.argv) > 1 else None))
