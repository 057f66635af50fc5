import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))


def golden_flow(g, as_torch=True):
    """Rebuild the list of per-layer weight tuples stored by make_golden.py."""
    import torch
    flow = []
    for li in range(int(g['n_layers'])):
        w = []
        pi = 0
        while f'w{li}_{pi}' in g:
            a = g[f'w{li}_{pi}']
            w.append(torch.from_numpy(a.copy()) if as_torch else a)
            pi += 1
        flow.append(tuple(w))
    return flow


@pytest.fixture(scope='session')
def golden():
    return load_golden
