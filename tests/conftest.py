import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no HIP device')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    out = {}
    for k in z.files:
        v = z[k]
        if v.dtype.kind in 'US':
            out[k] = [str(s) for s in v]
        elif v.dtype == np.int16:
            out[k] = torch.from_numpy(v.astype(np.int64))
        elif v.ndim == 0 and v.dtype.kind in 'iu':
            out[k] = int(v)
        else:
            out[k] = torch.from_numpy(np.asarray(v))
    return out


@pytest.fixture(scope='session')
def ops_golden():
    return load_golden('ops.npz')


@pytest.fixture(scope='session')
def mmd_golden():
    return load_golden('mmd.npz')


@pytest.fixture(scope='session')
def dev():
    return torch.device('cuda:0')


@pytest.fixture(autouse=True)
def _eager_net_mda_calls():
    """The model-level tests pin the call-by-call EAGER form of Net_MDA.forward (many of them spy on ops.knn or feed FPS
    starts through ops.START_PROVIDER, which a replayed call graph never reaches).  The per-call hipGraph replay
    (sug_amd.call_graphs, on by default in the product) is held against exactly that eager form, bit for bit, by
    tests/test_gpu_call_graphs.py, which switches it on for itself."""
    from sug_amd.model.Model import Net_MDA
    keep, Net_MDA.call_graphs = Net_MDA.call_graphs, False
    try:
        yield
    finally:
        Net_MDA.call_graphs = keep
