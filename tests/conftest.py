import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected explicitly with -m gpu; if someone runs the whole suite on a box
    # without a GPU they are skipped (never silently passed on a fallback).
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def missing_golden(what, cpu_test=False):
    """a golden file a parity test needs does not exist.  On a box WITH a GPU that is a failure -- the goldens are
    committed, a lost or renamed one must not turn a parity test into a silent green; on the CPU container (where the GPU
    tests do not run anyway) it is a skip.  ``cpu_test``: the caller is an oracle-vs-golden test of the CPU suite: a
    failure everywhere."""
    if cpu_test:
        pytest.fail('golden missing: %s' % what)
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.fail('golden missing on the GPU box: %s' % what)
    pytest.skip('golden not generated: %s' % what)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def fixture_cloud():
    import numpy as np
    return np.load(os.path.join(GOLDEN, 'cloud_abc_00994122.npy'))


def flip_logits(model, w, cfg, cloud, rng, queries, j):
    """the two sign logits that decide whether a sign flip against a golden is a tie (parity.is_tie, the rule of bench.py's
    self-check): the device's own for query ``j`` of ``queries`` (``rng`` positioned at the shape's first draw) and the
    CPU restatement's (oracle/torch_port.py: the reference's ATen ops) on the SAME network inputs"""
    from oracle.torch_port import TorchPort
    from points2surf_amd import engine
    patch, sub, one = engine.query_inputs(model, cloud, rng, queries, int(j))
    lg_dev = float(model.forward(patch, sub, one)[0][0, model.output_dim - 1])
    lg_cpu = float(TorchPort(w, cfg).forward(patch.cpu().numpy(), sub.cpu().numpy(), one.cpu().numpy())[0, model.output_dim - 1])
    return lg_dev, lg_cpu
