"""TEST INFRASTRUCTURE ONLY -- import harness for the *unmodified* reference.

Runs the reference's own Python code (``/root/reference/source/...``) on CPU in
this container so that the numpy oracle (``oracle/p2s_oracle.py``) and the golden
vectors under ``tests/golden/`` can be pinned to it.  Nothing in the product
path (``points2surf_amd``) may import this module, and it is never used on the
GPU box (``/root/reference`` does not exist there).

The reference needs five compatibility shims with the installed stack
(torch 2.10, numpy 2.2, scipy 1.15; trimesh / scikit-image absent).  All of them
are applied from the outside; no reference file is touched:

 1. ``np.int`` alias           (source/sdf.py:75 uses the removed ``np.int``)
 2. stub ``trimesh`` package   (top-level imports source/data_loader.py:8,
                                source/sdf.py:4, source/base/point_cloud.py:3)
                                + no-op ``sdf.visualize_query_points``
                                (source/points_to_surf_eval.py:219-222,230-234)
                                + ``trimesh.transformations.random_rotation_matrix`` /
                                ``transform_points`` restated (oracle/trimesh_restated.py) for the
                                GT-query evaluation pass (source/data_loader.py:381-393)
 3. ``cKDTree.query(n_jobs=)`` -> ``workers=`` (source/base/point_cloud.py:175,177)
 4. ``.cuda()`` no-ops on CPU  (source/points_to_surf_eval.py:167,362)
 5. ``torch.load(weights_only=False)`` (source/points_to_surf_eval.py:169,316)
"""
import os
import sys
import types
import contextlib

REFERENCE_ROOT = os.environ.get('P2S_REFERENCE_ROOT', '/root/reference')


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'source', 'points_to_surf_model.py'))


def _install_trimesh_stub():
    if 'trimesh' in sys.modules and not getattr(sys.modules['trimesh'], '_p2s_stub', False):
        return  # a real trimesh is installed: use it
    tm = types.ModuleType('trimesh')
    tm._p2s_stub = True

    class Trimesh:  # only ever used as a type annotation / by export paths we no-op
        def __init__(self, *a, **k):
            self.args, self.kwargs = a, k

        def export(self, *a, **k):
            return None

    tm.Trimesh = Trimesh
    for sub in ('transformations', 'repair', 'sample', 'proximity', 'primitives', 'caching'):
        m = types.ModuleType('trimesh.' + sub)
        setattr(tm, sub, m)
        sys.modules['trimesh.' + sub] = m
    # the GT-query evaluation pass (source/data_loader.py:381-393) needs two real functions: restated from
    # the published formulas (oracle/trimesh_restated.py -- parity pinned to that restatement, not to trimesh)
    from oracle import trimesh_restated as _tr
    for fn in ('random_rotation_matrix', 'random_quaternion', 'quaternion_matrix', 'transform_points',
               'identity_matrix'):
        setattr(tm.transformations, fn, getattr(_tr, fn))
    sys.modules['trimesh'] = tm


_installed = False
_saved = {}


def install():
    """Apply the five shims (idempotent) and make the reference importable."""
    global _installed
    if _installed:
        return
    import numpy as np
    import torch
    import scipy.spatial as spatial

    if not reference_available():
        raise RuntimeError('reference not found at %s' % REFERENCE_ROOT)

    # (1)
    if not hasattr(np, 'int'):
        np.int = int
    # (2)
    _install_trimesh_stub()
    # (3)
    _orig_query = spatial.cKDTree.query
    _orig_qbp = spatial.cKDTree.query_ball_point

    def _query(self, *a, n_jobs=None, **k):
        if n_jobs is not None:
            k.setdefault('workers', n_jobs)
        return _orig_query(self, *a, **k)

    def _qbp(self, *a, n_jobs=None, **k):
        if n_jobs is not None:
            k.setdefault('workers', n_jobs)
        return _orig_qbp(self, *a, **k)

    class _KD(spatial.cKDTree):
        query = _query
        query_ball_point = _qbp

    _saved['cKDTree'] = spatial.cKDTree
    spatial.cKDTree = _KD
    # (4)
    _saved['Module.cuda'] = torch.nn.Module.cuda
    _saved['Tensor.cuda'] = torch.Tensor.cuda
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.cuda = lambda self, *a, **k: self
    # (5)
    _saved['torch.load'] = torch.load

    def _load(*a, **k):
        k.setdefault('weights_only', False)
        k.setdefault('map_location', 'cpu')
        return _saved['torch.load'](*a, **k)

    torch.load = _load

    # make ``import source...`` resolve to the reference (and only to it)
    for name in [m for m in sys.modules if m == 'source' or m.startswith('source.')]:
        del sys.modules[name]
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    import warnings
    warnings.filterwarnings('ignore', category=DeprecationWarning)
    from source import sdf as ref_sdf  # noqa: E402
    ref_sdf.visualize_query_points = lambda *a, **k: None
    _installed = True


def uninstall():
    """Undo the monkey patches (the reference stays imported)."""
    global _installed
    if not _installed:
        return
    import torch
    import scipy.spatial as spatial
    spatial.cKDTree = _saved['cKDTree']
    torch.nn.Module.cuda = _saved['Module.cuda']
    torch.Tensor.cuda = _saved['Tensor.cuda']
    torch.load = _saved['torch.load']
    for name in [m for m in sys.modules if m == 'source' or m.startswith('source.')]:
        del sys.modules[name]
    if REFERENCE_ROOT in sys.path:
        sys.path.remove(REFERENCE_ROOT)
    _installed = False


@contextlib.contextmanager
def reference():
    install()
    try:
        yield
    finally:
        uninstall()
