"""CPU: the reference's OWN, unmodified ``full_eval.py`` driven end to end through the drop-in package (boundary B1):
``python full_eval.py`` semantics with the drop-in's ``source`` directory in front of the reference checkout on
``sys.path`` -- GT-query pass, ``eval_predictions``, reconstruction pass, ``implicit_surface_to_mesh_directory``
(``--workers 7``), ``mesh_comparison``.  No GPU here, so the device calls of points2surf_amd.engine / .metrics are
replaced by the CPU oracles (test infrastructure); what is exercised is every line of host plumbing between the
reference driver and the C ABI wrappers: argument mutation, the two passes on one ``opt`` object, file layout, the
namespace bridge for ``source.sdf`` / ``source.base.evaluation``, no fork of the device stage.
(The numerics of the same sequence are checked on the GPU in tests/test_gpu_fulleval.py.)"""
import os
import shutil
import sys
import types

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(REPO, 'points2surf_amd', 'dropin')
REFERENCE = os.environ.get('P2S_REFERENCE_ROOT', '/root/reference')
FIX = os.path.join(REPO, 'tests', 'golden', 'abc_minimal')

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REFERENCE, 'full_eval.py')), reason='reference checkout not present')


def _install_cpu_engine(monkeypatch):
    """replace the device entry points by the oracles (numpy); tensors stay on the CPU"""
    import torch
    from oracle import p2s_oracle as O, mc_oracle as MC, metrics_oracle as MO
    from points2surf_amd import engine, metrics, ply

    class Cloud:
        def __init__(self, pts, device=None):
            self.pts_np = np.ascontiguousarray(pts[:, :3], dtype=np.float32)
            self.n = self.pts_np.shape[0]

        def query_grid(self, res, eps):
            return torch.from_numpy(O.query_grid(self.pts_np, res, eps)[0])

        def close(self):
            pass

    class Handle:
        def __init__(self, *a, **k):
            self.sub_sample_size, self.device = 1000, torch.device('cpu')

        def close(self):
            pass

    def fake_sdf(q):          # any deterministic field with both signs: a sphere around the cloud centre
        return (0.45 - np.linalg.norm(q, axis=1)).astype(np.float32)

    def infer_shape(model, cloud, rng, res, eps, q_begin=0, q_end=-1, chunk=0, want_queries=True, n_queries=None, rng_patch=None,
                    want_logits=False):
        assert not want_logits                           # only with P2S_TIE_REPORT
        q = O.query_grid(cloud.pts_np, res, eps)[0]
        q = q[q_begin:(q.shape[0] if q_end < 0 else q_end)]
        return torch.from_numpy(fake_sdf(q)), torch.from_numpy(q)

    def infer_queries(model, cloud, rng_sub, rng_rot, queries, chunk=0):
        assert rng_rot is not None                      # the GT-query pass passes the rotation generator
        return torch.from_numpy(fake_sdf(queries.numpy()))

    def sdf_volume(q, d, res, sigma, thr, clamp=True):
        return torch.from_numpy(O.sdf_volume(np.asarray(q), np.asarray(d), res, sigma, thr, clamp).astype(np.float32)), 1

    def marching_cubes(vol, model_space=True, fix_inversion=True):
        v, f, inv = MC.marching_cubes(vol.numpy(), model_space, fix_inversion)
        return torch.from_numpy(v), torch.from_numpy(f), inv

    def mesh_distances(file_in, file_ref, samples_per_model=10000, seed=0, device=None):
        rs = np.random.RandomState(seed)
        sets = []
        for path in (file_in, file_ref):
            v, f = ply.read_ply(path)
            if len(v) == 0 or len(f) == 0:
                return -1.0, -1.0, -1.0, -1.0
            sets.append(MO.sample_surface_even(v, f, 300, rs))
        return MO.mesh_distances(*sets)

    monkeypatch.setattr(engine, 'select_device', lambda idx: torch.device('cpu'))
    monkeypatch.setattr(engine, 'upload', lambda a, dev: torch.from_numpy(np.ascontiguousarray(a)))
    monkeypatch.setattr(engine, 'Model', Handle)
    monkeypatch.setattr(engine, 'Rng', Handle)
    monkeypatch.setattr(engine, 'Cloud', Cloud)
    monkeypatch.setattr(engine, 'infer_shape', infer_shape)
    monkeypatch.setattr(engine, 'infer_queries', infer_queries)
    monkeypatch.setattr(engine, 'sdf_volume', sdf_volume)
    monkeypatch.setattr(engine, 'marching_cubes', marching_cubes)
    monkeypatch.setattr(metrics, 'mesh_distances', mesh_distances)
    monkeypatch.setattr(torch.cuda, 'synchronize', lambda *a, **k: None)


def test_unmodified_full_eval_py_runs_through_the_dropin(tmp_path, monkeypatch):
    import torch
    from points2surf_amd import synth
    saved = {k: v for k, v in sys.modules.items() if k == 'source' or k.startswith('source.') or k in ('full_eval', 'trimesh')}
    for k in saved:
        del sys.modules[k]
    if not hasattr(np, 'int'):
        monkeypatch.setattr(np, 'int', int, raising=False)      # the only shim the reference needs here (numpy >= 1.24)
    _install_cpu_engine(monkeypatch)
    indir = str(tmp_path / 'datasets')
    shutil.copytree(FIX, os.path.join(indir, 'abc_minimal'))
    modeldir = str(tmp_path / 'models')
    os.makedirs(modeldir)
    w, cfg = synth.make_weights('p2s_max')
    torch.save(synth.to_torch_state_dict(w), os.path.join(modeldir, 'p2s_max_model.pth'))
    import argparse
    torch.save(argparse.Namespace(
        outputs=['imp_surf_magnitude', 'imp_surf_sign', 'patch_pts_ids', 'p_index'], points_per_patch=300, patch_center='mean',
        sub_sample_size=1000, patch_radius=0.0, uniform_subsample=1, fixed_subsample=0, net_size=1024, use_point_stn=0,
        use_feat_stn=1, sym_op='max', single_transformer=0, shared_transformer=0, batchSize=501),
        os.path.join(modeldir, 'p2s_max_params.pth'))
    sys.path.insert(0, REFERENCE)
    sys.path.insert(0, DROPIN)                               # the drop-in's ``source`` shadows the reference's
    try:
        import full_eval                                     # /root/reference/full_eval.py, unmodified
        assert full_eval.__file__.startswith(REFERENCE)
        assert full_eval.points_to_surf_eval.__file__.startswith(DROPIN)
        assert full_eval.sdf.__file__.startswith(DROPIN) and full_eval.evaluation.__file__.startswith(DROPIN)
        outdir = str(tmp_path / 'results')
        opt = full_eval.points_to_surf_eval.parse_arguments([
            '--indir', indir, '--outdir', outdir, '--dataset', 'abc_minimal/testset.txt', '--modeldir', modeldir,
            '--models', 'p2s_max', '--query_grid_resolution', '32', '--epsilon', '3', '--certainty_threshold', '13',
            '--sigma', '5', '--workers', '7', '--batchSize', '501', '--cache_capacity', '5'])
        full_eval.full_eval(opt)                             # the whole driver, both passes + meshing + metrics
    finally:
        sys.path.remove(DROPIN)
        sys.path.remove(REFERENCE)
        for k in [k for k in sys.modules if k == 'source' or k.startswith('source.') or k == 'full_eval']:
            del sys.modules[k]
        sys.modules.update(saved)
    root = os.path.join(outdir, 'p2s_max_model', 'abc_minimal')
    shape = '00994122_57d9d4755722f9d2d7436f0a_trimesh_000'
    for rel in ('eval/eval/%s.xyz.npy', 'eval/eval/%s.xyz.txt', 'eval/vis/%s.ply', 'rec/eval/%s.xyz.npy',
                'rec/dist_ms/%s.xyz.npy', 'rec/query_pts_ms/%s.xyz.npy', 'rec/query_pts_ms_vis/%s.ply', 'rec/vol/%s.off',
                'rec/mesh/%s.ply'):
        assert os.path.isfile(os.path.join(root, rel % shape)), rel
    assert np.load(os.path.join(root, 'eval', 'eval', shape + '.xyz.npy')).shape == (2000,)
    assert np.load(os.path.join(root, 'rec', 'dist_ms', shape + '.xyz.npy')).shape == (2976,)
    csv = open(os.path.join(root, 'eval', 'rme_comp_res.csv')).read().split('\n')
    assert csv[0].replace(' ', '').startswith('file,mse,meangt,meanpred') and csv[1].startswith('00994122 5')
    rep = open(os.path.join(root, 'rec', 'hausdorff_dist_pred_rec.csv')).read().split('\n')
    assert rep[0].startswith('in mesh,ref mesh,Hausdorff dist new-ref') and len(rep) == 2
    assert float(rep[1].split(',')[4]) > 0.0
    from points2surf_amd import ply
    v, f = ply.read_ply(os.path.join(root, 'rec', 'mesh', shape + '.ply'))
    assert v.shape[0] > 100 and f.shape[0] > 200


def test_launcher_resolves_the_dropin_for_the_reference_cli():
    """``python full_eval.py`` would import the reference's own ``source`` (script directory first on sys.path); the
    launcher runs the same unmodified file with the drop-in in front: its argument parser answers ``--help``"""
    import subprocess
    env = dict(os.environ, PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, '-m', 'points2surf_amd.dropin.run', os.path.join(REFERENCE, 'full_eval.py'), '--help'],
                       cwd=REFERENCE, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert 'the engine has no CPU path' in r.stdout and '--query_grid_resolution' in r.stdout
    # and without a GPU the real run stops at the engine's own check, not at a missing trimesh / skimage import
    r = subprocess.run([sys.executable, '-m', 'points2surf_amd.dropin.run', os.path.join(REFERENCE, 'full_eval.py'),
                        '--indir', os.path.join(REPO, 'tests', 'golden'), '--dataset', 'abc_minimal/testset.txt',
                        '--modeldir', '/nonexistent', '--models', 'p2s_max', '--query_grid_resolution', '32', '--epsilon', '3'],
                       cwd=REFERENCE, env=env, capture_output=True, text=True, timeout=300)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and ('ROCm GPU' in r.stderr or 'No such file' in r.stderr), r.stderr[-1500:]
        assert 'trimesh' not in r.stderr and 'skimage' not in r.stderr
