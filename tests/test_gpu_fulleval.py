"""GPU: the exact call sequence of the reference's full_eval.py:24-49 through the drop-in -- the GT-query pass
(reconstruction=False; 05_query_pts, one random rotation per query) FIRST, then the reconstruction pass, with the
same mutated ``opt`` object -- on an abc_minimal-layout tree, against the golden the unmodified reference's own
full_eval.py wrote on CPU (oracle/make_golden_sizes.py fulleval)."""
import argparse
import json
import os
import shutil
import sys

import numpy as np
import pytest

from conftest import missing_golden

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(REPO, 'points2surf_amd', 'dropin')
GOLDEN = os.path.join(REPO, 'tests', 'golden')
FIX = os.path.join(GOLDEN, 'abc_minimal')
SEED = 40938661


@pytest.fixture()
def dropin_eval():
    for k in [k for k in sys.modules if k == 'source' or k.startswith('source.')]:
        del sys.modules[k]
    sys.path.insert(0, DROPIN)
    try:
        import source.points_to_surf_eval as ev
        yield ev
    finally:
        sys.path.remove(DROPIN)
        for k in [k for k in sys.modules if k == 'source' or k.startswith('source.')]:
            del sys.modules[k]


def _write_model_files(modeldir, name):
    import torch
    from points2surf_amd import synth
    w, cfg = synth.make_weights(name)
    os.makedirs(modeldir, exist_ok=True)
    torch.save(synth.to_torch_state_dict(w), os.path.join(modeldir, name + '_model.pth'))
    ns = argparse.Namespace(
        outputs=['imp_surf_magnitude', 'imp_surf_sign', 'patch_pts_ids', 'p_index'], points_per_patch=300,
        patch_center='mean', sub_sample_size=1000, patch_radius=float(cfg.get('patch_radius', 0.0)),
        uniform_subsample=int(cfg['uniform_subsample']),
        fixed_subsample=0, net_size=1024, use_point_stn=int(cfg['use_point_stn']), use_feat_stn=1, sym_op='max',
        single_transformer=0, shared_transformer=int(cfg['shared_transformer']), batchSize=501)
    torch.save(ns, os.path.join(modeldir, name + '_params.pth'))


def _eval_predictions_row(pred, gt):
    """the numbers evaluation.eval_predictions (reference source/base/evaluation.py:84-127) reports per shape"""
    nz = ((pred != 0.0) + (gt != 0.0)) > 0
    l2 = pred - gt
    return {'mse': (l2 * l2)[nz].mean(), 'mean_gt': gt.mean(), 'mean_pred': pred.mean(),
            'var_gt': (gt * gt).mean() - gt.mean() * gt.mean(), 'var_pred': (pred * pred).mean() - pred.mean() * pred.mean()}


@pytest.mark.parametrize('model', ['p2s_max', 'p2s_vanilla', 'p2s_medium_radius'])
def test_full_eval_call_sequence_matches_reference(model, dropin_eval, tmp_path):
    # fixed-radius model: distances are not scaled by a ~0.05 patch radius, |sdf| up to 1 -> absolute noise 20x (contract 1e-4)
    tol = 1e-4 if model.endswith('_radius') else 1e-5
    key = 'ref_fulleval_%s_abc3_grid32' % model
    if not os.path.isfile(os.path.join(GOLDEN, key + '.npz')):
        missing_golden(key + ' not generated')
    g = np.load(os.path.join(GOLDEN, key + '.npz'))
    with open(os.path.join(GOLDEN, 'meta_sizes.json')) as f:
        meta = json.load(f)[key]
    ev = dropin_eval
    indir_root = str(tmp_path / 'datasets')
    shutil.copytree(FIX, os.path.join(indir_root, 'abc_minimal'))
    modeldir = str(tmp_path / 'models')
    _write_model_files(modeldir, model)
    opt = ev.parse_arguments(['--indir', indir_root, '--outdir', str(tmp_path / 'results'), '--dataset',
                              'abc_minimal/abc3.txt', '--modeldir', modeldir, '--models', model,
                              '--query_grid_resolution', '32', '--epsilon', '3', '--certainty_threshold', '13',
                              '--sigma', '5', '--workers', '7', '--batchSize', '501', '--cache_capacity', '5'])

    # ---- full_eval.py:19-49, statement by statement (the reference file itself is not on the GPU box) ----
    indir_root = opt.indir
    outdir_root = os.path.join(opt.outdir, opt.models + os.path.splitext(opt.modelpostfix)[0])
    datasets = opt.dataset
    if not isinstance(datasets, list):
        datasets = [datasets]
    for dataset in datasets:
        opt.indir = os.path.join(indir_root, os.path.dirname(dataset))
        opt.outdir = os.path.join(outdir_root, os.path.dirname(dataset))
        opt.dataset = os.path.basename(dataset)
        assert os.path.exists(os.path.join(opt.indir, '05_query_dist'))
        opt.reconstruction = False
        ev.points_to_surf_eval(opt)
        res_dir_eval = os.path.join(opt.outdir, 'eval')
        opt.reconstruction = True
        ev.points_to_surf_eval(opt)
        res_dir_rec = os.path.join(opt.outdir, 'rec')
        # full_eval.py:51-64: the consumer stage, with --workers 7 (the reference would fork a process pool here)
        import source.sdf as sdf
        sdf.implicit_surface_to_mesh_directory(
            os.path.join(res_dir_rec, 'dist_ms'), os.path.join(res_dir_rec, 'query_pts_ms'),
            os.path.join(res_dir_rec, 'vol'), os.path.join(res_dir_rec, 'mesh'),
            opt.query_grid_resolution, opt.sigma, opt.certainty_threshold, opt.workers)

    from points2surf_amd import ply
    from oracle import mc_oracle
    from oracle import p2s_oracle
    with open(os.path.join(FIX, 'abc3.txt')) as f:
        names = [x.strip() for x in f if x.strip()]
    csv_rows = {l.split(',')[0].strip()[:8]: l.split(',') for l in meta['rme_comp_res_csv'].strip().split('\n')[1:]}
    for i, n in enumerate(names):
        pred = np.load(os.path.join(res_dir_eval, 'eval', n + '.xyz.npy'))
        ref = g['eval_%d' % i]
        d = float(np.abs(pred - ref).max())
        flips = int((np.sign(pred) != np.sign(ref)).sum())
        print('%s GT-query pass shape %d: max|dSDF| %.3g, sign flips %d/%d' % (model, i, d, flips, ref.size))
        assert pred.shape == ref.shape == (2000,) and d < tol and flips == 0
        assert os.path.isfile(os.path.join(res_dir_eval, 'eval', n + '.xyz.txt'))
        assert os.path.isfile(os.path.join(res_dir_eval, 'vis', n + '.ply'))
        assert not os.path.exists(os.path.join(res_dir_eval, 'dist_ms'))          # only written in reconstruction mode
        # what eval_predictions (full_eval.py:37-41) computes from our files == the reference's rme_comp_res.csv
        row = _eval_predictions_row(pred, np.load(os.path.join(FIX, '05_query_dist', n + '.ply.npy')))
        got = [c.strip() for c in csv_rows[n[:8]]]      # os.listdir order: match rows by shape id
        for j, k in enumerate(('mse', 'mean_gt', 'mean_pred', 'var_gt', 'var_pred')):
            assert abs(float(got[1 + j]) - row[k]) < 2e-5, (k, got[1 + j], row[k])
        rec = np.load(os.path.join(res_dir_rec, 'dist_ms', n + '.xyz.npy'))
        ref = g['rec_%d' % i]
        d = float(np.abs(rec - ref).max())
        flips = int((np.sign(rec) != np.sign(ref)).sum())
        print('%s reconstruction pass shape %d: max|dSDF| %.3g, sign flips %d/%d' % (model, i, d, flips, ref.size))
        assert rec.shape == ref.shape and d < tol and flips == 0
        for sub in ('eval', 'query_pts_ms'):
            assert os.path.isfile(os.path.join(res_dir_rec, sub, n + '.xyz.npy'))
        assert os.path.isfile(os.path.join(res_dir_rec, 'query_pts_ms_vis', n + '.ply'))
        # rec/mesh/<shape>.ply + rec/vol/<shape>.off: the mesh equals the CPU restatement of the consumer stage run on
        # the REFERENCE's SDF (reference add_samples/propagate semantics + the scikit-image-pinned iso-surface oracle)
        assert os.path.isfile(os.path.join(res_dir_rec, 'vol', n + '.off'))
        mv, mf = ply.read_ply(os.path.join(res_dir_rec, 'mesh', n + '.ply'))
        q = np.load(os.path.join(res_dir_rec, 'query_pts_ms', n + '.xyz.npy'))
        vol_ref = p2s_oracle.sdf_volume(q, ref, 32, 5, 13.0)
        ov, of, _ = mc_oracle.marching_cubes(vol_ref.astype(np.float32))
        ov, of = ply.merge_vertices(ov, of)           # Trimesh(process=True) of the reference (restated, unpinned)
        print('%s shape %d mesh: %d vertices, %d faces (oracle on the reference SDF: %d, %d)'
              % (model, i, mv.shape[0], mf.shape[0], ov.shape[0], of.shape[0]))
        assert (mv.shape[0], mf.shape[0]) == (ov.shape[0], of.shape[0])          # identical counts
        assert np.array_equal(mf, of) and np.abs(mv - ov).max() < 1e-4


def test_random_rotations_and_transform_match_the_oracle():
    """stage-wise: p2s_random_rotations == random_rotation_matrix(RandomState.rand(3)) (restated trimesh), and
    p2s_rotate_points == transform_points(...).astype(float32), bit for bit on the float32 outputs"""
    import torch
    from points2surf_amd import engine
    from oracle import trimesh_restated as T
    n = 5000
    rng = engine.Rng(SEED)
    rot = rng.random_rotations(n)
    rot2 = rng.random_rotations(7)                 # the stream continues
    torch.cuda.synchronize()
    rs = np.random.RandomState(SEED)
    ref = np.stack([T.random_rotation_matrix(rs.rand(3))[:3, :3] for _ in range(n + 7)])
    got = np.concatenate([rot.cpu().numpy(), rot2.cpu().numpy()])
    assert np.abs(got - ref).max() < 1e-14
    mt, pos = rng.get_state()                      # generator state == numpy's after 3 * (n + 7) doubles
    st = rs.get_state()
    assert np.array_equal(mt, st[1]) and pos == st[2]
    pts = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, (n, 11, 3)).astype(np.float32)).cuda()
    out = engine.rotate_points(rot, pts).cpu().numpy()
    m = np.zeros((4, 4))
    m[3, 3] = 1.0
    mism = 0
    for i in range(0, n, 7):
        m[:3, :3] = ref[i]
        e = T.transform_points(pts[i].cpu().numpy(), m).astype(np.float32)
        mism += int((e != out[i]).sum())
    assert mism <= 2, mism                         # float64 summation order may differ in the last bit: <= 1 ulp32, rare
    assert np.abs(out - np.einsum('nij,npj->npi', ref[:n], pts.cpu().numpy().astype(np.float64))).max() < 1e-6


def test_gt_query_pass_matches_oracle_without_rotation_too():
    """p2s_infer_queries with r_rot = NULL is plain inference at given points == p2s_infer_shape on the same points"""
    import torch
    from points2surf_amd import engine, synth
    w, cfg = synth.make_weights('p2s_max')
    model = engine.Model(w, cfg)
    cloud = engine.Cloud(np.load(os.path.join(FIX, '04_pts', '00994122_57d9d4755722f9d2d7436f0a_trimesh_000.xyz.npy')))
    sdf_a, q = engine.infer_shape(model, cloud, engine.Rng(SEED), 32, 3, q_end=700, chunk=256)
    sdf_b = engine.infer_queries(model, cloud, engine.Rng(SEED), None, q, chunk=256)
    torch.cuda.synchronize()
    assert torch.equal(sdf_a, sdf_b)
