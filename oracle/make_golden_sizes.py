"""TEST INFRASTRUCTURE ONLY -- goldens at the sizes the metric is quoted on, from the UNMODIFIED reference.

Run in the build container only (needs /root/reference; hours of CPU for the large jobs):

    python -m oracle.make_golden_sizes fixtures
    python -m oracle.make_golden_sizes fulleval  p2s_max     abc3    32
    python -m oracle.make_golden_sizes rec       p2s_max     testset 128
    python -m oracle.make_golden_sizes rec       p2s_vanilla abc3    64
    python -m oracle.make_golden_sizes rec       p2s_max     testset 256
    python -m oracle.make_golden_sizes rec       p2s_vanilla testset 32 1        (train --fixed_subsample 1)
    python -m oracle.make_golden_sizes rec       p2s_large_radius testset 128
    python -m oracle.make_golden_sizes rec       p2s_vanilla standin2 64

Jobs
  fixtures  copy the DATA of datasets/abc_minimal (clouds, GT query points / distances, meshes, shape
            lists) to tests/golden/abc_minimal/ so the GPU box has them (no reference source is copied).
  fulleval  run the reference's own ``full_eval.full_eval(opt)`` (full_eval.py:17-49): the GT-query pass
            (reconstruction=False, random rotation per query, data_loader.py:381-393), ``eval_predictions``,
            then the reconstruction pass.  The meshing / mesh-metric stages that follow (full_eval.py:51-75)
            need scikit-image + trimesh and are replaced by no-ops from the outside.
  rec       run ``points_to_surf_eval`` in reconstruction mode only.
  recsample the same with ``--sampling sequential_shapes_random_patches --patches_per_shape 150`` (the sampler's own
            RandomState picks 150 query indices per shape; the queries are evaluated in THAT order).
Datasets: ``testset`` = abc_minimal/testset.txt (1 shape); ``abc3`` = all three abc_minimal shapes in one
list (one dataset-wide RNG stream across the shapes, --workers 0); ``standin2`` = two stand-in clouds
(points2surf_amd/synth.py:standin_cloud, rotation seeds 0 / 1) as one dataset.
Weights: seeded synthetic (points2surf_amd/synth.py, seed 1234) -- no pretrained weights exist offline.
Output: tests/golden/ref_<job>_<model>_<dataset>_grid<res>.npz (float32 arrays) + an entry in
tests/golden/meta_sizes.json (counts, sha256 of the query points, reference wall time, queries/s).
"""
import hashlib
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

from oracle import ref_shims  # noqa: E402
from oracle.make_golden import train_namespace, sha, SEED_DATA  # noqa: E402
from points2surf_amd import synth  # noqa: E402

GOLDEN = os.path.join(REPO, 'tests', 'golden')
FIX = os.path.join(GOLDEN, 'abc_minimal')
ABC = os.path.join(ref_shims.REFERENCE_ROOT, 'datasets', 'abc_minimal')
ABC3 = ['00011084_fddd53ce45f640f3ab922328_trimesh_019',
        '00016513_3d6966cd42eb44ab8f4224f2_trimesh_053',
        '00994122_57d9d4755722f9d2d7436f0a_trimesh_000']


def update_meta(key, value):
    path = os.path.join(GOLDEN, 'meta_sizes.json')
    meta = {}
    if os.path.isfile(path):
        with open(path) as f:
            meta = json.load(f)
    meta[key] = value
    with open(path, 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def job_fixtures():
    for sub in ('03_meshes', '04_pts', '05_query_pts', '05_query_dist'):
        os.makedirs(os.path.join(FIX, sub), exist_ok=True)
        for f in sorted(os.listdir(os.path.join(ABC, sub))):
            dst = os.path.join(FIX, sub, f)
            shutil.copyfile(os.path.join(ABC, sub, f), dst)
            os.chmod(dst, 0o644)
    for f in ('testset.txt', 'trainset.txt', 'valset.txt'):
        shutil.copyfile(os.path.join(ABC, f), os.path.join(FIX, f))
        os.chmod(os.path.join(FIX, f), 0o644)
    with open(os.path.join(FIX, 'abc3.txt'), 'w') as f:
        f.write('\n'.join(ABC3) + '\n')
    np.save(os.path.join(GOLDEN, SMALL + '.xyz.npy'), small_cloud())
    print('fixtures ->', FIX)


def dataset_dir(tmp):
    """a writable dataset root: the reference's data directories (symlinks) + our abc3 list"""
    root = os.path.join(tmp, 'datasets', 'abc_minimal')
    os.makedirs(root)
    for sub in ('03_meshes', '04_pts', '05_query_pts', '05_query_dist'):
        os.symlink(os.path.join(ABC, sub), os.path.join(root, sub))
    shutil.copyfile(os.path.join(ABC, 'testset.txt'), os.path.join(root, 'testset.txt'))
    with open(os.path.join(root, 'abc3.txt'), 'w') as f:
        f.write('\n'.join(ABC3) + '\n')
    # the small cloud lives in its own tree (the reference's 04_pts is read-only)
    sroot = os.path.join(tmp, 'datasets', 'small')
    os.makedirs(os.path.join(sroot, '04_pts'))
    np.save(os.path.join(sroot, '04_pts', SMALL + '.xyz.npy'), small_cloud())
    with open(os.path.join(sroot, SMALL + '.txt'), 'w') as f:
        f.write(SMALL + '\n')
    # stand-in clouds of SURVEY 8d configs 3-5 (rotation seeds 0, 1 of the first two abc_minimal clouds in sorted file order)
    bases = [np.load(os.path.join(ABC, '04_pts', n + '.xyz.npy')) for n in sorted(ABC3)]
    synth.make_standin_dataset(os.path.join(tmp, 'datasets', 'standin'), bases, 2, list_name=STANDIN + '.txt')
    return os.path.join(tmp, 'datasets')


STANDIN = 'standin2'        # two stand-in clouds as one dataset (non-fixture geometry: rotated bbox / cell grid / query grid)
PATCHES_PER_SHAPE = 150     # job recsample
SMALL = 'small800'          # 800 points: fewer than the sub-sample size -> shuffle + pad branch (utils.py:221-226)


def small_cloud():
    cloud = np.load(os.path.join(ABC, '04_pts', ABC3[2] + '.xyz.npy'))
    return np.ascontiguousarray(cloud[::43][:800], dtype=np.float32)


def shapes_of(dataset):
    if dataset == SMALL:
        return [SMALL]
    if dataset == STANDIN:
        return ['standin_000', 'standin_001']
    if dataset == 'abc3':
        return ABC3
    with open(os.path.join(ABC, 'testset.txt')) as f:
        return [x.strip() for x in f if x.strip()]


def run(job, model, dataset, res, batch=500, fixed=0):
    import torch
    torch.set_num_threads(int(os.environ.get('P2S_GOLDEN_THREADS', os.cpu_count())))
    ref_shims.install()
    from source import points_to_surf_eval as ref_eval
    sys.path.insert(0, ref_shims.REFERENCE_ROOT)
    w, cfg = synth.make_weights(model, seed=1234)
    tmp = tempfile.mkdtemp(prefix='p2s_golden_')
    out = {}
    meta = {'model': model, 'dataset': dataset, 'grid': res, 'job': job, 'torch': torch.__version__,
            'numpy': np.__version__, 'threads': torch.get_num_threads(), 'batchSize': batch, 'seed': SEED_DATA}
    try:
        modeldir = os.path.join(tmp, 'models')
        os.makedirs(modeldir)
        torch.save(synth.to_torch_state_dict(w), os.path.join(modeldir, model + '_model.pth'))
        ns = train_namespace(cfg, batch=batch)
        ns.fixed_subsample = int(fixed)          # train --fixed_subsample 1: rng.seed(42) before every draw (utils.py:210-211)
        torch.save(ns, os.path.join(modeldir, model + '_params.pth'))
        indir_root = dataset_dir(tmp)
        outdir = os.path.join(tmp, 'out')
        sub = 'small' if dataset == SMALL else ('standin' if dataset == STANDIN else 'abc_minimal')
        args = ['--indir', indir_root, '--outdir', outdir, '--dataset', '%s/%s.txt' % (sub, dataset),
                '--modeldir', modeldir, '--models', model, '--query_grid_resolution', str(res),
                '--epsilon', '3', '--certainty_threshold', '13', '--sigma', '5', '--gpu_idx', '-1',
                '--workers', '0', '--batchSize', str(batch), '--cache_capacity', '5']
        if job == 'recsample':            # --sampling sequential_shapes_random_patches (source/points_to_surf_eval.py:126-136)
            args += ['--sampling', 'sequential_shapes_random_patches', '--patches_per_shape', str(PATCHES_PER_SHAPE)]
        opt = ref_eval.parse_arguments(args)
        names = shapes_of(dataset)
        t0 = time.time()
        if job == 'fulleval':
            import full_eval as ref_full_eval                # the reference's own driver, unmodified
            from source import sdf as ref_sdf
            from source.base import evaluation as ref_evaluation
            ref_sdf.implicit_surface_to_mesh_directory = lambda *a, **k: None       # needs skimage/trimesh
            ref_evaluation.mesh_comparison = lambda *a, **k: None                   # needs trimesh
            ref_full_eval.full_eval(opt)
            res_root = os.path.join(outdir, model + '_model', 'abc_minimal')
            for i, n in enumerate(names):
                out['eval_%d' % i] = np.load(os.path.join(res_root, 'eval', 'eval', n + '.xyz.npy')).astype(np.float32)
            with open(os.path.join(res_root, 'eval', 'rme_comp_res.csv')) as f:
                meta['rme_comp_res_csv'] = f.read()
        else:
            opt.indir = os.path.join(indir_root, sub)
            opt.outdir = os.path.join(outdir, model + '_model', sub)
            opt.dataset = dataset + '.txt'
            opt.reconstruction = True
            ref_eval.points_to_surf_eval(opt)
            res_root = opt.outdir
        meta['reference_seconds'] = time.time() - t0
        nq = 0
        for i, n in enumerate(names):
            d = np.load(os.path.join(res_root, 'rec', 'dist_ms', n + '.xyz.npy')).astype(np.float32)
            q = np.load(os.path.join(res_root, 'rec', 'query_pts_ms', n + '.xyz.npy'))
            out['rec_%d' % i] = d
            if job == 'recsample':        # <outdir>/rec/<shape>.idx: the sampled query indices, in evaluation order
                out['idx_%d' % i] = np.loadtxt(os.path.join(res_root, 'rec', n + '.idx'), dtype=np.int64).astype(np.int32)
            meta.setdefault('shapes', []).append({'name': n, 'queries': int(d.shape[0]), 'query_sha256': sha(q),
                                                  'pos_frac': float((d > 0).mean())})
            nq += d.shape[0]
        if job == 'fulleval':
            nq += sum(out['eval_%d' % i].shape[0] for i in range(len(names)))
        meta['queries_total'] = int(nq)
        meta['reference_queries_per_s'] = nq / meta['reference_seconds']
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    key = 'ref_%s_%s_%s_grid%d' % (job, model, dataset + ('_fixed' if fixed else ''), res)
    meta['fixed_subsample'] = int(fixed)
    np.savez_compressed(os.path.join(GOLDEN, key + '.npz'), **out)
    update_meta(key, meta)
    print(key, {k: v for k, v in meta.items() if k != 'rme_comp_res_csv'}, flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'fixtures':
        job_fixtures()
    else:
        run(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), fixed=int(sys.argv[5]) if len(sys.argv) > 5 else 0)
