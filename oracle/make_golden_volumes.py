"""TEST INFRASTRUCTURE ONLY -- the volumes the reference hands to ``measure.marching_cubes_lewiner`` (source/sdf.py:181-215),
built by the UNMODIFIED reference's ``sdf.add_samples_to_volume`` + ``sdf.propagate_sign`` + clamp (source/sdf.py:82-178,
199-201) from the committed SDF goldens of the unmodified reference.  Step 1 of the iso-surface goldens:

    python -m oracle.make_golden_volumes /tmp/vols.npz abc3_256 [more jobs]          (python3.10, needs /root/reference)
    /opt/conda/bin/python3.9 oracle/make_golden_mesh.py volumes /tmp/vols.npz        (scikit-image 0.18.3)

Jobs (name -> entries of the npz, = keys of tests/golden/meta_mesh.json):
  abc3_256     ref_rec_p2s_max_abc3_grid256.npz, the three abc_minimal clouds  -> p2s_max_abc3_<i>_grid256
  abc3_64      ref_rec_<model>_abc3_grid64.npz (both models)                   -> <model>_abc3_<i>_grid64
  testset_<r>  ref_rec_<model>_testset_grid<r>.npz (both models)               -> <model>_grid<r>
  standin2_64  ref_rec_<model>_standin2_grid64.npz (both models)               -> <model>_standin2_<i>_grid64
The query points are the reference's own (``sdf.get_voxel_centers_grid_smaller_pc``, source/sdf.py:46-70).
sigma = 5, certainty_threshold = 13 (experiments/eval_*.sh, full_eval.py:51-64)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
from oracle import ref_shims  # noqa: E402

GOLDEN = os.path.join(REPO, 'tests', 'golden')
ABC3 = ['00011084_fddd53ce45f640f3ab922328_trimesh_019', '00016513_3d6966cd42eb44ab8f4224f2_trimesh_053',
        '00994122_57d9d4755722f9d2d7436f0a_trimesh_000']


def volume(ref_sdf, pts, sdf, res, sigma=5, thr=13.0, eps=3):
    q = ref_sdf.get_voxel_centers_grid_smaller_pc(pts=pts, grid_resolution=res, distance_threshold_vs=eps)
    assert q.shape[0] == sdf.shape[0], (q.shape, sdf.shape)
    vol = np.zeros((res, res, res))
    vol = ref_sdf.add_samples_to_volume(vol, q, sdf)
    vol = ref_sdf.propagate_sign(vol, sigma, thr)
    vol[vol < -1.0] = -1.0
    vol[vol > 1.0] = 1.0
    assert np.array_equal(vol.astype(np.float32).astype(np.float64), vol)       # float32 is lossless here
    return vol.astype(np.float32)


def clouds_of(dataset):
    from points2surf_amd import synth
    abc = [np.load(os.path.join(GOLDEN, 'abc_minimal', '04_pts', n + '.xyz.npy'))[:, :3].astype(np.float32) for n in ABC3]
    if dataset == 'abc3':
        return abc
    if dataset == 'testset':
        return [abc[2]]
    if dataset == 'standin2':
        return [synth.standin_cloud(abc[i], i) for i in range(2)]
    raise SystemExit('unknown dataset ' + dataset)


def main():
    ref_shims.install()
    from source import sdf as ref_sdf
    out_path, jobs = sys.argv[1], sys.argv[2:]
    out = {}
    for job in jobs:
        dataset, res = job.rsplit('_', 1)
        res = int(res)
        models = ['p2s_max'] if job == 'abc3_256' else ['p2s_max', 'p2s_vanilla']
        for model in models:
            g = np.load(os.path.join(GOLDEN, 'ref_rec_%s_%s_grid%d.npz' % (model, dataset, res)))
            for i, pts in enumerate(clouds_of(dataset)):
                name = '%s_grid%d' % (model, res) if dataset == 'testset' else '%s_%s_%d_grid%d' % (model, dataset, i, res)
                out[name] = volume(ref_sdf, pts, g['rec_%d' % i], res)
                v = out[name]
                print(name, 'neg/zero/pos', int((v < 0).sum()), int((v == 0).sum()), int((v > 0).sum()), flush=True)
    np.savez_compressed(out_path, **out)


if __name__ == '__main__':
    main()
