"""GPU: row f-4 -- mesh metrics on the device against the CPU restatement (oracle/metrics_oracle.py: restated trimesh
sampling + scipy's own cKDTree / directed_hausdorff), on the abc_minimal ground-truth meshes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MESHES = os.path.join(REPO, 'tests', 'golden', 'abc_minimal', '03_meshes')
NAMES = ['00011084_fddd53ce45f640f3ab922328_trimesh_019', '00016513_3d6966cd42eb44ab8f4224f2_trimesh_053',
         '00994122_57d9d4755722f9d2d7436f0a_trimesh_000']


def _mesh(i):
    import torch
    from points2surf_amd import ply
    v, f = ply.read_ply(os.path.join(MESHES, NAMES[i] + '.ply'))
    return v.astype(np.float32), f.astype(np.int32), torch.from_numpy(v.astype(np.float32)).cuda(), torch.from_numpy(f.astype(np.int32)).cuda()


def test_random_sample_and_surface_sampling_match_the_oracle():
    import torch
    from points2surf_amd import engine, metrics
    from oracle import metrics_oracle as MO
    v, f, vt, ft = _mesh(2)
    rng = engine.Rng(123)
    pts, area, fid = metrics.sample_surface(vt, ft, 5000, rng, want_faces=True)
    torch.cuda.synchronize()
    rs = np.random.RandomState(123)
    ref, area_ref, idx_ref = MO.sample_surface(v, f, 5000, rs)
    assert abs(area - area_ref) < 1e-12 * area_ref
    same = fid.cpu().numpy() == idx_ref
    assert same.mean() > 0.999                               # cumulative areas: parallel scan vs np.cumsum, last-bit ties only
    assert np.abs(pts.cpu().numpy()[same] - ref[same]).max() < 1e-6
    mt, pos = rng.get_state()                                # 3 * 5000 doubles consumed, like numpy
    st = rs.get_state()
    assert np.array_equal(mt, st[1]) and pos == st[2]


def test_remove_close_matches_the_oracle():
    import ctypes
    import torch
    from points2surf_amd import engine, _lib
    from oracle import metrics_oracle as MO
    rs = np.random.RandomState(5)
    pts = rs.random_sample((6000, 3)).astype(np.float32)
    radius = 0.03
    keep_ref, mask = MO.remove_close(pts, radius)
    lib = _lib.load()
    out = torch.empty((6000, 3), dtype=torch.float32, device='cuda')
    n = ctypes.c_int64(0)
    src = torch.from_numpy(pts).cuda()
    _lib.check(lib.p2s_points_remove_close(engine._ptr(src), 6000, ctypes.c_double(radius), 6000, engine._ptr(out), ctypes.byref(n), 0,
                                           engine._stream_ptr(src.device)))
    assert n.value == keep_ref.shape[0] and 100 < n.value < 6000
    assert np.array_equal(out[:n.value].cpu().numpy(), keep_ref)


def test_mesh_distances_match_the_oracle_on_the_same_samples():
    import torch
    from points2surf_amd import engine, metrics
    from oracle import metrics_oracle as MO
    _, _, va, fa = _mesh(1)
    _, _, vb, fb = _mesh(2)
    rng = engine.Rng(9)
    sa = metrics.sample_surface_even(va, fa, 4000, rng)
    sb = metrics.sample_surface_even(vb, fb, 4000, rng)
    assert 3000 < sa.shape[0] <= 4000 and 3000 < sb.shape[0] <= 4000
    h_ab, s_ab = metrics.directed_stats(sa, sb)
    h_ba, s_ba = metrics.directed_stats(sb, sa)
    ref = MO.mesh_distances(sa.cpu().numpy(), sb.cpu().numpy())
    assert abs(h_ab - ref[0]) < 1e-9 and abs(h_ba - ref[1]) < 1e-9
    assert abs((s_ab + s_ba) - ref[3]) < 1e-7 * ref[3]
    # even sampling: no two samples closer than the rejection radius
    import scipy.spatial as spatial
    a = sa.cpu().numpy().astype(np.float64)
    d = spatial.cKDTree(a).query(a, 2)[0][:, 1]
    assert d.min() > 0.0


def test_mesh_comparison_writes_the_reference_csv(tmp_path):
    """full_eval.py:66-75: reconstructed meshes vs 03_meshes; a mesh compared with itself has distance ~ sampling noise"""
    import shutil
    from points2surf_amd import metrics
    new_dir = str(tmp_path / 'mesh')
    os.makedirs(new_dir)
    shutil.copy(os.path.join(MESHES, NAMES[2] + '.ply'), os.path.join(new_dir, NAMES[2] + '.ply'))
    ds = str(tmp_path / 'testset.txt')
    with open(ds, 'w') as f:
        f.write(NAMES[2] + '\n' + NAMES[0] + '\n')
    report = str(tmp_path / 'hausdorff_dist_pred_rec.csv')
    res = metrics.mesh_comparison(new_dir, MESHES, 7, report, samples_per_model=3000, dataset_file_abs=ds)
    lines = open(report).read().split('\n')
    assert lines[0].startswith('in mesh,ref mesh,Hausdorff dist new-ref')
    assert len(lines) == 3
    row = [l for l in lines[1:] if NAMES[2] in l][0].split(',')
    assert 0.0 < float(row[4]) < 0.05 and float(row[5]) > 0.0           # same surface, different samples
    missing = [l for l in lines[1:] if NAMES[0] in l][0].split(',')
    assert missing[2:] == ['-1', '-1', '-1', '-1']                         # in the dataset, never reconstructed
    assert len(res) == 2
