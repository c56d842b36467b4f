"""CPU: the report side of the drop-in's consumer stage -- PLY reader / writer, ``eval_predictions`` and the pairing
rules + CSV of ``mesh_comparison`` -- against the reference's own functions where the reference checkout is present
(its trimesh-dependent distance workers are replaced by the same deterministic stand-ins on both sides)."""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(REPO, 'points2surf_amd', 'dropin')
REFERENCE = os.environ.get('P2S_REFERENCE_ROOT', '/root/reference')
MESHES = os.path.join(REPO, 'tests', 'golden', 'abc_minimal', '03_meshes')


def test_ply_round_trip_and_reference_meshes(tmp_path):
    from points2surf_amd import ply
    rs = np.random.RandomState(0)
    v = rs.rand(50, 3).astype(np.float32)
    f = rs.randint(0, 50, (80, 3)).astype(np.int32)
    p = str(tmp_path / 'm.ply')
    ply.write_ply(p, v, f)
    v2, f2 = ply.read_ply(p)
    assert np.array_equal(v2.astype(np.float32), v) and np.array_equal(f2, f)
    col = rs.rand(50, 3)
    ply.write_ply(p, v, vertex_colors=col)
    v3, f3 = ply.read_ply(p)
    assert np.array_equal(v3.astype(np.float32), v) and f3.shape == (0, 3)
    head = open(p, 'rb').read(300)
    assert b'property uchar red' in head and b'format binary_little_endian 1.0' in head
    assert np.array_equal(ply.float_colors_to_rgba(np.array([[0.0, 0.5, 1.0]])), [[0, 128, 255, 255]])
    # the reference's ground-truth meshes (written by trimesh) parse with the same reader
    for name in sorted(os.listdir(MESHES)):
        vv, ff = ply.read_ply(os.path.join(MESHES, name))
        assert vv.shape[0] > 1000 and ff.shape[0] > 2000 and ff.max() == vv.shape[0] - 1 and np.abs(vv).max() < 1.0
    # ascii variant
    a = str(tmp_path / 'a.ply')
    with open(a, 'w') as fh:
        fh.write('ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n'
                 'element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n')
    va, fa = ply.read_ply(a)
    assert va.shape == (3, 3) and fa.tolist() == [[0, 1, 2]]


@pytest.fixture()
def both_evaluation_modules():
    if not os.path.isfile(os.path.join(REFERENCE, 'source', 'base', 'evaluation.py')):
        pytest.skip('reference checkout not present')
    saved = {k: v for k, v in sys.modules.items() if k == 'source' or k.startswith('source.')}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, REFERENCE)
    sys.path.insert(0, DROPIN)
    try:
        import source.base.evaluation as ours
        ref = sys.modules.get('source.base._reference_evaluation') or ours._ref
        assert ours.__file__.startswith(DROPIN) and ref is not None and ref.__file__.startswith(REFERENCE)
        yield ours, ref
    finally:
        sys.path.remove(DROPIN)
        sys.path.remove(REFERENCE)
        for k in [k for k in sys.modules if k == 'source' or k.startswith('source.')]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_eval_predictions_writes_the_reference_report(both_evaluation_modules, tmp_path):
    ours, ref = both_evaluation_modules
    rs = np.random.RandomState(1)
    pred, gt = str(tmp_path / 'pred'), str(tmp_path / 'gt')
    os.makedirs(pred)
    os.makedirs(gt)
    for name in ('00011084_fddd53ce45f640f3ab922328_trimesh_019', 'zz_shape', 'a_b_c_d_e_f'):
        np.save(os.path.join(pred, name + '.xyz.npy'), rs.normal(0, 0.1, 500).astype(np.float32))
        np.save(os.path.join(gt, name + '.ply.npy'), rs.normal(0, 0.1, 500).astype(np.float32))
    for unsigned in (False, True):
        a, b = str(tmp_path / ('ours_%d.csv' % unsigned)), str(tmp_path / ('ref_%d.csv' % unsigned))
        ours.eval_predictions(pred, gt, a, unsigned=unsigned)
        ref.eval_predictions(pred, gt, b, unsigned=unsigned)
        assert open(a).read() == open(b).read()


def test_mesh_comparison_pairing_and_csv_equal_the_reference(both_evaluation_modules, tmp_path, monkeypatch):
    ours, ref = both_evaluation_modules
    from points2surf_amd import metrics
    new_dir, ref_dir = str(tmp_path / 'new'), str(tmp_path / 'ref')
    os.makedirs(new_dir)
    os.makedirs(ref_dir)
    for n in ('shapeA', 'shapeB', 'extra'):
        open(os.path.join(new_dir, n + '.ply'), 'w').write('x')
    for n in ('shapeA', 'shapeB', 'missing'):
        open(os.path.join(ref_dir, n + '.ply'), 'w').write('x')
    ds = str(tmp_path / 'testset.txt')
    open(ds, 'w').write('shapeA\nshapeB\nmissing\n')

    def fake(file_in, file_ref):
        k = float(len(os.path.basename(file_in)))
        return 0.5 * k, 0.25 * k, 0.5 * k, 3.0 * k
    monkeypatch.setattr(metrics, 'mesh_distances', lambda a, b, samples_per_model=10000, seed=0, device=None: fake(a, b))
    monkeypatch.setattr(ref, '_hausdorff_distance_single_file', lambda a, b, n: (a, b) + fake(a, b)[:3])
    monkeypatch.setattr(ref, '_chamfer_distance_single_file', lambda a, b, n: (a, b, fake(a, b)[3]))
    a, b = str(tmp_path / 'ours.csv'), str(tmp_path / 'ref.csv')
    ours.mesh_comparison(new_dir, ref_dir, 1, a, samples_per_model=100, dataset_file_abs=ds)
    ref.mesh_comparison(new_meshes_dir_abs=new_dir, ref_meshes_dir_abs=ref_dir, num_processes=1, report_name=b,
                        samples_per_model=100, dataset_file_abs=ds)
    assert open(a).read() == open(b).read()
    assert open(a).read().count('\n') == 3 and ',-1,-1,-1,-1' in open(a).read()      # header + A + B + the missing one
    # without a dataset file the reference compares stems with full file names, finds nothing and raises: mirrored
    for mod, kw in ((ours, {}), (ref, {})):
        with pytest.raises(ValueError):
            mod.mesh_comparison(new_dir, ref_dir, 1, a, samples_per_model=100, dataset_file_abs=None)
    with pytest.raises(ValueError):
        ours.mesh_comparison(new_dir, ref_dir, 1, a, dataset_file_abs=str(tmp_path / 'nope.txt'))


def test_merge_vertices_groups_rows_exactly_like_the_row_wise_unique():
    """points2surf_amd/ply.py:merge_vertices (what Trimesh(process=True) does before the export, restated): the grouping by a
    64-bit row hash with exact verification gives the row-wise np.unique's result -- referenced vertices whose coordinates
    agree after rounding to 1e-8 become one vertex in order of first occurrence, faces re-indexed, none removed"""
    import numpy as np
    from points2surf_amd import ply
    rs = np.random.RandomState(3)
    n = 20000
    v = rs.uniform(-1, 1, (n, 3)).astype(np.float32)
    dup = rs.randint(0, n, 3000)
    v[dup] = v[(dup * 7) % n]                       # coincident vertices
    v[::500] = 0.0
    v[7] = -0.0                                     # -0.0 and 0.0 are the same vertex
    f = rs.randint(0, n, (30000, 3)).astype(np.int32)
    mv, mf = ply.merge_vertices(v, f)
    referenced = np.zeros(n, dtype=bool)
    referenced[f.reshape(-1)] = True
    rows = np.round(v[referenced].astype(np.float64) * 1e8).astype(np.int64)
    uniq, first = np.unique(rows, axis=0, return_index=True)
    assert mv.shape[0] == uniq.shape[0] and mf.shape == f.shape and mf.dtype == f.dtype
    # same geometry: every face corner keeps its (rounded) position
    assert np.array_equal(np.round(mv[mf].astype(np.float64) * 1e8), np.round(v[f].astype(np.float64) * 1e8))
    # order of first occurrence
    assert np.array_equal(np.round(mv.astype(np.float64) * 1e8).astype(np.int64), rows[np.sort(first)])
    # no two kept vertices coincide
    assert np.unique(np.round(mv.astype(np.float64) * 1e8).astype(np.int64), axis=0).shape[0] == mv.shape[0]
