"""CPU: the restated trimesh rotation (oracle/trimesh_restated.py) and the oracle's GT-query pass
(oracle/p2s_oracle.py:infer_queries) against the golden written by the reference's own full_eval.py
(oracle/make_golden_sizes.py fulleval: unmodified reference + the restated trimesh functions)."""
import os

import numpy as np
import pytest

from conftest import missing_golden

from oracle import p2s_oracle as O
from oracle import trimesh_restated as T

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
ABC3 = ['00011084_fddd53ce45f640f3ab922328_trimesh_019', '00016513_3d6966cd42eb44ab8f4224f2_trimesh_053',
        '00994122_57d9d4755722f9d2d7436f0a_trimesh_000']


def test_rotation_matrix_is_a_rotation_and_matches_documented_values():
    rs = np.random.RandomState(7)
    for _ in range(200):
        m = T.random_rotation_matrix(rs.rand(3))
        r = m[:3, :3]
        assert np.allclose(r @ r.T, np.eye(3), atol=1e-14) and abs(np.linalg.det(r) - 1.0) < 1e-13
        assert np.array_equal(m[3], [0, 0, 0, 1]) and np.array_equal(m[:3, 3], [0, 0, 0])
    # Gohlke transformations.py doctest values: random_quaternion uses rand -> (cos(t2) r2, sin(t1) r1, cos(t1) r1, sin(t2) r2)
    assert np.allclose(T.random_quaternion(np.array([0.0, 0.0, 0.0])), [0, 0, 1, 0])
    assert np.allclose(T.quaternion_matrix([1, 0, 0, 0]), np.identity(4))
    # quaternion_matrix([0.99810947, 0.06146124, 0, 0]) == rotation_matrix(0.123, [1, 0, 0])  (published doctest)
    c, s = np.cos(0.123), np.sin(0.123)
    assert np.allclose(T.quaternion_matrix([0.99810947, 0.06146124, 0, 0])[:3, :3], [[1, 0, 0], [0, c, -s], [0, s, c]], atol=1e-7)


def test_transform_points_is_the_float64_homogeneous_product():
    rs = np.random.RandomState(3)
    m = T.random_rotation_matrix(rs.rand(3))
    p = rs.rand(50, 3).astype(np.float32)
    out = T.transform_points(p, m)
    assert out.dtype == np.float64 and np.allclose(out, p.astype(np.float64) @ m[:3, :3].T, atol=1e-15)
    assert np.array_equal(T.transform_points(p, np.eye(4)), p.astype(np.float64))


def test_legacy_rand_matches_numpy():
    rs = np.random.RandomState(40938661)
    mt = O.LegacyMT19937(40938661)
    assert np.array_equal(np.concatenate([mt.rand(3) for _ in range(40)]), rs.rand(120))


@pytest.mark.parametrize('model', ['p2s_max', 'p2s_vanilla'])
def test_oracle_gt_query_pass_matches_reference_full_eval(model):
    from points2surf_amd import synth
    path = os.path.join(GOLDEN, 'ref_fulleval_%s_abc3_grid32.npz' % model)
    if not os.path.isfile(path):
        missing_golden(os.path.basename(path), cpu_test=True)
    g = np.load(path)
    w, cfg = synth.make_weights(model)
    n = 12
    pts = np.load(os.path.join(GOLDEN, 'abc_minimal', '04_pts', ABC3[0] + '.xyz.npy'))
    q = np.load(os.path.join(GOLDEN, 'abc_minimal', '05_query_pts', ABC3[0] + '.ply.npy'))[:n]
    sdf = O.infer_queries(w, cfg, pts, q, O.LegacyMT19937(40938661), O.LegacyMT19937(40938661))
    d = np.abs(sdf - g['eval_0'][:n]).max()
    assert d < 1e-5, d
    assert np.array_equal(np.sign(sdf), np.sign(g['eval_0'][:n]))


def test_oracle_small_cloud_shuffle_pad_matches_reference():
    """cloud with fewer points (800) than the sub-sample size: rng.shuffle of shape.pts IN PLACE + zero padding
    (reference source/base/utils.py:221-226); the oracle's literal restatement against the unmodified reference"""
    from points2surf_amd import synth
    path = os.path.join(GOLDEN, 'ref_rec_p2s_max_small800_grid16.npz')
    if not os.path.isfile(path):
        missing_golden(os.path.basename(path), cpu_test=True)
    g = np.load(path)
    w, cfg = synth.make_weights('p2s_max')
    pts = np.load(os.path.join(GOLDEN, 'small800.xyz.npy'))
    n = 10
    _, sdf = O.infer_shape(w, cfg, pts, 16, 3, O.LegacyMT19937(40938661), query_range=(0, n))
    assert np.abs(sdf - g['rec_0'][:n]).max() < 1e-5
    # the legacy shuffle itself against numpy
    rs, mt = np.random.RandomState(3), O.LegacyMT19937(3)
    a = np.arange(800 * 3, dtype=np.float32).reshape(800, 3)
    b = a.copy()
    rs.shuffle(a)
    mt.shuffle_rows(b)
    assert np.array_equal(a, b) and rs.randint(0, 99, 7).tolist() == mt.randint(99, 7).tolist()
