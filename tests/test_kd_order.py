"""The host-built tree order of the fixed-radius patches (p2s_kd_order_host, points2surf_amd/csrc/p2s_ball.hip) against
scipy ITSELF: ``cKDTree(pts, leafsize).indices`` -- the order in which ``query_ball_point`` reports the points of a
ball for one query (reference source/base/point_cloud.py:177, source/data_loader.py:40-42: leafsize 1000).  A host
function of the C ABI: no device needed."""
import glob
import os

import numpy as np
import pytest
from scipy import spatial

from points2surf_amd import engine

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLOUDS = sorted(glob.glob(os.path.join(REPO, 'tests', 'golden', 'abc_minimal', '04_pts', '*.xyz.npy')))


@pytest.mark.parametrize('path', CLOUDS, ids=[os.path.basename(p)[:8] for p in CLOUDS])
@pytest.mark.parametrize('leaf', [1000, 16])
def test_order_of_the_reference_clouds(path, leaf):
    pts = np.load(path).astype(np.float32)
    tree = spatial.cKDTree(pts, leaf)
    order, leaves = engine.kd_order(pts, leaf)
    assert np.array_equal(order, tree.indices)
    # leaves = the leaf nodes of scipy's tree, left to right
    want = []

    def walk(node):
        if node.split_dim < 0:
            want.append(node.start_idx)
        else:
            walk(node.lesser)
            walk(node.greater)
    walk(tree.tree)
    assert np.array_equal(leaves, want + [len(pts)])


@pytest.mark.parametrize('n', [1, 5, 999, 1000, 1001, 2500, 150000])
def test_order_with_ties_and_duplicates(n):
    """equal coordinates (scipy compares by coordinate alone: the introselect permutation decides) and the sliding step
    when every point of a node lies on one side of the median value"""
    r = np.random.RandomState(n)
    pts = r.rand(n, 3).astype(np.float32)
    pts[::7] = pts[0]
    pts[:, 1] = np.round(pts[:, 1], 2)
    for leaf in (1000, 10):
        order, _ = engine.kd_order(pts, leaf)
        assert np.array_equal(order, spatial.cKDTree(pts, leaf).indices), (n, leaf)
    flat = pts.copy()
    flat[:, 0] = 0.25
    flat[: n // 2, 1] = 0.5
    order, _ = engine.kd_order(flat, 10)
    assert np.array_equal(order, spatial.cKDTree(flat, 10).indices)


def test_ball_is_the_tree_order_filtered_by_distance():
    """the premise of the device kernels, checked against scipy: query_ball_point of ONE point = indices filtered by
    ((dx^2 + dy^2) + dz^2) <= r^2 in float64"""
    pts = np.load(CLOUDS[2]).astype(np.float32)
    tree = spatial.cKDTree(pts, 1000)
    P = pts.astype(np.float64)[tree.indices]
    r = np.random.RandomState(3)
    for rad in (0.05, 0.1, 0.2):
        for _ in range(40):
            q = (pts[r.randint(len(pts))] + r.randn(3) * 0.01).astype(np.float32)
            d = P - q.astype(np.float64)
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            assert np.array_equal(np.asarray(tree.query_ball_point(q, rad)), tree.indices[d2 <= rad * rad])
