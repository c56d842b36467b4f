"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the neighbour index the engine builds on the device
(points2surf_amd/csrc/p2s_cloud.hip: p2s_cloud_create).

The reference builds ``spatial.cKDTree(pts, leaf_size=1000)`` per shape (source/data_loader.py:40-42) and asks it
for the k nearest neighbours (source/base/point_cloud.py:170-175).  The engine replaces the tree by a uniform cell grid
over the bounding box: points counting-sorted by cell (stable: original order inside a cell), ``cell_start`` offsets
and a 3-D summed-area table of the per-cell counts.  What the kd-tree RETURNS (the k nearest ids, float64 ranking) is
pinned elsewhere (tests/test_gpu_parity.py against scipy's own cKDTree through the reference); this file pins the
data structure itself, bit for bit, in numpy float32 arithmetic (this was the host-side C++ builder of rounds 1-2).
"""
import numpy as np


def cell_coord(x, lo, inv, G):
    """floorf((x - lo) * inv) clamped to [0, G-1], all float32"""
    f = np.floor((x.astype(np.float32) - np.float32(lo)) * np.float32(inv))
    return np.clip(f, 0, G - 1).astype(np.int64)


def build(pts):
    pts = np.ascontiguousarray(pts, dtype=np.float32)[:, :3]
    n = pts.shape[0]
    lo = pts.min(axis=0)
    hi = pts.max(axis=0)
    G = int(np.ceil(np.sqrt(n / 32.0)))
    G = max(4, min(G, 128))
    ext = np.float32(max(np.float32(hi[0] - lo[0]), np.float32(hi[1] - lo[1]), np.float32(hi[2] - lo[2])))
    if not ext > 0:
        ext = np.float32(1.0)
    cell = np.float32(np.float32(ext / np.float32(G)) * np.float32(1.0001))
    inv = np.float32(np.float32(1.0) / cell)
    cx = cell_coord(pts[:, 0], lo[0], inv, G)
    cy = cell_coord(pts[:, 1], lo[1], inv, G)
    cz = cell_coord(pts[:, 2], lo[2], inv, G)
    cid = (cx * G + cy) * G + cz
    cnt = np.bincount(cid, minlength=G ** 3)
    cell_start = np.zeros(G ** 3 + 1, dtype=np.int32)
    cell_start[1:] = np.cumsum(cnt)
    order = np.argsort(cid, kind='stable')
    sat = np.zeros((G + 1, G + 1, G + 1), dtype=np.int64)
    sat[1:, 1:, 1:] = cnt.reshape(G, G, G).cumsum(0).cumsum(1).cumsum(2)
    return {'G': G, 'lo': lo.astype(np.float32), 'inv_cell': float(inv), 'cell_start': cell_start,
            'sat': sat.astype(np.int32).reshape(-1), 'sorted_xyz': pts[order], 'sorted_id': order.astype(np.int32)}
