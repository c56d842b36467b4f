"""CPU: the restatement of the neighbour index (oracle/cloud_index_oracle.py) is what it claims to be: a stable
counting sort by cell plus an inclusive 3-D summed-area table whose box counts equal brute force."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_index_oracle_invariants():
    from oracle import cloud_index_oracle as CO
    pts = np.load(os.path.join(GOLDEN, 'cloud_abc_00994122.npy')).astype(np.float32)
    ix = CO.build(pts)
    G, n = ix['G'], pts.shape[0]
    assert G == int(np.ceil(np.sqrt(n / 32.0)))
    cs = ix['cell_start']
    assert cs[0] == 0 and cs[-1] == n and np.all(np.diff(cs) >= 0)
    cid = (CO.cell_coord(pts[:, 0], ix['lo'][0], ix['inv_cell'], G) * G +
           CO.cell_coord(pts[:, 1], ix['lo'][1], ix['inv_cell'], G)) * G + CO.cell_coord(pts[:, 2], ix['lo'][2], ix['inv_cell'], G)
    sid = ix['sorted_id']
    assert np.array_equal(np.sort(sid), np.arange(n))
    scid = cid[sid]
    assert np.all(np.diff(scid) >= 0)                                  # sorted by cell ...
    same = np.diff(scid) == 0
    assert np.all(np.diff(sid)[same] > 0)                              # ... original order inside a cell
    for c in np.unique(scid)[:50]:
        assert np.all(scid[cs[c]:cs[c + 1]] == c)
    sat = ix['sat'].reshape(G + 1, G + 1, G + 1)
    assert sat[-1, -1, -1] == n and not sat[0].any() and not sat[:, 0].any() and not sat[:, :, 0].any()
    cx, cy, cz = cid // (G * G), (cid // G) % G, cid % G
    rs = np.random.RandomState(0)
    for _ in range(40):
        lo = rs.randint(0, G, 3)
        hi = np.minimum(lo + rs.randint(0, 6, 3), G - 1)
        brute = int(((cx >= lo[0]) & (cx <= hi[0]) & (cy >= lo[1]) & (cy <= hi[1]) & (cz >= lo[2]) & (cz <= hi[2])).sum())
        x0, y0, z0, x1, y1, z1 = lo[0], lo[1], lo[2], hi[0] + 1, hi[1] + 1, hi[2] + 1
        box = sat[x1, y1, z1] - sat[x0, y1, z1] - sat[x1, y0, z1] - sat[x1, y1, z0] + sat[x0, y0, z1] + sat[x0, y1, z0] + \
            sat[x1, y0, z0] - sat[x0, y0, z0]
        assert box == brute
