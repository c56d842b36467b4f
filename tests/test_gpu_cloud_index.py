"""GPU: the neighbour index (reference source/data_loader.py:40-42 builds a cKDTree here) is built on
the device -- bbox, cell histogram, summed-area table, stable counting sort -- and is bit-identical to the CPU
restatement oracle/cloud_index_oracle.py; the handle's memory comes from the per-device block cache."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
ABC3 = ['00011084_fddd53ce45f640f3ab922328_trimesh_019', '00016513_3d6966cd42eb44ab8f4224f2_trimesh_053',
        '00994122_57d9d4755722f9d2d7436f0a_trimesh_000']
SEED = 40938661


def _clouds():
    from points2surf_amd import synth
    out = [(n[:8], np.load(os.path.join(GOLDEN, 'abc_minimal', '04_pts', n + '.xyz.npy'))[:, :3].astype(np.float32)) for n in ABC3]
    out.append(('synthetic150k', synth.make_cloud(150000, seed=5)))
    rs = np.random.RandomState(3)
    clustered = np.concatenate([0.02 * rs.standard_normal((20000, 3)), np.repeat(rs.uniform(-0.4, 0.4, (50, 3)), 200, 0),
                                rs.uniform(-0.5, 0.5, (5000, 3))]).astype(np.float32)
    out.append(('clustered+duplicates', clustered))
    out.append(('all equal', np.full((3000, 3), 0.25, dtype=np.float32)))
    out.append(('tiny', rs.uniform(-0.5, 0.5, (5, 3)).astype(np.float32)))
    out.append(('600k (G clamps at 128)', rs.uniform(-0.5, 0.5, (600000, 3)).astype(np.float32)))
    return out


@pytest.mark.parametrize('case', range(8))
def test_device_index_equals_cpu_restatement(case):
    from points2surf_amd import engine
    from oracle import cloud_index_oracle as CO
    name, pts = _clouds()[case]
    cloud = engine.Cloud(pts)
    dev = cloud.index_export()
    ref = CO.build(pts)
    assert dev['G'] == ref['G'], name
    assert np.array_equal(dev['lo'], ref['lo']) and dev['inv_cell'] == ref['inv_cell'], name
    assert np.array_equal(dev['cell_start'], ref['cell_start']), name
    assert np.array_equal(dev['sat'], ref['sat']), name
    assert np.array_equal(dev['sorted_id'], ref['sorted_id']), name          # stable: original order inside a cell
    assert np.array_equal(dev['sorted_xyz'], ref['sorted_xyz']), name
    cloud.close()


def test_non_finite_point_is_rejected_with_its_index(fixture_cloud):
    from points2surf_amd import engine, _lib
    pts = np.array(fixture_cloud, dtype=np.float32)
    pts[1234, 1] = np.nan
    pts[20000, 0] = np.inf
    with pytest.raises(_lib.P2SError) as e:
        engine.Cloud(pts)
    assert e.value.code == -1 and 'point 1234' in str(e.value)
    engine.Cloud(fixture_cloud).close()                      # the failed create left the device usable


def test_handle_per_shape_allocates_nothing_once_warm(fixture_cloud):
    import torch
    from points2surf_amd import engine
    sizes = [34693, 30000, 34000, 28000]
    for n in sizes:                                          # warm the block cache
        c = engine.Cloud(fixture_cloud[:n])
        c.query_grid(64, 3)
        c.close()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for rep in range(20):
        c = engine.Cloud(fixture_cloud[:sizes[rep % 4]])
        q = c.query_grid(64, 3)
        assert q.shape[0] > 0
        c.close()
    torch.cuda.synchronize()
    assert abs(torch.cuda.mem_get_info()[0] - free0) <= (2 << 20)     # every block came from / went back to the cache
    engine.release_scratch()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] > free0 + (2 << 20)  # ... and the cache can be handed back to HIP


def test_small_cloud_second_larger_grid_on_the_same_handle():
    """VERDICT r2 weak #4 / ADVICE r2: a cloud with fewer points than the sub-sample keeps the permutation of shape.pts
    on its handle; asking the handle for a second, LARGER query grid used to free that buffer (use after free, then a
    double free in destroy)."""
    import torch
    from points2surf_amd import engine, synth
    pts = np.load(os.path.join(GOLDEN, 'small800.xyz.npy'))
    w, cfg = synth.make_weights('p2s_max')
    model = engine.Model(w, cfg)

    def run(grow):
        cloud = engine.Cloud(pts)
        rng = engine.Rng(SEED)
        a, _ = engine.infer_shape(model, cloud, rng, 16, 3)
        if grow:
            assert cloud.query_grid(64, 3).shape[0] > a.shape[0]      # re-allocates the grid cache of the handle
        b, _ = engine.infer_shape(model, cloud, rng, 16, 3)           # shuffles the permutation the first pass left
        torch.cuda.synchronize()
        out = (a.cpu().numpy(), b.cpu().numpy())
        cloud.close()
        return out
    a0, b0 = run(False)
    a1, b1 = run(True)
    assert np.array_equal(a0, a1) and np.array_equal(b0, b1)
    assert not np.array_equal(a0, b0)                                 # the second pass sees the permuted array
    ref = np.load(os.path.join(GOLDEN, 'ref_rec_p2s_max_small800_grid16.npz'))['rec_0']
    assert np.abs(a1 - ref).max() < 1e-4
