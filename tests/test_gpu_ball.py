"""a4, fixed-radius branch (p2s_{small,medium,large}_radius): the device patch selection (p2s_ball_count / p2s_ball_patch,
points2surf_amd/csrc/p2s_ball.hip) against the calls the reference makes -- scipy's query_ball_point and numpy's legacy
``RandomState.choice(np.arange(count), k, replace=False)`` (reference source/base/point_cloud.py:177-191,
source/data_loader.py:335-350) -- ids, patch-space points and the generator's position, bit for bit; then whole shapes
against goldens written by the unmodified reference."""
import json
import os

import numpy as np
import pytest

from conftest import missing_golden
from scipy import spatial

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ABC = os.path.join(REPO, 'tests', 'golden', 'abc_minimal', '04_pts')


def _reference_patches(seed, pts, tree, queries, radius, k, with_rotation=False):
    """source/base/point_cloud.py:177-191 + source/data_loader.py:340-350,384 with numpy's own generator"""
    from oracle import trimesh_restated as trafo
    rs = np.random.RandomState(seed)
    ids = np.zeros((len(queries), k), np.int32)
    patch = np.zeros((len(queries), k, 3), np.float32)
    rots = np.zeros((len(queries), 3, 3))
    counts = np.zeros(len(queries), np.int32)
    for i, q in enumerate(queries):
        pid = np.array(tree.query_ball_point(x=q, r=radius), dtype=np.int32)
        counts[i] = n = pid.shape[0]
        if n > k:
            pid = pid[rs.choice(np.arange(n), k, replace=False)]
        if n < k:
            pid = np.concatenate((pid, np.full(k - n, -1, np.int32))) if n else np.full(k, -1, np.int32)
        pad = pid == -1
        pid[pad] = 0
        ms = pts[pid, :]
        ms[pad, :] = q
        patch[i] = (ms - np.repeat(np.expand_dims(q, 0), k, axis=0)) / radius
        ids[i] = pid
        if with_rotation:
            rots[i] = trafo.random_rotation_matrix(rs.rand(3))[:3, :3]
    return ids, patch, counts, rots, rs


def _queries(pts, n, seed, spread=0.02):
    r = np.random.default_rng(seed)
    q = (pts[r.integers(0, len(pts), n)] + r.normal(0, spread, (n, 3))).astype(np.float32)
    q[0] = pts[5]                                   # on a cloud point
    q[1] = np.float32([3.0, 3.0, 3.0])              # empty ball: all padding
    return q


def _same_generator(rng_dev, rs):
    """both continue with the same 32-bit words"""
    mt, pos = rng_dev.get_state()
    probe = np.random.RandomState(0)
    probe.set_state(('MT19937', mt, pos, 0, 0.0))
    return np.array_equal(probe.randint(0, 2 ** 32, 8, dtype=np.uint32), rs.randint(0, 2 ** 32, 8, dtype=np.uint32))


@pytest.mark.parametrize('cloud_i', [0, 2])
@pytest.mark.parametrize('radius', [0.05, 0.1, 0.2])
def test_ball_count_is_scipys(cloud_i, radius):
    import torch
    from points2surf_amd import engine
    pts = np.load(os.path.join(ABC, sorted(os.listdir(ABC))[cloud_i])).astype(np.float32)
    tree = spatial.cKDTree(pts, 1000)
    cloud = engine.Cloud(pts)
    q = cloud.query_grid(24, 3).cpu().numpy()
    got = engine.ball_count(cloud, torch.from_numpy(q).cuda(), radius).cpu().numpy()
    want = tree.query_ball_point(q, radius, return_length=True)
    assert np.array_equal(got, want)
    assert want.max() > 300 or radius < 0.1
    cloud.close()


@pytest.mark.parametrize('radius,k,nq', [(0.05, 300, 96), (0.1, 300, 96), (0.2, 300, 40), (0.1, 75, 64), (0.3, 1200, 24),
                                         (2.0, 300, 6),        # the whole cloud in every ball: lists of 34,693, shuffled in place
                                         (1e-4, 300, 8)])      # (almost) empty balls: padding only
def test_ball_patch_matches_scipy_and_numpy_choice(fixture_cloud, radius, k, nq):
    import torch
    from points2surf_amd import engine
    tree = spatial.cKDTree(fixture_cloud, 1000)
    cloud = engine.Cloud(fixture_cloud)
    q = _queries(fixture_cloud, nq, 7)
    ids_r, patch_r, counts, _, rs = _reference_patches(99, fixture_cloud, tree, q, radius, k)
    assert (counts > k).any() or radius < 0.1
    assert radius != 2.0 or counts.max() == fixture_cloud.shape[0]
    rng = engine.Rng(99)
    # ragged calls: the stream continues across them
    parts = [engine.ball_patch(cloud, rng, torch.from_numpy(q[a:b]).cuda(), radius, k) for a, b in ((0, 3), (3, 40), (40, nq)) if b > a]
    rng.check()
    ids = np.concatenate([p[0].cpu().numpy() for p in parts])
    patch = np.concatenate([p[1].cpu().numpy() for p in parts])
    rad = np.concatenate([p[2].cpu().numpy() for p in parts])
    assert np.array_equal(ids, ids_r)
    assert np.array_equal(patch.view(np.uint32), patch_r.view(np.uint32))
    assert np.all(rad == 1.0)          # fixed-radius models do not rescale the distance (points_to_surf_eval.py:180,188)
    assert _same_generator(rng, rs)
    cloud.close()


def test_ball_patch_with_rotation_interleaves_the_rand3(fixture_cloud):
    """GT-query pass: per query the patch choice, THEN rand(3) from the same generator (source/data_loader.py:336,384)"""
    import torch
    from points2surf_amd import engine
    tree = spatial.cKDTree(fixture_cloud, 1000)
    cloud = engine.Cloud(fixture_cloud)
    q = _queries(fixture_cloud, 80, 11)
    ids_r, patch_r, counts, rots_r, rs = _reference_patches(5, fixture_cloud, tree, q, 0.1, 300, with_rotation=True)
    rng = engine.Rng(5)
    ids, patch, _, rot = engine.ball_patch(cloud, rng, torch.from_numpy(q).cuda(), 0.1, 300, with_rotation=True)
    rng.check()
    assert np.array_equal(ids.cpu().numpy(), ids_r)
    assert np.abs(rot.cpu().numpy() - rots_r).max() < 1e-14
    assert _same_generator(rng, rs)
    # advancing only (what a rank does for a shape it does not own) ends at the same position
    rng2 = engine.Rng(5)
    engine.ball_skip(cloud, rng2, torch.from_numpy(q).cuda(), 0.1, 300, with_rotation=True)
    rng2.check()
    assert np.array_equal(rng2.get_state()[0], rng.get_state()[0]) and rng2.get_state()[1] == rng.get_state()[1]
    cloud.close()


def test_ball_patch_long_lists_and_many_queries():
    """the 86,648-point cloud at r = 0.3: a third of the hit lists beyond the LDS capacity (global-memory shuffle), several thousand
    queries in one call (word staging of the chain wave wraps its ring many times)"""
    import torch
    from points2surf_amd import engine
    pts = np.load(os.path.join(ABC, sorted(os.listdir(ABC))[1])).astype(np.float32)
    assert pts.shape[0] == 86648
    tree = spatial.cKDTree(pts, 1000)
    cloud = engine.Cloud(pts)
    q = _queries(pts, 1500, 3, spread=0.01)
    ids_r, patch_r, counts, _, rs = _reference_patches(1, pts, tree, q, 0.3, 300)
    assert counts.max() > 8192 and (counts <= 8192).any()
    rng = engine.Rng(1)
    ids, patch, _, _ = engine.ball_patch(cloud, rng, torch.from_numpy(q).cuda(), 0.3, 300)
    rng.check()
    assert np.array_equal(ids.cpu().numpy(), ids_r)
    assert np.array_equal(patch.cpu().numpy().view(np.uint32), patch_r.view(np.uint32))
    assert _same_generator(rng, rs)
    cloud.close()


@pytest.mark.parametrize('model', ['p2s_small_radius', 'p2s_medium_radius', 'p2s_large_radius'])
def test_radius_models_match_the_reference_golden(fixture_cloud, golden_dir, model):
    """whole shape through p2s_infer_shape_ball vs the SDF the unmodified reference wrote (oracle/make_golden_sizes.py rec
    <model> testset 32): 1e-4 contract, no sign flips; the dataset's two generators end where numpy's do"""
    from points2surf_amd import engine, synth
    path = os.path.join(golden_dir, 'ref_rec_%s_testset_grid32.npz' % model)
    if not os.path.isfile(path):
        missing_golden('golden not generated yet: ' + os.path.basename(path))
    ref = np.load(path)['rec_0']
    w, cfg = synth.make_weights(model)
    m = engine.Model(w, cfg)
    cloud = engine.Cloud(fixture_cloud)
    with open(os.path.join(golden_dir, 'meta.json')) as f:
        seed = json.load(f)['seed_data']
    rng, rng_patch = engine.Rng(seed), engine.Rng(seed)
    sdf, q = engine.infer_shape(m, cloud, rng, 32, 3, rng_patch=rng_patch)
    sdf = sdf.cpu().numpy()
    assert sdf.shape == ref.shape
    assert np.abs(sdf - ref).max() < 1e-4, np.abs(sdf - ref).max()
    assert int(((sdf > 0) != (ref > 0)).sum()) == 0
    # p2s_infer_shape refuses the model: the patch generator is not optional
    with pytest.raises(Exception):
        engine.infer_shape(m, cloud, rng, 32, 3)
    m.close()
    cloud.close()


@pytest.mark.parametrize('model,chunk', [('p2s_small_radius', 0), ('p2s_medium_radius', 777), ('p2s_large_radius', 0)])
def test_radius_models_match_the_reference_at_128(fixture_cloud, golden_dir, model, chunk):
    """the fixed-radius models at a quoted size -- every one of the 68,088 queries of the 128^3 grid
    against the SDF the unmodified reference wrote (oracle/make_golden_sizes.py rec <model> testset 128; ~25 min of
    reference CPU each): the ball scan, the serial stream chain (ring wrap, 256-word blocks across thousands of queries)
    and the legacy shuffle at shape scale against the reference itself, not against numpy alone.  One model runs with
    chunks of 777 queries."""
    from points2surf_amd import engine, parity, synth
    path = os.path.join(golden_dir, 'ref_rec_%s_testset_grid128.npz' % model)
    if not os.path.isfile(path):
        missing_golden('golden not generated yet: ' + os.path.basename(path))
    ref = np.load(path)['rec_0']
    w, cfg = synth.make_weights(model)
    m = engine.Model(w, cfg)
    cloud = engine.Cloud(fixture_cloud)
    with open(os.path.join(golden_dir, 'meta.json')) as f:
        seed = json.load(f)['seed_data']
    rng, rng_patch = engine.Rng(seed), engine.Rng(seed)
    sdf, q = engine.infer_shape(m, cloud, rng, 128, 3, rng_patch=rng_patch, chunk=chunk)
    sdf = sdf.cpu().numpy()
    assert sdf.shape == ref.shape == (68088,)
    c = parity.compare_sdf(sdf, ref)
    print('%s 128^3 (chunk %d): max|dSDF| %.3g, sign flips %d / %d, positive in the reference %.1f %%'
          % (model, chunk, c['max_abs_dsdf'], c['flipped'].size, ref.size, 100.0 * (ref > 0).mean()))
    assert c['max_abs_dsdf'] < 1e-4 and c['flipped'].size == 0
    rng.check()
    rng_patch.check()
    m.close()
    cloud.close()


def test_radius_model_is_invariant_to_chunking_and_query_ranges(fixture_cloud):
    """size-independent property: where a query's random words start depends on the queries before it, never on how the
    pipeline batches them -- default chunks vs chunks of 777 queries vs two query ranges with the generators advanced
    past the first (what query-range sharding does) give the same SDF, bit for bit, at 64^3 (12k queries)"""
    import torch
    from points2surf_amd import engine, sharding, synth
    w, cfg = synth.make_weights('p2s_large_radius')
    m = engine.Model(w, cfg)
    cloud = engine.Cloud(fixture_cloud)

    def run(**kw):
        r1, r2 = engine.Rng(11), engine.Rng(11)
        sdf, _ = engine.infer_shape(m, cloud, r1, 64, 3, rng_patch=r2, **kw)
        return sdf.cpu().numpy(), r1.get_state(), r2.get_state()

    a, s1, s2 = run()
    b, t1, t2 = run(chunk=777)
    assert a.shape[0] > 10000 and np.array_equal(a, b)
    assert np.array_equal(s1[0], t1[0]) and s1[1] == t1[1] and np.array_equal(s2[0], t2[0]) and s2[1] == t2[1]
    # second half only: both generators skipped past the first half
    q = cloud.query_grid(64, 3)
    half = int(q.shape[0]) // 2 + 13
    r1, r2 = engine.Rng(11), engine.Rng(11)
    sharding.skip_queries(cloud, r1, cfg, q[:half], 1000, rng_patch=r2)
    c, _ = engine.infer_shape(m, cloud, r1, 64, 3, q_begin=half, rng_patch=r2)
    assert np.array_equal(c.cpu().numpy(), a[half:])
    m.close()
    cloud.close()


@pytest.mark.parametrize('n,seed', [(5, 0), (63, 1), (64, 2), (65, 3), (999, 4), (1001, 5), (2500, 6), (4097, 7)])
def test_ball_patch_on_odd_clouds(n, seed):
    """clouds around the wave and leaf sizes, with duplicated points and a rounded coordinate (ties in the tree build),
    k larger and smaller than the balls, queries on points and far away"""
    import torch
    from points2surf_amd import engine
    r = np.random.RandomState(seed)
    pts = (r.rand(n, 3).astype(np.float32) - 0.5)
    pts[::7] = pts[0]
    pts[:, 1] = np.round(pts[:, 1], 2)
    tree = spatial.cKDTree(pts, 1000)
    cloud = engine.Cloud(pts)
    q = np.concatenate([pts[r.randint(0, n, 20)], (r.rand(12, 3) - 0.5).astype(np.float32), np.float32([[4, 4, 4]])]).astype(np.float32)
    for radius, k in ((0.3, 16), (0.6, 300), (2.0, 50)):
        ids_r, patch_r, counts, _, rs = _reference_patches(seed + 100, pts, tree, q, radius, k)
        rng = engine.Rng(seed + 100)
        got = engine.ball_count(cloud, torch.from_numpy(q).cuda(), radius).cpu().numpy()
        assert np.array_equal(got, counts)
        ids, patch, _, _ = engine.ball_patch(cloud, rng, torch.from_numpy(q).cuda(), radius, k)
        rng.check()
        assert np.array_equal(ids.cpu().numpy(), ids_r), (n, radius, k)
        assert np.array_equal(patch.cpu().numpy().view(np.uint32), patch_r.view(np.uint32))
        assert _same_generator(rng, rs)
    cloud.close()


def test_ball_patch_on_a_million_point_scan():
    """a raw scan of 1,000,000 points (the reference's tooling stops at 150,000, make_pc_dataset.py:39; nothing in the
    fixed-radius path of data_loader.py does): balls of 20,000-80,000 hits, every list through the global-memory shuffle.
    Counts, ids, patches and generator position == scipy's tree order + numpy's legacy ``choice``"""
    import torch
    from points2surf_amd import engine, synth
    pts = synth.make_cloud(1000000, seed=31)
    tree = spatial.cKDTree(pts, 1000)
    cloud = engine.Cloud(pts)
    q = _queries(pts, 40, 9, spread=0.01)
    for radius in (0.1, 0.2):
        ids_r, patch_r, counts, _, rs = _reference_patches(6, pts, tree, q, radius, 300)
        assert counts.max() > 8192
        got = engine.ball_count(cloud, torch.from_numpy(q).cuda(), radius).cpu().numpy()
        assert np.array_equal(got, counts)
        rng = engine.Rng(6)
        ids, patch, _, _ = engine.ball_patch(cloud, rng, torch.from_numpy(q).cuda(), radius, 300)
        rng.check()
        assert np.array_equal(ids.cpu().numpy(), ids_r), radius
        assert np.array_equal(patch.cpu().numpy().view(np.uint32), patch_r.view(np.uint32))
        assert _same_generator(rng, rs)
    cloud.close()
