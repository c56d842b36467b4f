"""Stress / randomized checks of the data-path kernels against the oracle (GPU)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_knn_large_clustered_cloud_with_duplicates():
    """N = 150k (the reference's maximum, make_pc_dataset.py:39): uniform part + tight clusters + exact duplicates.
    Distance ties make the id choice implementation-defined (also in cKDTree), so distances are compared."""
    import torch
    from oracle import p2s_oracle as O
    from points2surf_amd import engine
    rng = np.random.default_rng(42)
    parts = [rng.uniform(-0.5, 0.5, (100000, 3)),
             0.002 * rng.standard_normal((30000, 3)) + rng.uniform(-0.4, 0.4, (30, 1, 3)).repeat(1000, 1).reshape(-1, 3),
             np.repeat(rng.uniform(-0.3, 0.3, (40, 3)), 500, axis=0)]          # 40 locations x 500 identical points
    pts = np.concatenate(parts).astype(np.float32)
    assert pts.shape[0] == 150000
    cloud = engine.Cloud(pts)
    q = np.concatenate([pts[rng.integers(0, pts.shape[0], 150)] + rng.normal(0, 0.003, (150, 3)).astype(np.float32),
                        parts[2][::500][:20].astype(np.float32),                # queries exactly on duplicate stacks
                        rng.uniform(-0.9, 0.9, (30, 3)).astype(np.float32)]).astype(np.float32)
    ids, _, rad = cloud.knn_patch(torch.from_numpy(q).cuda(), 300)
    ids = ids.cpu().numpy()
    ref = O.knn_ids(pts, q, 300)
    p64, q64 = pts.astype(np.float64), q.astype(np.float64)
    d_dev = ((p64[ids] - q64[:, None, :]) ** 2).sum(-1)
    d_ref = ((p64[ref] - q64[:, None, :]) ** 2).sum(-1)
    assert np.array_equal(np.sort(d_dev, axis=1), np.sort(d_ref, axis=1))       # the same 300 distances
    assert (np.diff(d_dev, axis=1) >= 0).all()                                  # ascending, like cKDTree
    for row in ids:
        assert len(set(row.tolist())) == 300                                    # no point twice
    no_tie = np.array([len(np.unique(d)) == 300 for d in d_ref])
    assert np.array_equal(ids[no_tie], ref[no_tie])                             # without ties: identical ids
    r_ref, _ = O.patch_radius_and_ps(pts, ids, q)
    assert np.array_equal(rad.cpu().numpy(), r_ref)


@pytest.mark.parametrize('k', [1, 64, 300, 1200])
def test_knn_selection_kernel_returns_the_same_set_as_the_sorting_kernel(k):
    """the pipeline's kNN selects the k smallest distances by bisection instead of sorting: same points (rows of the patch
    as a set), bit-identical radius -- on the fixture, on a clustered cloud with duplicate stacks (ties at the k-th
    distance fall back to the sort) and with queries sitting exactly on points"""
    import torch
    from points2surf_amd import engine
    rng = np.random.default_rng(7)
    fixture = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cloud_abc_00994122.npy')).astype(np.float32)
    clustered = np.concatenate([rng.uniform(-0.5, 0.5, (40000, 3)),
                                0.002 * rng.standard_normal((20000, 3)) + rng.uniform(-0.4, 0.4, (20, 1, 3)).repeat(1000, 1).reshape(-1, 3),
                                np.repeat(rng.uniform(-0.3, 0.3, (30, 3)), 400, axis=0)]).astype(np.float32)
    for pts in (fixture, clustered):
        cloud = engine.Cloud(pts)
        q = np.concatenate([pts[rng.integers(0, pts.shape[0], 300)] + rng.normal(0, 0.004, (300, 3)),
                            pts[rng.integers(0, pts.shape[0], 40)],                  # distance 0 to a point / a stack
                            rng.uniform(-0.9, 0.9, (60, 3))]).astype(np.float32)
        qd = torch.from_numpy(q).cuda()
        _, patch_sorted, rad_sorted = cloud.knn_patch(qd, k)
        patch_set, rad_set = cloud.knn_patch_set(qd, k)
        torch.cuda.synchronize()
        a, b = patch_sorted.cpu().numpy(), patch_set.cpu().numpy()
        assert np.array_equal(rad_sorted.cpu().numpy(), rad_set.cpu().numpy())
        for i in range(q.shape[0]):
            ra = a[i][np.lexsort(a[i].T[::-1])]
            rb = b[i][np.lexsort(b[i].T[::-1])]
            assert np.array_equal(ra, rb, equal_nan=True), (i, k)
        cloud.close()


def test_query_grid_random_clouds_match_oracle():
    import torch
    from oracle import p2s_oracle as O
    from points2surf_amd import engine
    rng = np.random.default_rng(7)
    for trial in range(12):
        n = int(rng.integers(1, 4000))
        res = int(rng.choice([8, 17, 32, 50, 64, 96]))
        eps = int(rng.integers(1, 8))
        scale = rng.choice([0.1, 0.5, 0.99])
        pts = (rng.uniform(-1, 1, (n, 3)) * scale).astype(np.float32)
        if trial % 3 == 0:
            pts[: max(1, n // 10)] = np.float32(0.999)               # points in the last voxel slab (dropped by [:-1])
            pts[-1] = np.float32(-1.0)                               # exactly on the lower border
        q_ref, _ = O.query_grid(pts, res, eps)
        q = engine.Cloud(pts).query_grid(res, eps).cpu().numpy()
        assert q.shape == q_ref.shape, (trial, n, res, eps)
        assert np.array_equal(q, q_ref), (trial, n, res, eps)


def test_forward_extreme_but_finite_inputs(fixture_cloud):
    """large coordinates / tiny radius do not break parity with the oracle"""
    import torch
    from oracle import p2s_oracle as O
    from points2surf_amd import engine, synth
    w, cfg = synth.make_weights('p2s_max')
    m = engine.Model(w, cfg)
    rng = np.random.default_rng(3)
    B = 5
    patch = rng.uniform(-1, 1, (B, 300, 3)).astype(np.float32)
    patch[0] *= 1e-6                                                  # degenerate tiny patch
    patch[1, :, 0] = 1.0                                              # all points on a plane
    sub = rng.uniform(-50, 50, (B, 1000, 3)).astype(np.float32)      # far outside the unit cube
    q = rng.uniform(-1, 1, (B, 3)).astype(np.float32)
    ref = O.model_forward(w, cfg, patch, sub, q)
    t = lambda a: torch.from_numpy(a).cuda()
    logits, _ = m.forward(t(patch), t(sub), t(q))
    err = np.abs(logits.cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())
    assert err < 1e-5, err


@pytest.mark.parametrize('name,n_pts', [('p2s_max', 2000000), ('p2s_vanilla', 185000), ('p2s_vanilla', 470000)])
def test_scans_beyond_the_reference_cap(name, n_pts):
    """raw scans larger than what make_pc_dataset.py:39 lets through (150,000 points) -- nothing in points_to_surf_eval /
    data_loader.py limits the cloud: 2,000,000 points for the uniform sub-sample; 185,000 and 470,000 for the
    distance-weighted one (its per-query found-bitmap lives in the ids kernel's LDS: up to 475,040 points -- see the next
    tests).  kNN ids, radius, the
    sub-sample (``randint`` draws / ``choice(p, replace=False)``), generator position and SDF of the GT-query pass
    against the oracle"""
    import torch
    from oracle import p2s_oracle as O
    from points2surf_amd import engine, synth
    w, cfg = synth.make_weights(name)
    model = engine.Model(w, cfg)
    pts = synth.make_cloud(n_pts, seed=9)
    rng = np.random.default_rng(3)
    nq = 48
    q = (pts[rng.integers(0, pts.shape[0], nq)] + rng.normal(0, 0.004, (nq, 3))).astype(np.float32)
    cloud = engine.Cloud(pts)
    ids, _, rad = cloud.knn_patch(torch.from_numpy(q).cuda(), 300)
    ref_ids = O.knn_ids(pts, q, 300)
    r_ref, _ = O.patch_radius_and_ps(pts, ref_ids, q)
    assert np.array_equal(ids.cpu().numpy(), ref_ids) and np.array_equal(rad.cpu().numpy(), r_ref)
    r_dev, r_cpu = engine.Rng(77), O.LegacyMT19937(77)
    sdf = engine.infer_queries(model, cloud, r_dev, None, torch.from_numpy(q).cuda()).cpu().numpy()
    ref = O.infer_queries(w, cfg, pts, q, r_cpu, None)
    nxt = r_dev.subsample_uniform(cloud, 1, 7, want_pts=False)[0].cpu().numpy().reshape(-1)
    assert np.array_equal(nxt, r_cpu.randint(pts.shape[0], 7))             # the same number of draws consumed
    assert np.abs(sdf - ref).max() < 1e-5 and np.array_equal(np.sign(sdf), np.sign(ref))
    cloud.close()
    model.close()


@pytest.mark.parametrize('serial', [False, True])
def test_weighted_subsample_of_a_300k_cloud_also_in_order(serial, monkeypatch):
    """a 300,000-point cloud through the speculative offsets pass and, with ``P2S_WC_SERIAL=1``, through the chain kernel's
    in-order remainder path alone (the whole algorithm per query).  ids and stream position == numpy's
    ``choice(N, 1000, replace=False, p)`` as the oracle restates it"""
    import torch
    from oracle import p2s_oracle as O
    from points2surf_amd import engine, synth
    if serial:
        monkeypatch.setenv('P2S_WC_SERIAL', '1')
    pts = synth.make_cloud(300000, seed=12)
    rng = np.random.default_rng(4)
    nq = 70                                                            # >= 64: the chain claims its own CU
    q = (pts[rng.integers(0, pts.shape[0], nq)] + rng.normal(0, 0.01, (nq, 3))).astype(np.float32)
    cloud = engine.Cloud(pts)
    r_dev, r_cpu = engine.Rng(21), O.LegacyMT19937(21)
    ids = r_dev.subsample_weighted(cloud, torch.from_numpy(q).cuda(), 1000, want_pts=False)[0].cpu().numpy()
    ref = np.stack([O.subsample_ids(r_cpu, pts, q[i], 1000, uniform=False) for i in range(nq)])
    assert np.array_equal(ids.reshape(nq, 1000), ref)
    nxt = r_dev.subsample_uniform(cloud, 1, 7, want_pts=False)[0].cpu().numpy().reshape(-1)
    assert np.array_equal(nxt, r_cpu.randint(pts.shape[0], 7))
    cloud.close()


@pytest.mark.parametrize('n_pts,what', [(476000, 'LDS bitmap'), (2000000, 'summation nodes')])
def test_weighted_subsample_refuses_clouds_beyond_its_limit_loudly(n_pts, what):
    """the distance-weighted sub-sample (p2s_vanilla and the ablations with uniform_subsample = 0) keeps one bit per
    cloud point in LDS: clouds of more than 475,040 points (the reference's tooling stops at 150,000) are refused with
    an error that names the limit -- before a single random word is consumed -- and never answered approximately"""
    import torch
    from points2surf_amd import engine, synth, _lib
    pts = synth.make_cloud(n_pts, seed=2)
    cloud = engine.Cloud(pts)
    r = engine.Rng(5)
    q = torch.from_numpy(pts[:8].copy()).cuda()
    with pytest.raises(_lib.P2SError, match=what):
        r.subsample_weighted(cloud, q, 1000)
    got = r.subsample_uniform(cloud, 1, 9, want_pts=False)[0].cpu().numpy().reshape(-1)
    assert np.array_equal(got, np.random.RandomState(5).randint(0, n_pts, 9))      # the stream is where it was
    cloud.close()
