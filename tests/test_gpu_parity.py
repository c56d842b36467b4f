"""GPU parity tests proper: the HIP path (through the C ABI) against the reference's golden vectors
and against the numpy oracle on seeded inputs.  Run with ``pytest -m gpu`` on an MI355X."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SDF_TOL = 1e-4       # contractual bound (BASELINE.json north_star): |dSDF| <= 1e-4 fp32
SDF_TOL_TIGHT = 1e-5  # what fp32 re-association noise actually allows at this weight scale
LOGIT_TOL = 1e-4


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope='module')
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return torch


@pytest.fixture(scope='module')
def meta(golden_dir):
    with open(os.path.join(golden_dir, 'meta.json')) as f:
        return json.load(f)


@pytest.fixture(scope='module')
def cloud_dev(torch_cuda, fixture_cloud):
    from points2surf_amd import engine
    return engine.Cloud(fixture_cloud)


def _model(name):
    from points2surf_amd import engine, synth
    w, cfg = synth.make_weights(name)
    return engine.Model(w, cfg), w, cfg


@pytest.fixture(scope='module')
def model_max(torch_cuda):
    return _model('p2s_max')


@pytest.fixture(scope='module')
def model_vanilla(torch_cuda):
    return _model('p2s_vanilla')


def test_native_library_is_loaded(torch_cuda):
    from points2surf_amd import _lib
    lib = _lib.load()
    assert lib.p2s_device_count() >= 1
    with open('/proc/self/maps') as f:
        assert 'libp2s_hip.so' in f.read()


# ---------------------------------------------------------------- a1 query grid
@pytest.mark.parametrize('res,eps', [(32, 3), (64, 3), (128, 3), (256, 3), (64, 4), (32, 5)])
def test_query_grid_bit_exact(cloud_dev, meta, res, eps):
    q = cloud_dev.query_grid(res, eps).cpu().numpy()
    g = meta['query_grids']['%d_%d' % (res, eps)]
    assert q.shape[0] == g['count']
    assert _sha(q) == g['sha256']


def test_query_grid_rejects_points_outside_volume(torch_cuda):
    from points2surf_amd import engine, _lib
    pts = np.array([[0.0, 0.0, 0.0], [1.5, 0.0, 0.0]] * 200, dtype=np.float32)
    c = engine.Cloud(pts)
    with pytest.raises(_lib.P2SError):
        c.query_grid(32, 3)


# ---------------------------------------------------------------- a4/a5 kNN patch
def test_knn_matches_reference_golden(cloud_dev, golden_dir, torch_cuda, meta):
    g = np.load(os.path.join(golden_dir, 'ref_p2s_max_grid32.npz'))
    q = cloud_dev.query_grid(32, 3)[:meta['nq']]
    ids, patch, rad = cloud_dev.knn_patch(q, 300)
    assert np.array_equal(ids.cpu().numpy(), g['knn_ids'])
    assert np.array_equal(rad.cpu().numpy(), g['radius'])              # bit-exact fp32
    assert np.array_equal(patch.cpu().numpy()[:4], g['patch_ps_head'])


def test_knn_matches_oracle_random_queries(cloud_dev, fixture_cloud, torch_cuda):
    from oracle import p2s_oracle as O
    rng = np.random.default_rng(5)
    # queries near the surface, far outside the cloud, and exactly on points
    q = np.concatenate([
        fixture_cloud[rng.integers(0, fixture_cloud.shape[0], 300)] + rng.normal(0, 0.02, (300, 3)).astype(np.float32),
        rng.uniform(-1, 1, (100, 3)).astype(np.float32),
        fixture_cloud[rng.integers(0, fixture_cloud.shape[0], 50)],
    ]).astype(np.float32)
    for k in (300, 1, 64, 1200):
        ids, patch, rad = cloud_dev.knn_patch(torch_cuda.from_numpy(q).cuda(), k)
        ref = O.knn_ids(fixture_cloud, q, k)
        assert np.array_equal(ids.cpu().numpy(), ref), 'k=%d' % k
        r_ref, ps_ref = O.patch_radius_and_ps(fixture_cloud, ref, q)
        assert np.array_equal(rad.cpu().numpy(), r_ref)
        # k=1 with a query exactly on a point: r = 0 -> 0/0 = NaN in numpy and on the device alike
        assert np.array_equal(patch.cpu().numpy(), ps_ref, equal_nan=True)


def test_knn_small_and_degenerate_clouds(torch_cuda):
    from oracle import p2s_oracle as O
    from points2surf_amd import engine, _lib
    rng = np.random.default_rng(9)
    pts = rng.uniform(-0.5, 0.5, (700, 3)).astype(np.float32)
    pts[:350, 2] = 0.25                                     # half of the points on one plane
    c = engine.Cloud(pts)
    q = rng.uniform(-0.6, 0.6, (64, 3)).astype(np.float32)
    ids, _, rad = c.knn_patch(torch_cuda.from_numpy(q).cuda(), 300)
    ref = O.knn_ids(pts, q, 300)
    assert np.array_equal(ids.cpu().numpy(), ref)
    with pytest.raises(_lib.P2SError):                      # N < k: the reference raises IndexError
        engine.Cloud(pts[:100]).knn_patch(torch_cuda.from_numpy(q).cuda(), 300)


# ---------------------------------------------------------------- a6 sub-sample RNG
def test_rng_matches_numpy_legacy_stream(cloud_dev, fixture_cloud, golden_dir, torch_cuda):
    from points2surf_amd import engine
    kat = np.load(os.path.join(golden_dir, 'numpy_legacy_rng_kat.npz'))
    r = engine.Rng(40938661)
    ids, pts = r.subsample_uniform(cloud_dev, 2, 1000)
    assert np.array_equal(ids.cpu().numpy().reshape(-1).astype(np.int64), kat['randint_34693'])
    assert np.array_equal(pts.cpu().numpy().reshape(-1, 3), fixture_cloud[kat['randint_34693']])
    # chunk invariance + continuation across calls with ragged sizes vs numpy itself
    r = engine.Rng(123)
    ref = np.random.RandomState(123)
    for nq, n in ((1, 1000), (7, 1000), (3, 17), (64, 1000), (1, 1)):
        got = r.subsample_uniform(cloud_dev, nq, n, want_pts=False)[0].cpu().numpy()
        assert np.array_equal(got.reshape(-1), ref.randint(0, fixture_cloud.shape[0], nq * n))
    mt, pos = r.get_state()
    st = ref.get_state()
    assert np.array_equal(mt, st[1]) and pos == st[2]
    r2 = engine.Rng(0)
    r2.set_state(mt, pos)
    assert np.array_equal(r2.subsample_uniform(cloud_dev, 5, 1000, want_pts=False)[0].cpu().numpy().reshape(-1),
                          ref.randint(0, fixture_cloud.shape[0], 5000))


def test_rng_other_cloud_sizes(torch_cuda):
    from points2surf_amd import engine
    for n_pts, seed in ((1000, 1), (65536, 2), (65537, 3), (150000, 4), (2, 5)):
        pts = np.random.default_rng(seed).uniform(-0.5, 0.5, (n_pts, 3)).astype(np.float32)
        c = engine.Cloud(pts)
        r = engine.Rng(seed)
        n = min(1000, n_pts)
        got = r.subsample_uniform(c, 9, n, want_pts=False)[0].cpu().numpy().reshape(-1)
        assert np.array_equal(got, np.random.RandomState(seed).randint(0, n_pts, 9 * n)), n_pts


# ---------------------------------------------------------------- a8/a9 network
@pytest.mark.parametrize('name', ['p2s_max', 'p2s_vanilla'])
def test_forward_matches_reference_golden(name, cloud_dev, fixture_cloud, golden_dir, torch_cuda, meta, request):
    model, w, cfg = request.getfixturevalue('model_max' if name == 'p2s_max' else 'model_vanilla')
    g = np.load(os.path.join(golden_dir, 'ref_%s_grid32.npz' % name))
    nq = meta['nq']
    q = cloud_dev.query_grid(32, 3)[:nq]
    _, patch, rad = cloud_dev.knn_patch(q, 300)
    sub = cloud_dev.gather(torch_cuda.from_numpy(g['sub_ids']).cuda())
    sub_before = sub.clone()
    fl, fg = model.features(patch, sub, q)
    logits, sdf = model.forward(patch, sub, q, rad, want_logits=True, want_sdf=True)
    assert torch_cuda.equal(sub, sub_before)                 # engine does not mutate the caller's tensor
    err_l = np.abs(fl.cpu().numpy() - g['feat_local']).max() / np.abs(g['feat_local']).max()
    err_g = np.abs(fg.cpu().numpy() - g['feat_global']).max() / np.abs(g['feat_global']).max()
    print('%s: rel feature err local %.3g global %.3g' % (name, err_l, err_g))
    assert err_l < 1e-5 and err_g < 1e-5
    dl = np.abs(logits.cpu().numpy() - g['logits']).max()
    ds = np.abs(sdf.cpu().numpy() - g['sdf_full'][:nq]).max()
    print('%s: max|dlogit| %.3g  max|dSDF| %.3g' % (name, dl, ds))
    assert dl < LOGIT_TOL
    assert ds < SDF_TOL_TIGHT < SDF_TOL
    assert np.array_equal(np.sign(sdf.cpu().numpy()), np.sign(g['sdf_full'][:nq]))


@pytest.mark.parametrize('name,B', [('p2s_max', 193), ('p2s_max', 1), ('p2s_vanilla', 70)])
def test_forward_matches_oracle_ragged_batches(name, B, fixture_cloud, torch_cuda, request):
    from oracle import p2s_oracle as O
    model, w, cfg = request.getfixturevalue('model_max' if name == 'p2s_max' else 'model_vanilla')
    rng = np.random.default_rng(B)
    q = (fixture_cloud[rng.integers(0, fixture_cloud.shape[0], B)] + rng.normal(0, 0.01, (B, 3))).astype(np.float32)
    ids = O.knn_ids(fixture_cloud, q, 300)
    r, ps = O.patch_radius_and_ps(fixture_cloud, ids, q)
    sub = fixture_cloud[rng.integers(0, fixture_cloud.shape[0], (B, 1000))]
    ref = O.model_forward(w, cfg, ps, sub, q)
    t = lambda a: torch_cuda.from_numpy(np.ascontiguousarray(a)).cuda()
    logits, sdf = model.forward(t(ps), t(sub), t(q), t(r), want_logits=True, want_sdf=True)
    assert np.abs(logits.cpu().numpy() - ref).max() < LOGIT_TOL
    assert np.abs(sdf.cpu().numpy() - O.post_process(ref, r)).max() < SDF_TOL_TIGHT


@pytest.mark.parametrize('encoder', [0, 4])
@pytest.mark.parametrize('k,n_sub', [(20, 1000), (40, 1000), (48, 1008), (49, 1009), (64, 1024), (100, 960), (112, 976), (113, 977)])
def test_max_pool_at_every_kind_of_last_tile(k, n_sub, encoder, fixture_cloud, torch_cuda):
    """r06: the fp32 chain kernel runs the last 64-point tile of an item with at most 48 points as 32 + 16 rows (the 16 on
    the 4-block MFMA, two k partial sums added before the pool; p2s_chain_conv3.inl).  Patch / sub-sample sizes on both
    sides of that switch -- 20 and 40 points (ONE tile, tail), 48 | 49 points in the last tile (tail | full), 64 (no
    padding), 100 -> 36 and 112 -> 48 (tail), 113 -> 49 (full); sub-samples ending with 40, 48, 49, 64, 0 (960 = 15 full
    tiles), 16 and 17 points -- against the numpy restatement of the reference's forward, fp32 and fp16-pair encoders."""
    from oracle import p2s_oracle as O
    from points2surf_amd import engine, synth
    w, cfg = synth.make_weights('p2s_max')
    cfg = dict(cfg, points_per_patch=k, sub_sample_size=n_sub)
    model = engine.Model(w, dict(cfg, encoder_bf16=encoder))
    B = 29
    rng = np.random.default_rng(1000 * k + n_sub)
    q = (fixture_cloud[rng.integers(0, fixture_cloud.shape[0], B)] + rng.normal(0, 0.01, (B, 3))).astype(np.float32)
    ids = O.knn_ids(fixture_cloud, q, k)
    r, ps = O.patch_radius_and_ps(fixture_cloud, ids, q)
    sub = fixture_cloud[rng.integers(0, fixture_cloud.shape[0], (B, n_sub))]
    ref = O.model_forward(w, cfg, ps, sub, q)
    t = lambda a: torch_cuda.from_numpy(np.ascontiguousarray(a)).cuda()
    logits, sdf = model.forward(t(ps), t(sub), t(q), t(r), want_logits=True, want_sdf=True)
    assert np.abs(logits.cpu().numpy() - ref).max() < LOGIT_TOL, np.abs(logits.cpu().numpy() - ref).max()
    assert np.abs(sdf.cpu().numpy() - O.post_process(ref, r)).max() < SDF_TOL_TIGHT
    model.close()


@pytest.mark.parametrize('k', [64, 75, 96, 300])
def test_sum_pool_masks_the_padded_rows_of_the_last_point_tile(k, fixture_cloud, torch_cuda):
    """sym_op='sum' (reference source/points_to_surf_model.py:213-214): the last 64-point tile of an item is padded with
    copies of its last point -- harmless for a max, wrong for a sum unless masked.  Patches of 64 (no padding), 75 (11
    valid rows: row block 1 empty), 96 (exactly one row block) and 300 points; the 1000-point sub-sample ends with 40
    valid rows.  Against the numpy restatement of the reference's forward."""
    from oracle import p2s_oracle as O
    from points2surf_amd import engine, synth
    w, cfg = synth.make_weights('p2s_max_sum')
    cfg = dict(cfg, points_per_patch=k)
    model = engine.Model(w, cfg)
    B = 37
    rng = np.random.default_rng(k)
    q = (fixture_cloud[rng.integers(0, fixture_cloud.shape[0], B)] + rng.normal(0, 0.01, (B, 3))).astype(np.float32)
    ids = O.knn_ids(fixture_cloud, q, k)
    r, ps = O.patch_radius_and_ps(fixture_cloud, ids, q)
    sub = fixture_cloud[rng.integers(0, fixture_cloud.shape[0], (B, 1000))]
    ref = O.model_forward(w, cfg, ps, sub, q)
    t = lambda a: torch_cuda.from_numpy(np.ascontiguousarray(a)).cuda()
    logits, sdf = model.forward(t(ps), t(sub), t(q), t(r), want_logits=True, want_sdf=True)
    assert np.abs(logits.cpu().numpy() - ref).max() < LOGIT_TOL, np.abs(logits.cpu().numpy() - ref).max()
    assert np.abs(sdf.cpu().numpy() - O.post_process(ref, r)).max() < SDF_TOL_TIGHT
    model.close()


def test_nan_input_maps_to_one(model_max, fixture_cloud, torch_cuda):
    model, w, cfg = model_max
    B = 2
    patch = torch_cuda.zeros((B, 300, 3), device='cuda')
    patch[0, 0, 0] = float('nan')
    sub = torch_cuda.zeros((B, 1000, 3), device='cuda')
    q = torch_cuda.zeros((B, 3), device='cuda')
    rad = torch_cuda.ones((B,), device='cuda')
    _, sdf = model.forward(patch, sub, q, rad, want_logits=False, want_sdf=True)
    assert sdf[0].item() == 1.0                       # reference: NaN -> 1.0 (points_to_surf_eval.py:205-207)
    assert np.isfinite(sdf[1].item())


# ---------------------------------------------------------------- whole path
def test_infer_shape_matches_reference_full_grid32(model_max, cloud_dev, golden_dir, meta, torch_cuda):
    from points2surf_amd import engine
    model, w, cfg = model_max
    g = np.load(os.path.join(golden_dir, 'ref_p2s_max_grid32.npz'))
    rng = engine.Rng(meta['seed_data'])
    sdf, q = engine.infer_shape(model, cloud_dev, rng, 32, 3, chunk=1000)
    sdf = sdf.cpu().numpy()
    ref = g['sdf_full']
    assert sdf.shape == ref.shape
    assert np.array_equal(q.cpu().numpy(), np.load(os.path.join(golden_dir, 'query_grid_32_3.npy')))
    d = np.abs(sdf - ref)
    flips = int((np.sign(sdf) != np.sign(ref)).sum())
    print('infer_shape grid32: max|dSDF| %.3g mean %.3g sign flips %d / %d' % (d.max(), d.mean(), flips, ref.size))
    assert d.max() < SDF_TOL_TIGHT < SDF_TOL
    assert flips == 0


def test_infer_shape_chunking_and_ranges_are_consistent(model_max, cloud_dev, meta, torch_cuda):
    """size-independent properties: the result does not depend on the internal chunk size, and a
    query sub-range continues the RNG stream exactly (shape/query sharding relies on this)."""
    from points2surf_amd import engine
    model, w, cfg = model_max
    a, _ = engine.infer_shape(model, cloud_dev, engine.Rng(7), 32, 3, chunk=4096)
    b, _ = engine.infer_shape(model, cloud_dev, engine.Rng(7), 32, 3, chunk=333)
    assert torch_cuda.equal(a, b)
    r = engine.Rng(7)
    p1, _ = engine.infer_shape(model, cloud_dev, r, 32, 3, q_begin=0, q_end=1500)
    p2, _ = engine.infer_shape(model, cloud_dev, r, 32, 3, q_begin=1500, q_end=-1)
    assert torch_cuda.equal(torch_cuda.cat([p1, p2]), a)


def test_rng_parallel_jump_ahead_matches_numpy(cloud_dev, fixture_cloud, torch_cuda):
    """large requests take the GF(2) jump-ahead path (sessions of 512 streams): same stream as numpy / the serial kernel,
    including the resume position, across ragged sizes and across the two code paths"""
    from points2surf_amd import engine
    n_pts = fixture_cloud.shape[0]
    par = engine.Rng(2024, parallel=True)
    ser = engine.Rng(2024, parallel=False)
    ref = np.random.RandomState(2024)
    for nq, n in ((3, 1000), (4096, 1000), (5, 7), (5000, 1000), (450, 1000), (1, 1)):
        a = par.subsample_uniform(cloud_dev, nq, n, want_pts=False)[0].cpu().numpy().reshape(-1)
        b = ser.subsample_uniform(cloud_dev, nq, n, want_pts=False)[0].cpu().numpy().reshape(-1)
        r = ref.randint(0, n_pts, nq * n)
        assert np.array_equal(b, r), ('serial', nq, n)
        assert np.array_equal(a, r), ('parallel', nq, n)
        mt, pos = par.get_state()
        st = ref.get_state()
        assert np.array_equal(mt, st[1]) and pos == st[2], (nq, n)
    par.check()


def test_rng_parallel_worst_case_acceptance(torch_cuda):
    """N just above a power of two: acceptance ~50 %, the request still fits one super-segment"""
    from points2surf_amd import engine
    n_pts = 32769
    pts = np.random.default_rng(0).uniform(-0.5, 0.5, (n_pts, 3)).astype(np.float32)
    c = engine.Cloud(pts)
    r = engine.Rng(77, parallel=True)
    got = r.subsample_uniform(c, 4096, 1000, want_pts=False)[0].cpu().numpy().reshape(-1)
    assert np.array_equal(got, np.random.RandomState(77).randint(0, n_pts, 4096 * 1000))
    r.check()
