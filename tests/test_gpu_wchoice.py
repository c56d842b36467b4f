"""a6, distance-weighted mode (p2s_vanilla): the device implementation of ``rng.choice(N, n, replace=False, p)``
(p2s_subsample_weighted) against the unmodified reference's golden ids, numpy's legacy RandomState and the oracle.
Bit-exact: ids AND the position of the shared MT19937 stream afterwards."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return torch


@pytest.fixture(scope='module')
def meta(golden_dir):
    with open(os.path.join(golden_dir, 'meta.json')) as f:
        return json.load(f)


def _numpy_reference(seed, pts, queries, n):
    """the reference's call sequence (source/base/utils.py:200-219) on numpy's own legacy generator"""
    from oracle import p2s_oracle as O
    rs = np.random.RandomState(seed)
    out = np.empty((queries.shape[0], n), dtype=np.int64)
    for i, q in enumerate(queries):
        out[i] = rs.choice(pts.shape[0], size=n, replace=False, p=O.dist_prob(pts, q))
    return out, rs


def _same_stream_position(rng_dev, rs):
    mt, pos = rng_dev.get_state()
    st = rs.get_state()
    if pos == 624 or st[2] == 624:          # numpy twists lazily; compare through the next draws instead
        return None
    return np.array_equal(mt, st[1]) and pos == st[2]


def test_weighted_ids_match_reference_golden(fixture_cloud, golden_dir, meta, torch_cuda):
    from points2surf_amd import engine
    g = np.load(os.path.join(golden_dir, 'ref_p2s_vanilla_grid32.npz'))
    cloud = engine.Cloud(fixture_cloud)
    q = cloud.query_grid(32, 3)[:meta['nq']]
    r = engine.Rng(meta['seed_data'])
    ids, pts = r.subsample_weighted(cloud, q, 1000)
    r.check()
    assert np.array_equal(ids.cpu().numpy(), g['sub_ids'])
    assert np.array_equal(pts.cpu().numpy(), fixture_cloud[g['sub_ids']])


def test_weighted_matches_numpy_legacy_choice_and_stream_position(fixture_cloud, torch_cuda):
    from points2surf_amd import engine
    rng = np.random.default_rng(5)
    cloud = engine.Cloud(fixture_cloud)
    q = (fixture_cloud[rng.integers(0, fixture_cloud.shape[0], 48)] + rng.normal(0, 0.02, (48, 3))).astype(np.float32)
    q[0] = fixture_cloud[17]                 # query on a cloud point: d = 0 -> p = clip(1) -> fine
    r = engine.Rng(2024)
    # ragged calls: the stream continues across them
    got = np.concatenate([r.subsample_weighted(cloud, torch_cuda.from_numpy(q[a:b]).cuda(), 1000, want_pts=False)[0]
                          .cpu().numpy() for a, b in ((0, 1), (1, 8), (8, 48))])
    r.check()
    ref, rs = _numpy_reference(2024, fixture_cloud, q, 1000)
    assert np.array_equal(got, ref)
    # both generators continue identically (uniform draws after the weighted ones)
    nxt = r.subsample_uniform(cloud, 3, 1000, want_pts=False)[0].cpu().numpy().reshape(-1)
    assert np.array_equal(nxt, rs.randint(0, fixture_cloud.shape[0], 3000))
    same = _same_stream_position(r, rs)
    assert same is None or same


@pytest.mark.parametrize('n_pts,n_sel,nq,seed', [
    (1000, 1000, 6, 1),        # every point selected: many redraw rounds, found set grows to the whole cloud
    (1500, 1000, 12, 2),       # heavy collisions
    (8200, 1000, 10, 3),       # np.sum crosses its 8192-element buffer boundary by 8 elements
    (8191, 250, 10, 4),        # odd sizes: leaf with a non-multiple-of-8 tail, per-thread draw counts < 4
    (150001, 1000, 5, 5),      # largest cloud size of the data sets (19 buffer chunks)
    (20000, 1, 9, 6),          # a single draw per query
])
def test_weighted_other_sizes(n_pts, n_sel, nq, seed, torch_cuda):
    from points2surf_amd import engine
    g = np.random.default_rng(seed)
    pts = g.uniform(-0.6, 0.6, (n_pts, 3)).astype(np.float32)
    q = g.uniform(-0.7, 0.7, (nq, 3)).astype(np.float32)
    cloud = engine.Cloud(pts)
    r = engine.Rng(seed)
    got = r.subsample_weighted(cloud, torch_cuda.from_numpy(q).cuda(), n_sel, want_pts=False)[0].cpu().numpy()
    r.check()
    ref, rs = _numpy_reference(seed, pts, q, n_sel)
    assert np.array_equal(got, ref)
    nxt = r.subsample_uniform(cloud, 1, min(n_pts, 500), want_pts=False)[0].cpu().numpy().reshape(-1)
    assert np.array_equal(nxt, rs.randint(0, n_pts, min(n_pts, 500)))


def test_weighted_request_splitting_is_invisible(fixture_cloud, torch_cuda, monkeypatch):
    """more queries than one raw request of random words serves: results identical to one numpy stream"""
    from points2surf_amd import engine
    rng = np.random.default_rng(11)
    cloud = engine.Cloud(fixture_cloud)
    q = (fixture_cloud[rng.integers(0, fixture_cloud.shape[0], 30)] + rng.normal(0, 0.05, (30, 3))).astype(np.float32)
    monkeypatch.setenv('P2S_WCHOICE_QUERIES', '7')      # 7 queries per request -> 5 requests
    r = engine.Rng(99)
    got = r.subsample_weighted(cloud, torch_cuda.from_numpy(q).cuda(), 1000, want_pts=False)[0].cpu().numpy()
    r.check()
    ref, rs = _numpy_reference(99, fixture_cloud, q, 1000)
    assert np.array_equal(got, ref)
    nxt = r.subsample_uniform(cloud, 1, 100, want_pts=False)[0].cpu().numpy().reshape(-1)
    assert np.array_equal(nxt, rs.randint(0, fixture_cloud.shape[0], 100))


def test_weighted_degenerate_cloud_is_reported(torch_cuda):
    from points2surf_amd import engine, _lib
    pts = np.zeros((1200, 3), dtype=np.float32)         # all distances 0 -> numpy raises (NaN probabilities)
    cloud = engine.Cloud(pts)
    r = engine.Rng(1)
    r.subsample_weighted(cloud, torch_cuda.zeros((2, 3), device='cuda'), 1000, want_pts=False)
    with pytest.raises(_lib.P2SError):
        r.check()


def test_infer_shape_vanilla_matches_reference_full_grid32(fixture_cloud, golden_dir, meta, torch_cuda):
    """whole path for p2s_vanilla (QSTN + distance-weighted sub-sample) on the device vs the unmodified reference"""
    from points2surf_amd import engine, synth
    w, cfg = synth.make_weights('p2s_vanilla')
    model = engine.Model(w, cfg)
    cloud = engine.Cloud(fixture_cloud)
    g = np.load(os.path.join(golden_dir, 'ref_p2s_vanilla_grid32.npz'))
    rng = engine.Rng(meta['seed_data'])
    sdf, q = engine.infer_shape(model, cloud, rng, 32, 3, chunk=1000)
    rng.check()
    sdf = sdf.cpu().numpy()
    ref = g['sdf_full']
    d = np.abs(sdf - ref)
    flips = int((np.sign(sdf) != np.sign(ref)).sum())
    print('vanilla infer_shape grid32: max|dSDF| %.3g mean %.3g sign flips %d / %d' % (d.max(), d.mean(), flips, ref.size))
    assert d.max() < 1e-5
    assert flips == 0
    # chunking / query ranges do not change the stream
    rng2 = engine.Rng(meta['seed_data'])
    a, _ = engine.infer_shape(model, cloud, rng2, 32, 3, q_begin=0, q_end=1234, chunk=500)
    b, _ = engine.infer_shape(model, cloud, rng2, 32, 3, q_begin=1234, q_end=-1, chunk=4096)
    assert np.array_equal(np.concatenate([a.cpu().numpy(), b.cpu().numpy()]), sdf)


def test_rng_sessions_mixed_calls_match_numpy(fixture_cloud, torch_cuda):
    """one generated session serves many calls: consecutive takes without touching the state in between, small
    requests inside an open session, a switch to another cloud size, weighted draws in between, and the lazily
    committed state -- all identical to one numpy RandomState"""
    from points2surf_amd import engine
    from oracle import p2s_oracle as O
    n_pts = fixture_cloud.shape[0]
    cloud = engine.Cloud(fixture_cloud)
    small_pts = np.random.default_rng(3).uniform(-0.5, 0.5, (20011, 3)).astype(np.float32)
    small = engine.Cloud(small_pts)
    r = engine.Rng(31337)
    ref = np.random.RandomState(31337)

    def uni(c, n, nq, k):
        got = r.subsample_uniform(c, nq, k, want_pts=False)[0].cpu().numpy().reshape(-1)
        assert np.array_equal(got, ref.randint(0, n, nq * k)), (n, nq, k)

    uni(cloud, n_pts, 600, 1000)          # opens a session
    uni(cloud, n_pts, 4096, 1000)         # take
    uni(cloud, n_pts, 3, 11)              # small request served by the open session
    uni(cloud, n_pts, 1, 1)
    uni(small, 20011, 500, 1000)          # other modulus: close (commit) + new session
    q = (fixture_cloud[[5, 77, 4000]] + np.float32(0.01)).astype(np.float32)
    got = r.subsample_weighted(cloud, torch_cuda.from_numpy(q).cuda(), 1000, want_pts=False)[0].cpu().numpy()
    want = np.stack([ref.choice(n_pts, size=1000, replace=False, p=O.dist_prob(fixture_cloud, qq)) for qq in q])
    assert np.array_equal(got, want)      # raw-word session after a value session
    uni(cloud, n_pts, 700, 1000)          # and back
    mt, pos = r.get_state()               # closes the session: state = numpy's
    st = ref.get_state()
    assert np.array_equal(mt, st[1]) and pos == st[2]
    uni(cloud, n_pts, 2, 5)               # serial kernel on the committed state
    r.check()


def test_speculative_offsets_across_blocks_match_numpy_and_serial_kernel(fixture_cloud, torch_cuda, monkeypatch):
    """the parallel offsets pass (wc_spec_kernel + wc_chain_kernel, blocks of 2048 queries) against numpy's stream for
    4,300 consecutive grid queries (three blocks), and against the in-order path of the chain kernel (P2S_WC_SERIAL) including
    the skip path
    (NULL ids: stream advanced only)"""
    from points2surf_amd import engine
    cloud = engine.Cloud(fixture_cloud)
    q = cloud.query_grid(64, 3)[1000:5300].contiguous()
    nq = int(q.shape[0])
    r = engine.Rng(4242)
    got = r.subsample_weighted(cloud, q, 1000, want_pts=False)[0].cpu().numpy()
    r.check()
    ref, rs = _numpy_reference(4242, fixture_cloud, q.cpu().numpy(), 1000)
    assert np.array_equal(got, ref)
    tail = r.subsample_uniform(cloud, 1, 777, want_pts=False)[0].cpu().numpy().reshape(-1)
    assert np.array_equal(tail, rs.randint(0, fixture_cloud.shape[0], 777))
    # skip path: no ids kernel cross-check behind it -> compare the stream position with the full path
    r2 = engine.Rng(4242)
    r2.skip(cloud, 1000, query_ms=q)
    r2.check()
    tail2 = r2.subsample_uniform(cloud, 1, 777, want_pts=False)[0].cpu().numpy().reshape(-1)
    assert np.array_equal(tail2, tail)
    monkeypatch.setenv('P2S_WC_SERIAL', '1')
    r3 = engine.Rng(4242)
    got3 = r3.subsample_weighted(cloud, q, 1000, want_pts=False)[0].cpu().numpy()
    r3.check()
    assert np.array_equal(got3, got) and nq == 4300


def test_weighted_at_the_size_cap(torch_cuda):
    """the clouds the weighted sub-sample takes (LDS bitmap of the in-place algorithm): 170,000 points work
    and match numpy, 260,000 too; 480,000 are refused
    with an error code, not a wrong result (more sizes: tests/test_gpu_stress.py)"""
    from points2surf_amd import engine, _lib
    g = np.random.default_rng(17)
    pts = g.uniform(-0.7, 0.7, (170000, 3)).astype(np.float32)
    q = g.uniform(-0.5, 0.5, (3, 3)).astype(np.float32)
    cloud = engine.Cloud(pts)
    r = engine.Rng(3)
    got = r.subsample_weighted(cloud, torch_cuda.from_numpy(q).cuda(), 1000, want_pts=False)[0].cpu().numpy()
    r.check()
    ref, rs = _numpy_reference(3, pts, q, 1000)
    assert np.array_equal(got, ref)
    pts2 = g.uniform(-0.7, 0.7, (260000, 3)).astype(np.float32)
    mid = engine.Cloud(pts2)
    r2 = engine.Rng(3)
    got2 = r2.subsample_weighted(mid, torch_cuda.from_numpy(q).cuda(), 1000, want_pts=False)[0].cpu().numpy()
    r2.check()
    assert np.array_equal(got2, _numpy_reference(3, pts2, q, 1000)[0])
    big = engine.Cloud(g.uniform(-0.7, 0.7, (480000, 3)).astype(np.float32))
    with pytest.raises(_lib.P2SError):
        engine.Rng(3).subsample_weighted(big, torch_cuda.from_numpy(q).cuda(), 1000, want_pts=False)


@pytest.mark.parametrize('n_pts,nq', [(1100, 2300), (2500, 2300), (6000, 4300)])
def test_skip_path_equals_ids_path_on_collision_heavy_clouds(n_pts, nq, torch_cuda):
    """the NULL-ids skip (hand-off mode, query-range sharding) trusts the offsets pass -- tentative verdicts, band
    look-ups, jump tables -- without the ids kernel's ``base + used != next`` cross-check behind it.  Clouds barely larger
    than the sub-sample make collisions the rule (1000 of 1100 points: hundreds of redraws per query over many rounds, every
    candidate 'undecided' -> the complete algorithm in place; 2500 / 6000: ~200 / ~80 first-round collisions, the band
    kernel's territory) over more than one block of 2048 queries: the generator state after the skip must equal the state
    after the ids path AND numpy's, and the ids numpy's."""
    from points2surf_amd import engine
    g = np.random.default_rng(n_pts)
    pts = (g.normal(0, 0.25, (n_pts, 3)) * np.array([1.0, 0.6, 0.3])).astype(np.float32)
    q = (pts[g.integers(0, n_pts, nq)] + g.normal(0, 0.03, (nq, 3))).astype(np.float32)
    cloud = engine.Cloud(pts)
    qd = torch_cuda.from_numpy(q).cuda()
    r = engine.Rng(99)
    ids = r.subsample_weighted(cloud, qd, 1000, want_pts=False)[0].cpu().numpy()
    r.check()
    r2 = engine.Rng(99)
    r2.skip(cloud, 1000, query_ms=qd)
    r2.check()
    mt, pos = r.get_state()
    mt2, pos2 = r2.get_state()
    assert pos == pos2 and np.array_equal(mt, mt2)
    ref, rs = _numpy_reference(99, pts, q, 1000)
    assert np.array_equal(ids, ref)
    tail = r2.subsample_uniform(cloud, 1, 64, want_pts=False)[0].cpu().numpy().reshape(-1)
    assert np.array_equal(tail, rs.randint(0, n_pts, 64))
