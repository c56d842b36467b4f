"""Full-size (BASELINE.json configs[2]: 256^3, k=300, n=1000) checks through size-independent properties and
spot checks against the oracle at known positions of the RNG stream."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def setup(fixture_cloud):
    import torch
    from points2surf_amd import engine, synth
    w, cfg = synth.make_weights('p2s_max')
    model = engine.Model(w, cfg)
    cloud = engine.Cloud(fixture_cloud)
    return torch, engine, model, cloud, w, cfg


def test_full_256_grid_properties_and_oracle_spot_checks(setup, fixture_cloud, golden_dir):
    torch, engine, model, cloud, w, cfg = setup
    from oracle import p2s_oracle as O
    seed = 40938661
    sdf, q = engine.infer_shape(model, cloud, engine.Rng(seed), 256, 3)
    Q = q.shape[0]
    assert Q == 307237                                     # reference count for this shape (golden meta / SURVEY)
    s = sdf.cpu().numpy()
    qn = q.cpu().numpy()
    assert np.isfinite(s).all()
    # |sdf| = tanh^2(.) * r <= r: bounded by the largest kNN radius of the shape
    _, _, rad = cloud.knn_patch(q[::997], 300, want_ids=False, want_patch=False)
    assert np.abs(s).max() <= float(rad.max().item()) * 1.5 + 1e-6
    # queries are distinct voxel centres in C order of the voxel index
    vox = np.floor((qn.astype(np.float64) + 1) / 2 * 256).astype(np.int64)
    lin = (vox[:, 0] * 256 + vox[:, 1]) * 256 + vox[:, 2]
    assert (np.diff(lin) > 0).all()
    # chunk invariance at full size: internal batching must not change a single bit
    sdf2, _ = engine.infer_shape(model, cloud, engine.Rng(seed), 256, 3, chunk=1500, want_queries=False)
    assert torch.equal(sdf, sdf2)
    # spot checks against the oracle: queries at stream offsets far into the shape
    for q0 in (0, 150000, Q - 8):
        rng = O.LegacyMT19937(seed)
        skip = q0 * 1000
        while skip > 0:                                   # advance the oracle's stream past the first q0 queries
            n = min(skip, 5_000_000)
            rng.randint(fixture_cloud.shape[0], n)
            skip -= n
        ids = O.knn_ids(fixture_cloud, qn[q0:q0 + 8], 300)
        r, ps = O.patch_radius_and_ps(fixture_cloud, ids, qn[q0:q0 + 8])
        sub = fixture_cloud[np.stack([rng.randint(fixture_cloud.shape[0], 1000) for _ in range(8)])]
        ref = O.post_process(O.model_forward(w, cfg, ps, sub, qn[q0:q0 + 8]), r)
        assert np.abs(s[q0:q0 + 8] - ref).max() < 1e-5, q0
        assert np.array_equal(np.sign(s[q0:q0 + 8]), np.sign(ref))


def test_query_sharding_of_one_shape_matches_whole(setup):
    """intra-shape sharding (few shapes, many GPUs): contiguous query ranges with the RNG stream carried over
    reproduce the single-call result (reference semantics: one stream in query order)."""
    torch, engine, model, cloud, w, cfg = setup
    whole, _ = engine.infer_shape(model, cloud, engine.Rng(11), 128, 3, want_queries=False)
    Q = whole.shape[0]
    cuts = [0, Q // 3, Q // 3 + 1, Q - 5, Q]
    r = engine.Rng(11)
    parts = [engine.infer_shape(model, cloud, r, 128, 3, q_begin=a, q_end=b, want_queries=False)[0]
             for a, b in zip(cuts[:-1], cuts[1:])]
    assert torch.equal(torch.cat(parts), whole)


def test_volume_of_full_256_inference_is_consistent(setup):
    """row f-1 at full size: every sample keeps its value, borders are outside, propagation only fills zeros"""
    torch, engine, model, cloud, w, cfg = setup
    sdf, q = engine.infer_shape(model, cloud, engine.Rng(3), 256, 3)
    vol, iters = engine.sdf_volume(q, sdf, 256, 5, 13.0)
    v = vol.cpu().numpy()
    qn = q.cpu().numpy()
    vox = np.floor((qn.astype(np.float64) + 1) / 2 * 256).astype(np.int64)
    inner = ((vox > 0) & (vox < 255)).all(axis=1)
    got = v[vox[inner, 0], vox[inner, 1], vox[inner, 2]]
    assert np.array_equal(got, np.clip(sdf.cpu().numpy()[inner], -1, 1))
    assert (v[0] == -1).all() and (v[-1] == -1).all() and (v[:, 0] == -1).all() and (v[:, :, -1] == -1).all()
    filled = np.ones_like(v, dtype=bool)
    filled[vox[:, 0], vox[:, 1], vox[:, 2]] = False
    assert np.isin(v[filled], (-1.0, 0.0, 1.0)).all()
    assert iters >= 2


def test_grid_512_config5_size(setup):
    """BASELINE configs[4] size (grid 512): count equals the reference's for this shape, values sane"""
    torch, engine, model, cloud, w, cfg = setup
    q = cloud.query_grid(512, 3)
    assert q.shape[0] == 757499
    sdf, _ = engine.infer_shape(model, cloud, engine.Rng(5), 512, 3, q_begin=700000, q_end=-1, want_queries=False)
    assert sdf.shape[0] == 57499 and bool(torch.isfinite(sdf).all())
