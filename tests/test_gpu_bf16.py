"""bf16 encoder mode (BASELINE.json configs[3]: "bf16 encoder + fp32 decoder"; cfg encoder_bf16=1): the kernel against
an executable model of its arithmetic, and its deviation from the exact fp32 path on the reference fixture."""
import json
import os

import numpy as np
import pytest

from conftest import missing_golden, flip_logits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return torch


def _models(name):
    from points2surf_amd import engine, synth
    w, cfg = synth.make_weights(name)
    cfg16 = dict(cfg)
    cfg16['encoder_bf16'] = True
    return w, cfg, engine.Model(w, cfg), engine.Model(w, cfg16)


def test_bf16_encoder_matches_its_arithmetic_model(fixture_cloud, torch_cuda):
    from oracle import p2s_oracle as O
    from tests import bf16_model as bm
    w, cfg, m32, m16 = _models('p2s_max')
    rng = np.random.default_rng(2)
    B = 24
    q = (fixture_cloud[rng.integers(0, fixture_cloud.shape[0], B)] + rng.normal(0, 0.01, (B, 3))).astype(np.float32)
    ids = O.knn_ids(fixture_cloud, q, 300)
    _, ps = O.patch_radius_and_ps(fixture_cloud, ids, q)
    sub = fixture_cloud[rng.integers(0, fixture_cloud.shape[0], (B, 1000))]
    t = lambda a: torch_cuda.from_numpy(np.ascontiguousarray(a)).cuda()
    fl, fg = m16.features(t(ps), t(sub), t(q))
    w32 = {k: np.asarray(v, dtype=np.float32) for k, v in w.items()}
    ref_l = bm.encoder_features(w32, 'feat_local', ps.astype(np.float32))
    ref_g = bm.encoder_features(w32, 'feat_global', (sub - q[:, None, :]).astype(np.float32))
    el = np.abs(fl.cpu().numpy() - ref_l).max() / np.abs(ref_l).max()
    eg = np.abs(fg.cpu().numpy() - ref_g).max() / np.abs(ref_g).max()
    f32l, f32g = m32.features(t(ps), t(sub), t(q))
    dl = np.abs(fl.cpu().numpy() - f32l.cpu().numpy()).max() / np.abs(ref_l).max()
    dg = np.abs(fg.cpu().numpy() - f32g.cpu().numpy()).max() / np.abs(ref_g).max()
    print('bf16 kernel vs its model: rel err local %.3g global %.3g ; bf16 vs fp32 features: %.3g %.3g' % (el, eg, dl, dg))
    # the model differs from the kernel only by fp32 summation order (which can flip single bf16 roundings)
    assert el < 4e-3 and eg < 4e-3
    assert dl < 0.1 and dg < 0.1


@pytest.mark.parametrize('name', ['p2s_max', 'p2s_vanilla'])
def test_bf16_deviation_from_fp32_on_reference_fixture(name, fixture_cloud, golden_dir, torch_cuda):
    """whole shape (2976 queries, grid 32) against the unmodified reference's SDF: the deviation of the two logits is
    bounded; the sign (hence the SDF, = magnitude * sign) can only flip where the reference's sign logit is itself
    within that deviation of zero"""
    from points2surf_amd import engine
    w, cfg, m32, m16 = _models(name)
    with open(os.path.join(golden_dir, 'meta.json')) as f:
        meta = json.load(f)
    ref = np.load(os.path.join(golden_dir, 'ref_%s_grid32.npz' % name))['sdf_full']
    cloud = engine.Cloud(fixture_cloud)
    q = cloud.query_grid(32, 3)
    _, patch, rad = cloud.knn_patch(q, 300)
    rng = engine.Rng(meta['seed_data'])
    if cfg.get('uniform_subsample'):
        _, sub = rng.subsample_uniform(cloud, int(q.shape[0]), 1000)
    else:
        _, sub = rng.subsample_weighted(cloud, q, 1000)
    l32, s32 = m32.forward(patch, sub, q, rad, want_logits=True, want_sdf=True)
    l16, s16 = m16.forward(patch, sub, q, rad, want_logits=True, want_sdf=True)
    l32, l16, s32, s16 = (t.cpu().numpy() for t in (l32, l16, s32, s16))
    assert np.abs(s32 - ref).max() < 1e-5                      # the fp32 path is the reference
    dl = np.abs(l16 - l32)
    flips = np.sign(s16) != np.sign(ref)
    dm = np.abs(np.abs(s16) - np.abs(ref))
    print('%s bf16 encoder vs reference: max|dlogit| %.3g mean %.3g (logit range %.2f); |SDF| magnitude max dev %.3g mean '
          '%.3g; sign flips %d / %d, largest |sign logit| among them %.3g'
          % (name, dl.max(), dl.mean(), np.abs(l32).max(), dm.max(), dm.mean(), int(flips.sum()), ref.size,
             np.abs(l32[flips, 1]).max() if flips.any() else 0.0))
    assert np.isfinite(s16).all()
    assert dl.max() < 0.25 and dl.mean() < 0.04          # measured: 0.088 / 0.014 (p2s_max), logit range 7.7
    assert not flips.any() or np.abs(l32[flips, 1]).max() <= dl.max()
    assert flips.mean() < 0.03


@pytest.mark.parametrize('pieces,name', [(2, 'p2s_max'), (3, 'p2s_max'), (2, 'p2s_vanilla'), (3, 'p2s_vanilla'),
                                         (3, 'p2s_vanilla_mixed'), (4, 'p2s_max'), (4, 'p2s_vanilla'), (4, 'p2s_vanilla_mixed')])
def test_split_bf16_against_the_reference(pieces, name, golden_dir, torch_cuda):
    """split precision (cfg encoder_bf16 = 2 / 3: every operand as 2 / 3 bf16 pieces, 3 / 6 bf16 MFMAs per product,
    fp32 accumulate): the WHOLE 128^3 grid (68,088 queries; the 2,976 of grid 32 if that golden is absent) against
    the unmodified reference.  The north_star's contract is |dSDF| <= 1e-4; sign flips are counted and reported
    (3 pieces = 24 mantissa bits must have none; 2 pieces = 16 bits may flip queries whose sign logit is ~1e-4)."""
    import torch
    from points2surf_amd import engine, synth
    w, cfg = synth.make_weights(name)
    m = engine.Model(w, dict(cfg, encoder_bf16=pieces))
    fix = os.path.join(golden_dir, 'abc_minimal', '04_pts', '00994122_57d9d4755722f9d2d7436f0a_trimesh_000.xyz.npy')
    big = os.path.join(golden_dir, 'ref_rec_%s_testset_grid128.npz' % name)
    if os.path.isfile(big):
        ref, res = np.load(big)['rec_0'], 128
    else:
        ref, res = np.load(os.path.join(golden_dir, 'ref_%s_grid32.npz' % name))['sdf_full'], 32
    cloud = engine.Cloud(np.load(fix))
    sdf, _ = engine.infer_shape(m, cloud, engine.Rng(40938661), res, 3)
    torch.cuda.synchronize()
    sdf = sdf.cpu().numpy()
    d = np.abs(sdf - ref)
    flips = np.sign(sdf) != np.sign(ref)
    # magnitude and sign are separate logits: a flipped sign turns d into 2 |SDF| whatever the magnitude -- report the
    # magnitude deviation separately
    dm = np.abs(np.abs(sdf) - np.abs(ref))
    print('%s encoder mode %d, grid %d: max |d|SDF|| %.3g (mean %.3g), max |dSDF| %.3g, sign flips %d / %d'
          % (name, pieces, res, dm.max(), dm.mean(), d.max(), int(flips.sum()), ref.size))
    assert dm.max() < 1e-4
    if pieces in (3, 4):     # 3 bf16 pieces (6 MFMAs per product) / the fp16 pair (3 MFMAs per product)
        assert d.max() < 1e-4 and not flips.any()              # inside the contract: the fast exact modes
    else:
        # 16 mantissa bits move the sign logit by ~1e-4: about one query in 30,000 has a sign logit that small and
        # flips -- measured 2 / 68,088 (p2s_max) -- so two pieces are OUTSIDE the contract; bounded here, documented
        assert flips.sum() <= 8
    if name == 'p2s_vanilla_mixed':      # the weight set whose sign decision is tight: half the queries positive
        assert 0.3 < float((ref > 0).mean()) < 0.7


def test_fp16_pair_mode_repairs_activations_beyond_the_half_range(fixture_cloud, torch_cuda):
    """encoder_bf16 = 4 carries activations as fp16 pairs (max 65504): a model whose activations leave that range must not
    pass silently.  r05: the affected queries came out as 1.0 and the call failed; r06: they are re-run through the fp32
    kernels inside the same call (tests/test_gpu_fp16_fallback.py) -- here EVERY query of the shape (first layer times 1e6,
    not compensated: another function, but the same one in every mode) and the result equals the fp32 encoders' bit for bit"""
    import torch
    from points2surf_amd import engine, synth
    w, cfg = synth.make_weights('p2s_max')
    w = dict(w)
    w['feat_local.conv0a.weight'] = (w['feat_local.conv0a.weight'] * np.float32(1e6)).astype(np.float32)
    cloud = engine.Cloud(fixture_cloud)
    out = {}
    for mode in (4, 0, 3):   # the same weights are fine in fp32 and in the bf16 split (bf16 has the fp32 exponent range)
        m = engine.Model(w, dict(cfg, encoder_bf16=mode))
        sdf, _ = engine.infer_shape(m, cloud, engine.Rng(40938661), 16, 3)
        torch.cuda.synchronize()
        assert torch.isfinite(sdf).all()
        out[mode] = (sdf.clone(), int(m.counters()['fallback_queries']))
        m.close()
    assert out[4][1] == out[4][0].shape[0] and out[0][1] == 0 and out[3][1] == 0
    assert torch.equal(out[4][0], out[0][0])


@pytest.mark.parametrize('name', ['p2s_max', 'p2s_vanilla'])
def test_fp16_pair_full_512_grid(name, golden_dir, torch_cuda):
    """BASELINE configs[3] / [4] together: the fp16-pair encoder over every one of the 757,499 queries of the 512^3 grid
    against the unmodified reference -- magnitudes within the 1e-4 contract, signs identical except fp32 TIES of the sign
    decision (|sign logit| < parity.TIE_LOGIT_SPLIT = 2e-5 on the device; p2s_max has two such queries in fp32 as well)"""
    import torch
    from points2surf_amd import engine, synth, parity
    path = os.path.join(golden_dir, 'ref_rec_%s_testset_grid512.npz' % name)
    if not os.path.isfile(path):
        missing_golden('512^3 golden not generated')
    ref = np.load(path)['rec_0']
    w, cfg = synth.make_weights(name)
    m = engine.Model(w, dict(cfg, encoder_bf16=4))
    fix = os.path.join(golden_dir, 'abc_minimal', '04_pts', '00994122_57d9d4755722f9d2d7436f0a_trimesh_000.xyz.npy')
    cloud = engine.Cloud(np.load(fix))
    sdf, q = engine.infer_shape(m, cloud, engine.Rng(40938661), 512, 3)
    torch.cuda.synchronize()
    c = parity.compare_sdf(sdf.cpu().numpy(), ref)
    print('%s fp16 pair, grid 512: max |dSDF| %.3g (magnitudes at flipped signs), sign flips %d / %d'
          % (name, c['max_abs_dsdf'], c['flipped'].size, ref.size))
    assert c['max_abs_dsdf'] < 1e-4 and c['flipped'].size <= 8
    for j in c['flipped']:
        lg = flip_logits(m, w, cfg, cloud, engine.Rng(40938661), q, int(j))     # device (fp16 pair) and CPU port (fp32)
        assert parity.is_tie(lg[0], lg[1], encoder_bf16=4), (int(j), lg)
