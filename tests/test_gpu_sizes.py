"""GPU parity at the sizes the metric is quoted on (VERDICT r1 'next' item 1): full query grids against goldens the
UNMODIFIED reference wrote on CPU (oracle/make_golden_sizes.py) -- every query of the 128^3 grid (BASELINE
configs[1]) and of the 256^3 grid (configs[2]) of the abc_minimal test shape, and the three abc_minimal clouds in one
dataset (one RNG stream across shapes) at grids 32 and 64.  Tolerance: the north_star's 1e-4 on the SDF, and ZERO
sign flips (what the mesh depends on)."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import missing_golden, flip_logits

from points2surf_amd import parity

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
FIX = os.path.join(GOLDEN, 'abc_minimal')
SEED = 40938661


def _names(dataset):
    with open(os.path.join(FIX, dataset + '.txt')) as f:
        return [x.strip() for x in f if x.strip()]


def _golden(job, model, dataset, res):
    key = 'ref_%s_%s_%s_grid%d' % (job, model, dataset, res)
    path = os.path.join(GOLDEN, key + '.npz')
    if not os.path.isfile(path):
        missing_golden('%s not generated (hours of reference CPU time; see oracle/make_golden_sizes.py)' % key)
    with open(os.path.join(GOLDEN, 'meta_sizes.json')) as f:
        meta = json.load(f)[key]
    return np.load(path), meta


def _clouds(dataset):
    """the clouds of a dataset in order; ``standin2``: two stand-in clouds of SURVEY 8d configs 3-5 (the first two
    abc_minimal clouds under seeded rotations, re-normalised: points2surf_amd/synth.py:standin_cloud, seeds 0 / 1)"""
    from points2surf_amd import synth
    if dataset == 'standin2':
        bases = [np.load(os.path.join(FIX, '04_pts', n + '.xyz.npy')) for n in sorted(_names('abc3'))]
        return [synth.standin_cloud(bases[i], i) for i in range(2)]
    return [np.load(os.path.join(FIX, '04_pts', n + '.xyz.npy')) for n in _names(dataset)]


def _run_dataset(model_name, dataset, res):
    """what points_to_surf_eval does for a dataset in reconstruction mode: one stream over all shapes"""
    import torch
    from points2surf_amd import engine, synth
    w, cfg = synth.make_weights(model_name)
    model = engine.Model(w, cfg)
    rng = engine.Rng(SEED)
    out = []
    for pts in _clouds(dataset):
        cloud = engine.Cloud(pts)
        sdf, q = engine.infer_shape(model, cloud, rng, res, 3)
        torch.cuda.synchronize()
        out.append((sdf.cpu().numpy(), q.cpu().numpy()))
        cloud.close()
    rng.check()
    model.close()
    return out


def _compare(out, g, meta, tol=1e-4, ties_ok=False):
    """max |dSDF| and sign flips against the reference.  ``ties_ok``: returns the flipped (shape, query) pairs instead of
    failing on them (the caller proves that each one is an fp32 tie); their magnitudes still have to agree"""
    worst, total, flipped = 0.0, 0, []
    for i, (sdf, q) in enumerate(out):
        ref = g['rec_%d' % i]
        assert sdf.shape == ref.shape, (sdf.shape, ref.shape)
        assert hashlib.sha256(np.ascontiguousarray(q).tobytes()).hexdigest() == meta['shapes'][i]['query_sha256']
        c = parity.compare_sdf(sdf, ref)
        flipped += [(i, int(j)) for j in c['flipped']]
        worst = max(worst, c['max_abs_dsdf'])
        total += sdf.size
    print('max|dSDF| %.3g (magnitudes at flipped signs), sign flips %d / %d queries' % (worst, len(flipped), total))
    assert worst < tol, worst
    if not ties_ok:
        assert not flipped, flipped
    return flipped


@pytest.mark.parametrize('model', ['p2s_max', 'p2s_vanilla'])
def test_full_grid128_matches_reference(model):
    """BASELINE configs[1]: every one of the 68,088 queries of the 128^3 grid"""
    g, meta = _golden('rec', model, 'testset', 128)
    _compare(_run_dataset(model, 'testset', 128), g, meta)


def test_full_grid128_sign_decision_with_power():
    """with the default synthetic weights p2s_vanilla's SDF is positive for 0.9 % of the 128^3
    queries -- almost every sign logit is far from zero, so "0 flips" said little.  ``p2s_vanilla_mixed`` = the same
    weights with the sign logit's bias moved by its median over this grid: the reference's golden is positive for
    49.96 % of the 68,088 queries, i.e. the decision ``sign logit >= 0`` (sdf_nn.py:16-21) is as tight as it can be --
    and still not one sign differs (weighted sub-sample, QSTN, fp32 encoders)."""
    g, meta = _golden('rec', 'p2s_vanilla_mixed', 'testset', 128)
    assert 0.3 < meta['shapes'][0]['pos_frac'] < 0.7
    out = _run_dataset('p2s_vanilla_mixed', 'testset', 128)
    _compare(out, g, meta)
    assert 0.3 < float((out[0][0] > 0).mean()) < 0.7


@pytest.mark.parametrize('res', [256])
def test_three_clouds_one_stream_256_matches_reference(res):
    """the benchmarked dataset -- all three abc_minimal clouds as ONE dataset at 256^3, 1,378,242
    queries from one continuous stream -- against the golden the unmodified reference wrote (~6 h of CPU).  All
    magnitudes within 1e-4; signs identical except fp32 TIES of the sign decision (about one query in 400,000 has a sign
    logit within the logit accuracy of zero, see test_full_grid512_matches_reference): each flipped query is re-run at
    its exact stream position and must have |sign logit| < parity.TIE_LOGIT_FP32 (1e-5; the known ties: 2.4e-7, 1.7e-6)."""
    import torch
    from points2surf_amd import engine, synth, sharding
    g, meta = _golden('rec', 'p2s_max', 'abc3', res)
    out = _run_dataset('p2s_max', 'abc3', res)
    flipped = _compare(out, g, meta, ties_ok=True)
    assert len(flipped) <= 16, flipped
    if flipped:
        w, cfg = synth.make_weights('p2s_max')
        model = engine.Model(w, cfg)
        names = _names('abc3')
        for si, j in flipped:
            rng = engine.Rng(SEED)
            for n in names[:si]:                               # the draws of the shapes before this one
                c = engine.Cloud(np.load(os.path.join(FIX, '04_pts', n + '.xyz.npy')))
                sharding.skip_shape_stream(c, rng, cfg, res, 3, model.sub_sample_size)
                c.close()
            cloud = engine.Cloud(np.load(os.path.join(FIX, '04_pts', names[si] + '.xyz.npy')))
            lg = flip_logits(model, w, cfg, cloud, rng, torch.from_numpy(out[si][1]).cuda(), j)
            print('shape %d query %d: sign logits device %.3g / CPU port %.3g, device sdf %.6g, reference %.6g'
                  % (si, j, lg[0], lg[1], out[si][0][j], g['rec_%d' % si][j]))
            assert parity.is_tie(lg[0], lg[1]), (si, j, lg)          # the same two-logit rule as bench.py's self-check
            cloud.close()


@pytest.mark.parametrize('model', ['p2s_max', 'p2s_vanilla'])
def test_full_grid256_matches_reference(model):
    """BASELINE configs[2] (the benchmarked workload): every one of the 307,237 queries of the 256^3 grid"""
    g, meta = _golden('rec', model, 'testset', 256)
    _compare(_run_dataset(model, 'testset', 256), g, meta)


@pytest.mark.parametrize('model', ['p2s_max', 'p2s_vanilla'])
def test_full_grid512_matches_reference(model):
    """BASELINE configs[4] (512^3, overlapped data path): every one of the 757,499 queries of the 512^3 grid.
    The sign is ``sign logit >= 0`` (sdf_nn.py:16-21): among 757k queries a few have a sign logit within fp32 noise of
    zero (2 here, |logit| < 1e-5 with a logit accuracy of ~1.5e-5) -- the reference's own answer for them depends on
    its batch composition and thread count (the golden run says +, the same ATen ops on the same inputs in another batch
    say -5.0e-6 and -7.4e-6: oracle/torch_port.py).  Such TIES are tolerated, if and only if the device's own sign logit AND the CPU
    port's on the same inputs are that close to zero (parity.is_tie); every other query must agree in sign, and all magnitudes within 1e-4."""
    import torch
    from points2surf_amd import engine, synth
    g, meta = _golden('rec', model, 'testset', 512)
    out = _run_dataset(model, 'testset', 512)
    flipped = _compare(out, g, meta, ties_ok=True)
    assert len(flipped) <= 8, flipped                      # ~1e-5 of the queries
    if flipped:
        w, cfg = synth.make_weights(model)
        model = engine.Model(w, cfg)
        cloud = engine.Cloud(np.load(os.path.join(FIX, '04_pts', _names('testset')[0] + '.xyz.npy')))
        for _, j in flipped:
            lg = flip_logits(model, w, cfg, cloud, engine.Rng(SEED), torch.from_numpy(out[0][1]).cuda(), j)
            print('query %d: sign logits device %.3g / CPU port %.3g, device sdf %.6g, reference %.6g'
                  % (j, lg[0], lg[1], out[0][0][j], g['rec_0'][j]))
            assert parity.is_tie(lg[0], lg[1]), (j, lg)              # the same two-logit rule as bench.py's self-check


@pytest.mark.parametrize('res', [32, 64])
@pytest.mark.parametrize('model', ['p2s_max', 'p2s_vanilla'])
def test_three_clouds_one_stream_matches_reference(model, res):
    """all three abc_minimal clouds (34,693 / 59,979 / 86,648 points) as ONE dataset: the second and third shape
    depend on the stream position the previous ones left (data_loader.py:274-277)"""
    job = 'fulleval' if res == 32 else 'rec'
    g, meta = _golden(job, model, 'abc3', res)
    _compare(_run_dataset(model, 'abc3', res), g, meta)


@pytest.mark.parametrize('encoder', [0, 3, 4])
def test_sym_op_sum_matches_reference(encoder):
    """train --sym_op sum (reference source/points_to_surf_model.py:170-175, :211-214; set by no experiment script, the
    last constructor branch of the model): PointNetfeat pools with ``torch.sum(x, 2)``, its STN keeps the max-pool.  The
    engine sums in the MFMA accumulators' row layout with the padded rows of the last point tile masked, and adds the
    bias once per point.  Whole grid-32 shape against the unmodified reference -- the fp32 kernel, the split bf16x3
    kernel and the fp16 pair kernel (three code paths)."""
    import torch
    from points2surf_amd import engine, synth
    key = 'ref_rec_p2s_max_sum_testset_grid32'
    if not os.path.isfile(os.path.join(GOLDEN, key + '.npz')):
        missing_golden(key + ' not generated')
    ref = np.load(os.path.join(GOLDEN, key + '.npz'))['rec_0']
    w, cfg = synth.make_weights('p2s_max_sum')
    assert cfg['sym_op'] == 'sum'
    m = engine.Model(w, dict(cfg, encoder_bf16=encoder))
    cloud = engine.Cloud(np.load(os.path.join(FIX, '04_pts', _names('testset')[0] + '.xyz.npy')))
    sdf, _ = engine.infer_shape(m, cloud, engine.Rng(SEED), 32, 3, chunk=700)
    torch.cuda.synchronize()
    sdf = sdf.cpu().numpy()
    c = parity.compare_sdf(sdf, ref)
    print('p2s_max_sum (encoder mode %d): max|dSDF| %.3g, flips %d / %d, positive fraction %.2f'
          % (encoder, c['max_abs_dsdf'], c['flipped'].size, sdf.size, (sdf > 0).mean()))
    assert c['max_abs_dsdf'] < 1e-4 and c['flipped'].size == 0 and 0.2 < (sdf > 0).mean() < 0.8
    # the max model with the same weights answers differently: the pool really is a sum
    m2 = engine.Model(w, dict(cfg, sym_op='max'))
    other, _ = engine.infer_shape(m2, cloud, engine.Rng(SEED), 32, 3, q_end=64)
    assert np.abs(other.cpu().numpy() - sdf[:64]).max() > 1e-3
    m.close()
    m2.close()
    cloud.close()


def test_sym_op_sum_with_the_single_encoder_matches_reference():
    """sym_op='sum' together with --single_transformer 1 (one PointNetfeat over cat(patch, sub-sample), reference
    source/points_to_surf_model.py:253-263, :213-214): the engine runs the two point sets as two branches and the decoder's
    first layer reads the SUM of their two sum-pools (GemmArgs.a2_add); QSTN + weighted sub-sample on top."""
    import torch
    from points2surf_amd import engine, synth
    key = 'ref_rec_p2s_shared_encoder_sum_testset_grid32'
    if not os.path.isfile(os.path.join(GOLDEN, key + '.npz')):
        missing_golden(key + ' not generated')
    ref = np.load(os.path.join(GOLDEN, key + '.npz'))['rec_0']
    w, cfg = synth.make_weights('p2s_shared_encoder_sum')
    m = engine.Model(w, cfg)
    cloud = engine.Cloud(np.load(os.path.join(FIX, '04_pts', _names('testset')[0] + '.xyz.npy')))
    sdf, _ = engine.infer_shape(m, cloud, engine.Rng(SEED), 32, 3)
    torch.cuda.synchronize()
    c = parity.compare_sdf(sdf.cpu().numpy(), ref)
    print('p2s_shared_encoder_sum: max|dSDF| %.3g, flips %d / %d, positive fraction %.2f'
          % (c['max_abs_dsdf'], c['flipped'].size, ref.size, (ref > 0).mean()))
    assert c['max_abs_dsdf'] < 1e-4 and c['flipped'].size == 0 and 0.1 < (ref > 0).mean() < 0.9
    m.close()
    cloud.close()


@pytest.mark.parametrize('model', ['p2s_max', 'p2s_vanilla'])
def test_standin_clouds_match_reference(model):
    """non-fixture geometry -- two STAND-IN clouds (abc_minimal clouds under
    seeded random rotations, re-normalised to the unit cube like make_pc_dataset.py:20-36; other bounding boxes, cell
    grids, query grids and MT masks than the fixtures) as one dataset at 64^3, against the unmodified reference.  Pins
    the generator (synth.standin_cloud: the reference ran on the same arrays) and the path on them."""
    g, meta = _golden('rec', model, 'standin2', 64)
    out = _run_dataset(model, 'standin2', 64)
    assert [o[0].shape[0] for o in out] == [s['queries'] for s in meta['shapes']]
    _compare(out, g, meta)


@pytest.mark.parametrize('model', ['p2s_max', 'p2s_vanilla'])
def test_fixed_subsample_matches_reference(model):
    """train --fixed_subsample 1 (ablation branch, reference source/base/utils.py:210-211): rng.seed(42) before every
    query's draw -- full grid-32 shape against the unmodified reference, plus the ids against numpy"""
    import torch
    from points2surf_amd import engine, synth
    from oracle import p2s_oracle as O
    key = 'ref_rec_%s_testset_fixed_grid32' % model
    if not os.path.isfile(os.path.join(GOLDEN, key + '.npz')):
        missing_golden(key + ' not generated')
    g = np.load(os.path.join(GOLDEN, key + '.npz'))
    w, cfg = synth.make_weights(model)
    cfg = dict(cfg, fixed_subsample=True)
    m = engine.Model(w, cfg)
    pts = np.load(os.path.join(FIX, '04_pts', _names('testset')[0] + '.xyz.npy'))
    cloud = engine.Cloud(pts)
    rng = engine.Rng(SEED)
    sdf, q = engine.infer_shape(m, cloud, rng, 32, 3, chunk=700)
    torch.cuda.synchronize()
    d = np.abs(sdf.cpu().numpy() - g['rec_0'])
    flips = int((np.sign(sdf.cpu().numpy()) != np.sign(g['rec_0'])).sum())
    print('%s fixed_subsample: max|dSDF| %.3g, flips %d' % (model, d.max(), flips))
    assert d.max() < 1e-5 and flips == 0
    # stage-wise: ids and the generator state afterwards == numpy
    qs = q[:5]
    r2 = engine.Rng(7)
    if cfg.get('uniform_subsample'):
        ids, _ = r2.subsample_fixed(cloud, 1000, n_queries=5, want_pts=False)
        rs = np.random.RandomState(42)
        want = np.tile(rs.randint(0, pts.shape[0], 1000), (5, 1))
    else:
        ids, _ = r2.subsample_fixed(cloud, 1000, query_ms=qs, want_pts=False)
        want = []
        for qq in qs.cpu().numpy():
            rs = np.random.RandomState(42)
            want.append(rs.choice(pts.shape[0], size=1000, replace=False, p=O.dist_prob(pts, qq)))
        want = np.stack(want)
    assert np.array_equal(ids.cpu().numpy(), want)
    nxt = r2.subsample_uniform(cloud, 1, 64, want_pts=False)[0].cpu().numpy().reshape(-1)
    assert np.array_equal(nxt, rs.randint(0, pts.shape[0], 64))           # generator = seed(42) + the last query's draws


@pytest.mark.parametrize('model', ['p2s_max', 'p2s_vanilla'])
def test_small_cloud_shuffle_and_pad_matches_reference(model):
    """a cloud with fewer points (800) than the sub-sample size (1000): the reference shuffles shape.pts in place per
    query (under the kd-tree) and pads with zeros (source/base/utils.py:221-226) -- all 590 queries of the grid-16
    shape against the unmodified reference, chunked so that the permutation is carried across chunks"""
    import torch
    from points2surf_amd import engine, synth
    key = 'ref_rec_%s_small800_grid16' % model
    if not os.path.isfile(os.path.join(GOLDEN, key + '.npz')):
        missing_golden(key + ' not generated')
    g = np.load(os.path.join(GOLDEN, key + '.npz'))
    pts = np.load(os.path.join(GOLDEN, 'small800.xyz.npy'))
    w, cfg = synth.make_weights(model)
    m = engine.Model(w, cfg)
    cloud = engine.Cloud(pts)
    rng = engine.Rng(SEED)
    sdf, q = engine.infer_shape(m, cloud, rng, 16, 3, chunk=128)
    torch.cuda.synchronize()
    sdf = sdf.cpu().numpy()
    d = np.abs(sdf - g['rec_0'])
    flips = int((np.sign(sdf) != np.sign(g['rec_0'])).sum())
    print('%s small cloud: max|dSDF| %.3g, flips %d / %d' % (model, d.max(), flips, sdf.size))
    # grid 16: patch radii (the SDF scale) are ~4x those of grid 64 -> the north_star's 1e-4 bound, not the 1e-5 the
    # finer grids meet
    assert sdf.shape == g['rec_0'].shape and d.max() < 1e-4 and flips == 0
    # the generator and the ids against numpy's legacy shuffle (fresh handles)
    cloud2 = engine.Cloud(pts)
    r2 = engine.Rng(11)
    ids, sub = r2.subsample_uniform(cloud2, 3, 1000)
    rs = np.random.RandomState(11)
    cur = pts.copy()
    for i in range(3):
        rs.shuffle(cur)
        assert np.array_equal(sub[i, :800].cpu().numpy(), cur) and float(sub[i, 800:].abs().max()) == 0.0
        assert (ids[i, 800:] == -1).all()
    mt, pos = r2.get_state()
    st = rs.get_state()
    assert np.array_equal(mt, st[1]) and pos == st[2]


@pytest.mark.parametrize('model', ['p2s_uniform', 'p2s_no_qstn', 'p2s_small_kNN', 'p2s_large_kNN', 'p2s_regression',
                                   'p2s_shared_encoder', 'p2s_max_no_feat_stn'])
def test_ablation_models_match_reference(model):
    """the paper's ablation models whose branches the engine implements (reference experiments/train_p2s_*.sh): QSTN
    inside feat_global (sees the sub-sample only; its rotation also turns the patch -- source/points_to_surf_model.py
    :283-284, :337-339), no QSTN with the weighted sub-sample, 75- and 1200-point patches, the regression model (ONE
    output: the signed distance, sdf_nn.py:6-8) and the single encoder over cat(patch, sub-sample) (--single_transformer
    1, :253-263, :320-323); p2s_max_no_feat_stn = --use_feat_stn 0 (set by no script; the engine runs its usual path with
    an all-zero STN whose output is exactly the identity).  Whole grid-32 shape against the unmodified reference."""
    import torch
    from points2surf_amd import engine, synth
    key = 'ref_rec_%s_testset_grid32' % model
    if not os.path.isfile(os.path.join(GOLDEN, key + '.npz')):
        missing_golden(key + ' not generated')
    g = np.load(os.path.join(GOLDEN, key + '.npz'))
    w, cfg = synth.make_weights(model)
    m = engine.Model(w, cfg)
    assert m.points_per_patch == cfg['points_per_patch']
    cloud = engine.Cloud(np.load(os.path.join(FIX, '04_pts', _names('testset')[0] + '.xyz.npy')))
    sdf, _ = engine.infer_shape(m, cloud, engine.Rng(SEED), 32, 3, chunk=1000)
    torch.cuda.synchronize()
    sdf = sdf.cpu().numpy()
    d = np.abs(sdf - g['rec_0'])
    flips = int((np.sign(sdf) != np.sign(g['rec_0'])).sum())
    print('%s: max|dSDF| %.3g, flips %d / %d, positive fraction %.2f' % (model, d.max(), flips, sdf.size, (sdf > 0).mean()))
    assert d.max() < 1e-4 and flips == 0 and 0.02 < (sdf > 0).mean() < 0.98
