"""GPU: the fp16-pair encoder (cfg.encoder_bf16 = 4, BASELINE configs[3]'s fast mode) is safe for an ARBITRARY checkpoint --
VERDICT r5 item 3.  fp16 ends at 65504; r05 poisoned a query whose activation passed 6e4 and failed the whole call.  Now
the 16-bit kernels flag such queries, their inputs are put aside per chunk, and at the end of the same call they run
through the fp32 kernels (p2s_api.hip: p2s_model_fallback_finish); ``counters()['fallback_queries']`` says how many.

The checkpoints here compute the SAME function as the synthetic p2s_max / p2s_vanilla weights -- one BatchNorm channel is
scaled by 2^k and the next layer's weights for that channel by 2^-k (ReLU and max-pool are positively homogeneous, powers
of two are exact) -- but their activations in that channel are 2^k times larger: k is searched until a few per cent of the
queries leave the half range.  Two sites: a per-point layer of the main trunk (flag raised by the chain kernel when it
writes the activation to LDS as an fp16 pair) and the pooled STN feature (flag raised by the fp16-pair head GEMM)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 40938661


def _rescaled(w, site, c, s):
    w2 = {k: v.copy() for k, v in w.items()}
    if site == 'chain':          # feat_local main trunk: bn2 output channel c times s, conv3's input column c divided by s
        w2['feat_local.bn2.weight'][c] *= s
        w2['feat_local.bn2.bias'][c] *= s
        w2['feat_local.conv3.weight'][:, c, :] /= s
    elif site == 'first':        # feat_local first layer (fp32 VALU, its output goes to LDS as an fp16 pair): bn0a channel c
        w2['feat_local.bn0a.weight'][c] *= s
        w2['feat_local.bn0a.bias'][c] *= s
        w2['feat_local.conv0b.weight'][:, c, :] /= s
    else:                        # feat_global STN trunk: pooled (ReLU'd) feature c times s, fc1's input column c divided by s
        w2['feat_global.stn2.bn3.weight'][c] *= s
        w2['feat_global.stn2.bn3.bias'][c] *= s
        w2['feat_global.stn2.fc1.weight'][:, c] /= s
    return w2


def _run(engine, w, cfg, cloud, enc, res=32, chunk=0):
    import torch
    m = engine.Model(w, dict(cfg, encoder_bf16=enc))
    sdf, _ = engine.infer_shape(m, cloud, engine.Rng(SEED), res, 3, chunk=chunk)
    torch.cuda.synchronize()
    n = int(m.counters()['fallback_queries'])
    m.close()
    return sdf.cpu().numpy(), n


@pytest.mark.parametrize('model,site', [('p2s_max', 'chain'), ('p2s_max', 'heads'), ('p2s_vanilla', 'chain')])
def test_queries_beyond_the_half_range_are_rerun_in_fp32(model, site, fixture_cloud, golden_dir):
    from points2surf_amd import engine, synth, parity, _lib
    w, cfg = synth.make_weights(model)
    cloud = engine.Cloud(fixture_cloud)
    base16, n0 = _run(engine, w, cfg, cloud, 4)
    assert n0 == 0                                         # the weights at hand: nothing is ever flagged
    nq = base16.shape[0]
    found = None
    for c in range(4):
        for k in range(6, 26):
            w2 = _rescaled(w, site, c, float(2 ** k))
            try:
                got, n = _run(engine, w2, cfg, cloud, 4)    # activations of any size: no error, no wrong value
            except _lib.P2SError as e:                      # ... until the WEIGHTS leave the half range: refused at creation
                assert 'does not fit the half range' in str(e)
                break
            if 0.002 * nq <= n <= 0.2 * nq:
                found = (c, k, n, got, w2)
                break
            if n > 0.2 * nq:
                break
        if found:
            break
    assert found, 'no (channel, scale) with 0.2 % .. 20 % of the queries beyond the half range'
    c, k, n, got, w2 = found
    ref32, n32 = _run(engine, w2, cfg, cloud, 0)           # the same checkpoint through the fp32 encoders
    assert n32 == 0
    cmp_ = parity.compare_sdf(got, ref32)
    same = int((got == ref32).sum())
    print('%s / %s: channel %d x 2^%d -> %d of %d queries re-run in fp32; max|dSDF| vs the fp32 encoders %.3g, flips %d, '
          'bit-identical values %d' % (model, site, c, k, n, nq, cmp_['max_abs_dsdf'], cmp_['flipped'].size, same))
    assert cmp_['max_abs_dsdf'] < 1e-4 and cmp_['flipped'].size == 0
    assert same >= n                                       # the re-run queries ARE the fp32 kernels' values
    # and the function is still the reference's: the golden of the unscaled weights
    g = np.load(os.path.join(golden_dir, 'ref_%s_grid32.npz' % model))['sdf_full']
    cg = parity.compare_sdf(got, g)
    assert cg['max_abs_dsdf'] < 1e-4 and cg['flipped'].size == 0
    # the same through several chunks (flagged queries of every chunk land at their own place of the call's output), and
    # through a query range that does not start at 0
    got7, n7 = _run(engine, w2, cfg, cloud, 4, chunk=700)
    assert n7 == n and np.array_equal(got7, got)
    import torch
    m = engine.Model(w2, dict(cfg, encoder_bf16=4))
    rng = engine.Rng(SEED)
    part, _ = engine.infer_shape(m, cloud, rng, 32, 3, q_begin=0, q_end=1000, chunk=300)
    rest, _ = engine.infer_shape(m, cloud, rng, 32, 3, q_begin=1000, q_end=nq, chunk=650)
    torch.cuda.synchronize()
    assert np.array_equal(np.concatenate([part.cpu().numpy(), rest.cpu().numpy()]), got)
    m.close()


def test_fallback_through_the_model_forward_boundary_and_its_capacity(fixture_cloud):
    """B2 (PointsToSurfModel.forward -> p2s_encode_decode): logits and SDF of flagged queries come from the fp32 kernels;
    more flagged queries in ONE call than the side buffers hold (16384) is still a loud error, never a wrong value"""
    import torch
    from points2surf_amd import engine, synth, _lib
    w, cfg = synth.make_weights('p2s_max')
    # a WEIGHT beyond the half range cannot be repaired per query: the model refuses the mode when it is created
    with pytest.raises(_lib.P2SError, match='does not fit the half range'):
        engine.Model(_rescaled(w, 'chain', 0, float(2 ** 30)), dict(cfg, encoder_bf16=4))
    w2 = w
    for c in range(8):                                     # first-layer channels times 2^24: every query leaves the half range
        w2 = _rescaled(w2, 'first', c, float(2 ** 24))
    m16 = engine.Model(w2, dict(cfg, encoder_bf16=4))
    m32 = engine.Model(w2, cfg)
    cloud = engine.Cloud(fixture_cloud)
    q = cloud.query_grid(32, 3)[:600].contiguous()
    rng = engine.Rng(SEED)
    _, sub = rng.subsample_uniform(cloud, 600, 1000)
    _, patch, rad = cloud.knn_patch(q, 300, want_ids=False)
    lg16, sdf16 = m16.forward(patch, sub, q, rad, want_sdf=True)
    lg32, sdf32 = m32.forward(patch, sub, q, rad, want_sdf=True)
    torch.cuda.synchronize()
    assert int(m16.counters()['fallback_queries']) == 600
    assert torch.equal(lg16, lg32) and torch.equal(sdf16, sdf32)
    big = 17000
    with pytest.raises(_lib.P2SError, match='16384'):
        m16.forward(patch[:1].expand(big, -1, -1).contiguous(), sub[:1].expand(big, -1, -1).contiguous(),
                    q[:1].expand(big, -1).contiguous(), rad[:1].expand(big).contiguous(), want_sdf=True)
    # the model is usable afterwards
    lg, _ = m16.forward(patch[:8], sub[:8], q[:8], rad[:8], want_sdf=True)
    torch.cuda.synchronize()
    assert torch.equal(lg, lg32[:8])
    m16.close()
    m32.close()


def test_logits_capture_of_the_pipeline(fixture_cloud):
    """p2s_model_capture_logits: the decoder's raw logits of a pipeline call (what the drop-in's tie report reads) equal the
    logits of the stage-wise path on the same inputs, and sign(SDF) = (sign logit >= 0)"""
    import torch
    from points2surf_amd import engine, synth
    w, cfg = synth.make_weights('p2s_max')
    m = engine.Model(w, cfg)
    cloud = engine.Cloud(fixture_cloud)
    sdf, q, lg = engine.infer_shape(m, cloud, engine.Rng(SEED), 32, 3, chunk=700, want_logits=True)
    torch.cuda.synchronize()
    assert lg.shape == (sdf.shape[0], 2)
    assert torch.equal(lg[:, 1] >= 0, sdf > 0)
    rng = engine.Rng(SEED)
    _, sub = rng.subsample_uniform(cloud, 64, 1000)
    _, patch, rad = cloud.knn_patch(q[:64], 300, want_ids=False)
    lg2, _ = m.forward(patch, sub, q[:64].contiguous())
    assert float((lg2 - lg[:64]).abs().max()) < 1e-5
    # one-shot: the next call does not write into the old buffer
    keep = lg.clone()
    engine.infer_shape(m, cloud, engine.Rng(SEED + 1), 32, 3)
    torch.cuda.synchronize()
    assert torch.equal(keep, lg)
    m.close()


def test_fallback_in_the_gt_query_pass_with_rotations(fixture_cloud):
    """the evaluation pass (p2s_infer_queries with the per-query random rotation, reference source/data_loader.py:381-393):
    the flagged queries' ROTATED inputs are what is put aside and re-run"""
    import torch
    from points2surf_amd import engine, synth
    w, cfg = synth.make_weights('p2s_max')
    w2 = w
    for c in range(2):
        w2 = _rescaled(w2, 'first', c, float(2 ** 22))
    cloud = engine.Cloud(fixture_cloud)
    g = np.random.default_rng(3)
    q = torch.from_numpy((fixture_cloud[g.integers(0, fixture_cloud.shape[0], 900)] + g.normal(0, 0.02, (900, 3))).astype(np.float32)).cuda()
    out = {}
    for enc in (4, 0):
        m = engine.Model(w2, dict(cfg, encoder_bf16=enc))
        sdf = engine.infer_queries(m, cloud, engine.Rng(SEED), engine.Rng(SEED), q, chunk=256)
        torch.cuda.synchronize()
        out[enc] = (sdf.cpu().numpy(), int(m.counters()['fallback_queries']))
        m.close()
    assert out[0][1] == 0 and out[4][1] > 100
    same = int((out[4][0] == out[0][0]).sum())
    assert same >= out[4][1] and float(np.abs(out[4][0] - out[0][0]).max()) < 1e-4
