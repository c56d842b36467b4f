"""Host logic: BN folding + MFMA fragment packing + blob offsets, checked by emulating the engine's
algebra in numpy (tests only) against the oracle."""
import numpy as np
import pytest

from oracle import p2s_oracle as O
from points2surf_amd import synth, weights


def test_pack_roundtrip_and_fragment_semantics():
    rng = np.random.default_rng(0)
    W = rng.standard_normal((128, 96)).astype(np.float32)
    p = weights.pack_b(W)
    assert np.array_equal(weights.unpack_b(p, 128, 96), W)
    # the kernels read: packed[((nt*KG + kg)*64 + lane)*4 + t] == W[8kg + 4(lane>>5) + t][32nt + (lane&31)]
    KG = 128 // 8
    for nt, kg, lane, t in [(0, 0, 0, 0), (2, 15, 63, 3), (1, 7, 33, 2), (0, 3, 31, 1)]:
        assert p[((nt * KG + kg) * 64 + lane) * 4 + t] == W[8 * kg + 4 * (lane >> 5) + t][32 * nt + (lane & 31)]


def _gemm(x, blob, off_w, off_b, K, N, relu):
    W = weights.unpack_b(blob[off_w:off_w + K * N], K, N)
    y = x @ W + blob[off_b:off_b + N]
    return np.maximum(y, 0) if relu else y


def _emulate_encoder(blob, o, x):
    """what p2s_chain_kernel / p2s_gemm_kernel / p2s_fold_kernel compute, for x [B,P,3]"""
    w0a = blob[o.w0a:o.w0a + 192].reshape(3, 64)
    h = np.maximum(x @ w0a + blob[o.b0a:o.b0a + 64], 0)
    h0 = _gemm(h, blob, o.w0b, o.b0b, 64, 64, True)
    s = _gemm(h0, blob, o.s1, o.sb1, 64, 64, True)
    s = _gemm(s, blob, o.s2, o.sb2, 64, 128, True)
    W3 = weights.unpack_b(blob[o.s3:o.s3 + 128 * 1024], 128, 1024)
    g = np.maximum((s @ W3).max(axis=1) + blob[o.sb3:o.sb3 + 1024], 0)      # pooled affine epilogue
    g = _gemm(g, blob, o.sf1, o.sfb1, 1024, 512, True)
    g = _gemm(g, blob, o.sf2, o.sfb2, 512, 256, True)
    T = _gemm(g, blob, o.sf3, o.sfb3, 256, 4096, False).reshape(-1, 64, 64)  # identity already in the bias
    W1 = weights.unpack_b(blob[o.m1t:o.m1t + 4096], 64, 64)                # [k=in i][n=out o] = W1f[o][i]
    W1p = np.einsum('bij,io->bjo', T, W1)                                    # W1'[k=j][n=o] = sum_i T[i][j] W1f[o][i]
    y = np.maximum(np.einsum('bpj,bjo->bpo', h0, W1p) + blob[o.mb1:o.mb1 + 64], 0)
    y = _gemm(y, blob, o.m2, o.mb2, 64, 128, True)
    M3 = weights.unpack_b(blob[o.m3:o.m3 + 128 * 1024], 128, 1024)
    return (y @ M3).max(axis=1) + blob[o.mb3:o.mb3 + 1024]


@pytest.mark.parametrize('model', ['p2s_max', 'p2s_vanilla'])
def test_engine_algebra_matches_oracle(model, fixture_cloud):
    w, cfg = synth.make_weights(model)
    blob, offs, mc = weights.build_blob(w, cfg)
    assert blob.dtype == np.float32 and mc.use_point_stn == int(cfg['use_point_stn'])
    B = 3
    q, _ = O.query_grid(fixture_cloud, 32, 3)
    q = q[100:100 + B]
    ids = O.knn_ids(fixture_cloud, q, 300)
    r, ps = O.patch_radius_and_ps(fixture_cloud, ids, q)
    sub = fixture_cloud[np.random.default_rng(1).integers(0, fixture_cloud.shape[0], (B, 1000))]
    ref_logits, ref_fl, ref_fg = O.model_forward(w, cfg, ps, sub, q, return_feats=True)

    patch, shape = ps, sub - q[:, None, :]
    if cfg['use_point_stn']:
        qo = offs.qstn
        both = np.concatenate([patch, shape], axis=1)
        c1 = blob[qo.c1:qo.c1 + 192].reshape(3, 64)
        h = np.maximum(both @ c1 + blob[qo.cb1:qo.cb1 + 64], 0)
        h = _gemm(h, blob, qo.c2, qo.cb2, 64, 128, True)
        W3 = weights.unpack_b(blob[qo.c3:qo.c3 + 128 * 1024], 128, 1024)
        g = np.maximum((h @ W3).max(axis=1) + blob[qo.cb3:qo.cb3 + 1024], 0)
        g = _gemm(g, blob, qo.f1, qo.fb1, 1024, 512, True)
        g = _gemm(g, blob, qo.f2, qo.fb2, 512, 256, True)
        quat = g @ blob[qo.f3:qo.f3 + 1024].reshape(256, 4) + blob[qo.fb3:qo.fb3 + 4]
        R = O.quat_to_rotmat(quat)
        patch = np.einsum('bij,bpj->bpi', R, patch)
        shape = np.einsum('bij,bpj->bpi', R, shape)
    fl = _emulate_encoder(blob, offs.enc[0], patch)
    fg = _emulate_encoder(blob, offs.enc[1], shape)
    assert np.abs(fl - ref_fl).max() < 1e-4 * np.abs(ref_fl).max()
    assert np.abs(fg - ref_fg).max() < 1e-4 * np.abs(ref_fg).max()
    d = np.concatenate([_gemm(fl, blob, offs.d1l, offs.db1l, 1024, 512, True),
                        _gemm(fg, blob, offs.d1g, offs.db1g, 1024, 512, True)], axis=1)
    d = _gemm(d, blob, offs.d2, offs.db2, 1024, 256, True)
    d = _gemm(d, blob, offs.d3, offs.db3, 256, 128, True)
    logits = d @ blob[offs.d4:offs.d4 + 256].reshape(128, 2) + blob[offs.db4:offs.db4 + 2]
    assert np.abs(logits - ref_logits).max() < 1e-4


def test_unsupported_configs_raise():
    w, cfg = synth.make_weights('p2s_max')
    for bad in (dict(sym_op='median'), dict(net_size=512), dict(output_dim=3)):
        c = dict(cfg)
        c.update(bad)
        with pytest.raises(ValueError):
            weights.build_blob(w, c)
    # sym_op='sum' (reference source/points_to_surf_model.py:172-173) is built, also together with single_transformer
    _, _, mc = weights.build_blob(w, dict(cfg, sym_op='sum'))
    assert mc.sym_sum == 1 and weights.build_blob(w, cfg)[2].sym_sum == 0
    ws, cs = synth.make_weights('p2s_shared_encoder_sum')
    mcs = weights.build_blob(ws, cs)[2]
    assert mcs.sym_sum == 1 and mcs.single_transformer == 1


def test_no_feat_stn_becomes_an_exact_identity_transform():
    """train --use_feat_stn 0: the blob holds an all-zero feature STN whose fc3 bias is the identity"""
    w, cfg = synth.make_weights('p2s_max_no_feat_stn')
    blob, offs, mc = weights.build_blob(w, cfg)
    for e in range(2):
        o = offs.enc[e]
        assert np.array_equal(blob[o.sfb3:o.sfb3 + 4096].reshape(64, 64), np.eye(64, dtype=np.float32))
        assert not blob[o.sf3:o.sf3 + 256 * 4096].any() and not blob[o.s3:o.s3 + 128 * 1024].any()


def test_qstn_weights_come_from_the_right_module():
    """shared transformer: model.point_stn; otherwise the QSTN of feat_global (reference points_to_surf_model.py
    :267-269, :283-284) -- the ablation models p2s_uniform / p2s_*_kNN; patch sizes other than 300 reach the cfg"""
    for name, key in (('p2s_vanilla', 'point_stn.fc3.bias'), ('p2s_uniform', 'feat_global.stn1.fc3.bias')):
        w, cfg = synth.make_weights(name)
        assert key in w and ('point_stn.fc3.bias' in w) == (name == 'p2s_vanilla')
        blob, offs, mc = weights.build_blob(w, cfg)
        assert mc.use_point_stn == 1 and mc.shared_transformer == int(name == 'p2s_vanilla')
        b3 = blob[offs.qstn.fb3:offs.qstn.fb3 + 4]
        assert np.allclose(b3, np.asarray(w[key], dtype=np.float32) + np.array([1, 0, 0, 0], np.float32))
    for name, k in (('p2s_small_kNN', 75), ('p2s_large_kNN', 1200), ('p2s_no_qstn', 300)):
        w, cfg = synth.make_weights(name)
        _, _, mc = weights.build_blob(w, cfg)
        assert mc.points_per_patch == k and mc.weighted_subsample == 1


def test_regression_and_single_transformer_blobs():
    """p2s_regression: fc4 has ONE row (output imp_surf); p2s_shared_encoder: both encoder slots hold feat_local_global,
    the QSTN is its stn1, the decoder's two fc1 blocks are the column halves of fc1_local_global (BN folded)"""
    w, cfg = synth.make_weights('p2s_regression')
    blob, offs, mc = weights.build_blob(w, cfg)
    assert mc.output_dim == 1 and mc.single_transformer == 0
    assert np.array_equal(blob[offs.d4:offs.d4 + 128], np.asarray(w['fc4.weight'], np.float32).reshape(-1))
    w, cfg = synth.make_weights('p2s_shared_encoder')
    blob, offs, mc = weights.build_blob(w, cfg)
    assert mc.single_transformer == 1 and mc.use_point_stn == 1 and mc.output_dim == 2
    e0, e1 = offs.enc[0], offs.enc[1]
    assert np.array_equal(blob[e0.m3:e0.m3 + 128 * 1024], blob[e1.m3:e1.m3 + 128 * 1024])
    W1, b1 = weights.fold_affine({k: np.asarray(v) for k, v in w.items()}, 'fc1_local_global', 'bn1_local_global')
    assert np.array_equal(weights.unpack_b(blob[offs.d1g:offs.d1g + 1024 * 512], 1024, 512), W1[512:].T.astype(np.float32))
    assert np.array_equal(blob[offs.db1l:offs.db1l + 512], b1[:512].astype(np.float32))
    assert np.allclose(blob[offs.qstn.fb3:offs.qstn.fb3 + 4],
                       np.asarray(w['feat_local_global.stn1.fc3.bias'], np.float32) + np.array([1, 0, 0, 0], np.float32))


def test_model_cfg_carries_the_sub_sample_mode():
    """p2s_max draws randint ids (--uniform_subsample 1), p2s_vanilla the distance-weighted choice"""
    from points2surf_amd import synth, weights
    for name, weighted in (('p2s_max', 0), ('p2s_vanilla', 1)):
        w, cfg = synth.make_weights(name)
        _, _, mc = weights.build_blob(w, cfg)
        assert mc.weighted_subsample == weighted
        assert mc.use_point_stn == weighted


def test_bf16_fragment_repacking_index_math():
    """mirror of p2s_pack_bf16_kernel (p2s_chain_bf16.hip): fp32 B fragments [N/32][K/8][2][32][4] -> bf16 fragments
    [N/32][K/16][64 lanes][8] with k = 16 kb + 8 (lane >> 5) + t, n = 32 nt + (lane & 31)"""
    from points2surf_amd import weights
    rng = np.random.default_rng(3)
    for K, N in ((64, 64), (64, 128), (128, 1024)):
        W = rng.standard_normal((K, N)).astype(np.float32)
        src = weights.pack_b(W)
        e = np.arange(K * N)
        t = e & 7
        lane = (e >> 3) & 63
        r = e >> 9
        kb, nt = r % (K // 16), r // (K // 16)
        k = 16 * kb + 8 * (lane >> 5) + t
        j = lane & 31
        kg, kk, ts = k >> 3, (k >> 2) & 1, k & 3
        si = (((nt * (K // 8) + kg) * 2 + kk) * 32 + j) * 4 + ts
        dst = src[si]                                             # what the kernel writes (before rounding)
        assert np.array_equal(dst, W[k, 32 * nt + j])
        # every (k, n) exactly once
        assert np.unique(k * N + 32 * nt + j).size == K * N
