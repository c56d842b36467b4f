"""Row f-1 on the GPU: sign propagation volume, bit-exact vs the reference golden and vs the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('model', ['p2s_max', 'p2s_vanilla'])
@pytest.mark.parametrize('sigma,thr', [(5, 13), (3, 5), (4, 9.5), (2, 3), (6, 40)])   # odd and EVEN kernels
def test_sdf_volume_matches_reference_golden(golden_dir, model, sigma, thr):
    from points2surf_amd import engine
    g = np.load(os.path.join(golden_dir, 'ref_volume_grid32.npz'))
    q = np.load(os.path.join(golden_dir, 'query_grid_32_3.npy'))
    sdf = np.load(os.path.join(golden_dir, 'ref_%s_grid32.npz' % model))['sdf_full']
    vol, iters = engine.sdf_volume(q, sdf, 32, sigma, thr)
    assert iters >= 1
    assert np.array_equal(vol.cpu().numpy(), g['%s_s%d_t%g' % (model, sigma, thr)])   # bit-exact


@pytest.mark.parametrize('res,sigma,thr', [(64, 5, 13.0), (48, 4, 9.5), (96, 5, 13.0)])
def test_sdf_volume_matches_oracle_analytic_shape(res, sigma, thr):
    """larger grids: analytic noisy sphere SDF on its own query grid (the CPU probe of SURVEY App. B)"""
    from oracle import p2s_oracle as O
    from points2surf_amd import engine, synth
    pts = synth.make_cloud(20000, seed=3, kind='sphere')
    q, _ = O.query_grid(pts, res, 3)
    rng = np.random.default_rng(0)
    d = (0.5 - np.linalg.norm(q, axis=1)).astype(np.float32)          # positive inside (reference convention)
    d += (0.002 * rng.standard_normal(d.shape)).astype(np.float32)
    ref, it_ref = None, None
    vol0 = np.zeros((res, res, res))
    vol0 = O.add_samples_to_volume(vol0, q, d)
    ref, it_ref = O.propagate_sign(vol0, sigma, thr, return_iters=True)
    ref = np.clip(ref, -1.0, 1.0)
    vol, iters = engine.sdf_volume(q, d, res, sigma, thr)
    assert iters == it_ref
    assert np.array_equal(vol.cpu().numpy().astype(np.float64), ref)
    # size-independent property: the result is a fixed point of one more application on its own samples' signs
    assert np.all(vol.cpu().numpy()[0] == -1.0) and np.all(vol.cpu().numpy()[:, :, -1] == -1.0)


def test_sdf_volume_errors():
    from points2surf_amd import engine, _lib
    q = np.array([[0.0, 0.0, 0.0], [1.5, 0.0, 0.0]], dtype=np.float32)
    with pytest.raises(_lib.P2SError):
        engine.sdf_volume(q, np.array([0.1, 0.2], dtype=np.float32), 32, 5, 13)
    # all-unknown volume terminates immediately with borders only
    vol, iters = engine.sdf_volume(np.zeros((0, 3), np.float32), np.zeros((0,), np.float32), 16, 5, 13)
    v = vol.cpu().numpy()
    assert iters == 1 and (v[1:-1, 1:-1, 1:-1] == 0).all() and (v[0] == -1).all()


@pytest.mark.parametrize('res,sigma,thr', [(128, 5, 13.0), (80, 3, 5.0), (64, 4, 9.5), (32, 2, 3.0), (64, 1, 1.0)])
def test_fused_sweeps_equal_the_three_pass_path(res, sigma, thr, monkeypatch):
    """the fused LDS-tiled sweep kernel with device-side termination (default) against the separable three-pass
    path with a host decision per sweep (P2S_VOLUME_GENERIC): same volume bit for bit, same number of sweeps --
    including partial tiles (res not a multiple of the 64-voxel tile) and even kernels"""
    import torch
    from points2surf_amd import engine, synth
    pts = synth.make_cloud(30000, seed=11)
    cloud = engine.Cloud(pts)
    q = cloud.query_grid(res, 3)
    qn = q.cpu().numpy()
    d = (0.33 - np.linalg.norm(qn, axis=1)).astype(np.float32)
    d += (0.004 * np.random.default_rng(1).standard_normal(d.shape)).astype(np.float32)
    d[::97] = 0.0                                   # exact zeros among the samples: "unknown initially" voxels
    dd = torch.from_numpy(d).cuda()
    vol_f, it_f = engine.sdf_volume(q, dd, res, sigma, thr)          # verdict through the host-mapped mailbox
    monkeypatch.setenv('P2S_VOLUME_NO_MAILBOX', '1')                 # ... and copied out behind batches of 16 sweeps: same result
    vol_b, it_b = engine.sdf_volume(q, dd, res, sigma, thr)
    assert it_f == it_b and torch.equal(vol_f, vol_b)
    monkeypatch.setenv('P2S_VOLUME_GENERIC', '1')
    vol_g, it_g = engine.sdf_volume(q, dd, res, sigma, thr)
    assert it_f == it_g, (it_f, it_g)
    assert torch.equal(vol_f, vol_g)
