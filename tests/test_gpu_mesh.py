"""GPU: row f-2 -- iso-surface extraction on the device (p2s_marching_cubes) against the CPU restatement
(oracle/mc_oracle.py): identical vertex and face arrays (deterministic emission order), identical counts on the
reference-generated sign volumes, and the invariants that need no scikit-image (closed, oriented 2-manifold) at the
sizes of the benchmark.  scikit-image's own Lewiner counts are unpinned (absent offline) -- see DESIGN.md."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mc_oracle as M   # noqa: E402


def _sphere_volume(res, r=0.6, noise=0.0, seed=0):
    g = (np.arange(res) + 0.5) / res * 2 - 1
    x, y, z = np.meshgrid(g, g, g, indexing='ij')
    v = (r - np.sqrt(x * x + y * y + z * z)).astype(np.float32)
    if noise:
        v += (noise * np.random.default_rng(seed).standard_normal(v.shape)).astype(np.float32)
    v = np.clip(v, -1, 1)
    v[0], v[-1], v[:, 0], v[:, -1], v[:, :, 0], v[:, :, -1] = -1, -1, -1, -1, -1, -1
    return v


def _device_mc(vol_np, **kw):
    import torch
    from points2surf_amd import engine
    v, f, inv = engine.marching_cubes(torch.from_numpy(np.ascontiguousarray(vol_np, dtype=np.float32)).cuda(), **kw)
    torch.cuda.synchronize()
    return v.cpu().numpy(), f.cpu().numpy(), inv


@pytest.mark.parametrize('res,noise,seed', [(16, 0.0, 0), (32, 0.02, 1), (24, 0.3, 2), (33, 0.3, 3), (48, 0.1, 4)])
def test_device_mesh_equals_oracle(res, noise, seed):
    vol = _sphere_volume(res, noise=noise, seed=seed)
    for model_space in (True, False):
        v_ref, f_ref, inv_ref = M.marching_cubes(vol, model_space=model_space)
        v, f, inv = _device_mc(vol, model_space=model_space)
        assert v.shape == v_ref.shape and f.shape == f_ref.shape, (v.shape, v_ref.shape, f.shape, f_ref.shape)
        assert np.array_equal(f, f_ref) and inv == inv_ref
        assert np.array_equal(v, v_ref)                      # float64 interpolation, one rounding: bit-equal
    chk = M.mesh_checks(v, f)
    assert chk['closed'] and chk['oriented'] and chk['unused_vertices'] == 0, chk


@pytest.mark.parametrize('model', ['p2s_max', 'p2s_vanilla'])
def test_mesh_counts_on_reference_volumes(golden_dir, model):
    """the volumes the UNMODIFIED reference's add_samples_to_volume + propagate_sign produced (ref_volume_grid32.npz):
    device counts == oracle counts; closed oriented manifold"""
    g = np.load(os.path.join(golden_dir, 'ref_volume_grid32.npz'))
    vol = g['%s_s5_t13' % model]
    v_ref, f_ref, _ = M.marching_cubes(vol)
    v, f, _ = _device_mc(vol)
    print('%s grid32: %d vertices, %d faces' % (model, v.shape[0], f.shape[0]))
    assert (v.shape[0], f.shape[0]) == (v_ref.shape[0], f_ref.shape[0]) and f.shape[0] > 100
    assert np.array_equal(f, f_ref) and np.array_equal(v, v_ref)
    chk = M.mesh_checks(v, f)
    assert chk['closed'] and chk['oriented'], chk


def test_empty_and_one_sided_volumes():
    v, f, _ = _device_mc(-np.ones((16, 16, 16), np.float32))
    assert v.shape == (0, 3) and f.shape == (0, 3)
    v, f, _ = _device_mc(np.ones((16, 16, 16), np.float32))
    assert v.shape == (0, 3) and f.shape == (0, 3)
    vol = -np.ones((8, 8, 8), np.float32)
    vol[3:5, 3:5, 3:5] = 1.0
    vol[3, 3, 3] = 0.0                                        # exact zero = outside
    v, f, _ = _device_mc(vol, model_space=False)
    v_ref, f_ref, _ = M.marching_cubes(vol, model_space=False)
    assert np.array_equal(v, v_ref) and np.array_equal(f, f_ref)


@pytest.mark.parametrize('res', [128, 256])
def test_cloud_to_mesh_on_device_is_a_closed_manifold(res, fixture_cloud):
    """whole consumer chain at the benchmark sizes: inference -> sign propagation -> iso-surface, all on the device;
    size-independent properties: every edge in exactly two faces, consistent orientation, positive volume after the
    inversion fix, vertices inside the unit cube the clouds are normalised to"""
    import torch
    from points2surf_amd import engine, synth
    w, cfg = synth.make_weights('p2s_max')
    model = engine.Model(w, cfg)
    cloud = engine.Cloud(fixture_cloud)
    sdf, q = engine.infer_shape(model, cloud, engine.Rng(40938661), res, 3)
    vol, iters = engine.sdf_volume(q, sdf, res, 5, 13.0)
    v, f, inv = engine.marching_cubes(vol)
    torch.cuda.synchronize()
    v, f = v.cpu().numpy(), f.cpu().numpy()
    chk = M.mesh_checks(v, f)
    print('grid %d: %d sweeps, %d vertices, %d faces, %d components, euler %d, inverted %s'
          % (res, iters, chk['V'], chk['F'], chk['components'], chk['euler'], inv))
    assert chk['closed'] and chk['oriented'] and chk['unused_vertices'] == 0 and chk['F'] > 1000
    vv = v.astype(np.float64)
    vol6 = np.einsum('ij,ij->i', vv[f[:, 0]], np.cross(vv[f[:, 1]], vv[f[:, 2]])).sum()
    assert vol6 > 0 and np.abs(v).max() < 1.0
