"""GPU: row f-2 -- iso-surface extraction on the device (p2s_marching_cubes) = scikit-image's
``marching_cubes_lewiner(volume, 0)`` (reference source/sdf.py:213-215):
  * against the CPU restatement oracle/lewiner_mc.c (itself pinned to scikit-image 0.18.3, tests/test_lewiner_oracle.py):
    identical vertex and face ARRAYS (same deterministic emission order);
  * against scikit-image directly: counts and canonical-mesh hashes of the goldens (tests/golden/meta_mesh.json) on the
    volumes of the reference, 32^3 ... 256^3, both models -- the device builds those volumes itself from the reference's
    SDF (p2s_sdf_volume, bit-identical, checked through the volume hash)."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mc_oracle as M, lewiner_oracle as LO   # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


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


def _meta(name):
    with open(os.path.join(GOLDEN, 'meta_mesh.json')) as f:
        return json.load(f)[name]


def _assert_is_the_scikit_image_mesh(name, vol, verts_index_space, faces):
    m = _meta(name)
    assert hashlib.sha256(np.ascontiguousarray(vol, dtype=np.float32).tobytes()).hexdigest() == m['volume_sha256']
    assert (verts_index_space.shape[0], faces.shape[0]) == (m['n_verts'], m['n_faces'])
    cv, cf = LO.canonical_mesh(verts_index_space, faces)
    assert hashlib.sha256(cv.tobytes()).hexdigest() == m['canonical_verts_sha256']
    assert hashlib.sha256(cf.tobytes()).hexdigest() == m['canonical_faces_sha256']


@pytest.mark.parametrize('res,noise,seed', [(16, 0.0, 0), (32, 0.02, 1), (24, 0.3, 2), (33, 0.3, 3), (48, 0.1, 4), (40, 2.0, 5)])
def test_device_mesh_equals_oracle(res, noise, seed):
    """noise 2.0: mostly saturated +-1 values -> exact ties of the face decider, every ambiguous case incl. tunnels"""
    vol = _sphere_volume(res, noise=noise, seed=seed)
    for model_space in (True, False):
        v_ref, f_ref, inv_ref = M.marching_cubes(vol, model_space=model_space)
        v, f, inv = _device_mc(vol, model_space=model_space)
        assert v.shape == v_ref.shape and f.shape == f_ref.shape, (v.shape, v_ref.shape, f.shape, f_ref.shape)
        assert np.array_equal(f, f_ref) and inv == inv_ref
        assert np.array_equal(v, v_ref)                      # same float64 formula, one rounding: bit-equal
    if noise <= 0.02:
        chk = M.mesh_checks(v, f)
        assert chk['closed'] and chk['oriented'] and chk['unused_vertices'] == 0, chk


@pytest.mark.parametrize('key', ['p2s_max_s5_t13', 'p2s_max_s2_t3', 'p2s_max_s6_t40', 'p2s_vanilla_s5_t13', 'p2s_vanilla_s3_t5',
                                 'p2s_vanilla_s4_t9.5'])
def test_reference_volumes_32_are_meshed_like_scikit_image(key):
    """volumes written by the UNMODIFIED reference (add_samples_to_volume + propagate_sign): the device mesh is
    scikit-image's mesh (arrays of the golden compared as sets of positions / oriented triangles)"""
    vol = np.load(os.path.join(GOLDEN, 'ref_volume_grid32.npz'))[key]
    name = 'ref32_' + key.replace('.', 'p')
    v, f, _ = _device_mc(vol, model_space=False, fix_inversion=False)
    g = np.load(os.path.join(GOLDEN, 'mesh_%s_skimage.npz' % name))
    ok, msg = LO.same_mesh(v, f, g['verts'], g['faces'])
    assert ok, msg
    _assert_is_the_scikit_image_mesh(name, vol, v, f)
    print('%s: %d vertices, %d faces == scikit-image' % (name, v.shape[0], f.shape[0]))


@pytest.mark.parametrize('model', ['p2s_max', 'p2s_vanilla'])
@pytest.mark.parametrize('res', [128, 256, 512])
def test_reference_sdf_to_mesh_counts_equal_scikit_image(model, res, fixture_cloud):
    """BASELINE north_star: 'bit-identical mesh vertex/face counts'.  The reference's full-grid SDF golden -> sign
    propagation ON THE DEVICE (volume hash = the volume scikit-image was given) -> iso-surface ON THE DEVICE: the counts
    and the whole canonical mesh equal what marching_cubes_lewiner returned for that volume."""
    import torch
    from points2surf_amd import engine
    sdf = np.load(os.path.join(GOLDEN, 'ref_rec_%s_testset_grid%d.npz' % (model, res)))['rec_0']
    cloud = engine.Cloud(fixture_cloud)
    q = cloud.query_grid(res, 3)
    vol, iters = engine.sdf_volume(q, torch.from_numpy(sdf).cuda(), res, 5, 13.0)
    v, f, inv = engine.marching_cubes(vol, model_space=False, fix_inversion=False)
    torch.cuda.synchronize()
    v, f = v.cpu().numpy(), f.cpu().numpy()
    _assert_is_the_scikit_image_mesh('%s_grid%d' % (model, res), vol.cpu().numpy(), v, f)
    print('%s %d^3: %d vertices, %d faces == scikit-image 0.18.3' % (model, res, v.shape[0], f.shape[0]))


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
def test_cloud_to_mesh_on_device(res, fixture_cloud):
    """whole consumer chain at the benchmark sizes: inference -> sign propagation -> iso-surface, all on the device, against
    the oracle on the same volume; positive volume after the inversion fix, vertices inside the unit cube"""
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
    v_ref, f_ref, inv_ref = M.marching_cubes(vol.cpu().numpy())
    assert np.array_equal(v, v_ref) and np.array_equal(f, f_ref) and inv == inv_ref
    chk = M.mesh_checks(v, f)
    print('grid %d: %d sweeps, %d vertices, %d faces, %d components, closed %s, inverted %s'
          % (res, iters, chk['V'], chk['F'], chk['components'], chk['closed'], inv))
    assert chk['unused_vertices'] == 0 and chk['F'] > 1000
    vv = v.astype(np.float64)
    vol6 = np.einsum('ij,ij->i', vv[f[:, 0]], np.cross(vv[f[:, 1]], vv[f[:, 2]])).sum()
    assert vol6 > 0 and np.abs(v).max() < 1.0
