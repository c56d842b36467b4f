"""CPU: the iso-surface oracle (oracle/mc_oracle.py) -- invariants that need no scikit-image -- and the library's
generated triangulation table (host part of p2s_mesh.hip, no GPU) against the oracle's independent generator."""
import ctypes

import numpy as np
import pytest

from oracle import mc_oracle as M


def _sphere_volume(res, r=0.6, noise=0.0, seed=0):
    g = (np.arange(res) + 0.5) / res * 2 - 1
    x, y, z = np.meshgrid(g, g, g, indexing='ij')
    v = (r - np.sqrt(x * x + y * y + z * z)).astype(np.float32)
    if noise:
        v += (noise * np.random.default_rng(seed).standard_normal(v.shape)).astype(np.float32)
    v = np.clip(v, -1, 1)
    v[0], v[-1], v[:, 0], v[:, -1], v[:, :, 0], v[:, :, -1] = -1, -1, -1, -1, -1, -1      # borders outside (sdf.py:149-154)
    return v


def test_table_generator_matches_library_for_every_configuration():
    from points2surf_amd import build, _lib
    build.build(verbose=False)
    lib = _lib.load()
    n = ctypes.c_int32(0)
    edges = (ctypes.c_int32 * 36)()
    checked = 0
    for case in range(256):
        for fb in range(64):
            assert lib.p2s_mc_table_entry(case | (fb << 8), ctypes.byref(n), edges) == 0
            want = M.cell_triangles(case, fb)
            got = [tuple(edges[3 * k + j] for j in range(3)) for k in range(n.value)]
            assert len(want) <= 12
            assert got == [tuple(t) for t in want], (case, fb, got, want)
            checked += 1
    assert checked == 16384
    # the classic counts of the unambiguous base cases: 1 corner -> 1 triangle, an edge pair -> 2, a face -> 2
    assert len(M.cell_triangles(0b00000001, 0)) == 1 and len(M.cell_triangles(0b00000011, 0)) == 2
    assert len(M.cell_triangles(0b00001111, 0)) == 2 and len(M.cell_triangles(0, 0)) == 0 == len(M.cell_triangles(255, 0))
    # an ambiguous face (corners 0 and 3 inside, same z face): two sheets either way, but different pairings
    a, b = M.cell_triangles(0b00001001, 0), M.cell_triangles(0b00001001, 1 << 4)
    assert len(a) == 2 and len(b) == 4 and a != b


@pytest.mark.parametrize('res,noise', [(16, 0.0), (24, 0.02), (20, 0.3)])
def test_oracle_mesh_is_closed_and_oriented(res, noise):
    """noise 0.3 makes a rough, many-component surface full of ambiguous faces: still a closed oriented 2-manifold"""
    v = _sphere_volume(res, noise=noise, seed=res)
    verts, faces, inverted = M.marching_cubes(v)
    chk = M.mesh_checks(verts, faces)
    assert chk['closed'] and chk['oriented'] and chk['unused_vertices'] == 0, chk
    assert chk['F'] > 0 and chk['euler'] % 2 == 0
    if noise == 0.0:
        assert chk['components'] == 1 and chk['euler'] == 2            # a sphere
        r = np.linalg.norm(verts, axis=1)
        assert abs(r.mean() - 0.6) < 0.02
    vv = verts.astype(np.float64)
    vol6 = np.einsum('ij,ij->i', vv[faces[:, 0]], np.cross(vv[faces[:, 1]], vv[faces[:, 2]])).sum()
    assert vol6 > 0                                                     # after fix_inversion the volume is positive


def test_oracle_handles_empty_and_exact_zero_volumes():
    v = -np.ones((8, 8, 8), np.float32)
    verts, faces, _ = M.marching_cubes(v)
    assert verts.shape == (0, 3) and faces.shape == (0, 3)
    v[3:5, 3:5, 3:5] = 1.0
    v[3, 3, 3] = 0.0                                                    # an exact zero counts as outside (v > 0 is inside)
    verts, faces, _ = M.marching_cubes(v, model_space=False)
    chk = M.mesh_checks(verts, faces)
    assert chk['closed'] and chk['oriented'] and chk['F'] > 0
    assert np.isfinite(verts).all()
