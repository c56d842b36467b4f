"""TEST INFRASTRUCTURE ONLY -- the iso-surface step as the reference runs it (reference source/sdf.py:211-225):

    v, f, normals, values = skimage.measure.marching_cubes_lewiner(volume, 0)
    v = (((v + 0.5) / float(grid_res)) - 0.5) * 2.0             (float32 array, python floats: float32 arithmetic)
    trimesh.repair.fix_inversion(mesh)                          (faces reversed when the signed volume is negative)

``marching_cubes`` = oracle/lewiner_mc.c (plain C restatement of scikit-image's Lewiner marching cubes, PINNED to
scikit-image 0.18.3: tests/test_lewiner_oracle.py, goldens from oracle/make_golden_mesh.py) + the two lines above.
The emission order is the device's, so device output is compared array for array.  ``mesh_checks``: topology invariants.
"""
import numpy as np

from . import lewiner_oracle as LO


def marching_cubes(volume, model_space=True, fix_inversion=True):
    """returns (verts [V,3] float32, faces [F,3] int32, inverted)"""
    vol = np.ascontiguousarray(volume, dtype=np.float32)
    res = vol.shape[0]
    verts, faces = LO.marching_cubes(vol)
    if model_space:
        verts = (((verts + np.float32(0.5)) / np.float32(res)) - np.float32(0.5)) * np.float32(2.0)
        verts = verts.astype(np.float32)
    inverted = False
    if fix_inversion and len(faces):
        v = verts.astype(np.float64)
        vol6 = np.einsum('ij,ij->i', v[faces[:, 0]], np.cross(v[faces[:, 1]], v[faces[:, 2]])).sum()
        if vol6 < 0:
            faces = np.ascontiguousarray(faces[:, [0, 2, 1]])
            inverted = True
    return verts, faces, inverted


def mesh_checks(verts, faces):
    """dict: closed (every undirected edge in exactly two faces), oriented (every directed edge once), euler, components"""
    f = np.asarray(faces, dtype=np.int64)
    if len(f) == 0:
        return {'closed': True, 'oriented': True, 'euler': 0, 'components': 0, 'V': len(verts), 'E': 0, 'F': 0}
    d = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    nv = int(max(len(verts), d.max() + 1))
    key_d = d[:, 0] * nv + d[:, 1]
    und = np.sort(d, axis=1)
    key_u = und[:, 0] * nv + und[:, 1]
    _, cnt_u = np.unique(key_u, return_counts=True)
    _, cnt_d = np.unique(key_d, return_counts=True)
    # connected components over the vertices that faces use (union-find on edges)
    parent = np.arange(nv)

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i
    for a, b in und[np.unique(key_u, return_index=True)[1]]:
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[ra] = rb
    usedv = np.unique(f)
    comps = len({find(i) for i in usedv})
    E = len(cnt_u)
    return {'closed': bool((cnt_u == 2).all()), 'oriented': bool((cnt_d == 1).all()), 'euler': int(len(usedv) - E + len(f)),
            'components': comps, 'V': int(len(verts)), 'E': int(E), 'F': int(len(f)), 'unused_vertices': int(len(verts) - len(usedv))}
