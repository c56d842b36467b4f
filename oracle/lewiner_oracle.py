"""TEST INFRASTRUCTURE ONLY -- ctypes face of oracle/lewiner_mc.c (the CPU restatement of
``skimage.measure.marching_cubes_lewiner(volume, 0)``, reference source/sdf.py:213-215) + mesh comparison helpers.

``build()`` compiles the C file with gcc into oracle/_build/ (also called by __graft_entry__.build())."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'lewiner_mc.c')
OUT_DIR = os.path.join(HERE, '_build')
LIB = os.path.join(OUT_DIR, 'liblewiner_oracle.so')

# Lewiner corner p -> (x, y, z) offsets; array axes are (z, y, x)
CORNER_XYZ = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]


def build(force=False):
    deps = [SRC, os.path.join(HERE, 'lewiner_tables.h')]
    if not force and os.path.isfile(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    subprocess.check_call(['gcc', '-O2', '-std=c99', '-fPIC', '-shared', '-ffp-contract=off', '-o', LIB, SRC, '-lm'])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB)
        L.lewiner_cell.argtypes = [ctypes.c_void_p] * 4
        L.lewiner_mc.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p,
                                 ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
        _lib = L
    return _lib


def cell(values8):
    """one cube, values in Lewiner corner order -> (tiling row, n_triangles, edge ids [n, 3])"""
    v = np.ascontiguousarray(values8, dtype=np.float32)
    row, nt = ctypes.c_int32(0), ctypes.c_int32(0)
    tri = np.zeros(36, dtype=np.int32)
    lib().lewiner_cell(v.ctypes.data, ctypes.addressof(row), ctypes.addressof(nt), tri.ctypes.data)
    return int(row.value), int(nt.value), tri[:3 * nt.value].reshape(-1, 3)


def cell_counts(values8):
    """(vertices, faces) scikit-image returns for the 2x2x2 volume holding this one cube"""
    _, nt, tri = cell(values8)
    return (len(set(tri.reshape(-1).tolist())), nt)


def cell_volume(values8):
    a = np.zeros((2, 2, 2), dtype=np.float32)
    for p, (x, y, z) in enumerate(CORNER_XYZ):
        a[z, y, x] = values8[p]
    return a


def marching_cubes(volume):
    """-> (verts [V, 3] float32 array-index coordinates, faces [F, 3] int32), device emission order"""
    vol = np.ascontiguousarray(volume, dtype=np.float32)
    res = vol.shape[0]
    assert vol.shape == (res, res, res)
    nv, nf = ctypes.c_longlong(0), ctypes.c_longlong(0)
    L = lib()
    L.lewiner_mc(vol.ctypes.data, res, None, 0, None, 0, ctypes.addressof(nv), ctypes.addressof(nf))
    verts = np.zeros((max(nv.value, 1), 3), dtype=np.float32)
    faces = np.zeros((max(nf.value, 1), 3), dtype=np.int32)
    rc = L.lewiner_mc(vol.ctypes.data, res, verts.ctypes.data, nv.value, faces.ctypes.data, nf.value, ctypes.addressof(nv),
                      ctypes.addressof(nf))
    assert rc == 0
    return verts[:nv.value], faces[:nf.value]


def canonical_mesh(verts, faces):
    """(sorted vertex positions [V, 3], sorted faces as position triples [F, 9]): every face rotated so that its
    lexicographically smallest vertex POSITION comes first (orientation kept), then all rows sorted.  Vertices may
    coincide (a grid value of exactly 0 puts several vertices on one grid point), so faces are compared by position,
    not by index.  Two meshes are the same surface iff both arrays are equal."""
    v = np.ascontiguousarray(verts, dtype=np.float32)
    tri = v[np.asarray(faces, dtype=np.int64)]                                    # [F, 3, 3]
    # order of the three vertices of a face by (x, y, z)
    k = np.lexsort((tri[..., 2], tri[..., 1], tri[..., 0]), axis=1)[:, 0]          # index of the smallest vertex
    idx = (k[:, None] + np.arange(3)[None, :]) % 3
    tri = np.take_along_axis(tri, idx[:, :, None], axis=1).reshape(-1, 9)
    tri = tri[np.lexsort(tri.T[::-1])]
    vs = v[np.lexsort(v.T[::-1])]
    return vs, tri


def same_mesh(va, fa, vb, fb):
    """exact comparison (positions bit for bit) of two meshes as sets of vertices / oriented triangles"""
    if va.shape != vb.shape or fa.shape != fb.shape:
        return False, 'counts differ: %s %s vs %s %s' % (va.shape, fa.shape, vb.shape, fb.shape)
    a, ta = canonical_mesh(va, fa)
    b, tb = canonical_mesh(vb, fb)
    if not np.array_equal(a, b):
        return False, '%d vertex positions differ (max %.3g)' % (int((a != b).any(axis=1).sum()), float(np.abs(a - b).max()))
    if not np.array_equal(ta, tb):
        return False, '%d faces differ' % int((ta != tb).any(axis=1).sum())
    return True, ''
