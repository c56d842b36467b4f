"""Per-shape result files through the native host writers of the C ABI (p2s_hostio.hip): the same
bytes as the reference's ``np.savetxt`` / ``mesh_io.write_off`` / the drop-in's PLY layout, without Python loops.  ctypes
releases the GIL for the call, so the drop-in's writer threads really run beside the inference loop."""
import ctypes
import os

import numpy as np

from . import _lib


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if shape is None else a.reshape(shape)


def _ensure_dir(path):
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)


def savetxt_f32(path, values):
    """``np.savetxt(path, values)`` for a 1-D float32 array (reference source/points_to_surf_eval.py:210)"""
    v = _f32(values).reshape(-1)
    _ensure_dir(path)
    _lib.check(_lib.load().p2s_write_txt_f32(os.fsencode(path), v.ctypes.data_as(ctypes.c_void_p), v.shape[0]))


def query_vis_ply(path, query_pts_ms, query_dist_ms):
    """``sdf.visualize_query_points`` (reference source/sdf.py:269-285): coloured point-cloud PLY"""
    d = _f32(query_dist_ms).reshape(-1)
    q = _f32(query_pts_ms, (-1, 3))
    if q.shape[0] != d.shape[0]:
        raise ValueError('query points %s vs distances %s' % (q.shape, d.shape))
    _ensure_dir(path)
    _lib.check(_lib.load().p2s_write_query_vis_ply(os.fsencode(path), q.ctypes.data_as(ctypes.c_void_p),
                                                   d.ctypes.data_as(ctypes.c_void_p), d.shape[0]))


def coff_samples(path, query_pts_ms, query_dist_ms):
    """the coloured-samples ``.off`` of ``implicit_surface_to_mesh`` (reference source/sdf.py:203-209 ->
    source/base/mesh_io.py:75-140)"""
    d = _f32(query_dist_ms).reshape(-1)
    q = _f32(query_pts_ms, (-1, 3))
    if q.shape[0] != d.shape[0]:
        raise ValueError('query points %s vs distances %s' % (q.shape, d.shape))
    if d.shape[0] == 0:
        return
    _ensure_dir(path)
    _lib.check(_lib.load().p2s_write_coff_samples(os.fsencode(path), q.ctypes.data_as(ctypes.c_void_p),
                                                  d.ctypes.data_as(ctypes.c_void_p), d.shape[0]))
