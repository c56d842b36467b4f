"""Python face of the HIP engine: thin, typed wrappers over the C ABI.

PyTorch-ROCm tensors are used only as containers (device memory + streams); all compute
happens in libp2s_hip.so.  Nothing here computes on the CPU and nothing falls back.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from .weights import build_blob


_JUMP_TABLES = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mt_jump_tables.npz')


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _f32c(t, device):
    if t.device != device:
        raise ValueError('tensor on %s, engine on %s' % (t.device, device))
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def select_device(index):
    """make ``cuda:index`` current (the drop-in's --gpu_idx / LOCAL_RANK) and return it"""
    if not torch.cuda.is_available():
        raise RuntimeError('points2surf_amd needs a ROCm GPU (gfx950); no CPU fallback exists')
    dev = torch.device('cuda', int(index))
    torch.cuda.set_device(dev)
    return dev


def upload(array, device):
    """host array -> contiguous device tensor (container only)"""
    return torch.from_numpy(np.ascontiguousarray(array)).to(device)


def release_scratch(device=None):
    """give the library's cached device memory (blocks of destroyed cloud handles, volume / iso-surface scratch) back
    to HIP (p2s_release_scratch)"""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    _lib.check(_lib.load().p2s_release_scratch(int(dev)))


class Model:
    """Engine-side model: BN-folded, MFMA-packed weights resident in HBM.

    Replaces ``make_regressor`` (reference source/points_to_surf_eval.py:150-171)."""

    def __init__(self, state_dict, cfg, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError('points2surf_amd needs a ROCm GPU (gfx950); no CPU fallback exists')
        self.lib = _lib.load()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.cfg = dict(cfg)
        blob, offs, mc = build_blob(state_dict, cfg)
        self._blob = np.ascontiguousarray(blob)
        self.points_per_patch = mc.points_per_patch
        self.sub_sample_size = mc.sub_sample_size
        self.output_dim = mc.output_dim
        self.uniform_subsample = bool(cfg.get('uniform_subsample', False))
        self.handle = ctypes.c_void_p()
        _lib.check(self.lib.p2s_model_create(
            ctypes.byref(mc), self._blob.ctypes.data_as(ctypes.c_void_p), self._blob.size, ctypes.byref(offs),
            self.device.index, ctypes.byref(self.handle)))

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.p2s_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, patch_pts_ps, pts_sub_sample_ms, query_ms, radius=None, want_logits=True, want_sdf=False):
        """a8 (+a9).  Returns (logits [B, output_dim] or None, sdf [B] or None).  Inputs are not modified."""
        dev = self.device
        patch = _f32c(patch_pts_ps, dev)
        sub = _f32c(pts_sub_sample_ms, dev)
        q = _f32c(query_ms, dev)
        B = patch.shape[0]
        if patch.shape != (B, self.points_per_patch, 3) or sub.shape != (B, self.sub_sample_size, 3) or q.shape != (B, 3):
            raise ValueError('bad input shapes %s %s %s' % (tuple(patch.shape), tuple(sub.shape), tuple(q.shape)))
        logits = torch.empty((B, self.output_dim), dtype=torch.float32, device=dev) if want_logits else None
        sdf = torch.empty((B,), dtype=torch.float32, device=dev) if want_sdf else None
        rad = _f32c(radius.reshape(-1), dev) if radius is not None else None
        if want_sdf and rad is None:
            raise ValueError('radius is required for the SDF output')
        with torch.cuda.device(dev):
            _lib.check(self.lib.p2s_encode_decode(self.handle, _ptr(patch), _ptr(sub), _ptr(q), _ptr(rad), B,
                                                  _ptr(logits), _ptr(sdf), _stream_ptr(dev)))
        return logits, sdf

    def features(self, patch_pts_ps, pts_sub_sample_ms, query_ms):
        dev = self.device
        patch = _f32c(patch_pts_ps, dev)
        sub = _f32c(pts_sub_sample_ms, dev)
        q = _f32c(query_ms, dev)
        B = patch.shape[0]
        fl = torch.empty((B, 1024), dtype=torch.float32, device=dev)
        fg = torch.empty((B, 1024), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(self.lib.p2s_encode_features(self.handle, _ptr(patch), _ptr(sub), _ptr(q), B, _ptr(fl), _ptr(fg),
                                                    _stream_ptr(dev)))
        return fl, fg

    def debug_fault_chunk(self, chunk_index):
        """test hook: the next pipeline call fails with P2S_EHIP before chunk ``chunk_index`` (-1 = off)"""
        _lib.check(self.lib.p2s_debug_fault_chunk(self.handle, int(chunk_index)))

    def set_profiling(self, on):
        _lib.check(self.lib.p2s_set_profiling(self.handle, int(bool(on))))

    def counters(self):
        c = _lib.Counters()
        _lib.check(self.lib.p2s_get_counters(self.handle, ctypes.byref(c)))
        return {n: getattr(c, n) for n, _ in c._fields_ if n != 'reserved'}


class Cloud:
    """Device-resident point cloud + cell index.  Replaces ``load_shape`` / cKDTree
    (reference source/data_loader.py:16-68)."""

    def __init__(self, pts, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError('points2surf_amd needs a ROCm GPU (gfx950); no CPU fallback exists')
        self.lib = _lib.load()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        if isinstance(pts, np.ndarray):
            pts = torch.from_numpy(np.ascontiguousarray(pts[:, :3], dtype=np.float32))
        self.pts = pts[:, :3].to(self.device, torch.float32).contiguous()
        self.n = int(self.pts.shape[0])
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.p2s_cloud_create(_ptr(self.pts), self.n, self.device.index, _stream_ptr(self.device),
                                                 ctypes.byref(self.handle)))

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.p2s_cloud_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def index_export(self):
        """test / diagnostic read-back of the device-built neighbour index: dict with G, lo [3], inv_cell,
        cell_start [G^3+1], sat [(G+1)^3], sorted_xyz [n,3] float32, sorted_id [n] int32"""
        G = ctypes.c_int32(0)
        geom = np.zeros(4, dtype=np.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.p2s_cloud_index_export(self.handle, ctypes.byref(G), geom.ctypes.data_as(ctypes.c_void_p),
                                                       None, None, None, _stream_ptr(self.device)))
            g = int(G.value)
            cell_start = np.zeros(g ** 3 + 1, dtype=np.int32)
            sat = np.zeros((g + 1) ** 3, dtype=np.int32)
            srt = np.zeros((self.n, 4), dtype=np.float32)
            _lib.check(self.lib.p2s_cloud_index_export(
                self.handle, None, None, cell_start.ctypes.data_as(ctypes.c_void_p), sat.ctypes.data_as(ctypes.c_void_p),
                srt.ctypes.data_as(ctypes.c_void_p), _stream_ptr(self.device)))
        return {'G': g, 'lo': geom[:3].copy(), 'inv_cell': float(geom[3]), 'cell_start': cell_start, 'sat': sat,
                'sorted_xyz': np.ascontiguousarray(srt[:, :3]), 'sorted_id': np.ascontiguousarray(srt[:, 3]).view(np.int32)}

    def count_queries(self, grid_resolution, epsilon):
        """a1: compute the query grid of (res, eps) on the handle (kept there for the pipeline) and return its size"""
        n = ctypes.c_int64(0)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.p2s_query_grid(self.handle, int(grid_resolution), int(epsilon), None, 0,
                                               ctypes.byref(n), _stream_ptr(self.device)), allow=(_lib.P2S_ECAPACITY,))
        return int(n.value)

    def query_grid(self, grid_resolution, epsilon):
        """a1 -> q [Q,3] float32 on the device (C order of the voxel index)."""
        n = ctypes.c_int64(0)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.p2s_query_grid(self.handle, int(grid_resolution), int(epsilon), None, 0,
                                               ctypes.byref(n), _stream_ptr(self.device)), allow=(_lib.P2S_ECAPACITY,))
            q = torch.empty((max(n.value, 1), 3), dtype=torch.float32, device=self.device)
            if n.value > 0:
                _lib.check(self.lib.p2s_query_grid(self.handle, int(grid_resolution), int(epsilon), _ptr(q), n.value,
                                                   ctypes.byref(n), _stream_ptr(self.device)))
        return q[:n.value]

    def knn_patch(self, queries, k, want_ids=True, want_patch=True):
        """a4+a5 -> (ids [Q,k] int32, patch_ps [Q,k,3], radius [Q])"""
        q = _f32c(queries, self.device)
        Q = q.shape[0]
        ids = torch.empty((Q, k), dtype=torch.int32, device=self.device) if want_ids else None
        patch = torch.empty((Q, k, 3), dtype=torch.float32, device=self.device) if want_patch else None
        rad = torch.empty((Q,), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.p2s_knn_patch(self.handle, _ptr(q), Q, int(k), _ptr(ids), _ptr(patch), _ptr(rad),
                                              _stream_ptr(self.device)))
        return ids, patch, rad

    def knn_patch_set(self, queries, k):
        """a4+a5 as the pipeline runs them: the k nearest points as a set -> (patch_ps [Q,k,3] rows in arbitrary order,
        radius [Q]); no sort (p2s_knn_patch_set)"""
        q = _f32c(queries, self.device)
        Q = q.shape[0]
        patch = torch.empty((Q, k, 3), dtype=torch.float32, device=self.device)
        rad = torch.empty((Q,), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.p2s_knn_patch_set(self.handle, _ptr(q), Q, int(k), _ptr(patch), _ptr(rad),
                                                  _stream_ptr(self.device)))
        return patch, rad

    def gather(self, ids):
        ids = ids.to(self.device, torch.int32).contiguous()
        out = torch.empty(tuple(ids.shape) + (3,), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.p2s_gather_points(self.handle, _ptr(ids), ids.numel(), _ptr(out),
                                                  _stream_ptr(self.device)))
        return out


class Rng:
    """Device twin of the dataset-wide ``np.random.RandomState(seed)`` used for the global
    sub-sample (reference source/data_loader.py:274-277)."""

    def __init__(self, seed, device=None, parallel=None):
        if not torch.cuda.is_available():
            raise RuntimeError('points2surf_amd needs a ROCm GPU (gfx950); no CPU fallback exists')
        self.lib = _lib.load()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.p2s_rng_create(ctypes.c_uint32(int(seed) & 0xffffffff), self.device.index,
                                               ctypes.byref(self.handle)))
            if parallel is None:
                parallel = True
            if parallel and os.path.isfile(_JUMP_TABLES):
                t = np.load(_JUMP_TABLES)
                # table entry m = jump of base*2^m blocks.  Streams of `bps` blocks need the entries from
                # log2(bps/base) upwards; fewer, longer streams mean fewer dependent jump rounds: 512 x 1024
                base = int(t['blocks_per_stream'])
                bps = 1024
                first = max(0, int(round(np.log2(bps / base))))
                levels = int(t['levels']) - first
                bps = base << first
                sup = [np.ascontiguousarray(t['jump_%d' % (first + m)], dtype=np.uint16) for m in range(levels)]
                counts = np.array([a.size for a in sup], dtype=np.int32)
                flat = np.ascontiguousarray(np.concatenate(sup))
                _lib.check(self.lib.p2s_rng_set_jump_tables(
                    self.handle, flat.ctypes.data_as(ctypes.c_void_p), counts.ctypes.data_as(ctypes.c_void_p), levels, bps))

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.p2s_rng_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_state(self):
        mt = np.zeros(624, dtype=np.uint32)
        pos = ctypes.c_int32(0)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.p2s_rng_get_state(self.handle, mt.ctypes.data_as(ctypes.c_void_p), ctypes.byref(pos),
                                                  _stream_ptr(self.device)))
        return mt, int(pos.value)

    def set_state(self, mt, pos):
        mt = np.ascontiguousarray(mt, dtype=np.uint32)
        assert mt.shape == (624,)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.p2s_rng_set_state(self.handle, mt.ctypes.data_as(ctypes.c_void_p), int(pos),
                                                  _stream_ptr(self.device)))

    def check(self):
        """raise if the parallel generator flagged an (astronomically unlikely) overflow of its super-segment"""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.p2s_rng_check(self.handle, _stream_ptr(self.device)))

    def random_rotations(self, n):
        """``n`` times ``trimesh.transformations.random_rotation_matrix(rng.rand(3))[:3, :3]`` from this stream
        (reference source/data_loader.py:384) -> [n, 3, 3] float64 device tensor"""
        rot = torch.empty((int(n), 3, 3), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.p2s_random_rotations(self.handle, int(n), _ptr(rot), _stream_ptr(self.device)))
        return rot

    def subsample_uniform(self, cloud, n_queries, n, want_pts=True):
        """a6 (uniform): ids [Q,n] int32 (+ gathered points [Q,n,3])"""
        ids = torch.empty((n_queries, n), dtype=torch.int32, device=self.device)
        pts = torch.empty((n_queries, n, 3), dtype=torch.float32, device=self.device) if want_pts else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.p2s_subsample_uniform(self.handle, cloud.handle, n_queries, int(n), _ptr(ids), _ptr(pts),
                                                      _stream_ptr(self.device)))
        return ids, pts

    def subsample_fixed(self, cloud, n, n_queries=0, query_ms=None, seed=42, want_pts=True):
        """a6 with train --fixed_subsample 1 (reference source/base/utils.py:210-211): ``rng.seed(42)`` before every
        query's draw.  ``query_ms`` None: uniform (n_queries identical rows); else the distance-weighted choice."""
        q = None if query_ms is None else _f32c(query_ms, self.device).reshape(-1, 3)
        nq = int(n_queries) if q is None else int(q.shape[0])
        ids = torch.empty((nq, n), dtype=torch.int32, device=self.device)
        pts = torch.empty((nq, n, 3), dtype=torch.float32, device=self.device) if want_pts else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.p2s_subsample_fixed(self.handle, cloud.handle, _ptr(q), nq, int(n),
                                                    ctypes.c_uint32(int(seed) & 0xffffffff), _ptr(ids), _ptr(pts),
                                                    _stream_ptr(self.device)))
        return ids, pts

    def skip(self, cloud, n, n_queries=0, query_ms=None):
        """advance the stream past queries without producing their ids: ``n_queries`` uniform draws of ``n`` ids, or
        the distance-weighted draws of the queries ``query_ms`` (query-range sharding, reference stream semantics)"""
        with torch.cuda.device(self.device):
            if query_ms is None:
                _lib.check(self.lib.p2s_subsample_uniform(self.handle, cloud.handle, int(n_queries), int(n), None, None,
                                                          _stream_ptr(self.device)))
            else:
                q = _f32c(query_ms, self.device).reshape(-1, 3)
                _lib.check(self.lib.p2s_subsample_weighted(self.handle, cloud.handle, _ptr(q), int(q.shape[0]), int(n),
                                                           None, None, _stream_ptr(self.device)))

    def subsample_weighted(self, cloud, query_ms, n, want_pts=True):
        """a6 (distance-weighted, p2s_vanilla): ids [Q,n] int32 = ``rng.choice(N, n, replace=False, p=dist_prob)``
        per query, in query order from the same stream (+ gathered points [Q,n,3])"""
        q = _f32c(query_ms, self.device).reshape(-1, 3)
        nq = int(q.shape[0])
        ids = torch.empty((nq, n), dtype=torch.int32, device=self.device)
        pts = torch.empty((nq, n, 3), dtype=torch.float32, device=self.device) if want_pts else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.p2s_subsample_weighted(self.handle, cloud.handle, _ptr(q), nq, int(n), _ptr(ids), _ptr(pts),
                                                       _stream_ptr(self.device)))
        return ids, pts


def query_inputs(model, cloud, rng, queries, index):
    """Diagnostic: the network inputs of query ``index`` of a shape whose queries (in order) are ``queries``, with
    ``rng`` positioned at the shape's first draw -- the stream is advanced past the queries before it (NULL-ids path),
    then a4..a6 for this one query.  Returns (patch_ps [1,k,3], sub_ms [1,n,3], query [1,3]) device tensors."""
    n = model.sub_sample_size
    q = _f32c(queries, model.device).reshape(-1, 3)
    if index > 0:
        if model.uniform_subsample:
            rng.skip(cloud, n, n_queries=int(index))
        else:
            rng.skip(cloud, n, query_ms=q[:index])
    one = q[index:index + 1].contiguous()
    if model.uniform_subsample:
        _, sub = rng.subsample_uniform(cloud, 1, n)
    else:
        _, sub = rng.subsample_weighted(cloud, one, n)
    _, patch, _ = cloud.knn_patch(one, model.points_per_patch, want_ids=False)
    return patch, sub.reshape(1, n, 3), one


def query_logits(model, cloud, rng, queries, index):
    """Diagnostic: the decoder logits of that query (see query_inputs).  Used to classify sign flips against a
    reference as fp32 ties (|sign logit| ~ 0)."""
    patch, sub, one = query_inputs(model, cloud, rng, queries, index)
    logits, _ = model.forward(patch, sub, one)
    return logits[0]


def kd_order(pts, leafsize=1000):
    """HOST: the index array of ``scipy.spatial.cKDTree(pts, leafsize)`` (p2s_kd_order_host) -- the order in which
    ``query_ball_point`` reports its hits.  Returns (order [n] int32, leaf_start [n_leaves + 1] int32)."""
    lib = _lib.load()
    p = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 3)
    n = p.shape[0]
    order = np.empty(n, np.int32)
    leaf = np.empty(n + 2, np.int32)
    nl = ctypes.c_int32(0)
    _lib.check(lib.p2s_kd_order_host(p.ctypes.data_as(ctypes.c_void_p), n, int(leafsize), order.ctypes.data_as(ctypes.c_void_p),
                                     leaf.ctypes.data_as(ctypes.c_void_p), n + 2, ctypes.byref(nl)))
    return order, leaf[:nl.value + 1].copy()


def ball_count(cloud, queries, radius):
    """len(kdtree.query_ball_point(q, radius)) per query (p2s_ball_count) -> int32 device tensor"""
    dev = cloud.device
    q = _f32c(queries, dev).reshape(-1, 3)
    out = torch.empty((max(int(q.shape[0]), 1),), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(cloud.lib.p2s_ball_count(cloud.handle, _ptr(q), int(q.shape[0]), float(radius), _ptr(out), _stream_ptr(dev)))
    return out[:q.shape[0]]


def ball_patch(cloud, rng, queries, radius, points_per_patch, with_rotation=False, want_ids=True):
    """the fixed-radius patch of every query, in order (p2s_ball_patch; reference source/base/point_cloud.py:177-191,
    source/data_loader.py:335-350): (ids [n,k] int32 or None, patch_ps [n,k,3], radius [n], rot [n,3,3] float64 or None).
    ``rng`` = the data set's first generator."""
    dev = cloud.device
    q = _f32c(queries, dev).reshape(-1, 3)
    n, k = int(q.shape[0]), int(points_per_patch)
    ids = torch.empty((max(n, 1), k), dtype=torch.int32, device=dev) if want_ids else None
    patch = torch.empty((max(n, 1), k, 3), dtype=torch.float32, device=dev)
    rad = torch.empty((max(n, 1),), dtype=torch.float32, device=dev)
    rot = torch.empty((max(n, 1), 3, 3), dtype=torch.float64, device=dev) if with_rotation else None
    with torch.cuda.device(dev):
        _lib.check(cloud.lib.p2s_ball_patch(rng.handle, cloud.handle, _ptr(q), n, float(radius), k, int(bool(with_rotation)),
                                            _ptr(ids), _ptr(patch), _ptr(rad), _ptr(rot), _stream_ptr(dev)))
    return (ids[:n] if ids is not None else None), patch[:n], rad[:n], (rot[:n] if rot is not None else None)


def ball_skip(cloud, rng, queries, radius, points_per_patch, with_rotation=False):
    """advance the first generator past the patch choices (and rotations) of ``queries`` without producing them"""
    dev = cloud.device
    q = _f32c(queries, dev).reshape(-1, 3)
    rot = torch.empty((max(int(q.shape[0]), 1), 9), dtype=torch.float64, device=dev) if with_rotation else None
    with torch.cuda.device(dev):
        _lib.check(cloud.lib.p2s_ball_patch(rng.handle, cloud.handle, _ptr(q), int(q.shape[0]), float(radius), int(points_per_patch),
                                            int(bool(with_rotation)), None, None, None, _ptr(rot), _stream_ptr(dev)))


def infer_shape(model, cloud, rng, grid_resolution, epsilon, q_begin=0, q_end=-1, chunk=0, want_queries=True,
                n_queries=None, rng_patch=None, want_logits=False):
    """Fused per-shape pipeline (p2s_infer_shape).  Returns (sdf [n] device tensor, q [n,3] or None).
    ``n_queries``: the size of the query grid if the caller already asked for it (Cloud.count_queries).
    ``rng_patch``: fixed-radius models -- the data set's first generator (patch choice).
    ``want_logits``: also return the decoder's raw logits [n, output_dim] as a third value (p2s_model_capture_logits)."""
    dev = model.device
    lib = model.lib
    with torch.cuda.device(dev):
        Q = cloud.count_queries(grid_resolution, epsilon) if n_queries is None else int(n_queries)
        qe = Q if q_end < 0 else q_end
        nq = max(qe - q_begin, 0)
        sdf = torch.empty((max(nq, 1),), dtype=torch.float32, device=dev)
        q = torch.empty((max(nq, 1), 3), dtype=torch.float32, device=dev) if want_queries else None
        done = ctypes.c_int64(0)
        logits = None
        if want_logits:
            logits = torch.empty((max(nq, 1), model.output_dim), dtype=torch.float32, device=dev)
            _lib.check(lib.p2s_model_capture_logits(model.handle, _ptr(logits), nq))
        _lib.check(lib.p2s_infer_shape_ball(model.handle, cloud.handle, rng.handle,
                                            rng_patch.handle if rng_patch is not None else None, int(grid_resolution),
                                            int(epsilon), int(q_begin), int(qe), int(chunk), _ptr(sdf), _ptr(q),
                                            ctypes.byref(done), _stream_ptr(dev)))
    if want_logits:
        return sdf[:nq], (q[:nq] if q is not None else None), logits[:nq]
    return sdf[:nq], (q[:nq] if q is not None else None)


def rotate_points(rot, pts):
    """``trimesh.transformations.transform_points(pts[i], M[i]).astype(float32)`` per item (reference
    source/data_loader.py:385-393): rot [n,3,3] float64, pts [n,P,3] float32 device tensors -> [n,P,3] float32"""
    lib = _lib.load()
    dev = pts.device
    pts = _f32c(pts, dev)
    rot = rot.to(dev, torch.float64).contiguous()
    n, P = int(pts.shape[0]), int(pts.shape[1])
    if rot.shape != (n, 3, 3) or pts.shape != (n, P, 3):
        raise ValueError('bad shapes %s %s' % (tuple(rot.shape), tuple(pts.shape)))
    out = torch.empty_like(pts)
    with torch.cuda.device(dev):
        _lib.check(lib.p2s_rotate_points(_ptr(rot), _ptr(pts), P, n, _ptr(out), _stream_ptr(dev)))
    return out


def infer_queries(model, cloud, rng_sub, rng_rot, queries, chunk=0):
    """GT-query evaluation pass for one shape (p2s_infer_queries): SDF at the given query points, with the
    reference's per-query random rotation when ``rng_rot`` is given (reference source/data_loader.py:365-393).
    Returns sdf [n] device tensor."""
    dev = model.device
    q = _f32c(queries, dev).reshape(-1, 3)
    nq = int(q.shape[0])
    sdf = torch.empty((max(nq, 1),), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(model.lib.p2s_infer_queries(model.handle, cloud.handle, rng_sub.handle,
                                               rng_rot.handle if rng_rot is not None else None, _ptr(q), nq,
                                               int(chunk), _ptr(sdf), _stream_ptr(dev)))
    return sdf[:nq]


def sdf_volume(query_pts_ms, query_dist_ms, grid_resolution, sigma, certainty_threshold, clamp=True):
    """add_samples_to_volume + propagate_sign (+ clamp) of reference source/sdf.py:82-178,199-201
    on the device.  Returns (volume [res,res,res] float32 device tensor, number of propagation sweeps)."""
    if not torch.cuda.is_available():
        raise RuntimeError('points2surf_amd needs a ROCm GPU (gfx950); no CPU fallback exists')
    lib = _lib.load()
    if isinstance(query_pts_ms, np.ndarray):
        query_pts_ms = torch.from_numpy(np.ascontiguousarray(query_pts_ms, dtype=np.float32)).cuda()
    if isinstance(query_dist_ms, np.ndarray):
        query_dist_ms = torch.from_numpy(np.ascontiguousarray(query_dist_ms, dtype=np.float32)).cuda()
    dev = query_pts_ms.device
    q = _f32c(query_pts_ms, dev)
    d = _f32c(query_dist_ms.reshape(-1), dev)
    if q.shape != (d.shape[0], 3):
        raise ValueError('bad shapes %s %s' % (tuple(q.shape), tuple(d.shape)))
    res = int(grid_resolution)
    vol = torch.empty((res, res, res), dtype=torch.float32, device=dev)
    iters = ctypes.c_int32(0)
    with torch.cuda.device(dev):
        _lib.check(lib.p2s_sdf_volume(_ptr(q), _ptr(d), d.shape[0], res, int(sigma), ctypes.c_float(certainty_threshold),
                                      int(bool(clamp)), dev.index, _ptr(vol), ctypes.byref(iters), _stream_ptr(dev)))
    return vol, int(iters.value)


def marching_cubes(vol, model_space=True, fix_inversion=True):
    """iso-surface of a [res]^3 float32 device volume at level 0 (what reference source/sdf.py:211-225
    obtains from scikit-image + trimesh).  Returns (verts [V,3] float32, faces [F,3] int32, inverted) device tensors;
    empty tensors when the volume holds no 0-level set."""
    if not torch.cuda.is_available():
        raise RuntimeError('points2surf_amd needs a ROCm GPU (gfx950); no CPU fallback exists')
    lib = _lib.load()
    dev = vol.device
    vol = _f32c(vol, dev)
    res = int(vol.shape[0])
    if vol.shape != (res, res, res):
        raise ValueError('bad volume shape %s' % (tuple(vol.shape),))
    nv, nf, inv = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int(0)
    # one call with room for a rough surface (16 res^2 vertices, 32 res^2 faces; a smooth one has ~0.6 / 1.2 res^2);
    # the exact sizes come back either way, a second call only if that was not enough
    cap_v, cap_f = 16 * res * res, 32 * res * res
    with torch.cuda.device(dev):
        for _ in range(2):
            verts = torch.empty((cap_v, 3), dtype=torch.float32, device=dev)
            faces = torch.empty((cap_f, 3), dtype=torch.int32, device=dev)
            rc = lib.p2s_marching_cubes(_ptr(vol), res, _ptr(verts), cap_v, _ptr(faces), cap_f, ctypes.byref(nv),
                                        ctypes.byref(nf), int(bool(model_space)), int(bool(fix_inversion)), ctypes.byref(inv),
                                        dev.index, _stream_ptr(dev))
            if rc != _lib.P2S_ECAPACITY:
                _lib.check(rc)
                break
            cap_v, cap_f = max(int(nv.value), 1), max(int(nf.value), 1)
        else:
            _lib.check(rc)
    return verts[:nv.value], faces[:nf.value], bool(inv.value)
