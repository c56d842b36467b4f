"""TEST INFRASTRUCTURE ONLY -- CPU (numpy) restatement of the Points2Surf SDF-inference path.

This is the *oracle* for the MI355X engine: a plain numpy restatement of what the
reference (ErlerPhilipp/points2surf) computes on the path named by
BASELINE.json's north_star.  It is NOT part of the product; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Pinning status: PINNED.  ``oracle/make_golden.py`` executes the unmodified
reference (through ``oracle/ref_shims.py``) in the build container on seeded
synthetic weights + the ``abc_minimal`` fixture cloud and commits the outputs to
``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function below
against those vectors (query grid, kNN ids, radius, patch, sub-sample ids,
logits, SDF).  Not pinned (not on the path, third-party code absent): marching
cubes (scikit-image) and trimesh mesh export -- see DESIGN.md.

Each function cites the reference file:line it follows (paths relative to the
reference root).
"""
import numpy as np

# --------------------------------------------------------------------------------------
# a6 prerequisite: NumPy legacy MT19937 stream (numpy/random/_mt19937 + legacy RandomState)
# Reference call sites: source/data_loader.py:272-277 (RandomState(seed)),
#                       source/base/utils.py:211,216,219,224 (seed/randint/choice/shuffle)
# Third-party algorithm restated: MT19937 (Matsumoto & Nishimura 1998) as frozen by
# NumPy's legacy-RandomState policy (requirements.txt:1 numpy>=1.18, un-pinned).
# --------------------------------------------------------------------------------------

_U = np.uint32
_MATRIX_A = _U(0x9908b0df)
_UPPER = _U(0x80000000)
_LOWER = _U(0x7fffffff)


def _mix(u, v):
    y = (u & _UPPER) | (v & _LOWER)
    return (y >> _U(1)) ^ np.where((y & _U(1)) != 0, _MATRIX_A, _U(0))


class LegacyMT19937:
    """Bit-exact model of ``np.random.RandomState(seed)``'s raw uint32 stream."""

    N = 624
    M = 397

    def __init__(self, seed):
        self.seed(seed)

    def seed(self, seed):
        # init_genrand (numpy _legacy_seeding with an integer seed)
        mt = np.empty(self.N, dtype=np.uint64)
        mt[0] = int(seed) & 0xffffffff
        for i in range(1, self.N):
            prev = int(mt[i - 1])
            mt[i] = (1812433253 * (prev ^ (prev >> 30)) + i) & 0xffffffff
        self.mt = mt.astype(np.uint32)
        self.pos = self.N  # "needs twist before first draw"
        self._tempered = None

    def _twist(self):
        mt = self.mt
        new = np.empty_like(mt)
        n, m = self.N, self.M
        # the recurrence has three dependent, internally-parallel phases
        new[0:n - m] = mt[m:n] ^ _mix(mt[0:n - m], mt[1:n - m + 1])                       # i in [0,227)
        new[n - m:2 * (n - m)] = new[0:n - m] ^ _mix(mt[n - m:2 * (n - m)],
                                                       mt[n - m + 1:2 * (n - m) + 1])       # [227,454)
        new[2 * (n - m):n - 1] = new[n - m:m - 1] ^ _mix(mt[2 * (n - m):n - 1],
                                                           mt[2 * (n - m) + 1:n])           # [454,623)
        new[n - 1] = new[m - 1] ^ _mix(mt[n - 1:n], new[0:1])[0]
        self.mt = new
        y = new.copy()
        y ^= y >> _U(11)
        y ^= (y << _U(7)) & _U(0x9d2c5680)
        y ^= (y << _U(15)) & _U(0xefc60000)
        y ^= y >> _U(18)
        self._tempered = y
        self.pos = 0

    def raw(self, count):
        """next ``count`` tempered uint32 words."""
        out = np.empty(count, dtype=np.uint32)
        done = 0
        while done < count:
            if self.pos >= self.N:
                self._twist()
            take = min(count - done, self.N - self.pos)
            out[done:done + take] = self._tempered[self.pos:self.pos + take]
            self.pos += take
            done += take
        return out

    def _unread(self, k):
        """push back the last k words (only within the current block)."""
        assert 0 <= k <= self.pos
        self.pos -= k

    # -- legacy RandomState.randint(0, high, size) for high-1 < 2**32 ------------------
    def randint(self, high, size):
        """``RandomState.randint(low=0, high=high, size=size)`` (dtype int64):
        masked rejection on successive 32-bit words (numpy _bounded_integers,
        legacy ``use_masked=True`` path).  Call site: source/base/utils.py:216."""
        rng = int(high) - 1
        if rng == 0:
            return np.zeros(size, dtype=np.int64)
        assert 0 < rng <= 0xffffffff
        mask = rng
        for s in (1, 2, 4, 8, 16):
            mask |= mask >> s
        out = np.empty(size, dtype=np.int64)
        done = 0
        while done < size:
            need = size - done
            # draw with head-room, keep the accepted prefix, push back the unread tail
            want = min(self.N, max(16, int(need * (mask + 1) / (rng + 1) * 1.1) + 8))
            if self.pos >= self.N:
                self._twist()
            want = min(want, self.N - self.pos)
            w = self._tempered[self.pos:self.pos + want] & _U(mask)
            ok = np.nonzero(w <= rng)[0]
            if ok.size >= need:
                used = int(ok[need - 1]) + 1
                out[done:] = w[ok[:need]]
                self.pos += used
                done = size
            else:
                out[done:done + ok.size] = w[ok]
                done += ok.size
                self.pos += want
        return out

    # -- legacy RandomState.shuffle of a 2-D array ----------------------------------------
    def shuffle_rows(self, x):
        """``RandomState.shuffle(x)`` for a 2-D array, in place (numpy mtrand ``shuffle``, ndim > 1 branch: for i in
        reversed(range(1, n)): j = rk_interval(i); swap rows i, j through a bounce buffer).  rk_interval(max):
        smallest 2^k - 1 >= max as mask, successive 32-bit words & mask until <= max.  Call site:
        source/base/utils.py:224 (clouds with fewer points than the sub-sample size)."""
        n = x.shape[0]
        for i in reversed(range(1, n)):
            mask = i
            for sft in (1, 2, 4, 8, 16):
                mask |= mask >> sft
            while True:
                v = int(self.raw(1)[0]) & mask
                if v <= i:
                    break
            if v != i:
                x[[i, v]] = x[[v, i]]
        return x

    # -- legacy RandomState.permutation(n) ------------------------------------------------
    def permutation(self, n):
        """``RandomState.permutation(n)`` = shuffle(arange(n)) (numpy mtrand: 1-D branch of ``shuffle``, the same
        i = n-1 .. 1, j = rk_interval(i) swaps as above) -- what ``rng.choice(np.arange(n), k, replace=False)`` takes its
        first k entries of.  Call site: source/base/point_cloud.py:183 (fixed-radius patches)."""
        a = np.arange(n, dtype=np.int64)
        for i in reversed(range(1, n)):
            mask = i
            for sft in (1, 2, 4, 8, 16):
                mask |= mask >> sft
            while True:
                v = int(self.raw(1)[0]) & mask
                if v <= i:
                    break
            a[i], a[v] = a[v], a[i]
        return a

    # -- legacy RandomState.rand / random_sample ---------------------------------------
    def rand(self, size):
        w = self.raw(2 * size).astype(np.uint64)
        a = w[0::2] >> np.uint64(5)
        b = w[1::2] >> np.uint64(6)
        return (a * np.float64(67108864.0) + b) / np.float64(9007199254740992.0)

    # -- legacy RandomState.choice(a, size, replace=False, p) ---------------------------
    def choice_noreplace(self, n, size, p):
        """Call site: source/base/utils.py:219 (distance-weighted sub-sample, p2s_vanilla)."""
        p = np.array(p, dtype=np.float64, copy=True)
        found = np.zeros(size, dtype=np.int64)
        n_uniq = 0
        while n_uniq < size:
            x = self.rand(size - n_uniq)
            if n_uniq > 0:
                p[found[0:n_uniq]] = 0
            cdf = np.cumsum(p)
            cdf /= cdf[-1]
            new = cdf.searchsorted(x, side='right')
            _, unique_indices = np.unique(new, return_index=True)
            unique_indices.sort()
            new = new.take(unique_indices)
            found[n_uniq:n_uniq + new.size] = new
            n_uniq += new.size
        return found


# --------------------------------------------------------------------------------------
# a1: near-surface query grid
# --------------------------------------------------------------------------------------

def model_space_to_volume_space(pts_ms, vol_res):
    """source/sdf.py:73-75 (fp32 arithmetic when pts is fp32, floor, integer cast)."""
    pts = np.asarray(pts_ms)
    pos = (pts + 1.0) / 2.0
    return np.floor(pos * vol_res).astype(np.int64)


def volume_space_to_model_space(pts_vs, vol_res):
    """source/sdf.py:78-79 (float64 arithmetic on integer indices)."""
    return ((pts_vs + 0.5) / vol_res) * 2.0 - 1.0


def _box_offsets(size):
    # scipy.ndimage.convolve(kernel of ones, origin 0): out[i] = sum_j in[i + size//2 - j], j=0..size-1
    return [size // 2 - j for j in range(size)]


def query_grid(pts, grid_resolution, epsilon):
    """source/sdf.py:46-70 ``get_voxel_centers_grid_smaller_pc``.

    occupancy volume -> box filter of ones(eps^3), mode='nearest' -> nonzero of
    ``[:-1,:-1,:-1]`` (C order) -> voxel centres as float32.
    Returns (q [Q,3] float32, vox [Q,3] int64)."""
    res = int(grid_resolution)
    pts = np.asarray(pts, dtype=np.float32)
    vs = model_space_to_volume_space(pts, res)
    occ = np.zeros((res, res, res), dtype=bool)
    occ[vs[:, 0], vs[:, 1], vs[:, 2]] = True          # numpy semantics incl. IndexError when outside
    # a box filter of a 0/1 volume with edge replication is non-zero exactly where the
    # OR over the (index-clamped) neighbourhood is set; separable per axis.
    near = occ
    for axis in range(3):
        acc = np.zeros_like(near)
        idx = np.arange(res)
        for o in _box_offsets(int(epsilon)):
            src = np.clip(idx + o, 0, res - 1)
            acc |= np.take(near, src, axis=axis)
        near = acc
    vox = np.stack(np.nonzero(near[:-1, :-1, :-1]), axis=1)
    q = volume_space_to_model_space(vox, res).astype(np.float32)
    return q, vox


# --------------------------------------------------------------------------------------
# a4 / a5: kNN patch, radius, patch space
# --------------------------------------------------------------------------------------

def knn_ids(pts, queries, k, chunk=256):
    """source/base/point_cloud.py:170-175 ``kdtree.query(x, k)`` restated as an exact
    brute-force search: float64 squared distances on the float64 copy of the float32
    cloud (cKDTree stores float64, source/data_loader.py:40-42), ascending by distance.
    Returns ids [Q,k] int32 (sorted by distance)."""
    p64 = np.asarray(pts, dtype=np.float64)
    q64 = np.asarray(queries, dtype=np.float64).reshape(-1, 3)
    n = p64.shape[0]
    if n < k:
        raise IndexError('cloud has fewer points (%d) than points_per_patch (%d)' % (n, k))
    out = np.empty((q64.shape[0], k), dtype=np.int32)
    for s in range(0, q64.shape[0], chunk):
        qq = q64[s:s + chunk]
        d0 = qq[:, None, 0] - p64[None, :, 0]
        d2 = d0 * d0
        d1 = qq[:, None, 1] - p64[None, :, 1]
        d2 += d1 * d1
        d1 = qq[:, None, 2] - p64[None, :, 2]
        d2 += d1 * d1
        part = np.argpartition(d2, k - 1, axis=1)[:, :k]
        dd = np.take_along_axis(d2, part, axis=1)
        order = np.argsort(dd, axis=1, kind='stable')
        out[s:s + chunk] = np.take_along_axis(part, order, axis=1)
    return out


def patch_radius_and_ps(pts, ids, query):
    """source/data_loader.py:341-350 + source/base/utils.py:62-69,80-88 (all float32):
    r = max_i ||q - p_i||_2  (norm = sqrt((dx*dx + dy*dy) + dz*dz), each op rounded to fp32)
    patch_ps = (p_i - q) / r."""
    pts = np.asarray(pts, dtype=np.float32)
    q = np.asarray(query, dtype=np.float32)
    p = pts[ids]                                   # [..., k, 3]
    d = q[..., None, :] - p
    s = d * d
    dist = np.sqrt((s[..., 0] + s[..., 1]) + s[..., 2])
    r = dist.max(axis=-1)
    ps = (p - q[..., None, :]) / r[..., None, None]
    return r.astype(np.float32), ps.astype(np.float32)


# --------------------------------------------------------------------------------------
# a6: global sub-sample
# --------------------------------------------------------------------------------------

def ball_patch(rng_patch, pts, tree, query, radius, points_per_patch):
    """The fixed-radius patch of ONE query: get_patch_kdtree with patch_radius > 0 (reference
    source/base/point_cloud.py:177-191) + the padding / patch-space steps of source/data_loader.py:340-350.
    ``tree`` = scipy.spatial.cKDTree(pts, 1000) (source/data_loader.py:40-42): scipy is the reference's own dependency
    and present in this image, so the ball -- and the ORDER of its points, which decides what a random choice of
    positions selects -- comes from the call the reference makes.  ``rng_patch`` = dataset.rng (LegacyMT19937).
    Returns (ids [k] int32 with 0 for padding, patch_ps [k,3] float32, number of points in the ball)."""
    ids = np.array(tree.query_ball_point(x=query, r=radius), dtype=np.int32)
    count = ids.shape[0]
    if count > points_per_patch:
        ids = ids[rng_patch.permutation(count)[:points_per_patch]]      # rng.choice(np.arange(count), k, replace=False)
    pad = np.zeros(points_per_patch, dtype=bool)
    if count < points_per_patch:
        pad[count:] = True
        ids = np.concatenate((ids, np.zeros(points_per_patch - count, np.int32)))
    patch_ms = pts[ids, :]
    patch_ms[pad, :] = query
    patch_ps = (patch_ms - np.repeat(np.expand_dims(query, 0), points_per_patch, axis=0)) / np.float32(radius)
    return ids, patch_ps.astype(np.float32), count


def dist_prob(pts, query):
    """source/base/utils.py:200-208 (float32 throughout; np.sum pairwise)."""
    pts = np.asarray(pts, dtype=np.float32)
    qp = np.broadcast_to(np.asarray(query, dtype=np.float32), pts.shape)
    dist = np.linalg.norm(qp - pts, axis=1)
    dn = dist / np.max(dist)
    prob = 1.0 - 1.5 * dn
    pc = np.clip(prob, 0.05, 1.0)
    return pc / np.sum(pc)


def subsample_ids(rng, pts, query, sub_sample_size, uniform, fixed=False):
    """source/base/utils.py:196-219: ids of the global sub-sample for ONE query.
    ``rng`` is a LegacyMT19937 shared by all queries of all shapes
    (source/data_loader.py:274-277)."""
    n = pts.shape[0]
    if n < sub_sample_size:
        raise ValueError('N < sub_sample_size: use subsample_points (the reference shuffles shape.pts in place)')
    if fixed:
        rng.seed(42)
    if uniform:
        return rng.randint(n, sub_sample_size)
    return rng.choice_noreplace(n, sub_sample_size, dist_prob(pts, query))


def subsample_points(rng, pts, query, sub_sample_size, uniform, fixed=False):
    """source/base/utils.py:196-227 literally: returns pts_sub_sample_ms [sub_sample_size, 3].  For a cloud with fewer
    points than the sub-sample size the reference SHUFFLES ``pts`` IN PLACE (``pts_ms[:, :3]`` is a view, :223-224) and
    pads with zeros -- ``pts`` is modified, exactly as shape.pts is under the reference's kd-tree."""
    if pts.shape[0] >= sub_sample_size:
        return pts[subsample_ids(rng, pts, query, sub_sample_size, uniform, fixed)]
    rng.shuffle_rows(pts)
    pad = np.zeros((sub_sample_size - pts.shape[0], 3), dtype=np.float32)
    return np.concatenate((pts[:, :3], pad), axis=0)


# --------------------------------------------------------------------------------------
# a8: the network (eval mode), a9: post-processing
# --------------------------------------------------------------------------------------

_BN_EPS = np.float32(1e-5)


def _bn(x, w, prefix, channel_axis):
    """torch BatchNorm1d in eval mode: (x - mean) / sqrt(var + eps) * gamma + beta."""
    shape = [1] * x.ndim
    shape[channel_axis] = -1
    mean = w[prefix + '.running_mean'].reshape(shape)
    var = w[prefix + '.running_var'].reshape(shape)
    gamma = w[prefix + '.weight'].reshape(shape)
    beta = w[prefix + '.bias'].reshape(shape)
    return (x - mean) / np.sqrt(var + _BN_EPS) * gamma + beta


def _conv(x, w, prefix):
    """Conv1d(kernel 1) on x [B, P, Cin] (points-major layout of the reference's [B,Cin,P])."""
    W = w[prefix + '.weight'][:, :, 0]              # [Cout, Cin]
    return x @ W.T + w[prefix + '.bias']


def _fc(x, w, prefix):
    return x @ w[prefix + '.weight'].T + w[prefix + '.bias']


def _relu(x):
    return np.maximum(x, np.float32(0))


def _stn_trunk(x, w, pre):
    """shared trunk of STN/QSTN: source/points_to_surf_model.py:41-63 / :100-123."""
    x = _relu(_bn(_conv(x, w, pre + '.conv1'), w, pre + '.bn1', 2))
    x = _relu(_bn(_conv(x, w, pre + '.conv2'), w, pre + '.bn2', 2))
    x = _relu(_bn(_conv(x, w, pre + '.conv3'), w, pre + '.bn3', 2))
    x = x.max(axis=1)                               # MaxPool1d over all points
    x = _relu(_bn(_fc(x, w, pre + '.fc1'), w, pre + '.bn4', 1))
    x = _relu(_bn(_fc(x, w, pre + '.fc2'), w, pre + '.bn5', 1))
    return _fc(x, w, pre + '.fc3')


def stn_forward(x, w, pre, dim):
    """source/points_to_surf_model.py:41-69: returns trans [B, dim, dim]."""
    t = _stn_trunk(x, w, pre)
    t = t + np.eye(dim, dtype=np.float32).reshape(1, dim * dim)
    return t.reshape(-1, dim, dim)


def quat_to_rotmat(q):
    """source/base/utils.py:13-46 ``batch_quat_to_rotmat`` (same index pattern)."""
    q = q.astype(np.float32)
    s = np.float32(2) / np.sum(q * q, axis=1)
    h = q[:, :, None] * q[:, None, :]
    out = np.empty((q.shape[0], 3, 3), dtype=np.float32)
    out[:, 0, 0] = 1 - (h[:, 2, 2] + h[:, 3, 3]) * s
    out[:, 0, 1] = (h[:, 1, 2] - h[:, 3, 0]) * s
    out[:, 0, 2] = (h[:, 1, 3] + h[:, 2, 0]) * s
    out[:, 1, 0] = (h[:, 1, 2] + h[:, 3, 0]) * s
    out[:, 1, 1] = 1 - (h[:, 1, 1] + h[:, 3, 3]) * s
    out[:, 1, 2] = (h[:, 2, 3] - h[:, 1, 0]) * s
    out[:, 2, 0] = (h[:, 1, 3] - h[:, 2, 0]) * s
    out[:, 2, 1] = (h[:, 2, 3] + h[:, 1, 0]) * s
    out[:, 2, 2] = 1 - (h[:, 1, 1] + h[:, 2, 2]) * s
    return out


def qstn_forward(x, w, pre):
    """source/points_to_surf_model.py:100-131: returns (R [B,3,3], quat [B,4])."""
    quat = _stn_trunk(x, w, pre) + np.array([1, 0, 0, 0], dtype=np.float32)
    return quat_to_rotmat(quat), quat


def pointnetfeat_forward(x, w, pre, use_point_stn, use_feat_stn=True, return_aux=False, sym_op='max'):
    """source/points_to_surf_model.py:177-234 (num_scales=1; sym_op 'max' or 'sum', :211-214 -- the STN / QSTN inside
    keep their max-pool whatever sym_op says, :47, :106).
    x: [B, P, 3] (points-major).  Returns feature [B, net_size] (+ trans)."""
    trans = None
    if use_point_stn:
        trans, _ = qstn_forward(x, w, pre + '.stn1')
        x = np.einsum('bij,bpj->bpi', trans, x)      # bmm(trans, x[:, :3, :])
    x = _relu(_bn(_conv(x, w, pre + '.conv0a'), w, pre + '.bn0a', 2))
    x = _relu(_bn(_conv(x, w, pre + '.conv0b'), w, pre + '.bn0b', 2))
    aux = {}
    if use_feat_stn:
        trans2 = stn_forward(x, w, pre + '.stn2', 64)
        if return_aux:
            aux['trans2'] = trans2
        x = np.einsum('bij,bpj->bpi', trans2, x)     # bmm(trans2, x)
    x = _relu(_bn(_conv(x, w, pre + '.conv1'), w, pre + '.bn1', 2))
    x = _relu(_bn(_conv(x, w, pre + '.conv2'), w, pre + '.bn2', 2))
    x = _bn(_conv(x, w, pre + '.conv3'), w, pre + '.bn3', 2)     # no ReLU before the pool
    x = x.max(axis=1) if sym_op == 'max' else x.sum(axis=1, dtype=np.float32)      # torch.sum(x, 2, keepdim=True)
    if return_aux:
        return x, trans, aux
    return x, trans


def model_forward(w, cfg, patch_pts_ps, pts_sub_sample_ms, query_ms, chunk=32, return_feats=False):
    """source/points_to_surf_model.py:296-352 ``PointsToSurfModel.forward`` (eval mode).

    w:   dict name -> float32 ndarray (state_dict without the ``module.`` prefix)
    cfg: dict with use_point_stn, shared_transformer (and use_feat_stn, default True)
    Returns logits [B, output_dim] float32.  Inputs are not modified (the reference
    translates pts_sub_sample_ms in place, :303)."""
    w = {k: np.asarray(v, dtype=np.float32) for k, v in w.items()}
    B = patch_pts_ps.shape[0]
    use_point_stn = bool(cfg.get('use_point_stn', False))
    shared = bool(cfg.get('shared_transformer', False))
    use_feat_stn = bool(cfg.get('use_feat_stn', True))
    single = bool(cfg.get('single_transformer', False))
    sym_op = cfg.get('sym_op', 'max')
    out = []
    feats = []
    for s in range(0, B, chunk):
        patch = np.asarray(patch_pts_ps[s:s + chunk], dtype=np.float32)
        shape = np.asarray(pts_sub_sample_ms[s:s + chunk], dtype=np.float32) \
            - np.asarray(query_ms[s:s + chunk], dtype=np.float32)[:, None, :]      # :303
        if single:                                                                # :320-323
            lg, _ = pointnetfeat_forward(np.concatenate([patch, shape], axis=1), w, 'feat_local_global',
                                         use_point_stn, use_feat_stn, sym_op=sym_op)
            f = _relu(_bn(_fc(lg, w, 'fc1_local_global'), w, 'bn1_local_global', 1))
            f = _relu(_bn(_fc(f, w, 'fc2'), w, 'bn2', 1))
            f = _relu(_bn(_fc(f, w, 'fc3'), w, 'bn3', 1))
            out.append(_fc(f, w, 'fc4'))
            if return_feats:
                feats.append((lg, lg))
            continue
        if use_point_stn and shared:                                              # :325-331
            both = np.concatenate([patch, shape], axis=1)
            trans, _ = qstn_forward(both, w, 'point_stn')
            shape = np.einsum('bij,bpj->bpi', trans, shape)
            patch = np.einsum('bij,bpj->bpi', trans, patch)
        g, trans_g = pointnetfeat_forward(shape, w, 'feat_global',
                                          use_point_stn and not shared, use_feat_stn, sym_op=sym_op)   # :333
        gfc = _relu(_bn(_fc(g, w, 'fc1_global'), w, 'bn1_global', 1))                   # :335
        if use_point_stn and not shared:                                                # :337-339
            patch = np.einsum('bij,bpj->bpi', trans_g, patch)
        l, _ = pointnetfeat_forward(patch, w, 'feat_local', False, use_feat_stn, sym_op=sym_op)        # :341
        lfc = _relu(_bn(_fc(l, w, 'fc1_local'), w, 'bn1_local', 1))                     # :343
        f = np.concatenate([lfc, gfc], axis=1)                                          # :346
        f = _relu(_bn(_fc(f, w, 'fc2'), w, 'bn2', 1))
        f = _relu(_bn(_fc(f, w, 'fc3'), w, 'bn3', 1))
        out.append(_fc(f, w, 'fc4'))
        if return_feats:
            feats.append((l, g))
    logits = np.concatenate(out, axis=0).astype(np.float32)
    if return_feats:
        return logits, np.concatenate([f[0] for f in feats]), np.concatenate([f[1] for f in feats])
    return logits


def post_process(logits, patch_radius):
    """source/points_to_surf_eval.py:184-196 + source/sdf_nn.py:11-21 + :263-273,205-207:
    sdf = tanh(l0)^2 * r * (l1 >= 0 ? +1 : -1); NaN -> 1.0."""
    logits = np.asarray(logits, dtype=np.float32)
    if logits.shape[1] == 1:      # outputs = ['imp_surf']: sdf_nn.post_process_distance (sdf_nn.py:6-8), * radius (:176-183)
        sdf = (np.tanh(logits[:, 0]) ** 2 * np.sign(logits[:, 0]) * np.asarray(patch_radius, dtype=np.float32)).astype(np.float32)
        sdf[np.isnan(sdf)] = 1.0
        return sdf
    mag = np.tanh(logits[:, 0]) ** 2 * np.asarray(patch_radius, dtype=np.float32)
    sign = np.where(logits[:, 1] >= 0, np.float32(1), np.float32(-1))
    sdf = (mag * sign).astype(np.float32)
    sdf[np.isnan(sdf)] = 1.0
    return sdf


# --------------------------------------------------------------------------------------
# end to end (a1..a10 minus file IO) for one shape
# --------------------------------------------------------------------------------------

def infer_shape(w, cfg, pts, grid_resolution, epsilon, rng, points_per_patch=300,
                sub_sample_size=1000, query_range=None, chunk=32, return_all=False, rng_patch=None):
    """What source/points_to_surf_eval.py:358-404 computes for one shape in
    reconstruction mode with --workers 0.  ``rng`` = LegacyMT19937 carried across shapes.
    ``query_range`` = (q0, q1) restricts to a prefix/sub-range of the query list (the RNG
    is then only advanced for those queries -- use a prefix for stream parity)."""
    pts = np.asarray(pts, dtype=np.float32)
    q_all, _ = query_grid(pts, grid_resolution, epsilon)
    q0, q1 = (0, q_all.shape[0]) if query_range is None else query_range
    q = q_all[q0:q1]
    uniform = bool(cfg.get('uniform_subsample', False))
    fixed = bool(cfg.get('fixed_subsample', False))
    radius = float(cfg.get('patch_radius', 0.0) or 0.0)
    if radius > 0.0:
        # fixed-radius models: the patch choice draws from dataset.rng (``rng_patch``), the sub-sample from
        # dataset.rng_global_sample (``rng``) -- source/data_loader.py:272-277,336,376
        from scipy import spatial
        tree = spatial.cKDTree(pts, 1000)
        ids = np.zeros((q.shape[0], points_per_patch), np.int32)
        patch_ps = np.zeros((q.shape[0], points_per_patch, 3), np.float32)
        counts = np.zeros(q.shape[0], np.int64)
        sub_ids = np.zeros((q.shape[0], sub_sample_size), np.int64)
        for i in range(q.shape[0]):
            ids[i], patch_ps[i], counts[i] = ball_patch(rng_patch, pts, tree, q[i], radius, points_per_patch)
            sub_ids[i] = subsample_ids(rng, pts, q[i], sub_sample_size, uniform, fixed)
        # fixed radius: the prediction is not rescaled (source/points_to_surf_eval.py:180,188 `if not fixed_radius`)
        r = np.ones(q.shape[0], np.float32)
        logits = model_forward(w, cfg, patch_ps, pts[sub_ids], q, chunk=chunk)
        sdf = post_process(logits, r)
        if return_all:
            return dict(q=q, knn_ids=ids, radius=r, patch_ps=patch_ps, sub_ids=sub_ids, logits=logits, sdf=sdf,
                        q_total=q_all.shape[0], ball_counts=counts)
        return q, sdf
    ids = knn_ids(pts, q, points_per_patch)
    if pts.shape[0] < sub_sample_size:
        # data_loader.py:322-421 query by query: the kd-tree holds a float64 COPY of the original order (cKDTree of a
        # float32 array), the patch is gathered from shape.pts -- which every previous query's sub-sample shuffled
        cur = pts.copy()
        r = np.zeros(q.shape[0], np.float32)
        patch_ps = np.zeros((q.shape[0], points_per_patch, 3), np.float32)
        sub = np.zeros((q.shape[0], sub_sample_size, 3), np.float32)
        for i in range(q.shape[0]):
            ri, pi = patch_radius_and_ps(cur, ids[i:i + 1], q[i:i + 1])
            r[i], patch_ps[i] = ri[0], pi[0]
            sub[i] = subsample_points(rng, cur, q[i], sub_sample_size, uniform, fixed)
        sub_ids = None
    else:
        r, patch_ps = patch_radius_and_ps(pts, ids, q)
        sub_ids = np.stack([subsample_ids(rng, pts, q[i], sub_sample_size, uniform, fixed)
                            for i in range(q.shape[0])]) if q.shape[0] else np.zeros((0, sub_sample_size), np.int64)
        sub = pts[sub_ids]
    logits = model_forward(w, cfg, patch_ps, sub, q, chunk=chunk)
    sdf = post_process(logits, r)
    if return_all:
        return dict(q=q, knn_ids=ids, radius=r, patch_ps=patch_ps, sub_ids=sub_ids, logits=logits, sdf=sdf,
                    q_total=q_all.shape[0])
    return q, sdf


def infer_queries(w, cfg, pts, queries, rng_sub, rng_rot=None, points_per_patch=300, sub_sample_size=1000, chunk=32,
                  return_all=False):
    """The GT-query evaluation pass for one shape (reference source/data_loader.py:322-421 with
    reconstruction=False, driven by source/points_to_surf_eval.py:358-404): query points are given
    (05_query_pts), and every query is augmented with a random rotation
        rand_rot = trimesh.transformations.random_rotation_matrix(self.rng.rand(3))        (:384)
    applied in float64 to the sub-sample (model space), the patch (patch space) and the query point (:385-391),
    each cast back to float32.  ``rng_sub`` = dataset.rng_global_sample, ``rng_rot`` = dataset.rng (both
    LegacyMT19937(seed), carried across shapes); rng_rot None = no rotation.  trimesh is restated in
    oracle/trimesh_restated.py (absent from the image)."""
    from oracle import trimesh_restated as trafo
    pts = np.asarray(pts, dtype=np.float32)
    q = np.asarray(queries, dtype=np.float32).reshape(-1, 3)
    radius = float(cfg.get('patch_radius', 0.0) or 0.0)
    n = q.shape[0]
    if radius > 0.0:
        # fixed-radius models: the patch choice of query i draws from dataset.rng (= rng_rot) BEFORE that query's
        # rand(3) (source/data_loader.py:336, :384); the distance output is not rescaled
        from scipy import spatial
        tree = spatial.cKDTree(pts, 1000)
        r = np.ones(n, np.float32)
        patch_ps = np.zeros((n, points_per_patch, 3), np.float32)
    else:
        ids = knn_ids(pts, q, points_per_patch)
        r, patch_ps = patch_radius_and_ps(pts, ids, q)
    uniform = bool(cfg.get('uniform_subsample', False))
    fixed = bool(cfg.get('fixed_subsample', False))
    sub = np.zeros((n, sub_sample_size, 3), np.float32)
    q_rot = q.copy()
    patch_rot = patch_ps.copy()
    rots = np.zeros((n, 3, 3))
    for i in range(n):
        if radius > 0.0:
            _, patch_ps[i], _ = ball_patch(rng_rot, pts, tree, q[i], radius, points_per_patch)
            patch_rot[i] = patch_ps[i]
        sub_ids = subsample_ids(rng_sub, pts, q[i], sub_sample_size, uniform, fixed)
        sub[i] = pts[sub_ids]
        if rng_rot is not None:
            rand_rot = trafo.random_rotation_matrix(rng_rot.rand(3))
            rots[i] = rand_rot[:3, :3]
            sub[i] = trafo.transform_points(sub[i], rand_rot).astype(np.float32)
            patch_rot[i] = trafo.transform_points(patch_ps[i], rand_rot).astype(np.float32)
            q_rot[i] = trafo.transform_points(np.expand_dims(q[i], 0), rand_rot)[0].astype(np.float32)
    logits = model_forward(w, cfg, patch_rot, sub, q_rot, chunk=chunk)
    sdf = post_process(logits, r)
    if return_all:
        return dict(radius=r, patch_ps=patch_rot, sub=sub, q=q_rot, rot=rots, logits=logits, sdf=sdf)
    return sdf


# --------------------------------------------------------------------------------------
# "next" row f-1: SDF samples -> dense volume -> iterative sign propagation
# --------------------------------------------------------------------------------------

def add_samples_to_volume(vol, pos_ms, val):
    """source/sdf.py:82-111.  The reference clusters samples by voxel with np.unique(axis=0) and keeps,
    per voxel, the sample closest to the voxel centre -- but it measures the distance of every sample to
    ITSELF (:94-95, always 0), so the first sample of each group wins; groups are cut from ``val`` in the
    ORIGINAL order by the sorted-unique counts (:98-101), which is only meaningful when ``pos`` is already
    sorted with one sample per voxel -- exactly what the query grid (a1) produces.  Restated literally."""
    res = vol.shape[0]
    pos_vs = model_space_to_volume_space(pos_ms, res)
    _, counts = np.unique(pos_vs, return_counts=True, axis=0)
    cuts = np.cumsum(counts)[:-1]
    vals = np.split(np.asarray(val), cuts)
    coords = np.split(pos_vs, cuts)
    first = np.array([c[0] for c in coords])
    vol[first[:, 0], first[:, 1], first[:, 2]] = np.array([v[0] for v in vals])
    return vol


def _box_sum_nearest(a, size):
    """sum over a size^3 box with edge replication == scipy.ndimage.convolve(a, ones, mode='nearest')"""
    res = a.shape[0]
    idx = np.arange(res)
    out = a
    for axis in range(3):
        acc = np.zeros_like(out)
        for o in _box_offsets(int(size)):
            acc = acc + np.take(out, np.clip(idx + o, 0, res - 1), axis=axis)
        out = acc
    return out


def propagate_sign(vol, sigma=5, certainty_threshold=13, return_iters=False):
    """source/sdf.py:114-178 (modifies and returns ``vol``, float64)."""
    sgn = np.sign(vol)
    unknown_initially = sgn == 0
    vol[0, :, :] = -1.0
    vol[-1, :, :] = -1.0
    vol[:, 0, :] = -1.0
    vol[:, -1, :] = -1.0
    vol[:, :, 0] = -1.0
    vol[:, :, -1] = -1.0
    iters = 0
    while True:
        unknown_before = sgn == 0
        if unknown_before.sum() == 0:
            break
        new = _box_sum_nearest(sgn, sigma)
        new[np.abs(new) < certainty_threshold] = 0.0
        new = np.sign(new)
        iters += 1
        if (new == 0).sum() >= unknown_before.sum():
            break
        sgn[unknown_initially] = new[unknown_initially]
    zero = vol == 0
    vol[zero] = sgn[zero]
    return (vol, iters) if return_iters else vol


def sdf_volume(query_pts_ms, query_dist_ms, grid_res, sigma, certainty_threshold, clamp=True):
    """the volume source/sdf.py:181-201 hands to marching cubes (float64 [res]^3)."""
    vol = np.zeros((grid_res, grid_res, grid_res))
    vol = add_samples_to_volume(vol, query_pts_ms, query_dist_ms)
    vol = propagate_sign(vol, sigma, certainty_threshold)
    if clamp:
        vol[vol < -1.0] = -1.0
        vol[vol > 1.0] = 1.0
    return vol
