"""Reference ``state_dict`` -> engine weight blob (one-time, host side).

* BatchNorm1d (eval: y = (x-mean)/sqrt(var+1e-5)*gamma+beta) is folded into the preceding
  Conv1d(k=1)/Linear in float64 and rounded once to float32.  Order in the reference is
  conv -> BN -> ReLU everywhere except PointNetfeat.conv3 (conv -> BN -> max, reference
  source/points_to_surf_model.py:201-203); gamma may be negative, hence the engine pools
  AFTER the folded affine.
* GEMM operands are packed in the B-fragment order of v_mfma_f32_32x32x2_f32 with the
  k-permutation the kernels use:  packed[n/32][k/8][lane][t] = W[k = 8*(k/8) + 4*(lane>>5) + t][n = 32*(n/32) + (lane&31)]
* the STN identity (``x + eye(64)``, :66-67) and the QSTN identity quaternion (:125-126) are
  folded into the fc3 bias.
"""
import ctypes

import numpy as np

from .model_spec import strip_module_prefix

BN_EPS = 1e-5

_ENC_FIELDS = ['w0a', 'b0a', 'w0b', 'b0b', 's1', 'sb1', 's2', 'sb2', 's3', 'sb3', 'sf1', 'sfb1', 'sf2', 'sfb2',
               'sf3', 'sfb3', 'm1t', 'mb1', 'm2', 'mb2', 'm3', 'mb3']
_QSTN_FIELDS = ['c1', 'cb1', 'c2', 'cb2', 'c3', 'cb3', 'f1', 'fb1', 'f2', 'fb2', 'f3', 'fb3']
_DEC_FIELDS = ['d1l', 'db1l', 'd1g', 'db1g', 'd2', 'db2', 'd3', 'db3', 'd4', 'db4']


class EncoderOffsets(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in _ENC_FIELDS]


class QstnOffsets(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in _QSTN_FIELDS]


class WeightOffsets(ctypes.Structure):
    """mirror of ``p2s_weight_offsets`` (include/p2s_hip.h)"""
    _fields_ = [('enc', EncoderOffsets * 2), ('qstn', QstnOffsets)] + [(n, ctypes.c_uint64) for n in _DEC_FIELDS]


class ModelCfg(ctypes.Structure):
    """mirror of ``p2s_model_cfg`` (include/p2s_hip.h)"""
    _fields_ = [('net_size', ctypes.c_int32), ('points_per_patch', ctypes.c_int32),
                ('sub_sample_size', ctypes.c_int32), ('output_dim', ctypes.c_int32),
                ('use_point_stn', ctypes.c_int32), ('shared_transformer', ctypes.c_int32),
                ('weighted_subsample', ctypes.c_int32), ('encoder_bf16', ctypes.c_int32),
                ('fixed_subsample', ctypes.c_int32), ('single_transformer', ctypes.c_int32),
                ('patch_radius', ctypes.c_double), ('sym_sum', ctypes.c_int32), ('reserved', ctypes.c_int32 * 3)]


def _np(v):
    if hasattr(v, 'detach'):
        v = v.detach().cpu().numpy()
    return np.asarray(v)


def fold_affine(w, lin, bn=None):
    """returns (W [out,in] float64, b [out] float64) of ``bn(lin(x))``."""
    W = _np(w[lin + '.weight']).astype(np.float64)
    if W.ndim == 3:
        W = W[:, :, 0]
    b = _np(w[lin + '.bias']).astype(np.float64)
    if bn is not None:
        gamma = _np(w[bn + '.weight']).astype(np.float64)
        beta = _np(w[bn + '.bias']).astype(np.float64)
        mean = _np(w[bn + '.running_mean']).astype(np.float64)
        var = _np(w[bn + '.running_var']).astype(np.float64)
        s = gamma / np.sqrt(var + BN_EPS)
        W = W * s[:, None]
        b = (b - mean) * s + beta
    return W, b


def pack_b(Wkn):
    """[K, N] -> flat packed B fragments [N/32][K/8][64][4] (float32)."""
    Wkn = np.asarray(Wkn, dtype=np.float32)
    K, N = Wkn.shape
    assert K % 8 == 0 and N % 32 == 0, (K, N)
    # k = 8*kg + 4*kk + t ; n = 32*nt + j ; lane = 32*kk + j
    a = Wkn.reshape(K // 8, 2, 4, N // 32, 32)          # [kg, kk, t, nt, j]
    a = a.transpose(3, 0, 1, 4, 2)                      # [nt, kg, kk, j, t]
    return np.ascontiguousarray(a).reshape(-1)


def unpack_b(packed, K, N):
    """inverse of pack_b (tests)."""
    a = np.asarray(packed, dtype=np.float32).reshape(N // 32, K // 8, 2, 32, 4)
    return np.ascontiguousarray(a.transpose(1, 2, 4, 0, 3)).reshape(K, N)


class _Blob:
    def __init__(self):
        self.parts = []
        self.n = 0

    def add(self, arr):
        arr = np.ascontiguousarray(np.asarray(arr, dtype=np.float32).reshape(-1))
        off = self.n
        self.parts.append(arr)
        pad = (-arr.size) % 64                       # keep every tensor 256-byte aligned
        if pad:
            self.parts.append(np.zeros(pad, dtype=np.float32))
        self.n += arr.size + pad
        return off

    def finish(self):
        return np.concatenate(self.parts) if self.parts else np.zeros(0, np.float32)


def _add_gemm(blob, w, lin, bn, extra_bias=None):
    W, b = fold_affine(w, lin, bn)
    if extra_bias is not None:
        b = b + extra_bias
    return blob.add(pack_b(W.T)), blob.add(b)


def _add_plain(blob, w, lin, bn, extra_bias=None):
    """small layers kept as plain [K][N] (first K=3 conv, fc4, QSTN fc3)."""
    W, b = fold_affine(w, lin, bn)
    if extra_bias is not None:
        b = b + extra_bias
    return blob.add(W.T), blob.add(b)


def build_blob(state_dict, cfg):
    """state_dict (reference key layout, with or without ``module.``) + cfg dict ->
    (blob float32 [n], WeightOffsets, ModelCfg)."""
    w = strip_module_prefix(state_dict)
    n = int(cfg.get('net_size', 1024))
    if n != 1024:
        raise ValueError('engine is specialised for net_size 1024 (got %d)' % n)
    use_point_stn = bool(cfg.get('use_point_stn', False))
    shared = bool(cfg.get('shared_transformer', False))
    if not bool(cfg.get('use_feat_stn', True)):
        # train --use_feat_stn 0 (set by none of the reference's scripts): PointNetfeat without the 64x64 feature
        # transform (reference source/points_to_surf_model.py:151-153,194-199).  The engine keeps its one code path and
        # gets the identity from an all-zero STN: trans = 0 + bias(I) exactly, and the fold W1' = conv1 . I is exact
        # (one non-zero product per sum).  The STN trunk still runs -- not optimised, no script uses it.
        w = dict(w)
        for pre in (('feat_local_global',) if bool(cfg.get('single_transformer', False)) else ('feat_local', 'feat_global')):
            s = pre + '.stn2'
            for name, shape in (('conv1', (64, 64, 1)), ('conv2', (128, 64, 1)), ('conv3', (1024, 128, 1)),
                                ('fc1', (512, 1024)), ('fc2', (256, 512)), ('fc3', (4096, 256))):
                w[s + '.' + name + '.weight'] = np.zeros(shape, np.float32)
                w[s + '.' + name + '.bias'] = np.zeros(shape[0], np.float32)
            for bn, c in (('bn1', 64), ('bn2', 128), ('bn3', 1024), ('bn4', 512), ('bn5', 256)):
                w[s + '.' + bn + '.weight'] = np.ones(c, np.float32)
                w[s + '.' + bn + '.bias'] = np.zeros(c, np.float32)
                w[s + '.' + bn + '.running_mean'] = np.zeros(c, np.float32)
                w[s + '.' + bn + '.running_var'] = np.ones(c, np.float32)
    single = bool(cfg.get('single_transformer', False))
    sym_op = cfg.get('sym_op', 'max')
    if sym_op not in ('max', 'sum'):
        raise ValueError("Unsupported symmetric operation: %s" % sym_op)        # reference points_to_surf_model.py:175

    blob = _Blob()
    offs = WeightOffsets()
    # single_transformer: ONE encoder over cat(patch, sub-sample); the engine runs it as its usual two branches
    # (patch points / sub-sample points) with the same weights and takes the max of the two pools
    for e, pre in enumerate(('feat_local_global', 'feat_local_global') if single else ('feat_local', 'feat_global')):
        o = offs.enc[e]
        o.w0a, o.b0a = _add_plain(blob, w, pre + '.conv0a', pre + '.bn0a')
        o.w0b, o.b0b = _add_gemm(blob, w, pre + '.conv0b', pre + '.bn0b')
        s = pre + '.stn2'
        o.s1, o.sb1 = _add_gemm(blob, w, s + '.conv1', s + '.bn1')
        o.s2, o.sb2 = _add_gemm(blob, w, s + '.conv2', s + '.bn2')
        o.s3, o.sb3 = _add_gemm(blob, w, s + '.conv3', s + '.bn3')
        o.sf1, o.sfb1 = _add_gemm(blob, w, s + '.fc1', s + '.bn4')
        o.sf2, o.sfb2 = _add_gemm(blob, w, s + '.fc2', s + '.bn5')
        o.sf3, o.sfb3 = _add_gemm(blob, w, s + '.fc3', None, extra_bias=np.eye(64).reshape(-1))
        o.m1t, o.mb1 = _add_gemm(blob, w, pre + '.conv1', pre + '.bn1')
        o.m2, o.mb2 = _add_gemm(blob, w, pre + '.conv2', pre + '.bn2')
        o.m3, o.mb3 = _add_gemm(blob, w, pre + '.conv3', pre + '.bn3')
    if use_point_stn:
        q = offs.qstn
        # shared: one QSTN over cat(patch, sub-sample) (model.point_stn); otherwise the QSTN of feat_global, which
        # sees the sub-sample only (reference source/points_to_surf_model.py:267-269, :283-284)
        s = 'feat_local_global.stn1' if single else ('point_stn' if shared else 'feat_global.stn1')
        q.c1, q.cb1 = _add_plain(blob, w, s + '.conv1', s + '.bn1')
        q.c2, q.cb2 = _add_gemm(blob, w, s + '.conv2', s + '.bn2')
        q.c3, q.cb3 = _add_gemm(blob, w, s + '.conv3', s + '.bn3')
        q.f1, q.fb1 = _add_gemm(blob, w, s + '.fc1', s + '.bn4')
        q.f2, q.fb2 = _add_gemm(blob, w, s + '.fc2', s + '.bn5')
        q.f3, q.fb3 = _add_plain(blob, w, s + '.fc3', None, extra_bias=np.array([1.0, 0, 0, 0]))
    if single:
        # fc1_local_global 1024 -> 1024 as two 512-column halves: the decoder's two fc1 GEMMs read the same feature
        W1, b1 = fold_affine(w, 'fc1_local_global', 'bn1_local_global')
        offs.d1l, offs.db1l = blob.add(pack_b(W1[:512].T)), blob.add(b1[:512])
        offs.d1g, offs.db1g = blob.add(pack_b(W1[512:].T)), blob.add(b1[512:])
    else:
        offs.d1l, offs.db1l = _add_gemm(blob, w, 'fc1_local', 'bn1_local')
        offs.d1g, offs.db1g = _add_gemm(blob, w, 'fc1_global', 'bn1_global')
    offs.d2, offs.db2 = _add_gemm(blob, w, 'fc2', 'bn2')
    offs.d3, offs.db3 = _add_gemm(blob, w, 'fc3', 'bn3')
    offs.d4, offs.db4 = _add_plain(blob, w, 'fc4', None)

    mc = ModelCfg()
    mc.net_size = n
    mc.points_per_patch = int(cfg.get('points_per_patch', 300))
    mc.sub_sample_size = int(cfg.get('sub_sample_size', 1000))
    mc.output_dim = int(cfg.get('output_dim', 2))
    mc.use_point_stn = int(use_point_stn)
    mc.shared_transformer = int(shared)
    mc.weighted_subsample = int(not bool(cfg.get('uniform_subsample', False)))
    mc.fixed_subsample = int(bool(cfg.get('fixed_subsample', False)))
    mc.single_transformer = int(single)
    mc.sym_sum = int(sym_op == 'sum')
    mc.patch_radius = max(float(cfg.get('patch_radius', 0.0) or 0.0), 0.0)     # the float64 of the reference's Python float
    mc.encoder_bf16 = int(cfg.get('encoder_bf16', 0) or 0)      # 0 fp32, 1 bf16, 2 / 3 split bf16 (pieces per operand), 4 fp16 pair
    if mc.output_dim not in (1, 2):
        raise ValueError('engine supports outputs imp_surf (pred_dim 1) or imp_surf_magnitude + imp_surf_sign (pred_dim 2)')
    return blob.finish(), offs, mc
