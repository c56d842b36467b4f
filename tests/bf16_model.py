"""Executable model of the bf16 encoder mode (p2s_chain_bf16.hip): the per-point layers of PointNetfeat with weights and
activations rounded to bf16 exactly where the kernel rounds them (fp32 accumulation), everything else float32.
Test helper: separates "the kernel computes what it is meant to" (tight tolerance against this model) from "how far is
bf16 from the fp32 reference" (reported, loose bound)."""
import numpy as np

from points2surf_amd.weights import fold_affine

F32 = np.float32


def bf16(x):
    """round-to-nearest-even to bfloat16, returned as float32"""
    u = np.ascontiguousarray(x, dtype=F32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(F32).reshape(np.shape(x))


def _affine(w, lin, bn):
    W, b = fold_affine(w, lin, bn)
    return W.astype(F32), b.astype(F32)


def encoder_features(w, pre, x):
    """x [B, P, 3] float32 (already centred) -> pooled feature [B, 1024]; mirrors one (item, encoder) workgroup"""
    relu = lambda v: np.maximum(v, F32(0))
    W0a, b0a = _affine(w, pre + '.conv0a', pre + '.bn0a')
    W0b, b0b = _affine(w, pre + '.conv0b', pre + '.bn0b')
    S1, sb1 = _affine(w, pre + '.stn2.conv1', pre + '.stn2.bn1')
    S2, sb2 = _affine(w, pre + '.stn2.conv2', pre + '.stn2.bn2')
    S3, sb3 = _affine(w, pre + '.stn2.conv3', pre + '.stn2.bn3')
    F1, fb1 = _affine(w, pre + '.stn2.fc1', pre + '.stn2.bn4')
    F2, fb2 = _affine(w, pre + '.stn2.fc2', pre + '.stn2.bn5')
    F3, fb3 = _affine(w, pre + '.stn2.fc3', None)
    M1, mb1 = _affine(w, pre + '.conv1', pre + '.bn1')
    M2, mb2 = _affine(w, pre + '.conv2', pre + '.bn2')
    M3, mb3 = _affine(w, pre + '.conv3', pre + '.bn3')

    def layer(h, W, b, act=True):                       # bf16 operands, fp32 accumulate, fp32 bias
        y = bf16(h) @ bf16(W).T + b
        return relu(y) if act else y

    h0 = bf16(relu(x @ W0a.T + b0a))                    # first layer fp32, stored as bf16
    h0 = bf16(layer(h0, W0b, b0b))
    # STN trunk -> pooled (bias / ReLU after the max) -> fp32 heads -> trans2
    t = bf16(layer(h0, S1, sb1))
    t = bf16(layer(t, S2, sb2))
    g = relu((bf16(t) @ bf16(S3).T).max(axis=1) + sb3)
    g = relu(g @ F1.T + fb1)
    g = relu(g @ F2.T + fb2)
    T = (g @ F3.T + fb3 + np.eye(64, dtype=F32).reshape(1, 4096)).reshape(-1, 64, 64)
    # main pass: conv1 applied to bmm(trans2, x) == (W1 . trans2) x, folded in fp32, then rounded to bf16
    W1p = np.einsum('oc,bcj->boj', M1, T).astype(F32)    # [B, 64 out, 64 in]
    h1 = bf16(relu(np.einsum('bpj,boj->bpo', h0, bf16(W1p)) + mb1))
    h2 = bf16(layer(h1, M2, mb2))
    return (bf16(h2) @ bf16(M3).T).max(axis=1) + mb3
