"""CPU checks of the executable model of the bf16 encoder (tests/bf16_model.py): with the rounding switched off it
is the oracle's PointNetfeat (so the GPU test compares the kernel with the right graph), and the rounding itself is
round-to-nearest-even bfloat16."""
import numpy as np

from oracle import p2s_oracle as O
from points2surf_amd import synth
from tests import bf16_model as bm


def test_bf16_rounding_is_nearest_even():
    x = np.array([1.0, 1.00390625, 1.01171875, 3.14159, -2.5, 1e-3, 0.0], dtype=np.float32)
    got = bm.bf16(x)
    assert got.dtype == np.float32
    assert got[0] == 1.0 and got[1] == 1.0            # 1 + 2^-8 is a tie: to even
    assert got[2] == np.float32(1.015625)             # 1 + 3 * 2^-8 is a tie: to even (mantissa ...10)
    assert np.all((got.view(np.uint32) & 0xffff) == 0)
    assert np.all(np.abs(got - x) <= np.abs(x) * 2.0 ** -8)


def test_model_without_rounding_is_the_oracle(monkeypatch):
    w, cfg = synth.make_weights('p2s_max')
    w32 = {k: np.asarray(v, dtype=np.float32) for k, v in w.items()}
    x = np.random.default_rng(0).normal(0, 0.3, (3, 300, 3)).astype(np.float32)
    ref, _ = O.pointnetfeat_forward(x, w32, 'feat_local', False, True)
    monkeypatch.setattr(bm, 'bf16', lambda v: np.asarray(v, dtype=np.float32))
    got = bm.encoder_features(w32, 'feat_local', x)
    assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-5
    monkeypatch.undo()
    q = bm.encoder_features(w32, 'feat_local', x)
    dev = np.abs(q - ref).max() / np.abs(ref).max()
    assert 1e-4 < dev < 5e-2                          # bf16 operands: a real but bounded deviation
