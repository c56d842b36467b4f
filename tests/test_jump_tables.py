"""The shipped GF(2) jump-ahead tables (points2surf_amd/mt_jump_tables.npz) against the oracle's MT19937:
every polynomial is recomputed (t^J mod phi, phi from Berlekamp-Massey on the oracle's output) and the first levels
are applied to a real window and compared with plain generation."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tools'))


def test_jump_tables_match_recomputation_and_plain_generation():
    import mt_jump as mj
    from oracle.p2s_oracle import LegacyMT19937
    t = np.load(os.path.join(REPO, 'points2surf_amd', 'mt_jump_tables.npz'))
    base, levels = int(t['blocks_per_stream']), int(t['levels'])
    assert base == mj.BLOCKS_PER_STREAM and levels == mj.LEVELS >= 13

    g = LegacyMT19937(5489)
    bits = (g.raw(2 * mj.DEG + 64) & 1).astype(np.uint8)
    C, L = mj.berlekamp_massey(bits.tolist())
    assert L == mj.DEG
    phi = mj.reverse_bits(C, mj.DEG + 1)

    n_plain = 6                                      # levels checked against plain generation (2048 blocks at most)
    x = mj.raw_sequence(4321, (base << (n_plain - 1)) * mj.N + mj.DEG + 2 * mj.N)
    for m in range(levels):
        J = base * (1 << m) * mj.N
        gp = mj.pow_t_mod(J, phi, mj.DEG)
        sup = np.array([i for i in range(mj.DEG) if (gp >> i) & 1], dtype=np.uint16)
        assert np.array_equal(sup, t['jump_%d' % m]), 'level %d differs from t^J mod phi' % m
        if m < n_plain:
            y = mj.apply_jump(x[:mj.DEG + mj.N], sup)
            assert np.array_equal(y, x[J:J + mj.N]), 'level %d: jump != plain generation' % m
