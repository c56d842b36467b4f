"""Offline GF(2) jump-ahead tables for MT19937 (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer 2008).

The raw state sequence x[n] of MT19937 obeys a linear recurrence over GF(2); with F = "advance the 624-word
window by one word", F^J = g_J(F) where g_J(t) = t^J mod phi(t) and phi is the minimal polynomial (degree
19937).  Applying g_J to a window W = x[n..n+623] is a XOR of the windows x[n+i..n+i+623] over the support of
g_J, i.e. a GF(2) convolution of ~20.6k generated words -- embarrassingly parallel on the device.

This script computes phi by Berlekamp-Massey on an output bit stream, the polynomials for jumps of
B * 2^m blocks (B = 64 blocks of 624 words, m = 0..12) and writes their supports to
points2surf_amd/mt_jump_tables.npz.  It self-checks every polynomial against straightforward generation.

    python tools/mt_jump.py
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

N, M = 624, 397
DEG = 19937
BLOCKS_PER_STREAM = 64
LEVELS = 13


def raw_sequence(seed, nwords):
    """untempered state words x[0..nwords) ; x[0..623] = the block after the first twist of init_genrand(seed)"""
    from oracle.p2s_oracle import LegacyMT19937
    g = LegacyMT19937(seed)
    nblocks = (nwords + N - 1) // N
    out = np.empty(nblocks * N, dtype=np.uint32)
    for b in range(nblocks):
        g._twist()
        out[b * N:(b + 1) * N] = g.mt
    return out[:nwords]


def berlekamp_massey(bits):
    """connection polynomial C (int, bit i = c_i, c_0 = 1) and linear complexity L of a GF(2) sequence"""
    C, B, L, m = 1, 1, 0, 1
    R = 0                                  # R = sum_i b_{n-i} << i  (reversed prefix)
    for n, b in enumerate(bits):
        R = (R << 1) | int(b)
        d = (C & R).bit_count() & 1
        if d:
            T = C
            C ^= B << m
            if 2 * L <= n:
                L = n + 1 - L
                B = T
                m = 1
            else:
                m += 1
        else:
            m += 1
    return C, L


def reverse_bits(p, nbits):
    return int(bin(p)[2:].zfill(nbits)[::-1], 2)


_SPREAD = None


def gf2_square(p):
    """square of a GF(2) polynomial = bit spreading"""
    global _SPREAD
    if _SPREAD is None:
        _SPREAD = [int(''.join(c + '0' for c in bin(v)[2:].zfill(8)), 2) >> 1 for v in range(256)]
    out = 0
    b = p.to_bytes((p.bit_length() + 7) // 8 or 1, 'little')
    acc = bytearray(2 * len(b))
    for i, v in enumerate(b):
        s = _SPREAD[v]
        acc[2 * i] = s & 0xff
        acc[2 * i + 1] = s >> 8
    return int.from_bytes(acc, 'little')


def gf2_mod(p, phi, deg):
    while p.bit_length() > deg:
        p ^= phi << (p.bit_length() - deg - 1)
    return p


def pow_t_mod(J, phi, deg):
    """t^J mod phi"""
    p = 1
    for bit in bin(J)[2:]:
        p = gf2_mod(gf2_square(p), phi, deg)
        if bit == '1':
            p = gf2_mod(p << 1, phi, deg)
    return p


def apply_jump(window_words_from, support):
    """window_words_from: raw words x[n .. n+DEG+N) ; returns the window x[n+J .. n+J+623] (exact, incl. word 0)"""
    X = np.asarray(window_words_from, dtype=np.uint32)
    win = np.lib.stride_tricks.sliding_window_view(X, N)[support]
    y = np.bitwise_xor.reduce(win, axis=0)
    # the low 31 bits of word 0 are not part of the 19937-bit state; they follow from words 396 and 623:
    #   x[n+623] = x[n+396] ^ mix(upper(x[n-1]), lower(x[n]))
    v = int(y[623]) ^ int(y[396])
    b0 = v >> 31
    yv = (((v ^ (0x9908b0df if b0 else 0)) << 1) | b0) & 0xffffffff
    y = y.copy()
    y[0] = (int(y[0]) & 0x80000000) | (yv & 0x7fffffff)
    return y


def main():
    t0 = time.time()
    from oracle.p2s_oracle import LegacyMT19937
    g = LegacyMT19937(5489)
    bits = (g.raw(2 * DEG + 64) & 1).astype(np.uint8)
    C, L = berlekamp_massey(bits.tolist())
    assert L == DEG, L
    phi = reverse_bits(C, DEG + 1)         # characteristic polynomial t^L * C(1/t)
    assert phi.bit_length() == DEG + 1
    print('minimal polynomial: degree %d, weight %d (%.1f s)' % (L, phi.bit_count(), time.time() - t0))

    supports = {}
    x = raw_sequence(1234, (BLOCKS_PER_STREAM << (LEVELS - 1)) * N + DEG + 2 * N)
    for m in range(LEVELS):
        J = BLOCKS_PER_STREAM * (1 << m) * N
        gpoly = pow_t_mod(J, phi, DEG)
        sup = np.array([i for i in range(DEG) if (gpoly >> i) & 1], dtype=np.uint16)
        y = apply_jump(x[:DEG + N], sup)
        assert np.array_equal(y, x[J:J + N]), 'jump %d failed' % J
        supports['jump_%d' % m] = sup
        print('level %d: jump %d blocks, support %d, verified (%.1f s)' % (m, J // N, sup.size, time.time() - t0))
    np.savez_compressed(os.path.join(REPO, 'points2surf_amd', 'mt_jump_tables.npz'),
                        blocks_per_stream=np.array(BLOCKS_PER_STREAM), levels=np.array(LEVELS), **supports)
    print('written points2surf_amd/mt_jump_tables.npz')


if __name__ == '__main__':
    main()
