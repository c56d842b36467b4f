"""CPU models of the sign-propagation kernel's bit / index arithmetic (no GPU)."""
import numpy as np


def test_sweep_kernel_byte_compare_and_staging_math():
    """CPU model of two pieces of index / bit arithmetic of vol_sweep_kernel (p2s_volume.hip):
    (1) lerp_ge_consts: bit 7 of v_lerp_u8's byte (x + K + R) >> 1 is x >= c for the biased sums x <= 250, any c;
    (2) the staging map (thread -> half an x plane's row, z dword; iteration -> plane, half) covers every dword of the
        halo'd tile exactly once and lands at LDS address row * 19 + d."""
    def consts(c):
        kk = 255 if c <= 0 else (0 if c >= 256 else 255 - c)
        return kk, (0 if c >= 256 else 1)
    x = np.arange(0, 251)
    for c in range(-5, 300):
        K, R = consts(c)
        assert np.array_equal((((x + K + R) >> 1) & 0x80) != 0, x >= c), c
    # thresholds the kernel derives: sigma 1..5, certainty threshold 0.5 .. 130
    for nt in (1, 2, 3, 4, 5):
        bias = nt ** 3
        for thr in (0.5, 1.0, 1.5, 13.0, 26.0, 124.9, 125.0, 130.0):
            T = int(np.ceil(min(thr, 1024.0))) if thr > 1.0 else 1
            a = np.arange(-bias, bias + 1)                       # the unbiased integer sums
            pos_ref, neg_ref = a >= max(thr, 1e-9) if thr > 1 else a > 0, a <= -thr if thr > 1 else a < 0
            Kp, Rp = consts(bias + T)
            Kn, Rn = consts(bias - T + 1)
            acc = a + bias
            pos = (((acc + Kp + Rp) >> 1) & 0x80) != 0
            neg = ~((((acc + Kn + Rn) >> 1) & 0x80) != 0)
            # the reference: new_sign = sign(sum) where |sum| >= threshold (source/sdf.py:147-150)
            ref = np.where(np.abs(a) < thr, 0, np.sign(a))
            assert np.array_equal(pos.astype(int) - neg.astype(int), ref), (nt, thr)
            del pos_ref, neg_ref
    VT_X, VT_Y, VT_Z, H = 8, 16, 64, 2
    VA_X, VA_Y, VA_ZD = VT_X + 2 * H, VT_Y + 2 * H, VT_Z // 4 + 2
    VA_ZS, ROWS = VA_ZD + 1, VA_Y // 2
    seen = np.zeros(VA_X * VA_Y * VA_ZS, dtype=int)
    for tid in range(ROWS * VA_ZD):
        r0, d = divmod(tid, VA_ZD)
        for it in range(2 * VA_X):
            ax, ay = it >> 1, r0 + (it & 1) * ROWS
            lds = r0 * VA_ZS + d + ((it >> 1) * VA_Y + (it & 1) * ROWS) * VA_ZS
            assert lds == (ax * VA_Y + ay) * VA_ZS + d
            seen[lds] += 1
    want = np.zeros_like(seen).reshape(VA_X * VA_Y, VA_ZS)
    want[:, :VA_ZD] = 1
    assert np.array_equal(seen, want.reshape(-1)) and ROWS * VA_ZD <= 256


def test_activation_mask_of_a_changed_voxel():
    """which of the 27 tiles around a tile must run in the next sweep when ONE voxel (x, y, z) of it changed: those whose
    halo'd box (halo 2) contains the voxel.  Model of the kernel's bit arithmetic (py * mz products, 3-bit slots):
    bit (dz+1) + 3 (dy+1) + 9 (dx+1)."""
    VT_X, VT_Y, VT_Z, H = 8, 16, 64, 2
    for x in range(VT_X):
        for y in range(VT_Y):
            for z in range(VT_Z):
                tzd, byte = z // 4, z % 4
                diff = 0xff << (8 * byte)                                    # the changed byte of this thread's dword
                d_all = diff
                d_xlo = diff if x < H else 0
                d_xhi = diff if x >= VT_X - H else 0
                z_lo = 0x0000ffff if tzd == 0 else 0
                z_hi = 0xffff0000 if tzd == VT_Z // 4 - 1 else 0
                py = (1 if y < H else 0) | 8 | (64 if y >= VT_Y - H else 0)
                mz = lambda d: (1 if d & z_lo else 0) | (2 if d else 0) | (4 if d & z_hi else 0)
                m = (py * mz(d_xlo)) | ((py * mz(d_all)) << 9) | ((py * mz(d_xhi)) << 18)
                want = 0
                for dx in (-1, 0, 1):
                    for dy in (-1, 0, 1):
                        for dz in (-1, 0, 1):
                            okx = dx == 0 or (dx < 0 and x < H) or (dx > 0 and x >= VT_X - H)
                            oky = dy == 0 or (dy < 0 and y < H) or (dy > 0 and y >= VT_Y - H)
                            okz = dz == 0 or (dz < 0 and z < H) or (dz > 0 and z >= VT_Z - H)
                            if okx and oky and okz:
                                want |= 1 << ((dz + 1) + 3 * (dy + 1) + 9 * (dx + 1))
                assert m == want, (x, y, z, bin(m), bin(want))
