"""CPU model of the device algorithm for the fixed-radius patches (points2surf_amd/csrc/p2s_ball.hip), the way
tests/wchoice_model.py models the weighted sub-sample: the numpy-legacy shuffle behind

    rng.choice(np.arange(point_count), points_per_patch, replace=False)        (reference source/base/point_cloud.py:181-183)
    = RandomState.permutation(point_count)[:points_per_patch]
    = for i = n-1 .. 1:  j = rk_interval(i)  (32-bit words, masked rejection);  swap(a[i], a[j])

resolved 64 raw words at a time: lane l accepts its word iff (w & mask) <= i - (accepted words before l) -- a fixed point
that is reached from the left (lane 0 is right after one evaluation, lane l after l + 1 at the latest) and detected when
an evaluation changes nothing.  A block ends early where the mask changes (i crosses a power of two)."""
import numpy as np


def smear(i):
    m = int(i)
    for s in (1, 2, 4, 8, 16):
        m |= m >> s
    return m


def walk_blocks(words, pos, n, lanes=64):
    """-> (position after the shuffle of n elements, [(i, j)] in execution order, blocks, evaluations)"""
    i = n - 1
    swaps = []
    blocks = evals = 0
    lane = np.arange(lanes)
    while i >= 1:
        mask = smear(i)
        lim = i - (mask >> 1)                       # steps left under this mask
        v = (words[pos:pos + lanes] & np.uint32(mask)).astype(np.int64)
        acc = v <= i
        while True:
            evals += 1
            before = np.concatenate(([0], np.cumsum(acc)[:-1]))
            acc2 = v <= i - before
            if np.array_equal(acc2, acc):
                break
            acc = acc2
        before = np.concatenate(([0], np.cumsum(acc)[:-1]))
        total = int(acc.sum())
        if total >= lim:
            last = int(lane[acc & (before == lim - 1)][0])
            consumed, steps = last + 1, lim
        else:
            consumed, steps = lanes, total
        for l in lane[acc & (before < steps)]:
            swaps.append((i - int(before[l]), int(v[l])))
        pos += consumed
        i -= steps
        blocks += 1
    return pos, swaps, blocks, evals


def permutation(words, pos, n):
    pos2, swaps, _, _ = walk_blocks(words, pos, n)
    a = np.arange(n, dtype=np.int64)
    for i, j in swaps:
        a[i], a[j] = a[j], a[i]
    return a, pos2


def raw_words(rs, count):
    """the next `count` 32-bit outputs of a RandomState WITHOUT advancing it"""
    st = rs.get_state()
    w = rs.randint(0, 2 ** 32, size=count, dtype=np.uint32)
    rs.set_state(st)
    return w
