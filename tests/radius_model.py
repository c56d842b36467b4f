"""CPU model of the device algorithm for the fixed-radius patches (points2surf_amd/csrc/p2s_ball.hip), the way
tests/wchoice_model.py models the weighted sub-sample: the numpy-legacy shuffle behind

    rng.choice(np.arange(point_count), points_per_patch, replace=False)        (reference source/base/point_cloud.py:181-183)
    = RandomState.permutation(point_count)[:points_per_patch]
    = for i = n-1 .. 1:  j = rk_interval(i)  (32-bit words, masked rejection);  swap(a[i], a[j])

resolved 64 raw words at a time (the kernels fetch 256 and resolve the four slots one after the other): word j is accepted iff
(w_j & mask(i - A_j)) <= i - A_j with A_j = accepted words before it -- a fixed point that is reached from the left (word 0
is right after one evaluation, word j after j + 1 at the latest) and detected when an evaluation changes nothing."""
import numpy as np


def smear(i):
    m = int(i)
    for s in (1, 2, 4, 8, 16):
        m |= m >> s
    return m


def _smear_v(x):
    x = x.copy()
    for s in (1, 2, 4, 8, 16):
        x |= x >> s
    return x


def walk_blocks(words, pos, n, lanes=64, guess=True):
    """-> (position after the shuffle of n elements, [(i, j)] in execution order, blocks, evaluations).
    Word j of a block belongs to step i - A_j: its mask and its threshold follow from the accepted words before it, so
    a block runs across the powers of two and only the end of the shuffle cuts it short."""
    i = n - 1
    swaps = []
    blocks = evals = 0
    lane = np.arange(lanes)
    while i >= 1:
        w = words[pos:pos + lanes].astype(np.int64)
        bits = int(smear(i)).bit_length()

        def evaluate(before):
            thr = i - before
            v = w & _smear_v(np.maximum(thr, 1))
            return (thr >= 1) & (v <= thr), v
        acc, v = evaluate((lane * (i + 1)) >> bits if guess else np.zeros(lanes, np.int64))
        while True:
            evals += 1
            before = np.concatenate(([0], np.cumsum(acc)[:-1]))
            acc2, v = evaluate(before)
            if np.array_equal(acc2, acc):
                break
            acc = acc2
        before = np.concatenate(([0], np.cumsum(acc)[:-1]))
        total = int(acc.sum())
        if total >= i:
            last = int(lane[acc & (before == i - 1)][0])
            consumed, steps = last + 1, i
        else:
            consumed, steps = lanes, total
        for l in lane[acc & (before < steps)]:
            swaps.append((i - int(before[l]), int(v[l])))
        pos += consumed
        i -= steps
        blocks += 1
    return pos, swaps, blocks, evals


def permutation(words, pos, n, lanes=64):
    pos2, swaps, _, _ = walk_blocks(words, pos, n, lanes=lanes)
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
