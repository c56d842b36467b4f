"""Executable model of the device algorithm for the distance-weighted sub-sample (p2s_wchoice.hip).

Test helper (CPU): mirrors, step by step, what the HIP kernels compute so that the *algorithmic* claims can be
checked against numpy without a GPU:
  * np.sum(float32) = 8192-element buffer chunks, each summed by numpy's pairwise routine (128-element leaves with
    8 strided accumulators), chunks added sequentially;
  * the float64 cumsum of the (float32) probabilities is exact, hence order independent;
  * searchsorted(cdf, x, 'right') through a guide table T[b] = #{cdf_i <= b/K};
  * iterations >= 2 (found entries zeroed) located through the sorted found list instead of a new cumsum.
"""
import numpy as np

F32 = np.float32
PW_BLOCK = 128
NP_BUFSIZE = 8192


def pairwise_plan(n):
    """numpy's float32 add.reduce over n contiguous elements as a data-flow plan: buffer chunks of 8192 elements are
    summed by the pairwise routine (split at n/2 rounded down to a multiple of 8 until <= 128 elements remain) and
    the chunk sums are chained left to right.  Returns leaves [(start, len, node)], ops [(level, dst, a, b)] sorted
    by level, the root node and the node count."""
    leaves, ops = [], []
    count = [0]

    def new_node():
        count[0] += 1
        return count[0] - 1

    def rec(start, m):
        if m <= PW_BLOCK:
            d = new_node()
            leaves.append((start, m, d))
            return d, 0
        n2 = m // 2
        n2 -= n2 % 8
        a, la = rec(start, n2)
        b, lb = rec(start + n2, m - n2)
        d = new_node()
        ops.append((max(la, lb) + 1, d, a, b))
        return d, max(la, lb) + 1

    acc = None
    for c0 in range(0, n, NP_BUFSIZE):
        r, lr = rec(c0, min(NP_BUFSIZE, n - c0))
        if acc is None:
            acc, lvl = r, lr
        else:
            lvl = max(lvl, lr) + 1
            d = new_node()
            ops.append((lvl, d, acc, r))
            acc = d
    ops.sort(key=lambda o: o[0])
    return leaves, ops, acc, count[0]


def leaf_sum(a):
    n = len(a)
    if n < 8:
        r = F32(0.0)
        for v in a:
            r = F32(r + v)
        return r
    body = n - n % 8
    r = a[:body].reshape(-1, 8)
    acc = r[0].copy()
    for row in r[1:]:
        acc = (acc + row).astype(F32)
    res = F32(F32(F32(acc[0] + acc[1]) + F32(acc[2] + acc[3])) + F32(F32(acc[4] + acc[5]) + F32(acc[6] + acc[7])))
    for v in a[body:]:
        res = F32(res + v)
    return res


def numpy_sum_f32(a, plan=None):
    leaves, ops, root, n_nodes = plan or pairwise_plan(len(a))
    nodes = np.zeros(n_nodes, dtype=F32)
    for s, m, d in leaves:
        nodes[d] = leaf_sum(a[s:s + m])
    for _, d, x, y in ops:
        nodes[d] = F32(nodes[x] + nodes[y])
    return nodes[root]


def probabilities(pts, q):
    """float32 p of source/base/utils.py:200-208 with the explicit summation order"""
    pts = np.asarray(pts, F32)
    q = np.asarray(q, F32)
    d = q[None, :] - pts
    sq = (d * d).astype(F32)
    dist = np.sqrt(F32(F32(sq[:, 0] + sq[:, 1]) + sq[:, 2])).astype(F32)
    dmax = dist.max()
    pc = np.clip((F32(1.0) - (F32(1.5) * (dist / dmax).astype(F32)).astype(F32)).astype(F32), F32(0.05), F32(1.0))
    return (pc / numpy_sum_f32(pc)).astype(F32)


class Tables:
    def __init__(self, p32, K=None):
        n = p32.size
        self.n = n
        self.K = K or (1 << int(np.ceil(np.log2(n))))
        p = p32.astype(np.float64)
        # any summation order gives the same float64 values (all partial sums are exact)
        blocks = [np.sum(p[i:i + 1000]) for i in range(0, n, 1000)]
        self.Stot = float(np.sum(np.array(blocks)[::-1]))
        self.S = np.cumsum(p)
        assert self.S[-1] == self.Stot
        cdf = self.S / self.Stot
        c = np.ceil(cdf * self.K).astype(np.int64)
        cprev = np.concatenate([[0], c[:-1]])
        self.T = np.zeros(self.K, dtype=np.int64)
        for i in np.nonzero(c > cprev)[0]:
            self.T[cprev[i]:min(c[i], self.K)] = i


def locate(tb, x, sid, V, C, Stot_cur):
    """smallest i with fl((S_i - C(i)) / Stot_cur) > x where the found ids `sid` (sorted) carry zero mass"""
    m = len(sid)
    k = 0
    if m:
        lo_k, hi_k = 0, m        # largest k in [0, m] with k == 0 or V[k-1]/Stot_cur <= x
        while lo_k < hi_k:
            mid = (lo_k + hi_k + 1) // 2
            if V[mid - 1] / Stot_cur <= x:
                lo_k = mid
            else:
                hi_k = mid - 1
        k = lo_k
    lo = sid[k - 1] + 1 if k else 0
    hi = sid[k] if k < m else tb.n
    Ck = C[k - 1] if k else 0.0
    t = x * Stot_cur + Ck
    b = min(tb.K - 1, max(0, int((t / tb.Stot) * tb.K)))
    i = min(max(int(tb.T[b]), lo), hi - 1)

    def pred(j):
        return (tb.S[j] - Ck) / Stot_cur > x
    if pred(i):
        while i > lo and pred(i - 1):
            i -= 1
    else:
        while True:
            i += 1
            assert i < hi
            if pred(i):
                break
    return i


def choice_noreplace(tb, rand, size):
    """device-algorithm version of RandomState.choice(n, size, replace=False, p); rand(m) -> m doubles"""
    found = []
    fS, fP = [], []
    sid = np.zeros(0, np.int64)
    V = C = np.zeros(0)
    Stot_cur = tb.Stot
    while len(found) < size:
        m = size - len(found)
        x = rand(m)
        bins = [locate(tb, xv, sid, V, C, Stot_cur) for xv in x]
        first = {}
        for d, b in enumerate(bins):
            if b not in first:
                first[b] = d
        for d, b in enumerate(bins):
            if first[b] == d:
                found.append(b)
                fS.append(tb.S[b])
                fP.append(tb.S[b] - (tb.S[b - 1] if b else 0.0))
        order = np.argsort(np.array(found))
        sid = np.array(found)[order]
        sS = np.array(fS)[order]
        C = np.cumsum(np.array(fP)[order])
        V = sS - C
        Stot_cur = tb.Stot - C[-1]
    return np.array(found, dtype=np.int64)


def spec_redraws(tb, xs, nsel, W, look=64):
    """Model of wc_spec_kernel: for the candidate starts d = 0..W-1 (in doubles) of one query, the number of
    redraws R(d) of ``choice`` when its first draw is xs[d], or 255 where the kernel leaves the decision to the
    complete algorithm.  xs: at least W + nsel + 2*look doubles of the stream."""
    cdf = tb.S / tb.Stot
    nb = W + nsel
    bins = np.searchsorted(cdf, xs[:nb], side='right')
    last = {}
    prev = np.full(nb, -1, dtype=np.int64)
    for e, b in enumerate(bins):
        prev[e] = last.get(b, -1)
        last[b] = e
    diff = np.zeros(W + 1, dtype=np.int64)
    for e in np.nonzero(prev >= 0)[0]:
        lo, hi = max(0, e - nsel + 1), min(prev[e], W - 1)
        if lo <= hi:
            diff[lo] += 1
            diff[hi + 1] -= 1
    m2 = np.cumsum(diff)[:W]
    p = np.diff(np.concatenate([[0.0], tb.S]))
    pm = float(p.max())
    denom = tb.Stot - nsel * pm
    ok = denom > 0.25 * tb.Stot
    wmax = (pm / denom) * (1.0 + 1e-9) if ok else 2.0
    nd = np.full(W + look, 255, dtype=np.int64)
    for r in range(W + look):
        e = nsel + r
        close = np.nonzero(np.abs(xs[e + 1:e + look] - xs[e]) <= wmax)[0]
        if close.size:
            nd[r] = close[0] + 1
    out = np.full(W, 255, dtype=np.int64)
    for d in range(W):
        m = int(m2[d])
        if m == 0:
            out[d] = 0
        elif m <= look and ok:
            bad = any(nd[d + e] != 255 and e + nd[d + e] < m for e in range(m))
            if not bad:
                out[d] = m
    return out


def modified_bin(tb, x, found_ids, p):
    """Model of wc_wave_bin (r05): searchsorted(cdf', x, 'right') for the cdf modified by a candidate's found set,
    located gap by gap with the exact predicate only -- smallest i with fl((S_i - C(i)) / St_cur) > x,
    C(i) = mass of the found ids <= i.  ``found_ids``: the distinct first-round bins; ``p``: float64 probabilities."""
    found = np.unique(np.asarray(found_ids, dtype=np.int64))
    Ctot = float(np.sum(p[found]))
    St_cur = tb.Stot - Ctot
    n = tb.n

    def gt(S):
        return S / St_cur > x
    t = x * St_cur + x * Ctot
    b = min(tb.K - 1, max(0, int((t / tb.Stot) * tb.K)))
    i = min(int(tb.T[b]), n - 1)
    for _ in range(12):
        below = found[found <= i]
        Cle = float(np.sum(p[below]))
        lo = int(below.max()) if below.size else -1
        above = found[found > i]
        hi = int(above.min()) if above.size else n
        if lo >= 0 and gt(tb.S[lo] - Cle):
            i = lo - 1
            assert i >= 0
            continue
        if hi < n and not gt(tb.S[hi - 1] - Cle):
            i = hi + 1
            assert i < n
            continue
        k0 = max(lo + 1, min(i - 24, hi - 64))
        for _w in range(64):
            ks = np.arange(k0, k0 + 64)
            pr = np.array([(k >= hi) or gt(tb.S[k] - Cle) for k in ks])
            if not pr.any():
                k0 += 64
            elif pr[0] and k0 > lo + 1:
                k0 = max(lo + 1, k0 - 63)
            else:
                return int(k0 + np.argmax(pr))
        return -1
    return -1


def spec_redraws_exact(tb, xs, nsel, W, look=64):
    """Model of wc_spec_kernel since r05: as spec_redraws, but a candidate whose round-2 draws hold close pairs is decided
    by looking the bins of those draws up exactly in the candidate's own modified cdf (modified_bin): m3 = round-2 draws
    that hit a bin another round-2 draw took first; R = m2 + m3 if m3 <= 1 or the m3 draws of round 3 are pairwise farther
    apart than the widest bin of any modified cdf."""
    out = spec_redraws(tb, xs, nsel, W, look)
    cdf = tb.S / tb.Stot
    nb = W + nsel
    bins = np.searchsorted(cdf, xs[:nb], side='right')
    p = np.diff(np.concatenate([[0.0], tb.S]))
    pm = float(p.max())
    denom = tb.Stot - nsel * pm
    if not denom > 0.25 * tb.Stot:
        return out
    wmax = (pm / denom) * (1.0 + 1e-9)
    for d in np.nonzero(out == 255)[0]:
        found = bins[d:d + nsel]
        m2 = nsel - np.unique(found).size
        if m2 == 0 or m2 > look:
            continue
        x2 = xs[d + nsel:d + nsel + m2]
        flagged = [j for j in range(m2) if any(j2 != j and abs(x2[j] - x2[j2]) <= wmax for j2 in range(m2))]
        b2 = {j: modified_bin(tb, float(x2[j]), found, p) for j in flagged}
        if any(v < 0 for v in b2.values()):
            continue
        m3 = len(flagged) - len(set(b2.values()))
        # what the kernel does: bins are monotone in x, so with the close draws sorted by value m3 = the neighbours (next
        # larger close draw within wmax; ties: the later draw) that share the bin of their predecessor -- one look-up per pair
        m3_pairs = 0
        for j in flagged:
            above = [(x2[e], e) for e in range(m2) if e != j and (x2[e] > x2[j] or (x2[e] == x2[j] and e > j))]
            if above:
                xe, e = min(above)
                if xe - x2[j] <= wmax:
                    m3_pairs += int(b2[j] == modified_bin(tb, float(xe), found, p))
        assert m3_pairs == m3, (d, m3_pairs, m3)
        if m3 <= 1:
            out[d] = m2 + m3
        else:
            x3 = xs[d + nsel + m2:d + nsel + m2 + m3]
            if all(abs(x3[j] - x3[j2]) > wmax for j in range(m3) for j2 in range(j)):
                out[d] = m2 + m3
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Range-split stream skip ("what would lift the 16-bit figure at 8 GPUs"): N ranks skip ONE shape.
#
# The only serial quantity of the weighted sub-sample is where every query's draws start: s_(q+1) = s_q + nsel + R_q(s_q),
# R_q(s) = redraws of ``choice`` for query q when its first double is the s-th of the stream -- a function of q and s alone.
# A rank that owns the queries [a, b) of a shape does not know s_a (it depends on every redraw before a), but it can walk
# EVERY candidate start of a window around the predicted one: walks from neighbouring starts meet within tens of queries
# (a start shifted by one double loses one first-round draw and gains one; the redraw count absorbs the difference with a
# few per cent probability per query) and a walk that has met another one costs nothing more.  The rank publishes the map
# {candidate start at a -> start at b}; the true path is the composition of the N maps, bit for bit the single stream.
# ---------------------------------------------------------------------------------------------------------------------
def redraws_at(tb, xs, s, nsel):
    """R_q(s) by the complete algorithm: doubles consumed beyond the first nsel when the query's first double is xs[s]"""
    pos = [s]

    def rand(m):
        a = xs[pos[0]:pos[0] + m]
        assert a.size == m, 'stream exhausted'
        pos[0] += m
        return a
    choice_noreplace(tb, rand, nsel)
    return pos[0] - s - nsel


def range_map(tbs, xs, starts, nsel):
    """walk every candidate start through the queries of a range.  tbs: the Tables of the range's queries in order;
    starts: candidate stream positions (doubles) of the range's first query.  Returns ({start: end}, evaluations of R,
    number of queries after which all walks had merged into one (None: never))."""
    cur = {int(s): int(s) for s in starts}           # start -> where its walk stands
    evals, merged_at = 0, None
    for j, tb in enumerate(tbs):
        step = {}
        for p in set(cur.values()):                  # walks that met share every later step
            step[p] = p + nsel + redraws_at(tb, xs, p, nsel)
            evals += 1
        cur = {s: step[p] for s, p in cur.items()}
        if merged_at is None and len(set(cur.values())) == 1:
            merged_at = j + 1
    return cur, evals, merged_at
