// Weighted sub-sample, stage 2 (included by p2s_wchoice.hip inside its anonymous namespace): the offsets pass -- where every
// query's draws start in the stream: speculation tables, jump tables, the tentative walk, the band of exact look-ups, the walk.
struct WcArgs {
    const double *S;          // [nq][n]
    const WcRec *R;           // [nq][K]
    const double *stot;       // [nq]
    const float *pmax;        // [nq][2] largest probability of the query; (unused int)
    const float *dsum;        // [nq][2] largest distance, np.sum of the clipped probabilities: p_i is re-derived from the cloud
    const float *pts;         // [n][3] the cloud
    const float *q;           // [nq][3] the queries of the batch
    const uint32_t *words;    // raw tempered words from the generator's position
    long long cap_words;      // words this request may consume
    int n, K, nq, nsel;
    long long *base;          // [nq] word offset of every query's first draw (offsets pass -> ids kernel)
    int32_t *ids_out;         // [nq][nsel]
    long long *meta;          // [0] words consumed (out), [1] sticky error
    long long *stats;         // development counters (null = off)
    int fixed;                // fixed_subsample: every query starts at word 0 of a freshly seeded generator
};

// ---------------------------------------------------------------------------------------------------------------
// offsets: where every query's draws start.
//
// A query consumes 2*nsel words for its first round plus 2 words per redraw, so the only serial dependence is the word
// offset: query q starts at s_q = s_{q-1} + 2 (nsel + R_{q-1}), R = redraws.  R_q is a function of the start alone.  With
// X_k the k-th double of the stream and bin_q() the query's cdf look-up,
//     m2(s) = nsel - #distinct{ bin_q(X_k) : s <= k < s + nsel }          (first-round draws hitting a taken bin)
// is what round 2 draws.  Round 2 draws from the MODIFIED cdf (found ids carry no mass), so its draws can only collide
// with each other, and only if two of them are closer than the widest bin of any modified cdf, wmax.  r05: such close
// pairs are no longer left to the complete algorithm -- the spec kernel looks their bins up EXACTLY in the cdf modified
// by the candidate's own found set (wc_wave_bin), which decides every candidate but the very few whose THIRD round has
// two draws within wmax of each other:
//     m3 = round-2 draws that hit a bin another round-2 draw took first;  R = m2 + m3 when m3 <= 1 or the m3 draws of
//     round 3 are pairwise farther apart than wmax.
//   wc_spec_kernel   one workgroup per query of a block of SP_B queries, all CUs: R_q(s) for the SP_W candidate
//                    starts around the predicted one (block start, exact, + sum of the expected collision counts mu
//                    of the queries before it, from the tables kernel).  m2 over a sliding window: every draw e with
//                    an earlier draw prev(e) in the same bin adds 1 to the starts in (e - nsel, prev(e)] -- a difference
//                    array + scan; draws sharing a bin are found through an LDS hash.  255 = undecided.  It also writes
//                    level 0 of the jump tables.
//   wc_jumpm_kernel  levels 1.. of the jump tables (below), four levels per launch
//   wc_chain_kernel  one workgroup: the walk s -> s + 2 (nsel + R_q(s)) through the tables; the (now very rare) undecided
//                    candidate runs the complete algorithm in place (wc_full_query); a start outside the window ends the
//                    block early, the next (spec, chain) pair resumes there; the LAST launch of a request takes whatever
//                    is still unresolved through the complete algorithm, query by query (normally nothing).
// The ids kernel re-derives every query's consumption and flags any disagreement (meta[1] = 4).
// ---------------------------------------------------------------------------------------------------------------
constexpr int SP_B = 2048;                           // queries per speculation block
constexpr int SP_W = 1024;                           // candidate starts per query
constexpr int SP_LOOK = 64;                          // round-2 draws per candidate the spec kernel handles
constexpr int SP_NB = SP_W + WC_MAX_SEL;             // draws whose bin is needed
constexpr int SP_NX = SP_NB + 2 * SP_LOOK;           // doubles held
constexpr int SP_HASH = 4096;
constexpr int SP_DMAX = 1024;                        // draws that share their bin with another draw of the window (more: undecided)

// ---- the walk s -> s + 2 (nsel + R_q(s)) without walking --------------------------------------------------------------
// In window coordinates (d = (s - klo[q]) / 2, candidate d of query q) one step is
//     next(q, d) = d + R_q(d) + nsel + (klo[q] - klo[q + 1]) / 2        if R_q(d) is decided and the result is a candidate of q + 1
// -- a table look-up.  2^k steps at once are the look-up J_k[q][d] with J_k[q] = J_{k-1}[q + 2^(k-1)] o J_{k-1}[q].  r05: level
// k is only kept for the queries q = 0 mod 2^k (a "ruler": 2 SP_B rows in all instead of SP_LEV x SP_B) -- the walk takes the
// largest aligned jump that is valid, i.e. ~2 log2(SP_B) look-ups per SEGMENT between two queries it has to resolve
// itself, and the word offsets of the queries inside a segment are filled in by all lanes afterwards (binary lifting
// from the segment start, whose alignment covers every level the fill needs).
constexpr int SP_LEV = 11;
static_assert((1 << SP_LEV) >= SP_B, "jump levels cover a block");
constexpr unsigned short SP_INV = 0xffffu;
__host__ __device__ inline size_t sp_lev_row(int k, int i) {         // row of level k, query i (i = 0 mod 2^k)
    return (size_t)(2 * SP_B - ((2 * SP_B) >> k)) + (size_t)(i >> k);
}
constexpr size_t SP_JUMP_ROWS = 2 * SP_B;

struct WcSpec {
    unsigned char *rtab;      // [SP_B][SP_W] redraws for candidate start klo + 2 d; 255 = undecided
    long long *klo;           // [SP_B] word offset of candidate 0
    long long *ctl;           // [0] first unresolved query, [1] its word offset
    const float *mu;          // [nq] expected first-round collisions
    unsigned short *jump;     // [SP_JUMP_ROWS][SP_W] ruler of jump tables
    long long *klo1;          // [SP_B] window origin of the NEXT query as this query's workgroup computed it
    short *dtil;              // [SP_B] window coordinate where the tentative walk passed every query (-1: not reached)
    int *save;                // [SP_B][SP_SAVE] dup list + first-round bins of every window (spec -> band kernel)
    unsigned char *scratch;   // wc_lds_bytes(n) bytes: the arrays of the complete algorithm for the chain kernel
};

// levels k0 + 1 .. min(k0 + JM_LEV, kmax) from level k0 in ONE launch (r05: one launch per level -- ten dependent launches
// between the spec / band kernel and the walk, each waiting ~50 us for a free workgroup slot next to the encoders).  A
// workgroup takes JM_ROWS consecutive rows of level k0 into LDS (32 KB: fits the slot one retiring encoder workgroup frees)
// and composes upwards in place: level k0 + t lives in the rows j * 2^t of the tile,
//     J_k[i] = J_(k-1)[i + 2^(k-1)] o J_(k-1)[i]          (rows i = 0 mod 2^k; invalid unless i + 2^k lands on a query of the block)
// -- the element (row, d) is read only by the lane that overwrites it, the other operand row (j * 2^t + 2^(t-1)) is never
// written at that level, so one barrier per level is all the ordering there is.
constexpr int JM_LEV = 4;
constexpr int JM_ROWS = 1 << JM_LEV;
__global__ __launch_bounds__(256) void wc_jumpm_kernel(WcArgs a, WcSpec sp, int k0, int kmax) {
    __shared__ __attribute__((aligned(16))) unsigned short jm[JM_ROWS * SP_W];
    if (a.meta[1] != 0) return;
    const long long qb = sp.ctl[0];
    if (qb >= a.nq) return;
    const int lim = (int)(a.nq - qb < SP_B ? a.nq - qb : SP_B);
    const int i0 = ((int)blockIdx.x * JM_ROWS) << k0;            // first query of this tile
    if (i0 >= lim) return;
    const int tid = threadIdx.x;
    for (int c = tid; c < JM_ROWS * SP_W / 8; c += 256) {        // 16 bytes per lane and step
        const int r = c / (SP_W / 8), i = i0 + (r << k0);
        uint4 v = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);      // SP_INV: no such row
        if (i < lim) v = *(const uint4 *)(sp.jump + sp_lev_row(k0, i) * SP_W + 8 * (c % (SP_W / 8)));
        *(uint4 *)(jm + 8 * c) = v;
    }
    __syncthreads();
    for (int t = 1; t <= JM_LEV && k0 + t <= kmax; ++t) {
        const int k = k0 + t, nrow = JM_ROWS >> t, hs = 1 << (t - 1);
        for (int e = tid; e < nrow * SP_W; e += 256) {
            const int slot = (e / SP_W) << t, d = e % SP_W;
            const int i = i0 + (slot << k0);
            if (i >= lim) continue;
            unsigned short v = SP_INV;
            if (i + (1 << k) <= lim - 1) {                       // lands on a query of the block
                const unsigned short m = jm[slot * SP_W + d];
                if (m != SP_INV) v = jm[(slot + hs) * SP_W + m];
            }
            jm[slot * SP_W + d] = v;
            sp.jump[sp_lev_row(k, i) * SP_W + d] = v;
        }
        __syncthreads();
    }
}

__global__ void wc_ctl_init_kernel(long long *ctl, const long long *meta) {
    ctl[0] = 0;
    ctl[1] = meta[0];
}

// ---- exact look-up in a candidate's own modified cdf, by a group of 8 lanes ------------------------------------------
// All look-ups of a query run side by side (64 groups per workgroup): the kernel is bound by the latency of its random
// table reads (~10 us each under load), so what counts is how many of them are in flight, not the instructions per look-up.
constexpr int WC_G = 4;                               // lanes per group
constexpr int WC_GS = 8;                              // S values per lane and round trip: WC_G * WC_GS = 32 ids
__device__ __forceinline__ double wc_grp_sum(double v) {
#pragma unroll
    for (int off = WC_G / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ int wc_grp_max(int v) {
#pragma unroll
    for (int off = WC_G / 2; off >= 1; off >>= 1) {
        const int u = __shfl_xor(v, off);
        v = u > v ? u : v;
    }
    return v;
}
__device__ __forceinline__ int wc_grp_min(int v) {
#pragma unroll
    for (int off = WC_G / 2; off >= 1; off >>= 1) {
        const int u = __shfl_xor(v, off);
        v = u < v ? u : v;
    }
    return v;
}

// The found set F of candidate start d = distinct first-round bins of the draws [d, d + nsel).
struct WcCand {
    const int *bins;          // LDS [SP_NB] first-round bin of every draw of the window
    const float *pw;          // LDS [SP_NB] its probability (float32 value, exact)
    const int *dl;            // LDS [ndup] draw | (previous draw of the same bin + 1) << 11
    int ndup, d, nsel;
};
// found mass at or below id i (Cle) and in all (Ctot), largest found id <= i (-1: none), smallest found id > i (n: none);
// every lane of the group gets the result.  Exact in any order: every partial sum is a multiple of the probabilities'
// last place below 2 (see the file header).
__device__ __forceinline__ void wc_grp_pass(const WcCand &c, int n, int i, int gl, double &Cle, double &Ctot, int &lo, int &hi) {
    double s = 0.0, st = 0.0;
    int l = -1, h = n;
    // (8 draws per lane in flight: one at a time the loop runs at the latency of its two LDS reads, 23 us per pass)
    for (int e0 = c.d + gl; e0 < c.d + c.nsel; e0 += 8 * WC_G) {
        int b[8];
        float pf[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + WC_G * u;
            const bool in = e < c.d + c.nsel;
            b[u] = in ? c.bins[e] : n;                   // past the window: above every id, no mass
            pf[u] = in ? c.pw[e] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double p = (double)pf[u];
            st += p;
            if (b[u] <= i) {
                s += p;
                l = b[u] > l ? b[u] : l;
            } else {
                h = b[u] < h ? b[u] : h;
            }
        }
    }
    for (int t = gl; t < c.ndup; t += WC_G) {            // a bin drawn more than once inside the window counts once
        const int e = c.dl[t] & 2047, pv = (c.dl[t] >> 11) - 1;
        if (e >= c.d && e < c.d + c.nsel && pv >= c.d) {
            const double p = (double)c.pw[e];
            st -= p;
            if (c.bins[e] <= i) s -= p;
        }
    }
    Cle = wc_grp_sum(s);
    Ctot = wc_grp_sum(st);
    lo = wc_grp_max(l);
    hi = wc_grp_min(h);
}
// searchsorted(cdf', x, 'right') for the cdf modified by the found set of a candidate: smallest i with
// fl((S_i - C(i)) / St_cur) > x, C(i) = found mass at or below i.  The predicate is monotone in i and every decision below
// is taken with the exact predicate (wc_gt on exact S, C values); approximations only choose where to look.  One group of
// WC_G lanes (gl = lane in the group, gsh = bit position of the group's lane 0 in the wave); i0 = first guess (the bin of x
// in the unmodified cdf); -1 = gave up (the caller leaves the candidate undecided).
// x2 >= x (a close round-2 draw): *same = it falls into the same bin, i.e. fl(V_bin / St_cur) > x2 as well.
__device__ __forceinline__ int wc_grp_bin(const WcCand &c, const double *__restrict__ Sq, int n, double Stot, double x, double x2,
                                          int i0, int gl, int gsh, int *same) {
    constexpr unsigned GM = (1u << WC_G) - 1u;
    int i = i0 > n - 1 ? n - 1 : (i0 < 0 ? 0 : i0);
    for (int it = 0; it < 12; ++it) {
        double Cle, Ctot;
        int lo, hi;
        wc_grp_pass(c, n, i, gl, Cle, Ctot, lo, hi);
        const double St_cur = Stot - Ctot;
        // the found-free gap (lo, hi) around i: pred(lo) must be false and pred(hi - 1) true for the answer to lie in it;
        // 32 ids of it in the same round trip (id k0 + WC_G j + gl in slot j), centred where the found mass below i says
        // the answer is: the first guess assumed the proportional share x Ctot
        const double s_lo = Sq[lo > 0 ? lo : 0], s_hm = Sq[hi - 1 > 0 ? hi - 1 : 0];
        int k0;
        {
            double sh = (Cle - x * Ctot) * (double)n / Stot;              // mass -> ids at the mean bin width
            sh = sh > 64.0 ? 64.0 : (sh < -64.0 ? -64.0 : sh);
            k0 = i + (it == 0 ? (int)sh : 0) - WC_G * WC_GS / 2 + 4;
        }
        k0 = k0 > hi - WC_G * WC_GS ? hi - WC_G * WC_GS : k0;
        k0 = k0 < lo + 1 ? lo + 1 : k0;
        double sk[WC_GS];
#pragma unroll
        for (int j = 0; j < WC_GS; ++j) sk[j] = (k0 + WC_G * j + gl < hi) ? Sq[k0 + WC_G * j + gl] : 0.0;
        if (lo >= 0 && wc_gt(s_lo - Cle, St_cur, x)) {          // V_lo = S_lo - C(lo), C(lo) = Cle: the answer is below lo
            i = lo - 1;
            if (i < 0) return -1;
            continue;
        }
        if (hi < n && !wc_gt(s_hm - Cle, St_cur, x)) {          // V_(hi-1) still <= target (and V_hi = V_(hi-1)): above hi
            i = hi + 1;
            if (i >= n) return -1;
            continue;
        }
        if (hi - lo < 2) return -1;                              // (cannot happen: the predicate changes inside the gap)
        // inside the gap C is constant
        for (int w = 0; w < 96; ++w) {
            int first = -1;
            double sv = 0.0;
#pragma unroll
            for (int j = WC_GS - 1; j >= 0; --j) {
                const bool pr = (k0 + WC_G * j + gl >= hi) || wc_gt(sk[j] - Cle, St_cur, x);
                const unsigned m = (unsigned)(__ballot(pr) >> gsh) & GM;
                if (m) {
                    first = WC_G * j + __builtin_ctz(m);
                    sv = sk[j];
                }
            }
            if (first < 0) {
                k0 += WC_G * WC_GS;                               // all <= target: further right (k0 < hi: pred(hi-1) holds)
            } else if (first == 0 && k0 > lo + 1) {
                k0 = k0 - (WC_G * WC_GS - 1) < lo + 1 ? lo + 1 : k0 - (WC_G * WC_GS - 1);     // the first id already beyond: further left
            } else {
                // the lane that holds the answer's S value also answers for x2
                const bool mine = (first & (WC_G - 1)) == gl;
                *same = ((unsigned)(__ballot(mine && wc_gt(sv - Cle, St_cur, x2)) >> gsh) & GM) ? 1 : 0;
                return k0 + first;
            }
#pragma unroll
            for (int j = 0; j < WC_GS; ++j) sk[j] = (k0 + WC_G * j + gl < hi) ? Sq[k0 + WC_G * j + gl] : 0.0;
        }
        return -1;
    }
    return -1;
}

// verdict byte of a candidate (rtab): 0 .. 126 = redraws R, decided; 128 + m2 = TENTATIVE: round 2 draws m2 doubles that hold
// a close pair -- R = m2 unless such a pair shares a bin (wc_band_kernel decides that for the candidates near the path);
// 255 = undecided (the chain runs the complete algorithm if the path gets there)
constexpr unsigned SP_TENT = 128u, SP_UND = 255u;
constexpr int SP_SAVE = 4 + SP_DMAX + SP_NB;         // ints per query handed from wc_spec_kernel to wc_band_kernel: ndup, dup list, bins
constexpr int SP_BAND = 144;                         // candidates per query decided exactly: [path - 72, path + 72)
constexpr int SP_BAND_LO = 72;

__global__ __launch_bounds__(256) void wc_spec_kernel(WcArgs a, WcSpec sp) {
    __shared__ double xs[SP_NX];
    __shared__ int bins[SP_NB];
    __shared__ __attribute__((aligned(16))) uint32_t hkey[SP_HASH];      // later: diff, nd, rt (offsets below)
    __shared__ int dl[SP_DMAX];                      // draw | bin << 11, later draw | (previous draw of the same bin + 1) << 11
    __shared__ int s_ndup, wsum[4];
    __shared__ float redf[2][4];
    if (a.meta[1] != 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long qb = sp.ctl[0], sb = sp.ctl[1];
    const int i = blockIdx.x;
    const long long q = qb + i;
    if (q >= a.nq) return;
    const int lim = (int)(a.nq - qb < SP_B ? a.nq - qb : SP_B);
    // predicted start: the block start (exact) + the expected redraws of the queries before this one; the same for the
    // next query (its workgroup sums in exactly this order), whose window origin level 0 of the jump tables refers to
    float part = 0.0f, part1 = 0.0f;
    for (int j = tid; j <= i; j += 256) {
        const float m = sp.mu[qb + j];
        if (j < i) part += m;
        part1 += m;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        part += __shfl_xor(part, off);
        part1 += __shfl_xor(part1, off);
    }
    if (lane == 0) {
        redf[0][wave] = part;
        redf[1][wave] = part1;
    }
    if (tid == 0) s_ndup = 0;
    for (int h = tid; h < SP_HASH; h += 256) hkey[h] = 0xffffffffu;
    __syncthreads();
    long long dpre = (long long)((redf[0][0] + redf[0][1]) + (redf[0][2] + redf[0][3]) + 0.5f) - SP_W / 2;
    dpre = dpre < 0 ? 0 : dpre;
    long long dpre1 = (long long)((redf[1][0] + redf[1][1]) + (redf[1][2] + redf[1][3]) + 0.5f) - SP_W / 2;
    dpre1 = dpre1 < 0 ? 0 : dpre1;
    const long long klo = sb + 2 * ((long long)i * a.nsel + dpre);
    const long long klo1 = sb + 2 * ((long long)(i + 1) * a.nsel + dpre1);
    if (tid == 0) {
        sp.klo[i] = klo;
        sp.klo1[i] = klo1;
    }
    const int nsel = a.nsel, nb = SP_W + nsel, nx = nb + 2 * SP_LOOK;
    const double *Sq = a.S + (size_t)q * a.n;
    const WcRec *Rq = a.R + (size_t)q * a.K;
    const double Stot = a.stot[q];
    // ---- doubles of the window and their bins (first-round look-up)
    for (int e = tid; e < nx; e += 256) {
        const long long w = klo + 2LL * e;
        double x = 2.0;                               // past the request: never "far" from anything -> undecided
        int bin = -1;
        if (w + 1 < a.cap_words) {
            const uint2 wp = *(const uint2 *)(a.words + w);
            x = wc_double(wp.x, wp.y);
            if (e < nb) {
                int bk = (int)(x * (double)a.K);
                bk = bk > a.K - 1 ? a.K - 1 : bk;
                bin = wc_finish(Sq, a.n, Rq[bk], 0, a.n, 0.0, Stot, x).bin;
            }
        }
        xs[e] = x;
        if (e < nb) bins[e] = bin;
    }
    __syncthreads();
    // ---- draws that share their bin with another draw of the window: hash bin -> count
    constexpr int PER = (SP_NB + 255) / 256;
    int slot[PER];
    bool overflow = false;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int e = tid + 256 * j;
        slot[j] = -1;
        if (e < nb && bins[e] >= 0) {
            const uint32_t bin = (uint32_t)bins[e];
            uint32_t h = (bin * 2654435761u) >> 20;
            for (;;) {
                uint32_t cur = hkey[h];
                if (cur == 0xffffffffu) {
                    const uint32_t old = atomicCAS(&hkey[h], 0xffffffffu, (bin << 10) | 1u);
                    if (old == 0xffffffffu) break;
                    cur = old;
                }
                if ((cur >> 10) == bin) {
                    if ((atomicAdd(&hkey[h], 1u) & 1023u) >= 1000u) overflow = true;     // count field about to overflow
                    break;
                }
                h = (h + 1) & (SP_HASH - 1);
            }
            slot[j] = (int)h;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        if (slot[j] >= 0 && (hkey[slot[j]] & 1023u) >= 2u) {
            const int k = atomicAdd(&s_ndup, 1);
            if (k < SP_DMAX) dl[k] = (tid + 256 * j) | (bins[tid + 256 * j] << 11);
        }
    }
    const bool any_over = __syncthreads_or(overflow ? 1 : 0) != 0;
    const bool undecidable = any_over || s_ndup > SP_DMAX;
    const int ndup = s_ndup < SP_DMAX ? s_ndup : SP_DMAX;
    // the hash is dead: its 16 KB now hold
    int *diff = (int *)hkey;                                              // [SP_W + 1]              0 .. 4100
    unsigned char *nd = (unsigned char *)hkey + 4112;                     // [SP_W + SP_LOOK]     4112 .. 5200
    for (int d = tid; d <= SP_W; d += 256) diff[d] = 0;
    // ---- previous draw of the same bin for every listed draw (registers; written back behind the barrier)
    int pv[SP_DMAX / 256];
#pragma unroll
    for (int j = 0; j < SP_DMAX / 256; ++j) {
        const int t = tid + 256 * j;
        pv[j] = -1;
        if (t < ndup) {
            const int e = dl[t] & 2047, b = dl[t] >> 11;
            int prev = -1;
            for (int u = 0; u < ndup; ++u) {
                const int oe = dl[u] & 2047, ob = dl[u] >> 11;
                if (ob == b && oe < e && oe > prev) prev = oe;
            }
            pv[j] = prev;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SP_DMAX / 256; ++j) {
        const int t = tid + 256 * j;
        if (t < ndup) {
            const int e = dl[t] & 2047, prev = pv[j];
            dl[t] = e | ((prev + 1) << 11);
            // ---- m2(d): draw e with an earlier same-bin draw prev counts for the starts d in (e - nsel, prev]
            if (prev >= 0) {
                const int lo = e - nsel + 1 > 0 ? e - nsel + 1 : 0;
                const int hi = prev < SP_W - 1 ? prev : SP_W - 1;
                if (lo <= hi) {
                    atomicAdd(&diff[lo], 1);
                    atomicAdd(&diff[hi + 1], -1);
                }
            }
        }
    }
    // ---- nd[e - nsel]: distance to the first later draw within reach of the widest bin of any modified cdf
    const double pm = (double)a.pmax[2 * q];
    const double denom = Stot - (double)nsel * pm;
    const bool dist_ok = denom > 0.25 * Stot;
    const double wmax = dist_ok ? (pm / denom) * (1.0 + 1e-9) : 2.0;
    for (int r = tid; r < SP_W + SP_LOOK; r += 256) {
        const int e = nsel + r;
        const double x = xs[e];
        unsigned long long cm = 0ull;                    // bit j: draw e + j lies within wmax (all 63 reads in flight: no early exit)
#pragma unroll 9
        for (int j = 1; j < SP_LOOK; ++j) cm |= (unsigned long long)(fabs(x - xs[e + j]) <= wmax) << j;
        nd[r] = (unsigned char)(cm ? __builtin_ctzll(cm) : 255);
    }
    __syncthreads();
    // ---- scan of the difference array (4 candidates per lane) and the verdict per candidate
    const int d0 = 4 * tid;
    int c[4];
    c[0] = diff[d0];
    c[1] = c[0] + diff[d0 + 1];
    c[2] = c[1] + diff[d0 + 2];
    c[3] = c[2] + diff[d0 + 3];
    int v = c[3];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(v, off);
        if (lane >= off) v += u;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    int run = v - c[3];
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wave) run += wsum[w];
    uint32_t packed = 0;
    ushort4 o;
    unsigned short *ov = (unsigned short *)&o;
    const bool lastq = i + 1 >= lim;                         // the step of the block's last query leaves the block: not a jump
    const int delta = nsel + (int)((klo - klo1) >> 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int d = d0 + j;
        const int m2 = run + c[j];
        unsigned r = SP_UND;
        int guess = 0;
        const long long s = klo + 2LL * d;
        // (room for the third round of a candidate that is decided later: m3 <= m2)
        if (!undecidable && s + 2LL * (nsel + 2 * m2) <= a.cap_words) {
            if (m2 == 0) {
                r = 0u;
            } else if (m2 <= SP_LOOK && dist_ok) {
                bool bad = false;
                for (int e = 0; e < m2; ++e) {
                    const int reach = nd[d + e];
                    if (reach != 255 && e + reach < m2) {
                        bad = true;
                        // a guess with the right mean: the pair shares a bin of the UNMODIFIED cdf (its bins in the modified
                        // one lie some ten ids away and are as wide on average).  A tentative walk that took every close
                        // pair for two bins would fall behind the real one by ~5 draws per 100 queries.
                        const int ea = d + nsel + e, eb = ea + reach;
                        guess += (eb < nb && bins[ea] == bins[eb]) ? 1 : 0;
                    }
                }
                r = bad ? SP_TENT + (unsigned)m2 : (unsigned)m2;
            }
        }
        packed |= r << (8 * j);
        // level 0 of the jump tables (step of candidate d into the window of the next query); a tentative candidate steps by
        // its guess -- good enough to find out WHERE the path runs (wc_chain_kernel, tentative)
        const unsigned rs = r == SP_UND ? r : (r & 127u) + (unsigned)guess;
        const int ndq = d + (int)rs + delta;
        ov[j] = (lastq || r == SP_UND || ndq < 0 || ndq >= SP_W) ? SP_INV : (unsigned short)ndq;
    }
    ((uint32_t *)(sp.rtab + (size_t)i * SP_W))[tid] = packed;
    *(ushort4 *)(sp.jump + sp_lev_row(0, i) * SP_W + d0) = o;
    // ---- what wc_band_kernel needs of this window: the dup list and the first-round bins
    int *sv = sp.save + (size_t)i * SP_SAVE;
    if (tid == 0) sv[0] = undecidable ? -1 : ndup;
    for (int t = tid; t < ndup; t += 256) sv[4 + t] = dl[t];
    for (int e = tid; e < nb; e += 256) sv[4 + SP_DMAX + e] = bins[e];
}

// The exact pass: the candidates of a query within [path - 72, path + 72) of where the tentative walk went through its
// window, if tentative, get their close round-2 pairs looked up in their own modified cdf.  The real path stays that close:
// it leaves the tentative one by one double per pair that does share a bin (~1 % of the queries) and the two re-merge
// within tens of queries (a start shifted by one draw loses one first-round draw and gains one, and the redraw counts
// absorb the difference with ~5 % probability per query).  A path that does escape meets a tentative / undecided verdict
// and the chain resolves that query itself.
constexpr int BD_NP = SP_BAND + 2 * SP_LOOK;          // round-2 positions a band touches (+ look-ahead)
constexpr int BD_PAIRS = 192, BD_TASKS = 384;
__global__ __launch_bounds__(256) void wc_band_kernel(WcArgs a, WcSpec sp) {
    __shared__ int bins[SP_NB];
    __shared__ int dl[SP_DMAX];
    __shared__ float pwb[SP_BAND + WC_MAX_SEL];      // probability of draw e at pwb[e - band_lo]
    __shared__ double xsb[BD_NP];                    // double of round-2 position r (draw nsel + r) at xsb[r - band_lo]
    __shared__ __attribute__((aligned(4))) unsigned char rt[SP_W];
    __shared__ int pl[BD_PAIRS], task[BD_TASKS], xl[SP_BAND], cres[SP_BAND];
    __shared__ int s_npair, s_ntask, s_nxl;
    if (a.meta[1] != 0) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const long long qb = sp.ctl[0];
    const int i = blockIdx.x;
    const long long q = qb + i;
    if (q >= a.nq) return;
    const int lim = (int)(a.nq - qb < SP_B ? a.nq - qb : SP_B);
    const int nsel = a.nsel, nb = SP_W + nsel;
    const long long klo = sp.klo[i], klo1 = sp.klo1[i];
    const int dt = sp.dtil[i];                                   // where the tentative walk passed (-1: it did not get here)
    const int *sv = sp.save + (size_t)i * SP_SAVE;
    const int ndup = sv[0];
    ((uint32_t *)rt)[tid] = ((const uint32_t *)(sp.rtab + (size_t)i * SP_W))[tid];
    if (tid == 0) s_npair = s_ntask = s_nxl = 0;
    const int blo = dt < 0 ? 0 : (dt - SP_BAND_LO > 0 ? dt - SP_BAND_LO : 0);
    const int bhi = dt < 0 ? 0 : (blo + SP_BAND < SP_W ? blo + SP_BAND : SP_W);       // candidates [blo, bhi)
    __syncthreads();
    // ---- tentative candidates of the band
    if (tid < bhi - blo && ndup >= 0) {
        const unsigned r = rt[blo + tid];
        if (r >= SP_TENT && r != SP_UND) xl[atomicAdd(&s_nxl, 1)] = (blo + tid) | ((int)(r - SP_TENT) << 16);
    }
    __syncthreads();
    const int nxl = s_nxl;
    if (a.stats && tid == 0) {
        atomicAdd((unsigned long long *)&a.stats[0], 1ull);
        atomicAdd((unsigned long long *)&a.stats[2], (unsigned long long)nxl);
    }
    if (nxl > 0) {
        const double *Sq = a.S + (size_t)q * a.n;
        const WcRec *Rq = a.R + (size_t)q * a.K;
        const double Stot = a.stot[q];
        const double pm = (double)a.pmax[2 * q];
        const double wmax = (pm / (Stot - (double)nsel * pm)) * (1.0 + 1e-9);          // (dist_ok held: the candidates are tentative)
        // ---- the window's state: dup list, first-round bins (wc_spec_kernel), the doubles of the band's round-2 positions
        for (int t = tid; t < ndup; t += 256) dl[t] = sv[4 + t];
        for (int e = tid; e < nb; e += 256) bins[e] = sv[4 + SP_DMAX + e];
        const int np = bhi - blo + 2 * SP_LOOK;
        for (int r = tid; r < np; r += 256) {
            const long long w = klo + 2LL * (nsel + blo + r);
            double x = 2.0;
            if (w + 1 < a.cap_words) {
                const uint2 wp = *(const uint2 *)(a.words + w);
                x = wc_double(wp.x, wp.y);
            }
            xsb[r] = x;
        }
        __syncthreads();
        // ---- probability of every draw a band candidate's first round holds, re-derived from the L2-resident cloud exactly
        // as the tables kernel did; the close pairs among the band's round-2 positions
        {
            const float qx = a.q[3 * q], qy = a.q[3 * q + 1], qz = a.q[3 * q + 2];
            const float dmax = a.dsum[2 * q], sum = a.dsum[2 * q + 1];
            for (int e = blo + tid; e < bhi + nsel; e += 256) {
                const int b = bins[e];
                pwb[e - blo] = b >= 0 ? wc_clip_prob(wc_dist1(a.pts, b, qx, qy, qz), dmax) / sum : 0.0f;
            }
        }
        for (int r = tid; r < bhi - blo + SP_LOOK; r += 256) {
            const double x = xsb[r];
            for (int j = 1; j < SP_LOOK; ++j) {
                if (fabs(x - xsb[r + j]) <= wmax) {
                    const int k = atomicAdd(&s_npair, 1);
                    if (k < BD_PAIRS) pl[k] = r | ((r + j) << 16);
                }
            }
        }
        for (int t = tid; t < nxl; t += 256) cres[t] = 0;
        __syncthreads();
        const int npair = s_npair < BD_PAIRS ? s_npair : BD_PAIRS;
        if (s_npair > BD_PAIRS)
            for (int t = tid; t < nxl; t += 256) cres[t] = 0x10000;           // pair list overflow: stays undecided
        __syncthreads();
        // ---- 8 lanes per candidate over the close pairs: a pair inside the candidate's round-2 range [d, d + m2) is a task if
        // its upper draw u2 (larger x; ties: the later draw) is the NEIGHBOUR of the lower one u1 -- no third draw of the range
        // lies between them (it would be within wmax of both, i.e. be listed with u1).  Bins are monotone in x, so m3 (round-2
        // draws that hit a bin another one took first) = the neighbour pairs that share a bin: ONE look-up per pair.
        for (int t = tid >> 3; t < nxl; t += 32) {
            const int d = (xl[t] & 0xffff) - blo, m2 = xl[t] >> 16;          // band coordinates
            for (int k = tid & 7; k < npair; k += 8) {
                const int r1 = pl[k] & 0xffff, r2 = pl[k] >> 16;
                if (r1 < d || r2 >= d + m2) continue;
                const double xa = xsb[r1], xb = xsb[r2];
                const bool up = xb >= xa;                                // r2 > r1: a tie counts the later draw as the larger
                const int u1 = up ? r1 : r2, u2 = up ? r2 : r1;
                const double x1 = up ? xa : xb, x2 = up ? xb : xa;
                bool between = false;
                for (int k2 = 0; k2 < npair; ++k2) {
                    const int q1 = pl[k2] & 0xffff, q2 = pl[k2] >> 16;
                    if (q1 != u1 && q2 != u1) continue;
                    const int z = q1 == u1 ? q2 : q1;
                    if (z == u2 || z < d || z >= d + m2) continue;
                    const double xz = xsb[z];
                    const bool above = xz > x1 || (xz == x1 && z > u1);
                    const bool below = xz < x2 || (xz == x2 && z < u2);
                    between |= above && below;
                }
                if (!between) {
                    const int kk = atomicAdd(&s_ntask, 1);
                    if (kk < BD_TASKS) task[kk] = t | ((u1 - d) << 8) | ((u2 - d) << 16);
                    else atomicOr(&cres[t], 0x10000);               // no room: stays undecided
                }
            }
        }
        __syncthreads();
        // ---- the look-ups, 64 at a time
        {
            const int ntask = s_ntask < BD_TASKS ? s_ntask : BD_TASKS;
            const int grp = tid / WC_G, gl = tid & (WC_G - 1), gsh = lane & ~(WC_G - 1);
            WcCand cd;
            cd.bins = bins;
            cd.pw = pwb - blo;                                       // indexed by the draw
            cd.dl = dl;
            cd.ndup = ndup;
            cd.nsel = nsel;
            for (int k = grp; k < ntask; k += 256 / WC_G) {
                const int t = task[k] & 255, j = (task[k] >> 8) & 255, j2 = task[k] >> 16;
                const int d = xl[t] & 0xffff;
                cd.d = d;
                const int e = d + nsel + j;
                const double x = xsb[d - blo + j];
                // first guess: the bin of x in the unmodified cdf -- known for the draws that are first-round draws of later
                // candidates (all but the last few of the window), else one guide look-up
                int i0;
                if (e < nb) {
                    i0 = bins[e];
                } else {
                    int bk = (int)(x * (double)a.K);
                    bk = bk > a.K - 1 ? a.K - 1 : bk;
                    i0 = Rq[bk];
                }
                int same = 0;
                const int b = wc_grp_bin(cd, Sq, a.n, Stot, x, xsb[d - blo + j2], i0, gl, gsh, &same);
                if (gl == 0) {
                    if (b < 0) atomicOr(&cres[t], 0x10000);
                    else if (same) atomicAdd(&cres[t], 1);
                }
            }
        }
        __syncthreads();
        // ---- one thread per candidate: R = m2 + m3 if m3 <= 1 or the m3 draws of round 3, right behind round 2, are pairwise
        // farther apart than wmax
        for (int t = tid; t < nxl; t += 256) {
            const int d = xl[t] & 0xffff, m2 = xl[t] >> 16;
            const int m3 = cres[t] & 0xffff;
            unsigned r = SP_UND;
            if (!(cres[t] >> 16)) {
                if (m3 <= 1) {
                    r = (unsigned)(m2 + m3);
                } else {
                    bool close = false;
                    for (int u = 0; u < m3; ++u)
                        for (int v2 = 0; v2 < u; ++v2) close |= fabs(xsb[d - blo + m2 + u] - xsb[d - blo + m2 + v2]) <= wmax;
                    if (!close) r = (unsigned)(m2 + m3);
                }
            }
            rt[d] = (unsigned char)(r > 126u ? SP_UND : r);
        }
        if (a.stats && tid == 0) atomicAdd((unsigned long long *)&a.stats[5], (unsigned long long)s_ntask);
        __syncthreads();
    }
    // ---- out: the verdicts, and level 0 of the jump tables again -- now only decided candidates step
    ((uint32_t *)(sp.rtab + (size_t)i * SP_W))[tid] = ((const uint32_t *)rt)[tid];
    {
        const bool lastq = i + 1 >= lim;
        const int delta = nsel + (int)((klo - klo1) >> 1);
        const int d0 = 4 * tid;
        ushort4 o;
        unsigned short *ov = (unsigned short *)&o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned r = rt[d0 + j];
            const int ndq = d0 + j + (int)r + delta;
            ov[j] = (lastq || r >= SP_TENT || ndq < 0 || ndq >= SP_W) ? SP_INV : (unsigned short)ndq;
        }
        *(ushort4 *)(sp.jump + sp_lev_row(0, i) * SP_W + d0) = o;
    }
}

// tentative != 0: the walk over wc_spec_kernel's verdicts with the tentative ones taken at face value -- it only records where
// the path passes every query's window (sp.dtil) for wc_band_kernel and commits nothing.
// last != 0: the final launch of a request -- whatever is unresolved behind its block goes through the complete algorithm,
// query by query.  serial != 0 (development / tests: P2S_WC_SERIAL): no speculation at all, every query that way.
// lds_arrays != 0: the launch carries wc_lds_bytes(n) of dynamic LDS for the complete algorithm's arrays (stream skipping: the
// chip is idle, 23 us per query); 0: they live in global memory (60 us per query) and the kernel keeps 8.7 KB of LDS.
__global__ __launch_bounds__(256) void wc_chain_kernel(WcArgs a, WcSpec sp, int last, int serial, int tentative, int lds_arrays) {
    // Small on purpose: next to the encoders the kernel must fit the slot ONE retiring encoder workgroup frees.  The complete
    // algorithm runs rarely since r05 (undecided candidates: more than 64 first-round collisions, the remainder).
    extern __shared__ __attribute__((aligned(16))) unsigned char wc_lds[];
    __shared__ int wsum[16];
    __shared__ double wsumd[16];
    __shared__ long long s_ev[3];
    __shared__ unsigned short s_mark[SP_B];          // window coordinate of query j where the walk KNEW it; SP_INV = jumped over
    __shared__ short s_from[SP_B];
    __shared__ short s_lane[256];
    if (a.meta[1] != 0) return;
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x;
    const long long qb = sp.ctl[0];
    if (qb >= a.nq) return;
    const WcLds l = wc_carve(lds_arrays ? wc_lds : sp.scratch, a.n);
    const int BW = (a.n + 31) >> 5;
    if (!tentative)
        for (int i = tid; i < BW; i += 256) l.bitmap[i] = 0;
    const int lim = serial ? 0 : (int)(a.nq - qb < SP_B ? a.nq - qb : SP_B);
    const long long *s_klo = sp.klo;
    for (int j = tid; j < SP_B; j += 256) s_mark[j] = SP_INV;
    long long s = sp.ctl[1];
    int i = 0;
    long long t_fb = 0, n_fb = 0;
    const long long t_start = a.stats ? wall_clock64() : 0;
    __syncthreads();
    bool in_block = lim > 0;
    for (;;) {
        int ev = 2;                                   // 1 undecided candidate, 2 end of block / outside the window, 3 words exhausted
        if (in_block) {
            if (tid == 0) {
                while (i < lim) {
                    if (s + 2LL * a.nsel > a.cap_words) {
                        ev = 3;
                        break;
                    }
                    long long d = (s - s_klo[i]) >> 1;
                    if (d < 0 || d >= SP_W) break;
                    // as far as aligned jumps go: the largest level whose jump from here is valid (validity of a jump =
                    // validity of every step in it), again from where it lands, until not even one step is
                    for (;;) {
                        s_mark[i] = (unsigned short)d;
                        int k = i ? __builtin_ctz(i) : SP_LEV - 1;
                        k = k > SP_LEV - 1 ? SP_LEV - 1 : k;
                        bool moved = false;
                        for (; k >= 0; --k) {
                            if (i + (1 << k) > lim - 1) continue;
                            const unsigned short v = sp.jump[sp_lev_row(k, i) * SP_W + d];
                            if (v != SP_INV) {
                                i += 1 << k;
                                d = v;
                                moved = true;
                                break;
                            }
                        }
                        if (!moved) break;
                    }
                    s = s_klo[i] + 2 * d;
                    if (s + 2LL * a.nsel > a.cap_words) {
                        ev = 3;
                        break;
                    }
                    const unsigned r = sp.rtab[(size_t)i * SP_W + d];
                    if (tentative ? r == SP_UND : r >= SP_TENT) {
                        ev = tentative ? 2 : 1;       // undecided (exact walk: also a tentative verdict outside the band)
                        if (a.stats && !tentative && sp.dtil[i] >= 0) {      // how far the real path is from the tentative one
                            const long long dl_ = d - sp.dtil[i];
                            atomicMax((unsigned long long *)&a.stats[9], (unsigned long long)(dl_ < 0 ? -dl_ : dl_));
                        }
                        break;
                    }
                    if (!tentative) a.base[qb + i] = s;
                    s += 2LL * (a.nsel + (int)(r & 127u));
                    ++i;
                }
                s_ev[0] = ev;
                s_ev[1] = i;
                s_ev[2] = s;
            }
            __syncthreads();
            ev = (int)s_ev[0];
            i = (int)s_ev[1];
            s = s_ev[2];
            __syncthreads();
            if (tentative && ev == 3) ev = 2;         // words exhausted: the exact walk reports it
        }
        if (ev == 2 && in_block) {
            in_block = false;
            // word offsets of the queries the walk jumped over: nearest known start at or below j (prefix maximum over the
            // marks), then j - start steps by binary lifting (the start's alignment covers every level needed: the jump
            // that passed over j left from it).  Queries >= i were not reached.
            for (int j0 = tid * (SP_B / 256); j0 < (tid + 1) * (SP_B / 256); ++j0) s_from[j0] = s_mark[j0] != SP_INV ? (short)j0 : (short)-1;
            __syncthreads();
            {   // prefix maximum: 8 consecutive entries per lane, then across lanes
                const int b0 = tid * (SP_B / 256);
                short run = -1;
                for (int j0 = b0; j0 < b0 + SP_B / 256; ++j0) {
                    run = s_from[j0] > run ? s_from[j0] : run;
                    s_from[j0] = run;
                }
                s_lane[tid] = run;
                __syncthreads();
                short before = -1;
                for (int t = 0; t < tid; ++t) before = s_lane[t] > before ? s_lane[t] : before;
                for (int j0 = b0; j0 < b0 + SP_B / 256; ++j0)
                    if (s_from[j0] < before) s_from[j0] = before;
                __syncthreads();
            }
            for (int j = tid; j < i; j += 256) {
                int ii = s_from[j];
                if (ii < 0) continue;                         // (cannot happen: query 0 of the block is always a start)
                int dd = s_mark[ii];
                int m = j - ii;
                for (int k = SP_LEV - 1; k >= 0 && m > 0 && dd != SP_INV; --k) {
                    if (m >= (1 << k)) {
                        dd = (ii & ((1 << k) - 1)) ? SP_INV : sp.jump[sp_lev_row(k, ii) * SP_W + dd];
                        ii += 1 << k;
                        m -= 1 << k;
                    }
                }
                // (a start itself: m = 0 from the beginning.  The ids kernel re-derives every query's consumption and
                //  flags any disagreement, so a wrong offset cannot pass silently.)
                if (tentative) sp.dtil[j] = (m == 0 && dd != SP_INV) ? (short)dd : (short)-1;
                else if (m == 0 && dd != SP_INV) a.base[qb + j] = s_klo[j] + 2LL * dd;
                else a.meta[1] = 4;
            }
            if (tentative)
                for (int j = tid; j < lim; j += 256)
                    if (j >= i) sp.dtil[j] = -1;
        }
        if (tentative) return;                        // nothing is committed: the exact walk follows
        // words exhausted: in the block (the walk found out) or for the next query of the remainder
        if (ev == 3 || (ev == 2 && (last || serial) && qb + i < a.nq && s + 2LL * a.nsel > a.cap_words)) {
            if (tid == 0) {
                a.meta[1] = 2;
                a.meta[0] = s;
            }
            return;
        }
        // remainder (last launch of a request, or P2S_WC_SERIAL): in order, the whole algorithm per query (80 us each) --
        // after the spare (spec, chain) pairs normally nothing
        if (ev == 2 && !((last || serial) && qb + i < a.nq)) break;
        // the complete algorithm for query qb + i at word s: an undecided candidate of the block, or the remainder
        const long long t0 = a.stats ? wall_clock64() : 0;
        WcQuery qa;
        qa.Sq = a.S + (size_t)(qb + i) * a.n;
        qa.Rq = a.R + (size_t)(qb + i) * a.K;
        qa.Stot = a.stot[qb + i];
        qa.words = a.words + s;
        qa.words_left = a.cap_words - s;
        qa.n = a.n;
        qa.K = a.K;
        qa.nsel = a.nsel;
        const long long used = wc_full_query<false>(qa, l, wsum, wsumd, nullptr);
        if (used < 0) {
            if (tid == 0) {
                a.meta[1] = 3;
                a.meta[0] = s;
            }
            return;
        }
        if (tid == 0) a.base[qb + i] = s;
        s += used;
        ++i;
        if (a.stats) {
            t_fb += wall_clock64() - t0;
            ++n_fb;
        }
        __syncthreads();
    }
    const long long qn = qb + i;
    if (tid == 0) {
        if (a.stats) {
            atomicAdd((unsigned long long *)&a.stats[12], (unsigned long long)n_fb);
            atomicAdd((unsigned long long *)&a.stats[13], (unsigned long long)t_fb);
            atomicAdd((unsigned long long *)&a.stats[14], (unsigned long long)(wall_clock64() - t_start));
            atomicAdd((unsigned long long *)&a.stats[11], 1ull);            // chain launches that did work
        }
        sp.ctl[0] = qn;
        sp.ctl[1] = s;
        if (qn >= a.nq) a.meta[0] = s;
    }
}
