// a6, distance-weighted mode (p2s_vanilla, uniform_subsample=0): per query
//     p   = clip(1 - 1.5 * d / max(d), 0.05, 1);  p /= np.sum(p)                 (float32, utils.py:200-208)
//     ids = RandomState.choice(N, size=n, replace=False, p=p)                     (utils.py:219)
// reproduced bit for bit on the device, consuming the dataset-wide MT19937 stream exactly as numpy's legacy
// `choice` does (rand(m) doubles, cumsum/searchsorted, first-occurrence unique, redraw of the missing ones).
//
// What makes this exact:
//  * np.sum(float32[N]) = 8192-element buffer chunks added left to right, each chunk summed by numpy's pairwise
//    routine (128-element leaves with 8 strided accumulators, split at n/2 rounded down to a multiple of 8).  The
//    host turns that into a data-flow plan per cloud size (leaves, level-ordered binary ops); wc_tables_kernel
//    executes it -- same association, same float32 roundings.
//  * the probabilities are float32 values >= 0.05/N; converted to float64 every partial sum of them is a multiple
//    of 2^-52 below 2, i.e. EXACT.  np.cumsum (sequential) therefore equals a parallel scan in any order, and
//    "zero the found entries and cumsum again" (iterations >= 2) equals S_i minus the found mass below i.
//  * cdf_i = fl(S_i / S_N) and searchsorted(cdf, x, 'right') are evaluated with the same IEEE operations; a guide
//    table over K ~ N buckets, R[b] = (i = #{cdf_i <= b/K}, S_{i-1}, S_i, S_{i+1}), answers most look-ups with one
//    32-byte load.
//
// Two kernels per chunk of queries:
//   wc_tables_kernel  one workgroup per query, fully parallel: distances, max, plan-ordered sum, exact prefix sums
//                     S[q][N] (float64), guide records R[q][K].  HBM-resident (288 GB: ~2.7 MB per query).
//   wc_choice_kernel  ONE workgroup walks the queries in order -- the number of random words a query consumes
//                     depends on its collisions, so the stream position is a true serial dependence -- but per
//                     query it only does ~1000 table look-ups, an LDS bitmap for duplicates and a rank sort.
// The random words come from the jump-ahead generator (p2s_rng.hip, raw request); the generator is advanced by the
// count the choice kernel reports.
#include "p2s_common.h"
#include "p2s_internal.h"
#include <vector>
#include <algorithm>
#include <cstring>
#include <cstdlib>

#pragma clang fp contract(off)

namespace {

constexpr int WC_MAX_SEL = 1024;     // sub_sample_size limit (LDS arrays)
constexpr int WC_MAX_NODES = 8192;   // plan nodes held in LDS -> clouds up to ~390k points
constexpr int PW_BLOCK = 128;        // numpy PW_BLOCKSIZE
constexpr int NP_BUFSIZE = 8192;     // numpy ufunc buffer size (np.getbufsize())

struct __attribute__((aligned(32))) WcRec {
    int i, pad;
    double sm1, s0, sp1;     // S_{i-1} (0 for i = 0), S_i, S_{i+1} (S_{n-1} for i = n-1)
};

struct WcPlanDev {
    const int *leaf;       // [L][3] start, len, node
    const int *ops;        // [O][3] dst, a, b   (sorted by level)
    const int *lvl_off;    // [levels + 1] op ranges per level
    int n_leaves, n_levels, root;
};

// ---------------------------------------------------------------------------------------------------------------
// tables: one workgroup per query
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wc_clip_prob(float d, float dmax) {
    const float dn = d / dmax;
    const float pr = 1.0f - 1.5f * dn;
    return fminf(fmaxf(pr, 0.05f), 1.0f);
}

__global__ __launch_bounds__(256) void wc_tables_kernel(const float *__restrict__ pts, int n, const float *__restrict__ q,
                                                        WcPlanDev plan, int K, float *__restrict__ dist_all,
                                                        double *__restrict__ S_all, WcRec *__restrict__ R_all,
                                                        double *__restrict__ stot_all, long long *__restrict__ err) {
    __shared__ float nodes[WC_MAX_NODES];
    __shared__ float red_f[4];
    __shared__ double red_d[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = blockIdx.x;
    float *dist = dist_all + (size_t)qi * n;
    double *S = S_all + (size_t)qi * n;
    WcRec *R = R_all + (size_t)qi * K;
    const float qx = q[3 * qi], qy = q[3 * qi + 1], qz = q[3 * qi + 2];

    // pass 1: d_i = ||q - p_i|| (np.linalg.norm(axis=1): ((dx^2 + dy^2) + dz^2), sqrt), max
    float mx = 0.0f;
    for (int i = tid; i < n; i += 256) {
        const float dx = qx - pts[3 * i], dy = qy - pts[3 * i + 1], dz = qz - pts[3 * i + 2];
        const float d = sqrtf((dx * dx + dy * dy) + dz * dz);
        dist[i] = d;
        mx = fmaxf(mx, d);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) red_f[wave] = mx;
    __syncthreads();
    const float dmax = fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3]));
    if (!(dmax > 0.0f) || !(dmax < 3.0e38f)) {       // numpy would raise (NaN probabilities); flag and bail out
        if (tid == 0) err[1] = 1;
        return;
    }

    // pass 2: np.sum(pc) in numpy's association.  8 lanes per leaf = the 8 strided accumulators
    {
        const int g = tid >> 3, k = tid & 7;
        for (int lf = g; lf < plan.n_leaves; lf += 32) {
            const int st = plan.leaf[3 * lf], len = plan.leaf[3 * lf + 1], nd = plan.leaf[3 * lf + 2];
            float res = 0.0f;
            if (len < 8) {
                if (k == 0)
                    for (int i = 0; i < len; ++i) res += wc_clip_prob(dist[st + i], dmax);
            } else {
                const int body = len - (len & 7);
                float r = wc_clip_prob(dist[st + k], dmax);
                for (int i = 8; i < body; i += 8) r += wc_clip_prob(dist[st + i + k], dmax);
                r = r + __shfl_xor(r, 1);            // (r0+r1), (r2+r3), ...
                r = r + __shfl_xor(r, 2);            // (r0+r1)+(r2+r3), (r4+r5)+(r6+r7)
                r = r + __shfl_xor(r, 4);
                res = r;
                if (k == 0)
                    for (int i = body; i < len; ++i) res += wc_clip_prob(dist[st + i], dmax);
            }
            if (k == 0) nodes[nd] = res;
        }
    }
    __syncthreads();
    for (int lv = 0; lv < plan.n_levels; ++lv) {
        for (int o = plan.lvl_off[lv] + tid; o < plan.lvl_off[lv + 1]; o += 256)
            nodes[plan.ops[3 * o]] = nodes[plan.ops[3 * o + 1]] + nodes[plan.ops[3 * o + 2]];
        __syncthreads();
    }
    const float sum = nodes[plan.root];

    // pass 3: p_i = pc_i / sum (float32, kept in place of the distance) and the total mass S_N (exact in any order)
    double acc = 0.0;
    for (int i = tid; i < n; i += 256) {
        const float pi = wc_clip_prob(dist[i], dmax) / sum;
        dist[i] = pi;
        acc += (double)pi;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) red_d[wave] = acc;
    __syncthreads();
    const double Stot = (red_d[0] + red_d[1]) + (red_d[2] + red_d[3]);
    __syncthreads();                                  // red_d is reused by the scan below

    // pass 4: prefix sums + guide records, tiles of 1024 elements (4 consecutive per lane)
    const double dK = (double)K;
    double carry = 0.0;
    for (int t0 = 0; t0 < n; t0 += 1024) {
        const int i0 = t0 + 4 * tid;
        double p[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) p[j] = (i0 + j < n) ? (double)dist[i0 + j] : 0.0;
        const double l1 = p[0], l2 = l1 + p[1], l3 = l2 + p[2], l4 = l3 + p[3];
        double v = l4;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double u = __shfl_up(v, off);
            if (lane >= off) v += u;
        }
        if (lane == 63) red_d[wave] = v;
        __syncthreads();
        double base = carry, total = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) base += red_d[w];
            total += red_d[w];
        }
        const double excl = base + (v - l4);          // S_{i0-1}
        const double s[6] = {excl, excl + l1, excl + l2, excl + l3, excl + l4, (excl + l4) + p[4]};
        int cprev = (int)ceil((excl / Stot) * dK);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (i0 + j < n) {
                S[i0 + j] = s[j + 1];
                const int c = (int)ceil((s[j + 1] / Stot) * dK);
                const int ce = c < K ? c : K;
                WcRec rec;
                rec.i = i0 + j;
                rec.pad = 0;
                rec.sm1 = s[j];
                rec.s0 = s[j + 1];
                rec.sp1 = s[j + 2];                   // past the end: p = 0 -> S_{n-1}
                for (int b = cprev; b < ce; ++b) R[b] = rec;
                cprev = c;
            }
        }
        carry += total;
        __syncthreads();
    }
    if (tid == 0) stot_all[qi] = Stot;
}

// ---------------------------------------------------------------------------------------------------------------
// choice: one workgroup, queries in order
// ---------------------------------------------------------------------------------------------------------------
struct WcArgs {
    const double *S;          // [nq][n]
    const WcRec *R;           // [nq][K]
    const double *stot;       // [nq]
    const uint32_t *words;    // raw tempered words from the generator's position
    long long cap_words;
    int n, K, nq, nsel;
    int32_t *ids_out;         // [nq][nsel]
    long long *meta;          // [0] words consumed (out), [1] sticky error
    long long *prof;          // development: per-phase cycle counters (null = off)
};

struct WcLoc {
    int bin;
    double s, sprev;          // S_bin, S_{bin-1}
};

// searchsorted(cdf', x, 'right') = smallest i with fl((S_i - C(i)) / Stot_cur) > x, where the m_found ids sid[]
// (ascending) carry no mass any more: V[k] = S' at sid[k], C[k] = found mass up to and including sid[k].
// Step 1 (LDS only): the gap between two found ids that holds the answer, and the guide bucket to fetch.
struct WcGap {
    int lo, hi, bucket;
    double Ck;
};
__device__ __forceinline__ WcGap wc_gap(int n, int K, double Stot, double Stot_cur, double x, int m_found, const int *sid,
                                        const double *sV, const double *sC) {
    WcGap g;
    g.lo = 0;
    g.hi = n;
    g.Ck = 0.0;
    if (m_found) {
        int a = 0, b = m_found;                       // largest k in [0, m] with k == 0 or V[k-1]/Stot_cur <= x
        while (a < b) {
            const int mid = (a + b + 1) >> 1;
            if (sV[mid - 1] / Stot_cur <= x) a = mid;
            else b = mid - 1;
        }
        if (a) {
            g.lo = sid[a - 1] + 1;
            g.Ck = sC[a - 1];
        }
        if (a < m_found) g.hi = sid[a];
    }
    const double t = x * Stot_cur + g.Ck;
    int b = (int)((t / Stot) * (double)K);
    g.bucket = b < 0 ? 0 : (b > K - 1 ? K - 1 : b);
    return g;
}
// Step 2: finish from the guide record (one 32-byte load already done by the caller)
__device__ __forceinline__ WcLoc wc_finish(const double *__restrict__ Sq, int n, int i, double s_m1, double s_0,
                                           double s_p1, int g_lo, int g_hi, double Ck, double Stot_cur, double x) {
    if (i < g_lo || i > g_hi - 1) {                   // guide points outside the gap (iterations >= 2 only)
        i = i < g_lo ? g_lo : g_hi - 1;
        s_m1 = (i > 0) ? Sq[i - 1] : 0.0;
        s_0 = Sq[i];
        s_p1 = Sq[i + 1 < n ? i + 1 : n - 1];
    }
#define WC_PRED(sv) ((((sv)-Ck) / Stot_cur) > x)
    if (WC_PRED(s_0)) {
        while (i > g_lo && WC_PRED(s_m1)) {
            --i;
            s_0 = s_m1;
            s_m1 = (i > 0) ? Sq[i - 1] : 0.0;
        }
        return {i, s_0, s_m1};
    }
    if (WC_PRED(s_p1)) return {i + 1, s_p1, s_0};
    i += 1;
    double prev = s_p1;
    for (;;) {
        ++i;
        if (i >= n) return {n - 1, prev, prev};       // unreachable for valid tables; keeps the loop finite
        const double sv = Sq[i];
        if (WC_PRED(sv)) return {i, sv, prev};
        prev = sv;
    }
#undef WC_PRED
}

#define WC_T(k)                                                     \
    do {                                                            \
        if (a.prof && tid == 0) {                                   \
            const long long t_ = wall_clock64();                    \
            a.prof[k] += t_ - t_last;                               \
            t_last = t_;                                            \
        }                                                           \
    } while (0)

__global__ __launch_bounds__(256) void wc_choice_kernel(WcArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wc_lds[];
    __shared__ int wsum[4];
    __shared__ double wsumd[4];
    __shared__ int s_ncoll;
    __builtin_amdgcn_s_setprio(3);
    const int BW = (a.n + 31) >> 5;
    double *fS = (double *)wc_lds;                    // found, in numpy's order
    double *fP = fS + WC_MAX_SEL;
    double *sV = fP + WC_MAX_SEL;                     // found, sorted by id
    double *sC = sV + WC_MAX_SEL;
    int *fid = (int *)(sC + WC_MAX_SEL);
    int *sid = fid + WC_MAX_SEL;
    int *coll_bin = sid + WC_MAX_SEL;
    int *coll_min = coll_bin + WC_MAX_SEL;
    uint32_t *bitmap = (uint32_t *)(coll_min + WC_MAX_SEL);
    int *wpre = (int *)(bitmap + BW);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    if (a.meta[1] != 0) return;                       // tables invalid (degenerate input) or an earlier failure
    for (int i = tid; i < BW; i += 256) bitmap[i] = 0;
    if (tid == 0) s_ncoll = 0;
    __syncthreads();

    long long o = 0;                                  // words consumed so far (uniform)
    long long t_last = a.prof ? wall_clock64() : 0;
    for (int q = 0; q < a.nq; ++q) {
        const double *Sq = a.S + (size_t)q * a.n;
        const WcRec *Rq = a.R + (size_t)q * a.K;
        const double Stot = a.stot[q];
        double Stot_cur = Stot;
        int n_uniq = 0, m_found = 0, rounds = 0;
        while (n_uniq < a.nsel) {
            const int m = a.nsel - n_uniq;
            const int per = (m + 255) >> 8;
            if (o + 2LL * m > a.cap_words || ++rounds > 64) {
                if (tid == 0) {
                    a.meta[1] = (rounds > 64) ? 3 : 2;
                    a.meta[0] = o;
                }
                return;
            }
            // locate the bins of rand(m): A the doubles (two words each), B gap + guide record, C finish.
            int bins[4];
            double sb[4], sp[4];
            unsigned valid = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bins[j] = 0;
                sb[j] = sp[j] = 0.0;
            }
            if (m_found == 0) {
                // first round (nothing found yet, every lane has up to 4 draws): straight-line code, lanes past the
                // last draw repeat it, so that all loads of a phase are in flight together
                uint2 wpair[4];
                int dcl[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int d = tid * per + j;
                    const bool ok = (j < per) & (d < m);
                    valid |= (unsigned)ok << j;
                    dcl[j] = ok ? d : m - 1;
                    wpair[j] = *(const uint2 *)(a.words + o + 2LL * dcl[j]);      // o is even: 8-byte aligned
                }
                double xs[4];
                int r_i[4];
                double r_sm1[4], r_s0[4], r_sp1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xs[j] = ((double)(wpair[j].x >> 5) * 67108864.0 + (double)(wpair[j].y >> 6)) / 9007199254740992.0;
                    int bk = (int)(xs[j] * (double)a.K);              // x * Stot / Stot: the bucket of x itself
                    bk = bk > a.K - 1 ? a.K - 1 : bk;
                    const double *rp = (const double *)(Rq + bk);
                    r_i[j] = *(const int *)rp;
                    r_sm1[j] = rp[1];
                    r_s0[j] = rp[2];
                    r_sp1[j] = rp[3];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if ((valid >> j) & 1u) {
                        const WcLoc L = wc_finish(Sq, a.n, r_i[j], r_sm1[j], r_s0[j], r_sp1[j], 0, a.n, 0.0, Stot, xs[j]);
                        bins[j] = L.bin;
                        sb[j] = L.s;
                        sp[j] = L.sprev;
                    }
                }
            } else {
                // redraw rounds: few draws (usually one per lane, < 64 lanes)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int d = tid * per + j;
                    if (j < per && d < m) {
                        valid |= 1u << j;
                        const uint2 wp = *(const uint2 *)(a.words + o + 2LL * d);
                        const double x = ((double)(wp.x >> 5) * 67108864.0 + (double)(wp.y >> 6)) / 9007199254740992.0;
                        const WcGap g = wc_gap(a.n, a.K, Stot, Stot_cur, x, m_found, sid, sV, sC);
                        const double *rp = (const double *)(Rq + g.bucket);
                        const WcLoc L = wc_finish(Sq, a.n, *(const int *)rp, rp[1], rp[2], rp[3], g.lo, g.hi, g.Ck, Stot_cur, x);
                        bins[j] = L.bin;
                        sb[j] = L.s;
                        sp[j] = L.sprev;
                    }
                }
            }
            WC_T(m_found ? 1 : 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if ((valid >> j) & 1u) {
                    const uint32_t bit = 1u << (bins[j] & 31);
                    const uint32_t old = atomicOr(&bitmap[bins[j] >> 5], bit);
                    if (old & bit) {
                        const int c = atomicAdd(&s_ncoll, 1);
                        coll_bin[c] = bins[j];
                        coll_min[c] = 0x7fffffff;
                    }
                }
            }
            __syncthreads();
            WC_T(2);
            const int nc = s_ncoll;
            unsigned keep = valid;
            if (nc) {
                // a bin drawn more than once keeps its FIRST draw (np.unique(return_index) + sort)
                int slot[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    slot[j] = -1;
                    if ((valid >> j) & 1u) {
                        for (int c = 0; c < nc; ++c)
                            if (coll_bin[c] == bins[j]) {
                                slot[j] = c;
                                break;
                            }
                        if (slot[j] >= 0) atomicMin(&coll_min[slot[j]], tid * per + j);
                    }
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (slot[j] >= 0 && coll_min[slot[j]] != tid * per + j) keep &= ~(1u << j);
            }
            WC_T(3);
            // ordered compaction of the kept draws behind the ones found so far
            const int cnt = __popc(keep);
            int excl = 0, wtot = 0;
#pragma unroll
            for (int bit = 0; bit < 3; ++bit) {
                const unsigned long long mk = __ballot((cnt >> bit) & 1);
                excl += (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u)) << bit;
                wtot += __popcll(mk) << bit;
            }
            if (lane == 0) wsum[wave] = wtot;
            __syncthreads();
            int base = n_uniq, total = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (w < wave) base += wsum[w];
                total += wsum[w];
            }
            int r = base + excl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if ((keep >> j) & 1u) {
                    fid[r] = bins[j];
                    fS[r] = sb[j];
                    fP[r] = sb[j] - sp[j];
                    ++r;
                }
            }
            n_uniq += total;
            o += 2LL * m;
            if (tid == 0) s_ncoll = 0;
            __syncthreads();
            WC_T(4);
            if (n_uniq < a.nsel) {
                // found ids in ascending order through the bitmap: rank = set bits below
                const int wper = (BW + 255) >> 8, w0 = tid * wper;
                int local = 0;
                for (int i = 0; i < wper; ++i)
                    if (w0 + i < BW) local += __popc(bitmap[w0 + i]);
                int v = local;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int u = __shfl_up(v, off);
                    if (lane >= off) v += u;
                }
                if (lane == 63) wsum[wave] = v;
                __syncthreads();
                int run = v - local;
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    if (w < wave) run += wsum[w];
                for (int i = 0; i < wper; ++i) {
                    if (w0 + i < BW) {
                        wpre[w0 + i] = run;
                        run += __popc(bitmap[w0 + i]);
                    }
                }
                __syncthreads();
                for (int e = tid; e < n_uniq; e += 256) {
                    const int f = fid[e];
                    const int rank = wpre[f >> 5] + __popc(bitmap[f >> 5] & ((1u << (f & 31)) - 1u));
                    sid[rank] = f;
                    sV[rank] = fS[e];
                    sC[rank] = fP[e];
                }
                __syncthreads();
                // C = inclusive scan of the found masses (exact), V = S - C
                const int e0 = 4 * tid;
                double c[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j] = (e0 + j < n_uniq) ? sC[e0 + j] : 0.0;
                const double l1 = c[0], l2 = l1 + c[1], l3 = l2 + c[2], l4 = l3 + c[3];
                double vv = l4;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const double u = __shfl_up(vv, off);
                    if (lane >= off) vv += u;
                }
                if (lane == 63) wsumd[wave] = vv;
                __syncthreads();
                double bs = vv - l4;
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    if (w < wave) bs += wsumd[w];
                const double cs[4] = {bs + l1, bs + l2, bs + l3, bs + l4};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (e0 + j < n_uniq) {
                        sC[e0 + j] = cs[j];
                        sV[e0 + j] -= cs[j];
                    }
                }
                __syncthreads();
                Stot_cur = Stot - sC[n_uniq - 1];
                m_found = n_uniq;
                WC_T(5);
            }
        }
        for (int e = tid; e < a.nsel; e += 256) {
            const int f = fid[e];
            a.ids_out[(size_t)q * a.nsel + e] = f;
            bitmap[f >> 5] = 0;
        }
        __syncthreads();
        WC_T(6);
    }
    if (tid == 0) a.meta[0] = o;
}

// ---------------------------------------------------------------------------------------------------------------
// host: summation plan of np.sum(float32[n])
// ---------------------------------------------------------------------------------------------------------------
struct PlanBuilder {
    std::vector<int> leaf;                 // triples
    std::vector<int> op_lvl, op_d, op_a, op_b;
    int nodes = 0;
    int new_node() { return nodes++; }
    // returns node id, sets level
    int rec(int start, int m, int &level) {
        if (m <= PW_BLOCK) {
            const int d = new_node();
            leaf.push_back(start);
            leaf.push_back(m);
            leaf.push_back(d);
            level = 0;
            return d;
        }
        int n2 = m / 2;
        n2 -= n2 % 8;
        int la = 0, lb = 0;
        const int a = rec(start, n2, la);
        const int b = rec(start + n2, m - n2, lb);
        const int d = new_node();
        level = std::max(la, lb) + 1;
        op_lvl.push_back(level);
        op_d.push_back(d);
        op_a.push_back(a);
        op_b.push_back(b);
        return d;
    }
};

int wc_build_plan(p2s_cloud_s *c) {
    if (c->wc_plan) return P2S_OK;
    const int n = c->d.n;
    PlanBuilder pb;
    int acc = -1, lvl = 0;
    for (int c0 = 0; c0 < n; c0 += NP_BUFSIZE) {
        int lr = 0;
        const int r = pb.rec(c0, std::min(NP_BUFSIZE, n - c0), lr);
        if (acc < 0) {
            acc = r;
            lvl = lr;
        } else {
            lvl = std::max(lvl, lr) + 1;
            const int d = pb.new_node();
            pb.op_lvl.push_back(lvl);
            pb.op_d.push_back(d);
            pb.op_a.push_back(acc);
            pb.op_b.push_back(r);
            acc = d;
        }
    }
    if (pb.nodes > WC_MAX_NODES) {
        p2s_set_error("weighted sub-sample: cloud of %d points needs %d summation nodes (> %d)", n, pb.nodes, WC_MAX_NODES);
        return P2S_ECAPACITY;
    }
    const int n_ops = (int)pb.op_d.size();
    const int n_levels = lvl;
    std::vector<int> order(n_ops);
    for (int i = 0; i < n_ops; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return pb.op_lvl[x] < pb.op_lvl[y]; });
    std::vector<int> blob(pb.leaf);
    const size_t ops_at = blob.size();
    std::vector<int> lvl_off(n_levels + 1, 0);
    for (int i = 0; i < n_ops; ++i) {
        const int o = order[i];
        blob.push_back(pb.op_d[o]);
        blob.push_back(pb.op_a[o]);
        blob.push_back(pb.op_b[o]);
        lvl_off[pb.op_lvl[o]]++;                // level L (1-based) counted at index L
    }
    // lvl_off[L] currently = number of ops at level L (index 0 unused = 0) -> exclusive offsets per level-1
    std::vector<int> offs(n_levels + 1, 0);
    for (int L = 1; L <= n_levels; ++L) offs[L] = offs[L - 1] + lvl_off[L];
    const size_t lvl_at = blob.size();
    blob.insert(blob.end(), offs.begin(), offs.end());
    if (hipMalloc(&c->wc_plan, blob.size() * 4) != hipSuccess) {
        (void)hipGetLastError();
        p2s_set_error("weighted sub-sample: hipMalloc of the summation plan failed");
        return P2S_ENOMEM;
    }
    P2S_HIP_CHECK(hipMemcpy(c->wc_plan, blob.data(), blob.size() * 4, hipMemcpyHostToDevice));
    c->wc_leaves = (int)(pb.leaf.size() / 3);
    c->wc_ops_at = (int)ops_at;      // int offsets into the blob
    c->wc_lvl_at = (int)lvl_at;
    c->wc_levels = n_levels;
    c->wc_root = acc;
    return P2S_OK;
}

int wc_reserve(p2s_rng_s *r, size_t nq, size_t n, size_t K) {
    if (nq <= r->wc_cap_q && n <= r->wc_cap_n && K <= r->wc_cap_k) return P2S_OK;
    if (r->wc_dist) (void)hipFree(r->wc_dist);
    if (r->wc_S) (void)hipFree(r->wc_S);
    if (r->wc_T) (void)hipFree(r->wc_T);
    if (r->wc_stot) (void)hipFree(r->wc_stot);
    r->wc_dist = nullptr; r->wc_S = nullptr; r->wc_T = nullptr; r->wc_stot = nullptr;
    r->wc_cap_q = r->wc_cap_n = r->wc_cap_k = 0;
    nq = std::max(nq, r->wc_cap_q);
    if (hipMalloc(&r->wc_dist, nq * n * 4) != hipSuccess || hipMalloc(&r->wc_S, nq * n * 8) != hipSuccess ||
        hipMalloc(&r->wc_T, nq * K * sizeof(WcRec)) != hipSuccess || hipMalloc(&r->wc_stot, nq * 8) != hipSuccess) {
        (void)hipGetLastError();
        p2s_set_error("weighted sub-sample: hipMalloc of the per-query tables failed (%zu queries x %zu points)", nq, n);
        return P2S_ENOMEM;
    }
    r->wc_cap_q = nq;
    r->wc_cap_n = n;
    r->wc_cap_k = K;
    return P2S_OK;
}

}  // namespace

void p2s_wc_free_rng(p2s_rng_s *r) {
    if (r->wc_dist) (void)hipFree(r->wc_dist);
    if (r->wc_S) (void)hipFree(r->wc_S);
    if (r->wc_T) (void)hipFree(r->wc_T);
    if (r->wc_stot) (void)hipFree(r->wc_stot);
}

extern "C" int p2s_subsample_weighted(p2s_rng_t r, p2s_cloud_t c, const float *q_dev, int64_t nq, int n_sel,
                                      int32_t *ids_out_dev, float *pts_out_dev, void *stream) {
    if (!r || !c || !q_dev || nq < 0 || n_sel < 1 || !ids_out_dev) {
        p2s_set_error("p2s_subsample_weighted: bad argument (q_dev and ids_out_dev are required)");
        return P2S_EINVAL;
    }
    if (n_sel > WC_MAX_SEL) {
        p2s_set_error("p2s_subsample_weighted: sub_sample_size %d > %d", n_sel, WC_MAX_SEL);
        return P2S_EINVAL;
    }
    const int n = c->d.n;
    if (n < n_sel) {
        p2s_set_error("p2s_subsample_weighted: cloud has %d points < sub_sample_size %d (shuffle+pad path unsupported)", n,
                      n_sel);
        return P2S_EINVAL;
    }
    if (r->levels == 0) {
        p2s_set_error("p2s_subsample_weighted: needs the jump-ahead tables (p2s_rng_set_jump_tables)");
        return P2S_EINVAL;
    }
    if (nq == 0) return P2S_OK;
    P2S_HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    int rc = wc_build_plan(c);
    if (rc) return rc;
    int K = 1024;
    while (K < n) K <<= 1;
    // queries per raw request: numpy draws 2 words per double, n_sel doubles + a few % redraws per query
    const long long cap = p2s_rng_raw_capacity(r);
    long long per_req = cap / (long long)(2.0 * n_sel * 1.15);
    const long long req_env = getenv("P2S_WCHOICE_QUERIES") ? atoll(getenv("P2S_WCHOICE_QUERIES")) : 0;   // tests: force splitting
    if (req_env > 0) per_req = std::min(per_req, req_env);
    if (per_req < 1) {
        p2s_set_error("p2s_subsample_weighted: jump tables too small for one query");
        return P2S_EINVAL;
    }
    per_req = std::min<long long>(per_req, nq);
    rc = wc_reserve(r, (size_t)per_req, (size_t)n, (size_t)K);
    if (rc) return rc;
    WcPlanDev plan;
    plan.leaf = c->wc_plan;
    plan.ops = c->wc_plan + c->wc_ops_at;
    plan.lvl_off = c->wc_plan + c->wc_lvl_at;
    plan.n_leaves = c->wc_leaves;
    plan.n_levels = c->wc_levels;
    plan.root = c->wc_root;
    const int BW = (n + 31) / 32;
    size_t lds = (size_t)WC_MAX_SEL * (4 * 8 + 4 * 4) + (size_t)BW * 8;
    // the choice kernel is one latency-bound workgroup running next to the MFMA-saturated encoders: give it a CU of
    // its own by claiming most of that CU's LDS (same placement trick as the serial generator)
    static const size_t hog = getenv("P2S_RNG_LDS_HOG") ? (size_t)atoi(getenv("P2S_RNG_LDS_HOG")) : 120 * 1024;
    if (nq >= 64) lds = std::max(lds, hog);
    if (lds > 150 * 1024) {
        p2s_set_error("p2s_subsample_weighted: cloud of %d points does not fit the LDS bitmap", n);
        return P2S_ECAPACITY;
    }
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)wc_choice_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr = true;
    }
    long long *meta = p2s_rng_raw_meta(r);
    for (int64_t done = 0; done < nq;) {
        const int cur = (int)std::min<int64_t>(per_req, nq - done);
        rc = p2s_rng_raw_begin(r, s);
        if (rc) return rc;
        hipLaunchKernelGGL(wc_tables_kernel, dim3(cur), dim3(256), 0, s, c->d.pts, n, q_dev + (size_t)done * 3, plan, K,
                           r->wc_dist, r->wc_S, (WcRec *)r->wc_T, r->wc_stot, meta);
        WcArgs a;
        a.S = r->wc_S;
        a.R = (const WcRec *)r->wc_T;
        a.stot = r->wc_stot;
        a.words = r->tmp;
        a.cap_words = cap;
        a.n = n;
        a.K = K;
        a.nq = cur;
        a.nsel = n_sel;
        a.ids_out = ids_out_dev + (size_t)done * n_sel;
        a.meta = meta;
        a.prof = nullptr;
        static long long *prof_dev = nullptr;
        if (getenv("P2S_WC_PROF")) {
            if (!prof_dev) (void)hipMalloc(&prof_dev, 16 * 8);
            (void)hipMemsetAsync(prof_dev, 0, 16 * 8, s);
            a.prof = prof_dev;
        }
        hipLaunchKernelGGL(wc_choice_kernel, dim3(1), dim3(256), lds, s, a);
        if (a.prof) {
            long long h[16];
            (void)hipMemcpyAsync(h, prof_dev, sizeof(h), hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            fprintf(stderr, "[wc prof] %d queries, 100 MHz ticks: locate1 %lld locate2+ %lld mark %lld resolve %lld compact %lld "
                            "sortprep %lld output %lld\n", cur, h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
        }
        P2S_LAUNCH_CHECK("weighted sub-sample kernels");
        rc = p2s_rng_raw_commit(r, s);
        if (rc) return rc;
        done += cur;
    }
    if (pts_out_dev) return p2s_gather_points(c, ids_out_dev, nq * n_sel, pts_out_dev, stream);
    return P2S_OK;
}
