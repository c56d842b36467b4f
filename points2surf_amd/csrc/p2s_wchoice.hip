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
//    table over K ~ N buckets, R[b] ~ #{S_i <= b/K}, tells every look-up where to START reading S (a hint: the decision is
//    always taken on S itself, walking either way if the hint is off).
//
// Three stages per batch of queries:
//   wc_tables_kernel   one workgroup per query, fully parallel: distances, max, plan-ordered sum, exact prefix sums
//                      S[q][N] (float64), guide R[q][K].  HBM-resident (288 GB: ~0.5 MB per query).
//   offsets pass       where every query's draws start in the stream -- the number of random words a query consumes
//                      depends on its collisions, so the stream position is a true serial dependence: speculation
//                      tables over candidate starts on all CUs (wc_spec_kernel), jump tables, one light chain
//                      workgroup (below).
//   wc_ids_kernel      one workgroup per query again: with the offsets known, the complete algorithm in parallel.
// The random words come from a raw session of the jump-ahead generator (p2s_rng.hip); the word cursor lives on the
// device and the generator is advanced to it when the session is closed.
#include "p2s_common.h"
#include "p2s_internal.h"
#include <vector>
#include <algorithm>
#include <cstring>
#include <cstdlib>

#pragma clang fp contract(off)

namespace {

constexpr int WC_MAX_SEL = 1024;     // sub_sample_size limit (LDS arrays)
constexpr int WC_MAX_NODES = 8192;   // plan nodes held in LDS -> clouds up to 524,288 points (the found-bitmap next to
                                     // the chain kernel's arrays is the tighter limit: 475,040 points)
constexpr int PW_BLOCK = 128;        // numpy PW_BLOCKSIZE
constexpr int NP_BUFSIZE = 8192;     // numpy ufunc buffer size (np.getbufsize())

// guide entry of bucket b of the cdf (x in [b/K, (b+1)/K)):  i = #{cdf_j <= b/K}, the first candidate -- a HINT: the look-up
// (wc_finish) starts there and decides with the exact predicate on S_(i-1), S_i, S_(i+1), one round trip in all but the
// rare buckets that hold three or more boundaries.  r05: 4 bytes per bucket.  Rounds 1-4 kept {cdf_i, i, more} = 16 bytes,
// which decided 95 % of the first-round look-ups without touching S -- and made the tables kernel write 1 MB per query:
// timing-only ablations (no divisions, no power sums, no S store) left its 2.85 ms per 4096 queries unchanged; it was
// bound by those 4.3 GB of record writes.
typedef int WcRec;

struct WcPlanDev {
    const int *leaf;       // [L][3] start, len, node
    const int *ops;        // [O][3] dst, a, b   (sorted by level)
    const int *lvl_off;    // [levels + 1] op ranges per level
    int n_leaves, n_levels, root, n_nodes;
};

// ---------------------------------------------------------------------------------------------------------------
// tables: one workgroup per query
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wc_clip_prob(float d, float dmax) {
    const float dn = d / dmax;
    const float pr = 1.0f - 1.5f * dn;
    return fminf(fmaxf(pr, 0.05f), 1.0f);
}

// distances of 4 consecutive points i0 .. i0+3 to the query: np.linalg.norm(axis=1) = sqrt((dx^2 + dy^2) + dz^2).
// The cloud (<= 1.8 MB) stays in L2 for every workgroup; three 16-byte loads per thread, coalesced.  Points past the
// end give d = 0 (callers mask them).
__device__ __forceinline__ void wc_dist4(const float *__restrict__ pts, int n, int i0, float qx, float qy, float qz, float (&d)[4]) {
    float c[12];
    if (i0 + 4 <= n) {
        const float4 a = *(const float4 *)(pts + 3 * (size_t)i0), b = *(const float4 *)(pts + 3 * (size_t)i0 + 4),
                     e = *(const float4 *)(pts + 3 * (size_t)i0 + 8);
        c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
        c[8] = e.x; c[9] = e.y; c[10] = e.z; c[11] = e.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = i0 + j < n;
            c[3 * j] = in ? pts[3 * (size_t)(i0 + j)] : qx;
            c[3 * j + 1] = in ? pts[3 * (size_t)(i0 + j) + 1] : qy;
            c[3 * j + 2] = in ? pts[3 * (size_t)(i0 + j) + 2] : qz;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float dx = qx - c[3 * j], dy = qy - c[3 * j + 1], dz = qz - c[3 * j + 2];
        d[j] = sqrtf((dx * dx + dy * dy) + dz * dz);
    }
}
// squared distances of 4 consecutive points (pass 1: max d = sqrtf(max d^2) -- sqrtf is monotone and correctly rounded, so
// the square root is taken once per query instead of once per point)
__device__ __forceinline__ void wc_dist4_sq(const float *__restrict__ pts, int n, int i0, float qx, float qy, float qz, float (&d2)[4]) {
    float c[12];
    if (i0 + 4 <= n) {
        const float4 a = *(const float4 *)(pts + 3 * (size_t)i0), b = *(const float4 *)(pts + 3 * (size_t)i0 + 4),
                     e = *(const float4 *)(pts + 3 * (size_t)i0 + 8);
        c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
        c[8] = e.x; c[9] = e.y; c[10] = e.z; c[11] = e.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = i0 + j < n;
            c[3 * j] = in ? pts[3 * (size_t)(i0 + j)] : qx;
            c[3 * j + 1] = in ? pts[3 * (size_t)(i0 + j) + 1] : qy;
            c[3 * j + 2] = in ? pts[3 * (size_t)(i0 + j) + 2] : qz;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float dx = qx - c[3 * j], dy = qy - c[3 * j + 1], dz = qz - c[3 * j + 2];
        d2[j] = (dx * dx + dy * dy) + dz * dz;
    }
}
__device__ __forceinline__ float wc_dist1(const float *__restrict__ pts, int i, float qx, float qy, float qz) {
    const float dx = qx - pts[3 * (size_t)i], dy = qy - pts[3 * (size_t)i + 1], dz = qz - pts[3 * (size_t)i + 2];
    return sqrtf((dx * dx + dy * dy) + dz * dz);
}

// One workgroup per query.  Nothing per-point is kept between the passes: every pass re-derives distance ->
// clipped probability -> normalised probability from the L2-resident cloud (a few dozen VALU instructions) instead of
// round-tripping a per-query float array through HBM, and issues the loads of 4096 points (12 x 16 bytes per
// thread) before it consumes any: the kernel was bound by the latency of ~800 dependent one-element iterations
// per thread (6.3 ms per 4096 queries), not by its 1.4 MB of output per query.
constexpr int WC_BATCH = 4;      // sub-tiles of 1024 points in flight per thread
__global__ __launch_bounds__(256) void wc_tables_kernel(const float *__restrict__ pts, int n, const float *__restrict__ q,
                                                        WcPlanDev plan, int K,
                                                        double *__restrict__ S_all, WcRec *__restrict__ R_all,
                                                        double *__restrict__ stot_all, float *__restrict__ pmax_all,
                                                        float *__restrict__ mu_all, float *__restrict__ dsum_all, int nsel,
                                                        long long *__restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) float wc_tab_lds[];
    float *pcs = wc_tab_lds;                       // [NP_BUFSIZE] clipped probabilities of one numpy buffer chunk
    float *nodes = wc_tab_lds + NP_BUFSIZE;        // [plan.n_nodes]
    __shared__ float red_f[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = blockIdx.x;
    double *S = S_all + (size_t)qi * n;
    WcRec *R = R_all + (size_t)qi * K;
    const float qx = q[3 * qi], qy = q[3 * qi + 1], qz = q[3 * qi + 2];

    // pass 1: max distance, through the squares
    float mx = 0.0f;
    for (int t0 = 0; t0 < n; t0 += 1024 * WC_BATCH) {
        float d[WC_BATCH][4];
#pragma unroll
        for (int u = 0; u < WC_BATCH; ++u) wc_dist4_sq(pts, n, t0 + 1024 * u + 4 * tid, qx, qy, qz, d[u]);
#pragma unroll
        for (int u = 0; u < WC_BATCH; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) mx = fmaxf(mx, d[u][j]);        // points past the end contribute 0
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) red_f[wave] = mx;
    __syncthreads();
    const float dmax = sqrtf(fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3])));
    if (!(dmax > 0.0f) || !(dmax < 3.0e38f)) {       // numpy would raise (NaN probabilities); flag and bail out
        if (tid == 0) err[1] = 1;
        return;
    }

    // pass 2: np.sum(pc) in numpy's association, one ufunc buffer chunk (8192 elements = 64 leaves of 128, fewer /
    // other sizes in the last one) at a time through LDS.  8 lanes per leaf = the 8 strided accumulators.
    {
        const int g = tid >> 3, k = tid & 7;
        for (int c0 = 0, lf0 = 0; c0 < n; c0 += NP_BUFSIZE, lf0 += NP_BUFSIZE / PW_BLOCK) {
#pragma unroll
            for (int h = 0; h < NP_BUFSIZE / (1024 * WC_BATCH); ++h) {
                float d[WC_BATCH][4];
#pragma unroll
                for (int u = 0; u < WC_BATCH; ++u) wc_dist4(pts, n, c0 + 1024 * (WC_BATCH * h + u) + 4 * tid, qx, qy, qz, d[u]);
#pragma unroll
                for (int u = 0; u < WC_BATCH; ++u) {
                    float4 v;
                    v.x = wc_clip_prob(d[u][0], dmax); v.y = wc_clip_prob(d[u][1], dmax);
                    v.z = wc_clip_prob(d[u][2], dmax); v.w = wc_clip_prob(d[u][3], dmax);
                    *(float4 *)(pcs + 1024 * (WC_BATCH * h + u) + 4 * tid) = v;          // past the end: never read
                }
            }
            __syncthreads();
            const int lf1 = (c0 + NP_BUFSIZE < n) ? lf0 + NP_BUFSIZE / PW_BLOCK : plan.n_leaves;
            for (int lf = lf0 + g; lf < lf1; lf += 32) {
                const int st = plan.leaf[3 * lf] - c0, len = plan.leaf[3 * lf + 1], nd = plan.leaf[3 * lf + 2];
                float res = 0.0f;
                if (len < 8) {
                    if (k == 0)
                        for (int i = 0; i < len; ++i) res += pcs[st + i];
                } else {
                    const int body = len - (len & 7);
                    float r = pcs[st + k];
                    for (int i = 8; i < body; i += 8) r += pcs[st + i + k];
                    r = r + __shfl_xor(r, 1);            // (r0+r1), (r2+r3), ...
                    r = r + __shfl_xor(r, 2);            // (r0+r1)+(r2+r3), (r4+r5)+(r6+r7)
                    r = r + __shfl_xor(r, 4);
                    res = r;
                    if (k == 0)
                        for (int i = body; i < len; ++i) res += pcs[st + i];
                }
                if (k == 0) nodes[nd] = res;
            }
            __syncthreads();
        }
    }
    for (int lv = 0; lv < plan.n_levels; ++lv) {
        for (int o = plan.lvl_off[lv] + tid; o < plan.lvl_off[lv + 1]; o += 256)
            nodes[plan.ops[3 * o]] = nodes[plan.ops[3 * o + 1]] + nodes[plan.ops[3 * o + 2]];
        __syncthreads();
    }
    const float sum = nodes[plan.root];
    if (tid == 0) {                                   // what turns a point into its probability again (wc_spec_kernel)
        dsum_all[2 * qi] = dmax;
        dsum_all[2 * qi + 1] = sum;
    }

    // pass 3: p_i = pc_i / sum (float32): prefix sums + guide, tiles of 1024 elements (4 consecutive per lane + the first of
    // the next lane); the distances of the next tile are in flight while this one is scanned.  The same pass collects the
    // power sums and the widest bin (r05: they had a pass of their own, 30 % of the kernel's instructions, only because the
    // guide wanted the total mass S_N up front -- the guide is a hint, every look-up decides on S itself (wc_finish), so
    // its buckets are cut at cdf ~ S_i instead of S_i / S_N: S_N = 1 to ~1e-6, less than a bucket at any cloud size).
    const double dK = (double)K;
    double carry = 0.0;
    double acc2 = 0.0, acc3 = 0.0;
    float pm = 0.0f;
    __shared__ double red_p4[2][4];
    float dn[5];
    wc_dist4(pts, n, 4 * tid, qx, qy, qz, (float(&)[4])dn);
    dn[4] = 4 * tid + 4 < n ? wc_dist1(pts, 4 * tid + 4, qx, qy, qz) : 0.0f;
    for (int t0 = 0; t0 < n; t0 += 1024) {
        const int i0 = t0 + 4 * tid;
        double p[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const float pf = (i0 + j < n) ? wc_clip_prob(dn[j], dmax) / sum : 0.0f;
            p[j] = (double)pf;
            if (j < 4) {
                acc2 += p[j] * p[j];                               // power sums: expected collisions of the first round (below)
                acc3 += p[j] * p[j] * p[j];
                pm = fmaxf(pm, pf);
            }
        }
        if (t0 + 1024 < n) {
            wc_dist4(pts, n, i0 + 1024, qx, qy, qz, (float(&)[4])dn);
            dn[4] = i0 + 1028 < n ? wc_dist1(pts, i0 + 1028, qx, qy, qz) : 0.0f;
        }
        const double l1 = p[0], l2 = l1 + p[1], l3 = l2 + p[2], l4 = l3 + p[3];
        double v = l4;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double u = __shfl_up(v, off);
            if (lane >= off) v += u;
        }
        double *rd = red_p4[(t0 >> 10) & 1];          // alternating buffers: ONE barrier per tile
        if (lane == 63) rd[wave] = v;
        __syncthreads();
        double base = carry, total = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) base += rd[w];
            total += rd[w];
        }
        const double excl = base + (v - l4);          // S_{i0-1}
        const double sv[6] = {excl, excl + l1, excl + l2, excl + l3, excl + l4, (excl + l4) + p[4]};
        int cc[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) cc[k] = (int)ceil(sv[k] * dK);            // first bucket whose lower edge is >= S_k
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = i0 + j;
            if (idx < n) {
                S[idx] = sv[j + 1];
                const int c = cc[j + 1];
                const int ce = (c < K && idx < n - 1) ? c : K;               // the last id takes every bucket that is left
                for (int b = cc[j]; b < ce; ++b) R[b] = idx;
            }
        }
        carry += total;
    }
    const double Stot = carry;                        // S_N, exact in any order (every thread holds it)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        acc2 += __shfl_xor(acc2, off);
        acc3 += __shfl_xor(acc3, off);
        pm = fmaxf(pm, __shfl_xor(pm, off));
    }
    __shared__ double red_2[4], red_3[4];
    __syncthreads();                                  // red_f: every wave is past dmax
    if (lane == 0) {
        red_2[wave] = acc2;
        red_3[wave] = acc3;
        red_f[wave] = pm;
    }
    __syncthreads();
    if (tid == 0) {
        // E[nsel - #distinct bins of nsel draws] = C(nsel,2) sum p^2 - C(nsel,3) sum p^3 + ...: where the speculation
        // windows of the offsets pass are centred (a prediction only -- never part of the result)
        const double s2 = ((red_2[0] + red_2[1]) + (red_2[2] + red_2[3])) / (Stot * Stot);
        const double s3 = ((red_3[0] + red_3[1]) + (red_3[2] + red_3[3])) / (Stot * Stot * Stot);
        const double ns = (double)nsel;
        mu_all[qi] = (float)(0.5 * ns * (ns - 1.0) * s2 - ns * (ns - 1.0) * (ns - 2.0) / 6.0 * s3);
        pmax_all[2 * qi] = fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3]));     // widest bin of the cdf x S_N
        pmax_all[2 * qi + 1] = 0.0f;
    }
    if (tid == 0) stot_all[qi] = Stot;
}

// ---------------------------------------------------------------------------------------------------------------
// choice: one workgroup, queries in order
// ---------------------------------------------------------------------------------------------------------------
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
// fl(S / St) > x without the division in all but razor-thin cases: S/St >= x(1+2^-52) rounds to at least the double
// above x, S/St < x rounds to at most x; t = fl(x*St) is within 2^-53 of x*St, so 1e-15 of slack decides both.
__device__ __forceinline__ bool wc_gt(double S, double St, double x) {
    const double t = x * St;
    if (S > t * (1.0 + 1e-15)) return true;
    if (S < t * (1.0 - 1e-15)) return false;
    return (S / St) > x;
}
// Step 2: exact answer from a starting index near it (S values fetched here; walks are short and rare)
__device__ __forceinline__ WcLoc wc_finish(const double *__restrict__ Sq, int n, int i, int g_lo, int g_hi, double Ck,
                                           double Stot_cur, double x) {
    i = i < g_lo ? g_lo : (i > g_hi - 1 ? g_hi - 1 : i);
    // ONE round trip decides the answers i .. i + 4 (a guide bucket holds one or two cdf boundaries on average; the six
    // values sit in one or two 64-byte sectors); anything else walks, one dependent read per step
    double s_m1 = (i > 0) ? Sq[i - 1] : 0.0;
    double s_0 = Sq[i];
    double sp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) sp[k] = Sq[i + 1 + k < n ? i + 1 + k : n - 1];
#define WC_PRED(sv) wc_gt((sv)-Ck, Stot_cur, x)
    if (WC_PRED(s_0)) {
        while (i > g_lo && WC_PRED(s_m1)) {
            --i;
            s_0 = s_m1;
            s_m1 = (i > 0) ? Sq[i - 1] : 0.0;
        }
        return {i, s_0, s_m1};
    }
    double prev = s_0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i + 1 + k >= n) return {n - 1, prev, prev};       // unreachable for valid tables
        if (WC_PRED(sp[k])) return {i + 1 + k, sp[k], prev};
        prev = sp[k];
    }
    i += 4;
    for (;;) {
        ++i;
        if (i >= n) return {n - 1, prev, prev};       // unreachable for valid tables; keeps the loop finite
        const double sv = Sq[i];
        if (WC_PRED(sv)) return {i, sv, prev};
        prev = sv;
    }
#undef WC_PRED
}

// ---------------------------------------------------------------------------------------------------------------
// LDS layout shared by the kernels below
// ---------------------------------------------------------------------------------------------------------------
constexpr int WC_HASH = 2048;        // open-addressing table: bin -> first draw index of the round
struct WcLds {
    double *fS, *sV, *sC;            // S at the found ids (numpy order) / V, C of the found ids sorted by id
    float *fP;                       // probability of the found ids (float32 values, exact)
    int *fid, *sid;                  // found ids in numpy's order / ascending
    uint32_t *hash;                  // (bin << 10 | draw) packed, 0xffffffff = empty
    uint32_t *bitmap;                // one bit per cloud point: found so far
    uint16_t *wpre;                  // set bits before each bitmap word
};
__host__ __device__ inline size_t wc_lds_bytes(int n) {
    const size_t BW = (size_t)((n + 31) >> 5);
    return (size_t)WC_MAX_SEL * (3 * 8 + 4 + 2 * 4) + WC_HASH * 4 + BW * 4 + ((BW * 2 + 15) & ~(size_t)15);
}
__device__ __forceinline__ WcLds wc_carve(unsigned char *base, int n) {
    const int BW = (n + 31) >> 5;
    WcLds l;
    l.fS = (double *)base;
    l.sV = l.fS + WC_MAX_SEL;
    l.sC = l.sV + WC_MAX_SEL;
    l.fP = (float *)(l.sC + WC_MAX_SEL);
    l.fid = (int *)(l.fP + WC_MAX_SEL);
    l.sid = l.fid + WC_MAX_SEL;
    l.hash = (uint32_t *)(l.sid + WC_MAX_SEL);
    l.bitmap = l.hash + WC_HASH;
    l.wpre = (uint16_t *)(l.bitmap + BW);
    return l;
}

__device__ __forceinline__ double wc_double(uint32_t w0, uint32_t w1) {      // numpy legacy random_sample
    return ((double)(w0 >> 5) * 67108864.0 + (double)(w1 >> 6)) / 9007199254740992.0;
}

// ---------------------------------------------------------------------------------------------------------------
// One query, start to end, by one workgroup: the reference algorithm (all rounds).  The bitmap must be all zero on
// entry and is all zero again on return.  Returns the number of random words consumed (uniform), or -1 if the words
// ran out / no progress (error code stored by the caller).  With WRITE the ids go to ids_out[0..nsel).
// ---------------------------------------------------------------------------------------------------------------
struct WcQuery {
    const double *Sq;
    const WcRec *Rq;
    double Stot;
    const uint32_t *words;       // word o of this query's first draw at words[0]
    long long words_left;        // words available from there
    int n, K, nsel;
};

template <bool WRITE>
__device__ __forceinline__ long long wc_full_query(const WcQuery &qa, const WcLds &l, int *wsum, double *wsumd, int32_t *ids_out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int BW = (qa.n + 31) >> 5;
    const double Stot = qa.Stot;
    double Stot_cur = Stot;
    int n_uniq = 0, m_found = 0, rounds = 0;
    long long o = 0;
    while (n_uniq < qa.nsel) {
        const int m = qa.nsel - n_uniq;
        const int per = (m + 255) >> 8;
        if (o + 2LL * m > qa.words_left || ++rounds > 64) return -1;     // uniform
        for (int i = tid; i < WC_HASH; i += 256) l.hash[i] = 0xffffffffu;
        // locate the bins of rand(m): A the doubles (two words each), B gap + guide record, C finish
        int bins[4];
        double sb[4], sp[4];
        unsigned valid = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bins[j] = 0;
            sb[j] = sp[j] = 0.0;
        }
        if (m_found == 0) {
            // first round (nothing found yet, every lane has up to 4 draws): straight-line code, lanes past the last
            // draw repeat it, so that all loads of a phase are in flight together
            uint2 wpair[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = tid * per + j;
                const bool ok = (j < per) & (d < m);
                valid |= (unsigned)ok << j;
                wpair[j] = *(const uint2 *)(qa.words + o + 2LL * (ok ? d : m - 1));      // o is even: 8-byte aligned
            }
            double xs[4];
            int st[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xs[j] = wc_double(wpair[j].x, wpair[j].y);
                int bk = (int)(xs[j] * (double)qa.K);                 // the bucket of x itself
                bk = bk > qa.K - 1 ? qa.K - 1 : bk;
                st[j] = qa.Rq[bk];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if ((valid >> j) & 1u) {
                    const WcLoc L = wc_finish(qa.Sq, qa.n, st[j], 0, qa.n, 0.0, Stot, xs[j]);
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
                    const uint2 wp = *(const uint2 *)(qa.words + o + 2LL * d);
                    const double x = wc_double(wp.x, wp.y);
                    const WcGap g = wc_gap(qa.n, qa.K, Stot, Stot_cur, x, m_found, l.sid, l.sV, l.sC);
                    const WcLoc L = wc_finish(qa.Sq, qa.n, qa.Rq[g.bucket], g.lo, g.hi, g.Ck, Stot_cur, x);
                    bins[j] = L.bin;
                    sb[j] = L.s;
                    sp[j] = L.sprev;
                }
            }
        }
        __syncthreads();                                  // hash cleared
        // a bin drawn more than once in this round keeps its FIRST draw (np.unique(return_index) + sort):
        // hash bin -> smallest draw index
        unsigned slot[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            slot[j] = 0;
            if ((valid >> j) & 1u) {
                const uint32_t bin = (uint32_t)bins[j];
                const uint32_t packed = (bin << 10) | (uint32_t)(tid * per + j);
                atomicOr(&l.bitmap[bin >> 5], 1u << (bin & 31));
                uint32_t h = (bin * 2654435761u) >> 21;
                for (;;) {
                    uint32_t cur = l.hash[h];
                    if (cur == 0xffffffffu) {
                        const uint32_t old = atomicCAS(&l.hash[h], 0xffffffffu, packed);
                        if (old == 0xffffffffu) break;
                        cur = old;
                    }
                    if ((cur >> 10) == bin) {
                        atomicMin(&l.hash[h], packed);
                        break;
                    }
                    h = (h + 1) & (WC_HASH - 1);
                }
                slot[j] = h;
            }
        }
        __syncthreads();
        unsigned keep = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (((valid >> j) & 1u) && (l.hash[slot[j]] & 1023u) == (uint32_t)(tid * per + j)) keep |= 1u << j;
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
                l.fid[r] = bins[j];
                l.fS[r] = sb[j];
                l.fP[r] = (float)(sb[j] - sp[j]);           // exact: the float32 probability itself
                ++r;
            }
        }
        n_uniq += total;
        o += 2LL * m;
        __syncthreads();
        if (n_uniq < qa.nsel) {
            // found ids in ascending order through the bitmap: rank = set bits below
            const int wper = (BW + 255) >> 8, w0 = tid * wper;
            int local = 0;
            for (int i = 0; i < wper; ++i)
                if (w0 + i < BW) local += __popc(l.bitmap[w0 + i]);
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
                    l.wpre[w0 + i] = (uint16_t)run;
                    run += __popc(l.bitmap[w0 + i]);
                }
            }
            __syncthreads();
            for (int e = tid; e < n_uniq; e += 256) {
                const int f = l.fid[e];
                const int rank = l.wpre[f >> 5] + __popc(l.bitmap[f >> 5] & ((1u << (f & 31)) - 1u));
                l.sid[rank] = f;
                l.sV[rank] = l.fS[e];
                l.sC[rank] = (double)l.fP[e];
            }
            __syncthreads();
            // C = inclusive scan of the found masses (exact), V = S - C
            const int e0 = 4 * tid;
            double c[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = (e0 + j < n_uniq) ? l.sC[e0 + j] : 0.0;
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
                    l.sC[e0 + j] = cs[j];
                    l.sV[e0 + j] -= cs[j];
                }
            }
            __syncthreads();
            Stot_cur = Stot - l.sC[n_uniq - 1];
            m_found = n_uniq;
        }
    }
    for (int e = tid; e < qa.nsel; e += 256) {
        const int f = l.fid[e];
        if (WRITE) ids_out[e] = f;
        l.bitmap[f >> 5] = 0;
    }
    __syncthreads();
    return o;
}

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

// ids: one workgroup per query, every query's word offset known -> the full algorithm in parallel
__global__ __launch_bounds__(256) void wc_ids_kernel(WcArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wc_lds[];
    __shared__ int wsum[16];
    __shared__ double wsumd[16];
    const int tid = threadIdx.x;
    if (a.meta[1] != 0) return;
    const WcLds l = wc_carve(wc_lds, a.n);
    const int BW = (a.n + 31) >> 5;
    for (int i = tid; i < BW; i += 256) l.bitmap[i] = 0;
    __syncthreads();
    const int q = blockIdx.x;
    const long long base = a.fixed ? 0 : a.base[q];
    WcQuery qa;
    qa.Sq = a.S + (size_t)q * a.n;
    qa.Rq = a.R + (size_t)q * a.K;
    qa.Stot = a.stot[q];
    qa.words = a.words + base;
    qa.words_left = a.cap_words - base;
    qa.n = a.n;
    qa.K = a.K;
    qa.nsel = a.nsel;
    const long long used = wc_full_query<true>(qa, l, wsum, wsumd, a.ids_out + (size_t)q * a.nsel);
    if (a.fixed) {
        // rng.seed(42) before every query: the generator ends where the LAST query of the call left it
        if (tid == 0 && used < 0) a.meta[1] = 3;
        if (tid == 0 && q == a.nq - 1 && used >= 0) a.meta[0] = used;
        return;
    }
    // cross-check against the offsets pass: both must agree on where the next query starts
    const long long next = (q + 1 < a.nq) ? a.base[q + 1] : a.meta[0];
    if (tid == 0 && (used < 0 || base + used != next)) a.meta[1] = 4;
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
    c->wc_plan = (int *)p2s_pool_alloc(c->device, blob.size() * 4);
    if (!c->wc_plan) {
        p2s_set_error("weighted sub-sample: device allocation of the summation plan failed");
        return P2S_ENOMEM;
    }
    P2S_HIP_CHECK(hipMemcpy(c->wc_plan, blob.data(), blob.size() * 4, hipMemcpyHostToDevice));
    c->wc_leaves = (int)(pb.leaf.size() / 3);
    c->wc_ops_at = (int)ops_at;      // int offsets into the blob
    c->wc_lvl_at = (int)lvl_at;
    c->wc_levels = n_levels;
    c->wc_root = acc;
    c->wc_nodes = pb.nodes;
    return P2S_OK;
}

// per-query tables of one batch: S, guide records, scalars [stot f64][base i64][pmax f32 x 2][dsum f32 x 2][mu f32 + pad]
struct WcBatchMem {
    double *S;
    WcRec *R;
    double *stot;
    long long *base;
    float *pmax, *dsum, *mu;
};
WcBatchMem wc_batch_mem(p2s_rng_s *r) {
    WcBatchMem m;
    m.S = r->wc_S;
    m.R = (WcRec *)r->wc_T;
    m.stot = r->wc_sc;
    m.base = (long long *)(m.stot + r->wc_cap_q);
    m.pmax = (float *)(m.base + r->wc_cap_q);
    m.dsum = m.pmax + 2 * r->wc_cap_q;
    m.mu = m.dsum + 2 * r->wc_cap_q;
    return m;
}

void wc_free(p2s_rng_s *r) {
    if (r->wc_S) (void)hipFree(r->wc_S);
    if (r->wc_T) (void)hipFree(r->wc_T);
    if (r->wc_sc) (void)hipFree(r->wc_sc);
    if (r->wc_spec) (void)hipFree(r->wc_spec);
    if (r->wc_J) (void)hipFree(r->wc_J);
    r->wc_S = nullptr;
    r->wc_T = nullptr;
    r->wc_sc = nullptr;
    r->wc_spec = nullptr;
    r->wc_J = nullptr;
    r->wc_cap_q = r->wc_cap_n = r->wc_cap_k = 0;
}

constexpr size_t WC_SCRATCH = 160 * 1024;            // >= wc_lds_bytes(n) of every cloud the ids kernel takes
constexpr size_t WC_SPEC_BYTES = 64 + (size_t)SP_B * 16 + (size_t)SP_B * SP_W + (size_t)SP_B * 2 + 64 + (size_t)SP_B * SP_SAVE * 4 + 64 +
                                 WC_SCRATCH;

int wc_reserve(p2s_rng_s *r, size_t nq, size_t n, size_t K) {
    if (nq <= r->wc_cap_q && n <= r->wc_cap_n && K <= r->wc_cap_k) return P2S_OK;
    nq = std::max(nq, r->wc_cap_q);
    n = std::max(n, r->wc_cap_n);
    K = std::max(K, r->wc_cap_k);
    wc_free(r);
    const bool ok = hipMalloc(&r->wc_J, SP_JUMP_ROWS * SP_W * 2) == hipSuccess && hipMalloc(&r->wc_spec, WC_SPEC_BYTES) == hipSuccess &&
                    hipMalloc(&r->wc_S, nq * n * 8) == hipSuccess && hipMalloc(&r->wc_T, nq * K * sizeof(WcRec)) == hipSuccess &&
                    hipMalloc(&r->wc_sc, nq * 40) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        wc_free(r);
        p2s_set_error("weighted sub-sample: hipMalloc of the per-query tables failed (%zu queries x %zu points)", nq, n);
        return P2S_ENOMEM;
    }
    r->wc_cap_q = nq;
    r->wc_cap_n = n;
    r->wc_cap_k = K;
    return P2S_OK;
}

}  // namespace

void p2s_wc_free_rng(p2s_rng_s *r) { wc_free(r); }

static int wc_subsample(p2s_rng_s *r, p2s_cloud_s *c, const float *q_dev, int64_t nq, int n_sel, int32_t *ids_out_dev,
                        float *pts_out_dev, void *stream, bool fixed, uint32_t fixed_seed);

extern "C" int p2s_subsample_weighted(p2s_rng_t r, p2s_cloud_t c, const float *q_dev, int64_t nq, int n_sel,
                                      int32_t *ids_out_dev, float *pts_out_dev, void *stream) {
    return wc_subsample(r, c, q_dev, nq, n_sel, ids_out_dev, pts_out_dev, stream, false, 0u);
}

int p2s_wc_subsample_fixed(p2s_rng_s *r, p2s_cloud_s *c, const float *q_dev, int64_t nq, int n_sel, uint32_t seed,
                           int32_t *ids_out_dev, float *pts_out_dev, hipStream_t s) {
    return wc_subsample(r, c, q_dev, nq, n_sel, ids_out_dev, pts_out_dev, (void *)s, true, seed);
}

static int wc_subsample(p2s_rng_s *r, p2s_cloud_s *c, const float *q_dev, int64_t nq, int n_sel, int32_t *ids_out_dev,
                        float *pts_out_dev, void *stream, bool fixed, uint32_t fixed_seed) {
    if (!r || !c || !q_dev || nq < 0 || n_sel < 1 || (!ids_out_dev && pts_out_dev) || (fixed && !ids_out_dev)) {
        p2s_set_error("p2s_subsample_weighted: bad argument (q_dev is required, pts_out_dev needs ids_out_dev)");
        return P2S_EINVAL;
    }
    if (n_sel > WC_MAX_SEL) {
        p2s_set_error("p2s_subsample_weighted: sub_sample_size %d > %d", n_sel, WC_MAX_SEL);
        return P2S_EINVAL;
    }
    const int n = c->d.n;
    if (n < n_sel) {         // reference source/base/utils.py:221-226: no weighting, shuffle (in place) + zero padding
        const int rc = p2s_subsample_shuffle_pad(r, c, nq, n_sel, nullptr, ids_out_dev, stream);
        if (rc || !pts_out_dev) return rc;
        return p2s_gather_points(c, ids_out_dev, nq * n_sel, pts_out_dev, stream);
    }
    if (r->levels_max == 0) {
        p2s_set_error("p2s_subsample_weighted: needs the jump-ahead tables (p2s_rng_set_jump_tables)");
        return P2S_EINVAL;
    }
    if (nq == 0) return P2S_OK;
    P2S_HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    p2s_cloud_note_stream(c, s);
    int rc = wc_build_plan(c);
    if (rc) return rc;
    // guide buckets: a power of two in (n / 2, n] -- one to two cdf boundaries per bucket, all decided by the look-up's one
    // read of S (wc_finish); twice as many buckets (as until r04) only make the tables kernel write more
    int K = 1024;
    while (2 * K < n) K <<= 1;
    // queries per batch (table memory: 8 n + 4 K bytes per query, ~0.6 MB at 50k points); random words come from the raw session
    long long per_req = 4096;
    const long long req_env = getenv("P2S_WCHOICE_QUERIES") ? atoll(getenv("P2S_WCHOICE_QUERIES")) : 0;   // tests
    if (req_env > 0) per_req = std::min<long long>(req_env, 16384);
    per_req = std::min<long long>(per_req, nq);
    const long long cap = p2s_rng_session_words(r);
    // numpy draws 2 words per double, n_sel doubles + a few % redraws per query
    const long long margin = 8LL * n_sel + 6 * 128 + 8192;
    if ((long long)(2.0 * n_sel * 1.15) + margin > cap) {
        p2s_set_error("p2s_subsample_weighted: jump tables too small for one query");
        return P2S_EINVAL;
    }
    // one bit per cloud point next to the arrays of one query in the ids kernel's LDS: the size limit of this mode
    // (475,040 points; stated at the prototype)
    const size_t lds_ids = wc_lds_bytes(n);
    const size_t lds_limit = 160 * 1024 - 4096 - SP_B * 8 - SP_B * 4 - 1024;
    static_assert(160 * 1024 - 4096 - SP_B * 8 - SP_B * 4 - 1024 <= WC_SCRATCH, "chain scratch");
    if (lds_ids > lds_limit) {
        p2s_set_error("p2s_subsample_weighted: cloud of %d points does not fit the LDS bitmap (limit: %d points)", n,
                      (int)((lds_limit - wc_lds_bytes(0) - 16) / 6 * 32));
        return P2S_ECAPACITY;
    }
    rc = wc_reserve(r, (size_t)per_req, (size_t)n, (size_t)K);
    if (rc) return rc;
    WcPlanDev plan;
    plan.leaf = c->wc_plan;
    plan.ops = c->wc_plan + c->wc_ops_at;
    plan.lvl_off = c->wc_plan + c->wc_lvl_at;
    plan.n_leaves = c->wc_leaves;
    plan.n_levels = c->wc_levels;
    plan.root = c->wc_root;
    plan.n_nodes = c->wc_nodes;
    WcSpec sp;
    sp.ctl = (long long *)r->wc_spec;
    sp.klo = sp.ctl + 8;
    sp.klo1 = sp.klo + SP_B;
    sp.rtab = (unsigned char *)(sp.klo1 + SP_B);
    sp.dtil = (short *)(sp.rtab + (size_t)SP_B * SP_W);
    sp.save = (int *)(((uintptr_t)(sp.dtil + SP_B) + 63) & ~(uintptr_t)63);
    sp.scratch = (unsigned char *)(((uintptr_t)(sp.save + (size_t)SP_B * SP_SAVE) + 63) & ~(uintptr_t)63);
    sp.jump = r->wc_J;
    const bool serial_only = getenv("P2S_WC_SERIAL") != nullptr;   // development / tests: every query through the in-order remainder path
    // stream skipping (no ids wanted): nothing else runs on the chip, the chain kernel may take the LDS the complete algorithm
    // likes; next to the encoders (ids wanted) it must stay small
    const size_t skip_lds = ids_out_dev ? 0 : lds_ids;
    {   // per device (a process may drive several); the call is cheap
        (void)hipFuncSetAttribute((const void *)wc_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 146 * 1024);
        (void)hipFuncSetAttribute((const void *)wc_ids_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void *)wc_tables_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (NP_BUFSIZE + WC_MAX_NODES) * 4);
    }
    long long *meta = p2s_rng_raw_meta(r);
    static long long *stats_dev = nullptr;
    const bool want_stats = getenv("P2S_WC_STATS") != nullptr;
    if (want_stats && !stats_dev) (void)hipMalloc(&stats_dev, 16 * 8);
    const WcBatchMem m = wc_batch_mem(r);
    for (int64_t done = 0; done < nq;) {
        const int cur = (int)std::min<int64_t>(per_req, nq - done);
        if (fixed) {
            // a fresh generator per batch: every query of the batch reads the same words from its start
            if ((rc = p2s_rng_reseed(r, fixed_seed, s))) return rc;
            rc = p2s_rng_session_raw(r, (long long)(2.0 * n_sel * 1.5) + margin, s);
        } else {
            rc = p2s_rng_session_raw(r, (long long)(2.0 * n_sel * 1.15 * cur) + margin, s);
        }
        if (rc) return rc;
        hipLaunchKernelGGL(wc_tables_kernel, dim3(cur), dim3(256), (size_t)(NP_BUFSIZE + ((plan.n_nodes + 3) & ~3)) * 4, s, c->d.pts, n,
                           q_dev + (size_t)done * 3, plan, K, m.S, m.R, m.stot, m.pmax, m.mu, m.dsum, n_sel, meta);
        WcArgs a;
        a.S = m.S;
        a.R = m.R;
        a.stot = m.stot;
        a.pmax = m.pmax;
        a.dsum = m.dsum;
        a.pts = c->d.pts;
        a.q = q_dev + (size_t)done * 3;
        a.words = r->tmp;
        a.cap_words = cap;
        a.n = n;
        a.K = K;
        a.nq = cur;
        a.nsel = n_sel;
        a.base = m.base;
        a.ids_out = ids_out_dev ? ids_out_dev + (size_t)done * n_sel : nullptr;
        a.meta = meta;
        a.stats = nullptr;
        a.fixed = fixed ? 1 : 0;
        sp.mu = m.mu;
        if (want_stats) {
            (void)hipMemsetAsync(stats_dev, 0, 16 * 8, s);
            a.stats = stats_dev;
        }
        if (fixed) {
            // no serial dependence between the queries: the ids kernel alone (it reports the last query's consumption)
        } else if (serial_only) {
            hipLaunchKernelGGL(wc_ctl_init_kernel, dim3(1), dim3(1), 0, s, sp.ctl, meta);
            hipLaunchKernelGGL(wc_chain_kernel, dim3(1), dim3(256), skip_lds, s, a, sp, 1, 1, 0, skip_lds ? 1 : 0);
        } else {
            // per block of SP_B queries: speculation tables on all CUs (tentative where round-2 draws lie close together), the
            // tentative walk to find out where the path runs, the exact decision for the candidates near it, the walk.  One
            // spare round for a block that ends early (start outside the window); the last chain launch also takes whatever
            // is still unresolved then
            hipLaunchKernelGGL(wc_ctl_init_kernel, dim3(1), dim3(1), 0, s, sp.ctl, meta);
            const int blk = std::min(cur, SP_B);
            const int rounds = (cur + SP_B - 1) / SP_B + (cur > SP_B / 2 ? 1 : 0);
            int kmax = 0;
            for (int k = 1; k < SP_LEV && (1 << k) < blk; ++k) kmax = k;
            auto jumps = [&]() {        // levels 1 .. kmax of the jump tables, JM_LEV per launch: 3 launches for a full block
                for (int k0 = 0; k0 < kmax; k0 += JM_LEV) {
                    const int rows = (blk + (1 << k0) - 1) >> k0;
                    hipLaunchKernelGGL(wc_jumpm_kernel, dim3((rows + JM_ROWS - 1) / JM_ROWS), dim3(256), 0, s, a, sp, k0, kmax);
                }
            };
            for (int pr = 0; pr < rounds; ++pr) {
                hipLaunchKernelGGL(wc_spec_kernel, dim3(blk), dim3(256), 0, s, a, sp);
                jumps();
                hipLaunchKernelGGL(wc_chain_kernel, dim3(1), dim3(256), 0, s, a, sp, 0, 0, 1, 0);
                hipLaunchKernelGGL(wc_band_kernel, dim3(blk), dim3(256), 0, s, a, sp);
                jumps();
                hipLaunchKernelGGL(wc_chain_kernel, dim3(1), dim3(256), skip_lds, s, a, sp, pr == rounds - 1 ? 1 : 0, 0, 0, skip_lds ? 1 : 0);
            }
        }
        if (ids_out_dev) hipLaunchKernelGGL(wc_ids_kernel, dim3(cur), dim3(256), lds_ids, s, a);   // NULL: advance the stream only
        P2S_LAUNCH_CHECK("weighted sub-sample kernels");
        if (want_stats) {
            long long h[16];
            (void)hipMemcpyAsync(h, stats_dev, sizeof(h), hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            fprintf(stderr, "[wc stats] %d queries: chain launches that did work %lld, queries through the complete algorithm %lld "
                            "taking %lld of %lld 10-ns ticks (largest distance from the tentative path there: %lld draws); band: %lld "
                            "windows, %lld tentative candidates, %lld exact look-ups\n", cur, h[11], h[12], h[13], h[14], h[9], h[0], h[2], h[5]);
        }
        done += cur;
    }
    if (pts_out_dev) return p2s_gather_points(c, ids_out_dev, nq * n_sel, pts_out_dev, stream);
    return P2S_OK;
}
