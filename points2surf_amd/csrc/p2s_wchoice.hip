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
// Three kernels per chunk of queries:
//   wc_tables_kernel   one workgroup per query, fully parallel: distances, max, plan-ordered sum, exact prefix sums
//                      S[q][N] (float64), guide records R[q][K].  HBM-resident (288 GB: ~1.4 MB per query).
//   wc_offsets_kernel  ONE workgroup walks the queries in order -- the number of random words a query consumes
//                      depends on its collisions, so the stream position is a true serial dependence -- and computes
//                      nothing but that: where every query's draws start.
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

// guide record of bucket b of the cdf (x in [b/K, (b+1)/K)):  i = #{cdf_j <= b/K} is the first candidate,
// c0 = cdf_i.  x < c0 -> bin i; otherwise bin i+1 unless `more` (a second boundary may lie inside the bucket).
struct __attribute__((aligned(16))) WcRec {
    double c0;
    int i, more;
};

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
                                                        float *__restrict__ mu_all, int nsel, long long *__restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) float wc_tab_lds[];
    float *pcs = wc_tab_lds;                       // [NP_BUFSIZE] clipped probabilities of one numpy buffer chunk
    float *nodes = wc_tab_lds + NP_BUFSIZE;        // [plan.n_nodes]
    __shared__ float red_f[4];
    __shared__ double red_d[4];
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

    // pass 3: p_i = pc_i / sum (float32) -> the total mass S_N (exact in any order), power sums, max
    double acc = 0.0, acc2 = 0.0, acc3 = 0.0;
    float pm = 0.0f;
    for (int t0 = 0; t0 < n; t0 += 1024 * WC_BATCH) {
        float d[WC_BATCH][4];
#pragma unroll
        for (int u = 0; u < WC_BATCH; ++u) wc_dist4(pts, n, t0 + 1024 * u + 4 * tid, qx, qy, qz, d[u]);
#pragma unroll
        for (int u = 0; u < WC_BATCH; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (t0 + 1024 * u + 4 * tid + j < n) {
                    const float pi = wc_clip_prob(d[u][j], dmax) / sum;
                    acc += (double)pi;
                    acc2 += (double)pi * (double)pi;              // power sums: expected collisions of the first round (below)
                    acc3 += (double)pi * (double)pi * (double)pi;
                    pm = fmaxf(pm, pi);
                }
            }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        acc += __shfl_xor(acc, off);
        acc2 += __shfl_xor(acc2, off);
        acc3 += __shfl_xor(acc3, off);
        pm = fmaxf(pm, __shfl_xor(pm, off));
    }
    __shared__ double red_2[4], red_3[4];
    if (lane == 0) {
        red_d[wave] = acc;
        red_2[wave] = acc2;
        red_3[wave] = acc3;
        red_f[wave] = pm;
    }
    __syncthreads();
    const double Stot = (red_d[0] + red_d[1]) + (red_d[2] + red_d[3]);
    if (tid == 0) {
        // E[nsel - #distinct bins of nsel draws] = C(nsel,2) sum p^2 - C(nsel,3) sum p^3 + ...: where the speculation
        // windows of the offsets pass are centred (a prediction only -- never part of the result)
        const double s2 = ((red_2[0] + red_2[1]) + (red_2[2] + red_2[3])) / (Stot * Stot);
        const double s3 = ((red_3[0] + red_3[1]) + (red_3[2] + red_3[3])) / (Stot * Stot * Stot);
        const double ns = (double)nsel;
        mu_all[qi] = (float)(0.5 * ns * (ns - 1.0) * s2 - ns * (ns - 1.0) * (ns - 2.0) / 6.0 * s3);
    }
    if (tid == 0) {
        // pmax, and the number of cells of the close-pair grid of the serial kernel: pitch 1/G >= pmax / S_N
        const float pmx = fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3]));
        double Gd = floor(Stot / ((double)pmx * (1.0 + 1e-9)));
        Gd = Gd < 1.0 ? 1.0 : (Gd > 65536.0 ? 65536.0 : Gd);
        pmax_all[2 * qi] = pmx;
        pmax_all[2 * qi + 1] = __int_as_float((int)Gd);
    }
    __syncthreads();                                  // red_d is reused by the scan below

    // pass 4: prefix sums + guide records, tiles of 1024 elements (4 consecutive per lane + the first of the next
    // lane); the distances of the next tile are in flight while this one is scanned
    const double dK = (double)K;
    double carry = 0.0;
    __shared__ double red_p4[2][4];
    float dn[5];
    wc_dist4(pts, n, 4 * tid, qx, qy, qz, (float(&)[4])dn);
    dn[4] = 4 * tid + 4 < n ? wc_dist1(pts, 4 * tid + 4, qx, qy, qz) : 0.0f;
    for (int t0 = 0; t0 < n; t0 += 1024) {
        const int i0 = t0 + 4 * tid;
        double p[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) p[j] = (i0 + j < n) ? (double)(wc_clip_prob(dn[j], dmax) / sum) : 0.0;
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
        double cd[6];
        int cc[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            cd[k] = sv[k] / Stot;                     // cdf value exactly as numpy computes it
            cc[k] = (int)ceil(cd[k] * dK);            // first bucket whose lower edge is >= cdf
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = i0 + j;
            if (idx < n) {
                S[idx] = sv[j + 1];
                const int c = cc[j + 1];
                const int ce = c < K ? c : K;
                WcRec rec;
                rec.c0 = cd[j + 1];
                rec.i = idx;
                rec.more = 0;
                for (int b = cc[j]; b < ce; ++b) {
                    // only the bucket that contains cdf_idx can hold further boundaries
                    rec.more = (b == c - 1 && idx + 1 < n && cc[j + 2] == c) ? 1 : 0;
                    R[b] = rec;
                }
            }
        }
        carry += total;
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
    double s_m1 = (i > 0) ? Sq[i - 1] : 0.0;
    double s_0 = Sq[i];
    const double s_p1 = Sq[i + 1 < n ? i + 1 : n - 1];
#define WC_PRED(sv) wc_gt((sv)-Ck, Stot_cur, x)
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
// first-round shortcut: bin from the guide record alone (unmodified cdf), -1 if a second boundary must be checked
__device__ __forceinline__ int wc_quick_bin(double c0, int i, int more, double x) {
    return x < c0 ? i : (more ? -1 : i + 1);
}
__device__ __forceinline__ int wc_finish_cold(const double *Sq, int n, int i, double Stot, double x) {
    return wc_finish(Sq, n, i, 0, n, 0.0, Stot, x).bin;
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
    // optional: the first round's look-ups, already done by the speculation pass for the draws of its window
    // (bin, S_bin, S_{bin-1} of draw d at pre_bin[d], pre_s[d]); null = look them up here
    const int *pre_bin;
    const double2 *pre_s;
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
        if (m_found == 0 && qa.pre_bin) {
            // first round from the speculation pass's window: three coalesced loads per draw instead of the guide
            // record + the S values around the bin (random accesses: most of this function's time under load)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = tid * per + j;
                if (j < per && d < m) {
                    valid |= 1u << j;
                    bins[j] = qa.pre_bin[d];
                    const double2 ss = qa.pre_s[d];
                    sb[j] = ss.x;
                    sp[j] = ss.y;
                }
            }
        } else if (m_found == 0) {
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
                const WcRec rec = qa.Rq[bk];
                const int qb = wc_quick_bin(rec.c0, rec.i, rec.more, xs[j]);
                st[j] = qb >= 0 ? qb : rec.i + 1;
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
                    const WcLoc L = wc_finish(qa.Sq, qa.n, qa.Rq[g.bucket].i, g.lo, g.hi, g.Ck, Stot_cur, x);
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
    const float *pmax;        // [nq][2] largest probability of the query; cells of its close-pair grid (int bits)
    const uint32_t *words;    // raw tempered words from the generator's position
    long long cap_words;      // words this request may consume
    long long alloc_words;    // words readable in the buffer (>= cap_words)
    int n, K, nq, nsel;
    long long *base;          // [nq] word offset of every query's first draw (offsets kernel -> ids kernel)
    int32_t *ids_out;         // [nq][nsel]
    long long *meta;          // [0] words consumed (out), [1] sticky error
    long long *stats;         // development counters (null = off): [0] fallbacks, [1] window misses
    const long long *ctl;     // serial kernel: start at query ctl[0], word ctl[1] (null = query 0, word meta[0])
    int fixed;                // fixed_subsample: every query starts at word 0 of a freshly seeded generator
};

// ---------------------------------------------------------------------------------------------------------------
// offsets: ONE workgroup walks the queries in order and determines only where each query's draws start.
//
// A query consumes 2*nsel words for its first round plus 2 words per redraw.  The number of redraws of round 2 is the
// number of first-round draws that hit an already taken bin (a bitmap of atomicOr's counts them); further rounds are
// needed only if two round-2 draws fall into the same bin, which is impossible when they are pairwise farther apart
// than the widest bin of the modified cdf -- checked on the x values alone.  Only if that test fails (~1 % of the
// queries) the whole algorithm is run here (wc_full_query) to get the exact count.
// The look-ups of query q+1's first round are software-pipelined: while query q is processed, the bins of a window
// of nsel + RMAX consecutive doubles starting where query q+1 would start if query q needed no redraw are computed
// (its true start is R_q doubles later; R_q <= RMAX, else the window is recomputed).  Random words are staged through
// an LDS ring two queries ahead.
// ---------------------------------------------------------------------------------------------------------------
constexpr int WC_RMAX = 128;                         // redraws per query covered by the pipelined windows (x2: two queries ahead)
constexpr int WC_SLOTS = 1280;                       // window slots: 5 per lane >= nsel + 2 * RMAX
constexpr int WC_RING = 16384;                       // words
constexpr int WC_CELLW = 2048;                       // words per cell bitmap (65536 cells)
constexpr int WC_LIST = 256;                         // look-ups per window that need a second round (one per lane; more are resolved at once)

// window of query t anchored at word `anchor`, synchronously: bins of the doubles anchor + 2e, e < win -> wbin_t[e]
__device__ __forceinline__ void wc_window_sync(const WcArgs &a, int t, long long anchor, int win, int *wbin_t) {
    const double *Sq = a.S + (size_t)t * a.n;
    const WcRec *Rq = a.R + (size_t)t * a.K;
    const double Stot = a.stot[t];
    for (int e = threadIdx.x; e < win; e += (int)blockDim.x) {
        const long long w = anchor + 2LL * e;
        int bin = 0;
        if (w + 1 < a.cap_words) {
            const uint2 wp = *(const uint2 *)(a.words + w);
            const double x = wc_double(wp.x, wp.y);
            int bk = (int)(x * (double)a.K);
            bk = bk > a.K - 1 ? a.K - 1 : bk;
            const WcRec rec = Rq[bk];
            bin = wc_quick_bin(rec.c0, rec.i, rec.more, x);
            if (bin < 0) bin = wc_finish(Sq, a.n, rec.i + 1, 0, a.n, 0.0, Stot, x).bin;
        }
        wbin_t[e] = bin;
    }
}

#ifndef P2S_WC_NT
#define P2S_WC_NT 512
#endif
constexpr int WC_NT = P2S_WC_NT;                     // lanes of the serial kernel (256..768 measured: 512 is fastest, 7.9 us per query)
constexpr int WC_SL = (WC_SLOTS + WC_NT - 1) / WC_NT;        // window slots per lane
constexpr int WC_MD = (WC_MAX_SEL + WC_NT - 1) / WC_NT;      // first-round draws per lane
static_assert(WC_NT % 64 == 0 && WC_NT >= 256 && WC_NT <= 768, "serial kernel: whole waves, one workgroup");
struct WSet {                                        // guide records of one window
    double xs[WC_SL], c0[WC_SL];
    int ii[WC_SL], mm[WC_SL];
};
__device__ __forceinline__ void wc_lds_barrier() {
    // LDS-only synchronisation: __syncthreads() would add s_waitcnt vmcnt(0) and drain the loads kept in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

__host__ __device__ inline size_t wc_offsets_lds_bytes(int n) {
    return wc_lds_bytes(n) + (size_t)WC_RING * 4 + 2 * WC_SLOTS * 4 + WC_LIST * 16;
}

__global__ __launch_bounds__(WC_NT) void wc_offsets_kernel(WcArgs a_in) {
    WcArgs a = a_in;
    long long ctl_base = -1;
    if (a.ctl) {
        // remainder after the speculative passes (normally nothing): continue at query ctl[0], word ctl[1]
        const long long q0 = a.ctl[0];
        if (q0 >= a.nq) return;
        ctl_base = a.ctl[1];
        a.S += (size_t)q0 * a.n;
        a.R += (size_t)q0 * a.K;
        a.stot += q0;
        a.pmax += 2 * q0;
        a.base += q0;
        a.nq -= (int)q0;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char wc_lds[];
    __shared__ int wsum[16];
    __shared__ double wsumd[16];
    __shared__ int s_cnt, s_unsafe, s_nlist;
    __shared__ long long s_stats[16];
    __shared__ double s_St[128];                     // per-query scalars, staged 64..128 queries ahead with VECTOR loads
    __shared__ float2 s_pm[128];                     // (uniform-address loads would be scalar loads: their latency would
                                                     //  land on the next LDS barrier's lgkmcnt(0))
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x;
    const WcLds l = wc_carve(wc_lds, a.n);
    uint32_t *ring = (uint32_t *)(wc_lds + wc_lds_bytes(a.n));
    int *wbin = (int *)(ring + WC_RING);             // [2][SLOTS] bins of the windows of queries q, q+1
    double *list_x = (double *)(wbin + 2 * WC_SLOTS);
    int *list_e = (int *)(list_x + WC_LIST);
    int *list_i = list_e + WC_LIST;
    const int BW = (a.n + 31) >> 5;
    const int win = a.nsel + 2 * WC_RMAX;            // window length in doubles
    if (a.meta[1] != 0) return;                      // tables invalid (degenerate input) or an earlier failure
    for (int i = tid; i < BW; i += WC_NT) l.bitmap[i] = 0;
    for (int i = tid; i < 2 * WC_CELLW; i += WC_NT) ((uint32_t *)wc_lds)[i] = 0;
    if (tid == 0) s_cnt = s_unsafe = s_nlist = 0;
    if (tid < 16) s_stats[tid] = 0;
    if (tid < 128) {
        const int qq = tid < a.nq ? tid : a.nq - 1;
        s_St[tid] = a.stot[qq];
        s_pm[tid] = ((const float2 *)a.pmax)[qq];
    }

    // ring: words [.., r_hi) of the request are resident at ring[w & (RING-1)]
    const long long base0 = ctl_base >= 0 ? ctl_base : a.meta[0];   // word cursor: where this call's first query starts
    long long r_hi = base0 & ~3LL;
    auto ring_fill = [&](long long upto) {           // synchronous (prologue / after falling behind)
        upto = upto < a.cap_words ? upto : a.cap_words;
        for (long long w = r_hi + 4 * tid; w < upto; w += 4 * WC_NT) {
            const uint4 v = *(const uint4 *)(a.words + w);
            *(uint4 *)(ring + (w & (WC_RING - 1))) = v;
        }
        if (upto > r_hi) r_hi = (upto + 3) & ~3LL;
    };
    // asynchronous part 1: x values from the ring, guide records requested (one 16-byte load per slot)
    WSet RA, RB;                                     // ping-pong: no register copies of values still in flight
#pragma unroll
    for (int j = 0; j < WC_SL; ++j) {
        RA.xs[j] = RA.c0[j] = RB.xs[j] = RB.c0[j] = 0.0;
        RA.ii[j] = RA.mm[j] = RB.ii[j] = RB.mm[j] = 0;
    }
    // Two draws can only share a bin if their x values are closer than the widest bin of the cdf, pmax / S_N.  On a
    // grid of G <= S_N / pmax cells such draws sit in the same or in adjacent cells, so a slot whose cell neighbourhood
    // holds no other slot of the window cannot collide with anything: it needs no look-up at all (ii = -1).  About
    // 20 % of the slots remain; only those fetch their guide record.
    uint32_t *cellA = (uint32_t *)wc_lds;            // overlays the arrays of the fallback algorithm: zero on entry
    uint32_t *cellB = cellA + WC_CELLW;
    auto issue = [&](int t, long long anchor, float2 pmg, double (&xs)[WC_SL], double (&c0)[WC_SL], int (&ii)[WC_SL],
                     int (&mm)[WC_SL]) {
        const WcRec *Rn = a.R + (size_t)t * a.K;
        const int G = __float_as_int(pmg.y);         // cells of the close-pair grid (tables kernel), <= 32 * WC_CELLW
        int cell[WC_SL];
        uint32_t bit[WC_SL];
#pragma unroll
        for (int j = 0; j < WC_SL; ++j) {                // straight-line: the LDS round trips of the 5 slots overlap
            const int e = tid * WC_SL + j;
            const long long w = anchor + 2LL * (e < win ? e : win - 1);
            xs[j] = wc_double(ring[w & (WC_RING - 1)], ring[(w + 1) & (WC_RING - 1)]);
            int c = (int)(xs[j] * (double)G);
            c = c > G - 1 ? G - 1 : c;
            cell[j] = c;
            bit[j] = e < win ? 1u << (c & 31) : 0u;  // slots past the window: no-op atomics
        }
        uint32_t dup[WC_SL];
#pragma unroll
        for (int j = 0; j < WC_SL; ++j) dup[j] = atomicOr(&cellA[cell[j] >> 5], bit[j]) & bit[j];
#pragma unroll
        for (int j = 0; j < WC_SL; ++j) atomicOr(&cellB[cell[j] >> 5], dup[j]);
        wc_lds_barrier();
        uint32_t nb[WC_SL];
#pragma unroll
        for (int j = 0; j < WC_SL; ++j) {
            const int c = cell[j];
            const int cm = c > 0 ? c - 1 : c, cp = c + 1 < G ? c + 1 : c;
            const uint32_t wB = cellB[c >> 5], wM = cellA[cm >> 5], wP = cellA[cp >> 5];
            uint32_t near = (wB >> (c & 31)) & 1u;
            near |= (c > 0) ? (wM >> (cm & 31)) & 1u : 0u;
            near |= (c + 1 < G) ? (wP >> (cp & 31)) & 1u : 0u;
            nb[j] = near & (bit[j] != 0u ? 1u : 0u);
        }
#pragma unroll
        for (int j = 0; j < WC_SL; ++j) {
            c0[j] = 0.0;
            ii[j] = -1;
            mm[j] = 0;
            if (nb[j]) {
                int bk = (int)(xs[j] * (double)a.K);
                bk = bk > a.K - 1 ? a.K - 1 : bk;
                const WcRec rec = Rn[bk];
                c0[j] = rec.c0;
                ii[j] = rec.i;
                mm[j] = rec.more;
            }
        }
        wc_lds_barrier();
#pragma unroll
        for (int j = 0; j < WC_SL; ++j) {                // (slots past the window alias the cell of the last slot)
            cellA[cell[j] >> 5] = 0;
            cellB[cell[j] >> 5] = 0;
        }
    };
    auto covered = [&](long long anchor) {           // all words of a window staged and inside the request
        return anchor + 2LL * win <= r_hi && anchor + 2LL * win <= a.cap_words;
    };

    long long base = base0;                          // word offset of the current query's first draw (uniform)
    long long anc0 = base0, anc1 = base0 + 2LL * a.nsel;    // anchors of the windows of queries q and q+1
    bool issued1 = false;
    ring_fill(base0 + 8LL * a.nsel + 6 * WC_RMAX + 4096);
    wc_lds_barrier();
    wc_window_sync(a, 0, base0, win, wbin);
    // per-query scalars, loaded three queries ahead of their first use
    double St0 = s_St[0], St1 = s_St[1], St2 = s_St[2];
    float2 pm0 = s_pm[0], pm1 = s_pm[1], pm2 = s_pm[2];
    double stage_St = 0.0;                           // values on their way to s_St / s_pm (loaded a step ago)
    float2 stage_pm = make_float2(0.0f, 0.0f);
    int stage_q = -1;
    if (a.nq > 1 && covered(anc1)) {
        issue(1, anc1, pm1, RA.xs, RA.c0, RA.ii, RA.mm);
        issued1 = true;
    }
    wc_lds_barrier();

    long long t_last = a.stats ? wall_clock64() : 0;
    const long long wall0 = t_last, clk0 = a.stats ? clock64() : 0;
#define WC_T(k)                                                     \
    do {                                                            \
        if (a.stats && tid == 0) {                                  \
            const long long t_ = wall_clock64();                    \
            s_stats[k] += t_ - t_last;                              \
            t_last = t_;                                            \
        }                                                           \
    } while (0)
    // one query; P = records of the window of query q+1 (requested a step ago), C = set to fill for query q+2
    auto step = [&](const int q, WSet &P, WSet &C) -> bool {
        if (tid == 0) a.base[q] = base;
        if (base + 2LL * a.nsel > a.cap_words) {     // uniform
            if (tid == 0) {
                a.meta[1] = 2;
                a.meta[0] = base;
            }
            return false;
        }
        const double St3 = s_St[(q + 3) & 127];
        const float2 pm3 = s_pm[(q + 3) & 127];
        if (stage_q >= 0 && tid < 64) {              // scalars requested a step ago -> LDS (read >= 60 steps from now)
            s_St[(stage_q + tid) & 127] = stage_St;
            s_pm[(stage_q + tid) & 127] = stage_pm;
        }
        stage_q = -1;
        if ((q & 63) == 0 && q + 128 - 64 < a.nq + 64) {
            // entries [q+64, q+128) replace [q-64, q) of the 128-entry ring
            stage_q = q + 64;
            if (tid < 64) {
                const int qq = q + 64 + tid < a.nq ? q + 64 + tid : a.nq - 1;
                stage_St = a.stot[qq];
                stage_pm = ((const float2 *)a.pmax)[qq];
            }
        }
        // ---- stage more words: loads now, LDS stores at the end of the iteration
        const long long fill_to = base + 8LL * a.nsel + 6 * WC_RMAX;
        const bool do_fill = r_hi < fill_to && r_hi + 8 * WC_NT <= a.alloc_words;   // uniform: a whole chunk, 8 words per lane
        const long long f0 = r_hi + 8 * tid;
        uint4 fill[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) fill[k] = do_fill ? *(const uint4 *)(a.words + f0 + 4 * k) : make_uint4(0, 0, 0, 0);

        // ---- A: window of query q+1 (records requested one iteration ago): most slots are decided by the record,
        //         the rest goes to a list whose S values are requested now and looked at after the work on query q
        const double *Sn = a.S + (size_t)(q + 1) * a.n;
        int *wn = wbin + ((q + 1) & 1) * WC_SLOTS;
        if (issued1) {
#pragma unroll
            for (int j = 0; j < WC_SL; ++j) {
                const int e = tid * WC_SL + j;
                if (e < win) {
                    const int qb = P.ii[j] < 0 ? -2 : wc_quick_bin(P.c0[j], P.ii[j], P.mm[j], P.xs[j]);
                    if (qb != -1) {
                        wn[e] = qb;                  // bin, or -2: cannot collide with any other draw
                    } else {
                        const int k = atomicAdd(&s_nlist, 1);
                        if (k < WC_LIST) {
                            list_e[k] = e;
                            list_i[k] = P.ii[j] + 1;
                            list_x[k] = P.xs[j];
                        } else {
                            wn[e] = wc_finish_cold(Sn, a.n, P.ii[j] + 1, St1, P.xs[j]);
                        }
                    }
                }
            }
        }
        wc_lds_barrier();
        const int nl = s_nlist < WC_LIST ? s_nlist : WC_LIST;
        double sl[1][8];
        int le[1], li[1];
        double lx[1];
#pragma unroll
        for (int u = 0; u < 1; ++u) {
            const int k = tid;                       // WC_LIST <= WC_NT: one entry per lane
            le[u] = -1;
            li[u] = 0;
            lx[u] = 0.0;
#pragma unroll
            for (int v = 0; v < 8; ++v) sl[u][v] = 0.0;
            if (k < nl) {
                le[u] = list_e[k];
                li[u] = list_i[k];
                lx[u] = list_x[k];
#pragma unroll
                for (int v = 0; v < 8; ++v) sl[u][v] = Sn[li[u] + v < a.n ? li[u] + v : a.n - 1];
            }
        }
        WC_T(2);
        // ---- I: request the records of the window of query q+2 (anchor: no redraws in q and q+1)
        const long long anc2 = base + 4LL * a.nsel;
        const bool issued2 = (q + 2 < a.nq) && covered(anc2);
        if (issued2) issue(q + 2, anc2, pm2, C.xs, C.c0, C.ii, C.mm);
        WC_T(3);

        // ---- Q: query q itself: number of distinct bins of its first round
        const int rho = (int)((base - anc0) >> 1);                       // redraw doubles skipped in the window
        const int *wb = wbin + (q & 1) * WC_SLOTS;
        int events = 0;
        int mbin[WC_MD];
#pragma unroll
        for (int k = 0; k < WC_MD; ++k) {            // nsel <= 1024 <= WC_MD * WC_NT, straight-line
            const int d = tid + WC_NT * k;
            const int bin = wb[rho + (d < a.nsel ? d : 0)];
            mbin[k] = d < a.nsel ? bin : -1;
        }
#pragma unroll
        for (int k = 0; k < WC_MD; ++k) {
            const uint32_t bit = mbin[k] >= 0 ? 1u << (mbin[k] & 31) : 0u;
            events += (atomicOr(&l.bitmap[mbin[k] >= 0 ? mbin[k] >> 5 : 0], bit) & bit) ? 1 : 0;
        }
        if (events) atomicAdd(&s_cnt, events);
        wc_lds_barrier();
        const int m2 = s_cnt;
        WC_T(4);
        long long used = 2LL * a.nsel + 2LL * m2;
        bool fallback = false;
        if (m2 > 0) {
            // round 2 draws m2 doubles right behind the first round; they are distinct for sure if pairwise farther
            // apart than the widest possible bin of the modified cdf
            const double pm = (double)pm0.x;
            const double denom = St0 - (double)a.nsel * pm;
            if (m2 > 64 || !(denom > 0.25 * St0) || base + used > r_hi) {
                fallback = true;
            } else {
                const double wmax = (pm / denom) * (1.0 + 1e-9);     // widest bin of the modified cdf (+ rounding slack)
                if (tid < 64) {                      // wave 0: lane e holds draw e, compared against all through readlane
                    const long long w = base + 2LL * a.nsel + 2LL * (tid < m2 ? tid : 0);
                    const double x = wc_double(ring[w & (WC_RING - 1)], ring[(w + 1) & (WC_RING - 1)]);
                    const int xl = __double2loint(x), xh = __double2hiint(x);
                    bool bad = false;
                    for (int e = 0; e < m2; ++e) {
                        const double y = __hiloint2double(__builtin_amdgcn_readlane(xh, e), __builtin_amdgcn_readlane(xl, e));
                        bad |= (e != tid) && (fabs(x - y) <= wmax);
                    }
                    if (bad && tid < m2) s_unsafe = 1;
                }
                wc_lds_barrier();
                fallback = s_unsafe != 0;
            }
        }
        WC_T(5);
        // clear the bits of this query (every set bit belongs to one of its bins)
#pragma unroll
        for (int k = 0; k < WC_MD; ++k)
            if (mbin[k] >= 0) l.bitmap[mbin[k] >> 5] = 0;
        wc_lds_barrier();
        if (tid == 0) s_cnt = s_unsafe = 0;
        if (fallback) {
            WcQuery qa;
            qa.Sq = a.S + (size_t)q * a.n;
            qa.Rq = a.R + (size_t)q * a.K;
            qa.Stot = St0;
            qa.words = a.words + base;
            qa.words_left = a.cap_words - base;
            qa.n = a.n;
            qa.K = a.K;
            qa.nsel = a.nsel;
            qa.pre_bin = nullptr;
            qa.pre_s = nullptr;
            used = wc_full_query<false>(qa, l, wsum, wsumd, nullptr);
            for (int i = tid; i < 2 * WC_CELLW; i += WC_NT) ((uint32_t *)wc_lds)[i] = 0;    // cell maps overlay its arrays
            wc_lds_barrier();
            if (tid == 0 && a.stats) s_stats[0] += 1;
            if (used < 0) {
                if (tid == 0) {
                    a.meta[1] = 3;
                    a.meta[0] = base;
                }
                return false;
            }
        }
        WC_T(6);
        // ---- B: second round of the window of query q+1
#pragma unroll
        for (int u = 0; u < 1; ++u) {
            if (le[u] >= 0) {
                int bin = -1;
#pragma unroll
                for (int v = 0; v < 8; ++v)
                    if (bin < 0 && li[u] + v < a.n && wc_gt(sl[u][v], St1, lx[u])) bin = li[u] + v;
                if (bin < 0) bin = wc_finish_cold(Sn, a.n, li[u] + 8, St1, lx[u]);
                wn[le[u]] = bin;
            }
        }
        const long long base_next = base + used;
        if (q + 1 < a.nq) {
            const long long rho1 = (base_next - anc1) >> 1;
            if (!issued1 || rho1 > 2 * WC_RMAX) {
                // window miss (more redraws than the window covers, or its words were not staged in time)
                wc_lds_barrier();
                wc_window_sync(a, q + 1, base_next, win, wbin + ((q + 1) & 1) * WC_SLOTS);
                anc1 = base_next;
                if (tid == 0 && a.stats) s_stats[1] += 1;
            }
        }
        WC_T(7);
        // ---- ring: store the words loaded at the top; refill synchronously if the pipeline fell behind
        if (do_fill) {
#pragma unroll
            for (int k = 0; k < 2; ++k) *(uint4 *)(ring + ((f0 + 4 * k) & (WC_RING - 1))) = fill[k];
            r_hi += 8 * WC_NT;
        }
        if (tid == 0) s_nlist = 0;
        wc_lds_barrier();
        if (r_hi < base_next + 6LL * a.nsel + 4 * WC_RMAX && r_hi < a.cap_words) {
            ring_fill(base_next + 8LL * a.nsel + 6 * WC_RMAX);
            wc_lds_barrier();
        }
        // rotate the pipeline
        base = base_next;
        anc0 = anc1;
        anc1 = anc2;
        issued1 = issued2;
        St0 = St1;
        St1 = St2;
        St2 = St3;
        pm0 = pm1;
        pm1 = pm2;
        pm2 = pm3;
        WC_T(8);
        return true;
    };
    for (int q = 0; q < a.nq; q += 2) {
        if (!step(q, RA, RB)) return;
        if (q + 1 < a.nq && !step(q + 1, RB, RA)) return;
    }
#undef WC_T
    if (tid == 0) a.meta[0] = base;
    if (a.stats && tid == 0) {
        s_stats[9] = wall_clock64() - wall0;
        s_stats[10] = clock64() - clk0;
    }
    wc_lds_barrier();
    if (a.stats && tid < 16) a.stats[tid] = s_stats[tid];
}

// The same contract as wc_offsets_kernel for clouds whose found-bitmap does not fit next to that kernel's ring and
// windows (more than 185,664 points): the complete algorithm query by query, in order, with the LDS layout of the ids
// kernel (bitmaps of up to ~570k points).  80 us per query instead of 8 -- but it only sees what the speculative chain
// left over (normally nothing), or everything when that chain is switched off.
__global__ __launch_bounds__(256) void wc_offsets_plain_kernel(WcArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wc_lds[];
    __shared__ int wsum[16];
    __shared__ double wsumd[16];
    if (a.meta[1] != 0) return;
    const int tid = threadIdx.x;
    long long q0 = 0, s = a.meta[0];
    if (a.ctl) {
        q0 = a.ctl[0];
        if (q0 >= a.nq) return;
        s = a.ctl[1];
    }
    const WcLds l = wc_carve(wc_lds, a.n);
    const int BW = (a.n + 31) >> 5;
    for (int i = tid; i < BW; i += 256) l.bitmap[i] = 0;
    __syncthreads();
    for (long long q = q0; q < a.nq; ++q) {
        if (s + 2LL * a.nsel > a.cap_words) {
            if (tid == 0) {
                a.meta[1] = 2;
                a.meta[0] = s;
            }
            return;
        }
        WcQuery qa;
        qa.Sq = a.S + (size_t)q * a.n;
        qa.Rq = a.R + (size_t)q * a.K;
        qa.Stot = a.stot[q];
        qa.words = a.words + s;
        qa.words_left = a.cap_words - s;
        qa.n = a.n;
        qa.K = a.K;
        qa.nsel = a.nsel;
        qa.pre_bin = nullptr;
        qa.pre_s = nullptr;
        const long long used = wc_full_query<false>(qa, l, wsum, wsumd, nullptr);
        if (used < 0) {
            if (tid == 0) {
                a.meta[1] = 3;
                a.meta[0] = s;
            }
            return;
        }
        if (tid == 0) a.base[q] = s;
        s += used;
        __syncthreads();
    }
    if (tid == 0) a.meta[0] = s;
}

// ---------------------------------------------------------------------------------------------------------------
// offsets, parallel: speculation tables + a light chain (default; the serial kernel above remains as the fallback).
//
// The only serial dependence is the word offset: query q starts at s_q = s_{q-1} + 2 (nsel + R_{q-1}), R = redraws.
// R_q is a function of the start alone: with X_k the k-th double of the stream and bin_q() the query's cdf look-up,
//     m2(s) = nsel - #distinct{ bin_q(X_k) : s <= k < s + nsel }          (first-round draws hitting a taken bin)
// and round 3 is impossible when the m2 redraws X_{s+nsel} .. X_{s+nsel+m2-1} are pairwise farther apart than the
// widest bin of the modified cdf (same tests as the serial kernel), so R(s) = m2(s) for such s.
//   wc_spec_kernel   one workgroup per query of a block of SP_B queries, all CUs: R_q(s) for the SP_W candidate
//                    starts around the predicted one (block start, exact, + sum of the expected collision counts mu
//                    of the queries before it, from the tables kernel).  m2 over a sliding window: every draw e with
//                    an earlier draw prev(e) in the same bin adds 1 to the starts in (e - nsel, prev(e)] -- a difference
//                    array + scan; draws sharing a bin are found through an LDS hash.  255 = undecided here.
//   wc_chain_kernel  one workgroup: s -> s + 2 (nsel + R_q(s)) is a table look-up per query; undecided candidates
//                    (~1-5 % of the queries) run the complete algorithm in place (wc_full_query); a start outside the
//                    window ends the block early, the next (spec, chain) pair resumes there.
// The ids kernel re-derives every query's consumption and flags any disagreement (meta[1] = 4).
// ---------------------------------------------------------------------------------------------------------------
#ifndef P2S_SP_B
#define P2S_SP_B 2048
#endif
constexpr int SP_B = P2S_SP_B;                       // queries per speculation block
constexpr int SP_W = 1024;                           // candidate starts per query
constexpr int SP_LOOK = 64;                          // round-2 draws decided by the distance test
constexpr int SP_NB = SP_W + WC_MAX_SEL;             // draws whose bin is needed
constexpr int SP_NX = SP_NB + 2 * SP_LOOK;           // doubles held
constexpr int SP_HASH = 4096;
constexpr int SP_DMAX = 1024;                        // draws that share their bin with another draw of the window
struct WcSpec {
    unsigned char *rtab;      // [SP_B][SP_W] redraws for candidate start klo + 2 d; 255 = undecided
    long long *klo;           // [SP_B] word offset of candidate 0
    long long *ctl;           // [0] first unresolved query, [1] its word offset
    const float *mu;          // [nq] expected first-round collisions
    int *win_bin;             // [SP_B][SP_NB] first-round bin of every draw of the window (-1: past the request)
    double2 *win_s;           // [SP_B][SP_NB] (S_bin, S_{bin-1}) of it: what the complete algorithm needs of round 1
    unsigned short *jump;     // [SP_LEV][SP_B][SP_W] jump tables (below); nullptr = walk query by query
};

// ---- the walk s -> s + 2 (nsel + R_q(s)) without walking --------------------------------------------------------------
// In window coordinates (d = (s - klo[q]) / 2, candidate d of query q) one step is
//     next(q, d) = d + R_q(d) + nsel + (klo[q] - klo[q + 1]) / 2        if R_q(d) is decided and the result is a candidate of q + 1
// -- a table look-up.  2^k steps at once are the look-up J_k[q][d] with J_k = J_{k-1} o J_{k-1} (all CUs, 2 M entries
// per level, 11 levels for a block of 2048 queries), so the one workgroup that owns the serial dependence no longer pays
// one dependent L2 round trip per QUERY (1.4 us each, 2.9 ms per block) but ~11 per SEGMENT between two queries it has
// to resolve itself (undecided candidates: ~5 % of the queries), and the word offsets of the queries inside a segment
// are filled in by all its lanes afterwards (binary lifting from the segment start).
constexpr int SP_LEV = 11;
static_assert((1 << SP_LEV) >= SP_B, "jump levels cover a block");
constexpr unsigned short SP_INV = 0xffffu;

__global__ __launch_bounds__(256) void wc_jump0_kernel(WcArgs a, WcSpec sp) {
    if (a.meta[1] != 0) return;
    const long long qb = sp.ctl[0];
    if (qb >= a.nq) return;
    const int lim = (int)(a.nq - qb < SP_B ? a.nq - qb : SP_B);
    const int i = blockIdx.x;
    unsigned short *J0 = sp.jump + (size_t)i * SP_W;
    if (i >= lim) return;
    const bool last = i + 1 >= lim;                      // the step of the block's last query leaves the block: not a jump
    const int delta = last ? 0 : a.nsel + (int)((sp.klo[i] - sp.klo[i + 1]) >> 1);
    for (int d = threadIdx.x; d < SP_W; d += 256) {
        const unsigned r = sp.rtab[(size_t)i * SP_W + d];
        const int nd = d + (int)r + delta;
        J0[d] = (last || r == 255u || nd < 0 || nd >= SP_W) ? SP_INV : (unsigned short)nd;
    }
}

// level k from level k - 1: 2^k steps = 2^(k-1) steps twice
__global__ __launch_bounds__(256) void wc_jumpk_kernel(WcArgs a, WcSpec sp, int k) {
    if (a.meta[1] != 0) return;
    const long long qb = sp.ctl[0];
    if (qb >= a.nq) return;
    const int lim = (int)(a.nq - qb < SP_B ? a.nq - qb : SP_B);
    const int i = blockIdx.x, half = 1 << (k - 1);
    if (i >= lim) return;
    const unsigned short *Jp = sp.jump + (size_t)(k - 1) * SP_B * SP_W;
    unsigned short *Jk = sp.jump + (size_t)k * SP_B * SP_W + (size_t)i * SP_W;
    const bool reach = i + 2 * half <= lim - 1;          // lands on a query of the block
    for (int d = threadIdx.x; d < SP_W; d += 256) {
        unsigned short v = SP_INV;
        if (reach) {
            const unsigned short m = Jp[(size_t)i * SP_W + d];
            if (m != SP_INV) v = Jp[(size_t)(i + half) * SP_W + m];
        }
        Jk[d] = v;
    }
}

__global__ void wc_ctl_init_kernel(long long *ctl, const long long *meta) {
    ctl[0] = 0;
    ctl[1] = meta[0];
}

__global__ __launch_bounds__(256) void wc_spec_kernel(WcArgs a, WcSpec sp) {
    __shared__ double xs[SP_NX];
    __shared__ int bins[SP_NB];
    __shared__ __attribute__((aligned(16))) uint32_t hkey[SP_HASH];      // later: diff[SP_W + 1] and nd[SP_W + SP_LOOK]
    __shared__ int2 dl[SP_DMAX];
    __shared__ int s_ndup, wsum[4];
    __shared__ float redf[4];
    if (a.meta[1] != 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long qb = sp.ctl[0], sb = sp.ctl[1];
    const int i = blockIdx.x;
    const long long q = qb + i;
    if (q >= a.nq) return;
    // predicted start: the block start (exact) + the expected redraws of the queries before this one
    float part = 0.0f;
    for (int j = tid; j < i; j += 256) part += sp.mu[qb + j];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) redf[wave] = part;
    if (tid == 0) s_ndup = 0;
    for (int h = tid; h < SP_HASH; h += 256) hkey[h] = 0xffffffffu;
    __syncthreads();
    long long dpre = (long long)((redf[0] + redf[1]) + (redf[2] + redf[3]) + 0.5f) - SP_W / 2;
    dpre = dpre < 0 ? 0 : dpre;
    const long long klo = sb + 2 * ((long long)i * a.nsel + dpre);
    if (tid == 0) sp.klo[i] = klo;
    const int nsel = a.nsel, nb = SP_W + nsel, nx = nb + 2 * SP_LOOK;
    const double *Sq = a.S + (size_t)q * a.n;
    const WcRec *Rq = a.R + (size_t)q * a.K;
    const double Stot = a.stot[q];
    // ---- doubles of the window and their bins (first-round look-up, as wc_window_sync)
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
                const WcRec rec = Rq[bk];
                bin = wc_quick_bin(rec.c0, rec.i, rec.more, x);
                double2 ss;
                if (bin < 0) {
                    const WcLoc L = wc_finish(Sq, a.n, rec.i + 1, 0, a.n, 0.0, Stot, x);
                    bin = L.bin;
                    ss = make_double2(L.s, L.sprev);
                } else {
                    ss = make_double2(Sq[bin], bin > 0 ? Sq[bin - 1] : 0.0);
                }
                sp.win_s[(size_t)i * SP_NB + e] = ss;
            }
        }
        xs[e] = x;
        if (e < nb) {
            bins[e] = bin;
            sp.win_bin[(size_t)i * SP_NB + e] = bin;
        }
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
            if (k < SP_DMAX) dl[k] = make_int2(tid + 256 * j, bins[tid + 256 * j]);
        }
    }
    const bool any_over = __syncthreads_or(overflow ? 1 : 0) != 0;
    const bool undecidable = any_over || s_ndup > SP_DMAX;
    const int ndup = s_ndup < SP_DMAX ? s_ndup : SP_DMAX;
    int *diff = (int *)hkey;                                       // [SP_W + 1]
    unsigned char *nd = (unsigned char *)(diff + SP_W + 4);        // [SP_W + SP_LOOK]
    for (int d = tid; d <= SP_W; d += 256) diff[d] = 0;
    __syncthreads();
    // ---- m2(d): draw e with an earlier same-bin draw prev counts for the starts d in (e - nsel, prev]
    for (int t = tid; t < ndup; t += 256) {
        const int e = dl[t].x, b = dl[t].y;
        int prev = -1;
        for (int u = 0; u < ndup; ++u) {
            const int2 o = dl[u];
            if (o.y == b && o.x < e && o.x > prev) prev = o.x;
        }
        if (prev >= 0) {
            const int lo = e - nsel + 1 > 0 ? e - nsel + 1 : 0;
            const int hi = prev < SP_W - 1 ? prev : SP_W - 1;
            if (lo <= hi) {
                atomicAdd(&diff[lo], 1);
                atomicAdd(&diff[hi + 1], -1);
            }
        }
    }
    // ---- nd[e - nsel]: distance to the first later draw within reach of the widest bin of the modified cdf
    const double pm = (double)a.pmax[2 * q];
    const double denom = Stot - (double)nsel * pm;
    const bool dist_ok = denom > 0.25 * Stot;
    const double wmax = dist_ok ? (pm / denom) * (1.0 + 1e-9) : 2.0;
    for (int r = tid; r < SP_W + SP_LOOK; r += 256) {
        const int e = nsel + r;
        const double x = xs[e];
        int best = 255;
        for (int j = 1; j < SP_LOOK; ++j) {
            if (fabs(x - xs[e + j]) <= wmax) {
                best = j;
                break;
            }
        }
        nd[r] = (unsigned char)best;
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
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int d = d0 + j;
        const int m2 = run + c[j];
        unsigned r = 255u;
        const long long s = klo + 2LL * d;
        if (!undecidable && s + 2LL * (nsel + m2) <= a.cap_words) {
            if (m2 == 0) {
                r = 0u;
            } else if (m2 <= SP_LOOK && dist_ok) {
                bool bad = false;
                for (int e = 0; e < m2; ++e) {
                    const int reach = nd[d + e];
                    bad |= (reach != 255) && (e + reach < m2);
                }
                if (!bad) r = (unsigned)m2;
            }
        }
        packed |= r << (8 * j);
    }
    ((uint32_t *)(sp.rtab + (size_t)i * SP_W))[tid] = packed;
}

__global__ __launch_bounds__(256) void wc_chain_kernel(WcArgs a, WcSpec sp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wc_lds[];
    __shared__ int wsum[16];
    __shared__ double wsumd[16];
    __shared__ long long s_klo[SP_B];
    __shared__ long long s_ev[3];
    if (a.meta[1] != 0) return;
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x;
    const long long qb = sp.ctl[0];
    if (qb >= a.nq) return;
    const WcLds l = wc_carve(wc_lds, a.n);
    const int BW = (a.n + 31) >> 5;
    for (int i = tid; i < BW; i += 256) l.bitmap[i] = 0;
    const int lim = (int)(a.nq - qb < SP_B ? a.nq - qb : SP_B);
    for (int i = tid; i < lim; i += 256) s_klo[i] = sp.klo[i];
    long long s = sp.ctl[1];
    int i = 0;
    long long t_fb = 0, n_fb = 0;
    const long long t_start = a.stats ? wall_clock64() : 0;
    // jump mode: s_mark[j] = window coordinate of query j where the walk KNEW it (block start, after a query it resolved
    // itself); the queries in between are filled in at the end
    __shared__ unsigned short s_mark[SP_B];
    const bool jumping = sp.jump != nullptr;
    if (jumping)
        for (int j = tid; j < SP_B; j += 256) s_mark[j] = SP_INV;
    __syncthreads();
    for (;;) {
        if (tid == 0) {
            int ev = 2;                               // 1 undecided candidate, 2 end of block / outside the window, 3 words exhausted
            while (i < lim) {
                if (s + 2LL * a.nsel > a.cap_words) {
                    ev = 3;
                    break;
                }
                long long d = (s - s_klo[i]) >> 1;
                if (d < 0 || d >= SP_W) break;
                if (jumping) {
                    // as far as whole jumps go (validity of a jump = validity of every step in it, so the greedy
                    // descent through the levels ends on the last query before an undecided / outside step)
                    s_mark[i] = (unsigned short)d;
                    for (int k = SP_LEV - 1; k >= 0; --k) {
                        if (i + (1 << k) > lim - 1) continue;
                        const unsigned short v = sp.jump[((size_t)k * SP_B + i) * SP_W + d];
                        if (v != SP_INV) {
                            i += 1 << k;
                            d = v;
                        }
                    }
                    s = s_klo[i] + 2 * d;
                    if (s + 2LL * a.nsel > a.cap_words) {
                        ev = 3;
                        break;
                    }
                }
                const unsigned r = sp.rtab[(size_t)i * SP_W + d];
                if (r == 255u) {
                    ev = 1;
                    break;
                }
                a.base[qb + i] = s;
                s += 2LL * (a.nsel + (int)r);
                ++i;
            }
            s_ev[0] = ev;
            s_ev[1] = i;
            s_ev[2] = s;
        }
        __syncthreads();
        const int ev = (int)s_ev[0];
        i = (int)s_ev[1];
        s = s_ev[2];
        __syncthreads();
        if (ev == 3) {
            if (tid == 0) {
                a.meta[1] = 2;
                a.meta[0] = s;
            }
            return;
        }
        if (ev != 1) break;
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
        {
            const long long d = (s - s_klo[i]) >> 1;          // inside the window (checked by the walk above)
            qa.pre_bin = sp.win_bin + ((size_t)i * SP_NB + d);
            qa.pre_s = sp.win_s + ((size_t)i * SP_NB + d);
        }
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
    if (jumping) {
        // word offsets of the queries the walk jumped over: nearest known start at or below j (prefix maximum over the
        // marks), then j - start steps by binary lifting.  Queries >= i were not reached.
        __shared__ short s_from[SP_B];
        __syncthreads();
        for (int j0 = tid * (SP_B / 256); j0 < (tid + 1) * (SP_B / 256); ++j0) s_from[j0] = s_mark[j0] != SP_INV ? (short)j0 : (short)-1;
        __syncthreads();
        {   // prefix maximum: 8 consecutive entries per lane, then across lanes
            __shared__ short s_lane[256];
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
                    dd = sp.jump[((size_t)k * SP_B + ii) * SP_W + dd];
                    ii += 1 << k;
                    m -= 1 << k;
                }
            }
            // (a start itself: m = 0 from the beginning.  The ids kernel re-derives every query's consumption and flags
            //  any disagreement, so a wrong offset cannot pass silently.)
            if (m == 0 && dd != SP_INV) a.base[qb + j] = s_klo[j] + 2LL * dd;
            else if (j < i) a.meta[1] = 4;
        }
    }
    if (tid == 0) {
        if (a.stats) {
            atomicAdd((unsigned long long *)&a.stats[12], (unsigned long long)n_fb);
            atomicAdd((unsigned long long *)&a.stats[13], (unsigned long long)t_fb);
            atomicAdd((unsigned long long *)&a.stats[14], (unsigned long long)(wall_clock64() - t_start));
        }
        sp.ctl[0] = qb + i;
        sp.ctl[1] = s;
        if (qb + i >= a.nq) a.meta[0] = s;
        if (a.stats) {
            atomicAdd((unsigned long long *)&a.stats[11], 1ull);            // (spec, chain) pairs that did work
        }
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
    qa.pre_bin = nullptr;
    qa.pre_s = nullptr;
    const long long used = wc_full_query<true>(qa, l, wsum, wsumd, a.ids_out + (size_t)q * a.nsel);
    if (a.fixed) {
        // rng.seed(42) before every query: the generator ends where the LAST query of the call left it
        if (tid == 0 && used < 0) a.meta[1] = 3;
        if (tid == 0 && q == a.nq - 1 && used >= 0) a.meta[0] = used;
        return;
    }
    // cross-check against the offsets kernel: both must agree on where the next query starts
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

int wc_reserve(p2s_rng_s *r, size_t nq, size_t n, size_t K) {
    if (nq <= r->wc_cap_q && n <= r->wc_cap_n && K <= r->wc_cap_k) return P2S_OK;
    if (r->wc_S) (void)hipFree(r->wc_S);
    if (r->wc_T) (void)hipFree(r->wc_T);
    if (r->wc_stot) (void)hipFree(r->wc_stot);
    if (r->wc_J) (void)hipFree(r->wc_J);
    r->wc_S = nullptr; r->wc_T = nullptr; r->wc_stot = nullptr; r->wc_J = nullptr;
    r->wc_cap_q = r->wc_cap_n = r->wc_cap_k = 0;
    nq = std::max(nq, r->wc_cap_q);
    if (hipMalloc(&r->wc_S, nq * n * 8) != hipSuccess ||
        hipMalloc(&r->wc_T, nq * K * sizeof(WcRec)) != hipSuccess ||
        hipMalloc(&r->wc_J, (size_t)SP_LEV * SP_B * SP_W * 2) != hipSuccess ||
        hipMalloc(&r->wc_stot, nq * 32 + (size_t)SP_B * SP_W + (size_t)SP_B * 8 + 64 + (size_t)SP_B * SP_NB * 20 + 64) != hipSuccess) {
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
    if (r->wc_S) (void)hipFree(r->wc_S);
    if (r->wc_T) (void)hipFree(r->wc_T);
    if (r->wc_stot) (void)hipFree(r->wc_stot);
    if (r->wc_J) (void)hipFree(r->wc_J);
}

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
    int K = 1024;
    while (K < n) K <<= 1;
    // queries per batch (table memory: ~1.4 MB per query at 50k points); random words come from the raw session
    long long per_req = 4096;
    const long long req_env = getenv("P2S_WCHOICE_QUERIES") ? atoll(getenv("P2S_WCHOICE_QUERIES")) : 0;   // tests
    if (req_env > 0) per_req = std::min<long long>(req_env, 16384);
    per_req = std::min<long long>(per_req, nq);
    const long long cap = p2s_rng_session_words(r);
    // numpy draws 2 words per double, n_sel doubles + a few % redraws per query; the serial kernel stages words
    // a few queries ahead of the one it works on
    const long long margin = 8LL * n_sel + 6 * 128 + 8192;
    if ((long long)(2.0 * n_sel * 1.15) + margin > cap) {
        p2s_set_error("p2s_subsample_weighted: jump tables too small for one query");
        return P2S_EINVAL;
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
    // per-query scalars share one allocation: [stot f64][base i64][pmax f32, cells i32][mu f32, pad], followed by the
    // speculation block: [ctl i64 x 8][klo i64 x SP_B][rtab u8 x SP_B x SP_W]
    double *stot = r->wc_stot;
    long long *base = (long long *)(stot + r->wc_cap_q);
    float *pmax = (float *)(base + r->wc_cap_q);
    float *mu = pmax + 2 * r->wc_cap_q;
    WcSpec sp;
    sp.ctl = (long long *)(mu + 2 * r->wc_cap_q);
    sp.klo = sp.ctl + 8;
    sp.rtab = (unsigned char *)(sp.klo + SP_B);
    sp.mu = mu;
    sp.win_s = (double2 *)(((uintptr_t)(sp.rtab + (size_t)SP_B * SP_W) + 15) & ~(uintptr_t)15);
    sp.win_bin = (int *)(sp.win_s + (size_t)SP_B * SP_NB);
    sp.jump = getenv("P2S_WC_NO_JUMP") ? nullptr : r->wc_J;                 // development / A-B: walk query by query
    const bool serial_only = getenv("P2S_WC_SERIAL") != nullptr;            // development / A-B: the serial kernel alone
    const size_t lds_ids = wc_lds_bytes(n);
    size_t lds_off = wc_offsets_lds_bytes(n);
    // the offsets kernel is one latency-bound workgroup running next to the MFMA-saturated encoders: give it a CU of
    // its own by claiming most of that CU's LDS (same placement trick as the serial generator)
    const size_t hog = getenv("P2S_RNG_LDS_HOG") ? (size_t)atoi(getenv("P2S_RNG_LDS_HOG")) : 120 * 1024;
    // clouds beyond 185,664 points: the serial kernel's ring + windows leave no room for the found-bitmap; the plain
    // kernel (bitmap + the arrays of one query, like the ids kernel) takes its place
    const bool big = lds_off > 160 * 1024 - 4096;
    const size_t lds_chain_max = 160 * 1024 - 4096 - SP_B * 8 - SP_B * 4 - 1024;      // the chain kernel's static arrays
    if (lds_ids > lds_chain_max) {
        p2s_set_error("p2s_subsample_weighted: cloud of %d points does not fit the LDS bitmap (limit: %d points)", n,
                      (int)((lds_chain_max - wc_lds_bytes(0) - 16) / 6 * 32));
        return P2S_ECAPACITY;
    }
    if (nq >= 64 && !big) lds_off = std::max(lds_off, hog);
    // the chain workgroup of a full block claims a CU of its own (LDS no encoder workgroup fits next to): sharing a
    // CU with the encoders' MFMA-saturated waves it gets an instruction issued every ~200 cycles (measured: 1.4 us
    // per query of the walk whether the table sits in global memory, LDS or registers; 80 us per fallback instead of
    // 14.5) -- with blocks of 2048 queries the wait for a drained CU is paid twice per chunk
    const size_t lds_chain = nq >= 64 ? std::max(lds_ids, getenv("P2S_WC_CHAIN_LDS") ? (size_t)atoi(getenv("P2S_WC_CHAIN_LDS")) : (size_t)100000) : lds_ids;
    {   // per device (a process may drive several); the call is cheap
        (void)hipFuncSetAttribute((const void *)wc_offsets_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        (void)hipFuncSetAttribute((const void *)wc_ids_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void *)wc_offsets_plain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void *)wc_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_chain_max);
        (void)hipFuncSetAttribute((const void *)wc_tables_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (NP_BUFSIZE + WC_MAX_NODES) * 4);
    }
    long long *meta = p2s_rng_raw_meta(r);
    static long long *stats_dev = nullptr;
    const bool want_stats = getenv("P2S_WC_STATS") != nullptr;
    if (want_stats && !stats_dev) (void)hipMalloc(&stats_dev, 16 * 8);
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
                           q_dev + (size_t)done * 3, plan, K, r->wc_S, (WcRec *)r->wc_T, stot, pmax, mu, n_sel, meta);
        WcArgs a;
        a.S = r->wc_S;
        a.R = (const WcRec *)r->wc_T;
        a.stot = stot;
        a.pmax = pmax;
        a.words = r->tmp;
        a.cap_words = cap;
        a.alloc_words = cap + 624;
        a.n = n;
        a.K = K;
        a.nq = cur;
        a.nsel = n_sel;
        a.base = base;
        a.ids_out = ids_out_dev ? ids_out_dev + (size_t)done * n_sel : nullptr;
        a.meta = meta;
        a.stats = nullptr;
        a.ctl = nullptr;
        a.fixed = fixed ? 1 : 0;
        if (want_stats) {
            (void)hipMemsetAsync(stats_dev, 0, 16 * 8, s);
            a.stats = stats_dev;
        }
        if (fixed) {
            // no serial dependence between the queries: the ids kernel alone (it reports the last query's consumption)
        } else if (serial_only) {
            if (big)
                hipLaunchKernelGGL(wc_offsets_plain_kernel, dim3(1), dim3(256), lds_ids, s, a);
            else
                hipLaunchKernelGGL(wc_offsets_kernel, dim3(1), dim3(WC_NT), lds_off, s, a);
        } else {
            // speculation tables on all CUs + a light chain per block of SP_B queries; two spare pairs for blocks that
            // end early (start outside the window); whatever is still unresolved then goes through the serial kernel
            hipLaunchKernelGGL(wc_ctl_init_kernel, dim3(1), dim3(1), 0, s, sp.ctl, meta);
            const int pairs = (cur + SP_B - 1) / SP_B + (cur > SP_B / 2 ? 2 : 0);
            for (int pr = 0; pr < pairs; ++pr) {
                hipLaunchKernelGGL(wc_spec_kernel, dim3(std::min(cur, SP_B)), dim3(256), 0, s, a, sp);
                if (sp.jump) {
                    hipLaunchKernelGGL(wc_jump0_kernel, dim3(std::min(cur, SP_B)), dim3(256), 0, s, a, sp);
                    for (int k = 1; k < SP_LEV && (1 << k) < std::min(cur, SP_B); ++k)
                        hipLaunchKernelGGL(wc_jumpk_kernel, dim3(std::min(cur, SP_B)), dim3(256), 0, s, a, sp, k);
                }
                hipLaunchKernelGGL(wc_chain_kernel, dim3(1), dim3(256), pr < (cur + SP_B - 1) / SP_B ? lds_chain : lds_ids, s, a, sp);
            }
            a.ctl = sp.ctl;
            if (big)
                hipLaunchKernelGGL(wc_offsets_plain_kernel, dim3(1), dim3(256), lds_ids, s, a);
            else
                hipLaunchKernelGGL(wc_offsets_kernel, dim3(1), dim3(WC_NT), wc_offsets_lds_bytes(n), s, a);
            a.ctl = nullptr;
        }
        if (ids_out_dev) hipLaunchKernelGGL(wc_ids_kernel, dim3(cur), dim3(256), lds_ids, s, a);   // NULL: advance the stream only
        P2S_LAUNCH_CHECK("weighted sub-sample kernels");
        if (want_stats) {
            long long h[16];
            (void)hipMemcpyAsync(h, stats_dev, sizeof(h), hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            fprintf(stderr, "[wc stats] %d queries: %lld full-algorithm fallbacks, %lld window misses; 10-ns ticks: "
                            "A %lld issue %lld mark %lld sepcheck %lld clear+fallback %lld B %lld ring %lld; kernel %lld ticks = "
                            "%lld shader clocks; chain passes that did work %lld, in-place fallbacks %lld taking %lld of %lld 10-ns ticks\n", cur, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11], h[12], h[13], h[14]);
        }
        done += cur;
    }
    if (pts_out_dev) return p2s_gather_points(c, ids_out_dev, nq * n_sel, pts_out_dev, stream);
    return P2S_OK;
}
