// Weighted sub-sample, stage 1 (included by p2s_wchoice.hip inside its anonymous namespace): the per-query tables --
// distances, numpy's float32 sum in numpy's association, exact float64 prefix sums S, the guide R.
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
