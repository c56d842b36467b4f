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

#include "p2s_wc_tables.inl"
#include "p2s_wc_choice.inl"
#include "p2s_wc_offsets.inl"
#include "p2s_wc_ids.inl"

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
