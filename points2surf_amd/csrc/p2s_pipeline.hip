// p2s_infer_shape: the batch loop of the reference's points_to_surf_eval for ONE shape in
// reconstruction mode (source/points_to_surf_eval.py:358-404), entirely on the device:
//   query grid -> per chunk { kNN patch + radius, MT19937 sub-sample + gather, encoders + decoder } -> sdf
//
// Two HIP streams: the data path of chunks i+1, i+2 (latency-bound select/gather work that needs a
// handful of CUs; the MT19937 recurrence is serial) runs on an auxiliary stream while the MFMA-bound
// encoders of chunk i own the rest of the chip.  Double-buffered; events order buffer reuse.
#include "p2s_common.h"
#include "p2s_internal.h"
#include <algorithm>
#include <cstring>
#include <cstdlib>

namespace {

struct PipeBuffers {
    float *q = nullptr;             // [Q][3]
    float *patch[2] = {};           // [C][k][3]
    float *radius[2] = {};          // [C]
    int32_t *sub_ids[2] = {};       // [C][n]
    float *sub[2] = {};             // [C][n][3]
    hipEvent_t ready[2] = {};       // data path of the buffer finished (aux stream)
    hipEvent_t freed[2] = {};       // encoders finished reading the buffer (main stream)
    hipEvent_t grid = nullptr;
};

void free_pipe(PipeBuffers &b) {
    if (b.q) (void)hipFree(b.q);
    for (int i = 0; i < 2; ++i) {
        if (b.patch[i]) (void)hipFree(b.patch[i]);
        if (b.radius[i]) (void)hipFree(b.radius[i]);
        if (b.sub_ids[i]) (void)hipFree(b.sub_ids[i]);
        if (b.sub[i]) (void)hipFree(b.sub[i]);
        if (b.ready[i]) (void)hipEventDestroy(b.ready[i]);
        if (b.freed[i]) (void)hipEventDestroy(b.freed[i]);
    }
    if (b.grid) (void)hipEventDestroy(b.grid);
    b = PipeBuffers();
}

}  // namespace

extern "C" int p2s_infer_shape(p2s_model_t m, p2s_cloud_t c, p2s_rng_t r, int res, int eps, int64_t q_begin,
                               int64_t q_end, int chunk, float *sdf_out_dev, float *q_out_dev, int64_t *n_done,
                               void *stream) {
    if (!m || !c || !r || !sdf_out_dev) {
        p2s_set_error("p2s_infer_shape: null argument");
        return P2S_EINVAL;
    }
    const bool weighted = m->cfg.weighted_subsample != 0;   // p2s_vanilla: choice(p, replace=False) per query
    P2S_HIP_CHECK(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    const int k = m->cfg.points_per_patch, n = m->cfg.sub_sample_size;
    if (chunk <= 0) chunk = m->max_chunk;
    chunk = std::min(chunk, m->max_chunk);
    if (getenv("P2S_NO_OVERLAP")) m->overlap = false;   // development knob: single-stream pipeline
    if (m->overlap && !m->aux) {
        // high queue priority: the data-path kernels are tiny next to the encoder kernel and must not queue
        // behind its ~8k workgroups for a free CU slot.  (Giving the stream its own CUs with a CU mask was
        // measured slower: the masked compute stream lost far more than the masked-off CUs.)
        int lo = 0, hi = 0;
        P2S_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        P2S_HIP_CHECK(hipStreamCreateWithPriority(&m->aux, hipStreamNonBlocking, hi));
    }
    hipStream_t sa = m->overlap ? m->aux : s;

    p2s_prof_reset(m);

    int64_t Q = 0;
    const int eg0 = p2s_prof_mark(m, s);
    int rc = p2s_query_grid(c, res, eps, nullptr, 0, &Q, stream);
    if (rc != P2S_OK && rc != P2S_ECAPACITY) return rc;
    if (q_end < 0) q_end = Q;
    if (q_begin < 0 || q_begin > q_end || q_end > Q) {
        p2s_set_error("p2s_infer_shape: query range [%lld,%lld) outside [0,%lld]", (long long)q_begin,
                      (long long)q_end, (long long)Q);
        return P2S_EINVAL;
    }
    PipeBuffers b;
    auto fail = [&](int code) {
        (void)hipStreamSynchronize(s);
        if (sa != s) (void)hipStreamSynchronize(sa);
        free_pipe(b);
        return code;
    };
    const int64_t nq = q_end - q_begin;
    if (n_done) *n_done = 0;
    if (Q == 0 || nq == 0) return fail(P2S_OK);
    const int C = (int)std::min<int64_t>(chunk, nq);
    const int nbuf = (nq > C) ? 2 : 1;
    bool ok = hipMalloc(&b.q, (size_t)Q * 12) == hipSuccess && hipEventCreateWithFlags(&b.grid, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < nbuf && ok; ++i) {
        ok = hipMalloc(&b.patch[i], (size_t)C * k * 12) == hipSuccess && hipMalloc(&b.radius[i], (size_t)C * 4) == hipSuccess &&
             hipMalloc(&b.sub_ids[i], (size_t)C * n * 4) == hipSuccess && hipMalloc(&b.sub[i], (size_t)C * n * 12) == hipSuccess &&
             hipEventCreateWithFlags(&b.ready[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&b.freed[i], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        p2s_set_error("p2s_infer_shape: allocation of pipeline buffers failed");
        (void)hipGetLastError();
        return fail(P2S_ENOMEM);
    }
    rc = p2s_query_grid(c, res, eps, b.q, Q, &Q, stream);
    if (rc) return fail(rc);
    p2s_prof_span(m, ST_GRID, eg0, p2s_prof_mark(m, s));
    rc = p2s_model_reserve(m, C);
    if (rc) return fail(rc);
    if (sa != s) {
        P2S_HIP_CHECK(hipEventRecord(b.grid, s));
        P2S_HIP_CHECK(hipStreamWaitEvent(sa, b.grid, 0));
    }

    const int64_t nchunks = (nq + C - 1) / C;
    auto produce = [&](int64_t ci) -> int {       // sub-sample ids of chunk ci on the aux stream
        const int bi = (int)(ci % nbuf);
        const int64_t q0 = q_begin + ci * C;
        const int cur = (int)std::min<int64_t>(C, q_end - q0);
        if (ci >= nbuf && sa != s) P2S_HIP_CHECK(hipStreamWaitEvent(sa, b.freed[bi], 0));
        const int e0 = p2s_prof_mark(m, sa);
        const int rc2 = weighted ? p2s_subsample_weighted(r, c, b.q + (size_t)q0 * 3, cur, n, b.sub_ids[bi], nullptr, sa)
                                 : p2s_subsample_uniform(r, c, cur, n, b.sub_ids[bi], nullptr, sa);
        if (rc2) return rc2;
        p2s_prof_span(m, ST_SUB, e0, p2s_prof_mark(m, sa));
        if (sa != s) P2S_HIP_CHECK(hipEventRecord(b.ready[bi], sa));
        return P2S_OK;
    };

    // prologue: up to nbuf chunks of ids in flight
    for (int64_t ci = 0; ci < std::min<int64_t>(nbuf, nchunks); ++ci)
        if ((rc = produce(ci))) return fail(rc);
    for (int64_t ci = 0; ci < nchunks; ++ci) {
        const int bi = (int)(ci % nbuf);
        const int64_t q0 = q_begin + ci * C;
        const int cur = (int)std::min<int64_t>(C, q_end - q0);
        const float *qc = b.q + (size_t)q0 * 3;
        const int ek0 = p2s_prof_mark(m, s);
        rc = p2s_knn_patch(c, qc, cur, k, nullptr, b.patch[bi], b.radius[bi], s);
        if (rc) return fail(rc);
        p2s_prof_span(m, ST_KNN, ek0, p2s_prof_mark(m, s));
        if (sa != s) P2S_HIP_CHECK(hipStreamWaitEvent(s, b.ready[bi], 0));
        rc = p2s_gather_points(c, b.sub_ids[bi], (int64_t)cur * n, b.sub[bi], s);
        if (rc) return fail(rc);
        if (sa != s) P2S_HIP_CHECK(hipEventRecord(b.freed[bi], s));
        rc = p2s_run_chunk(m, b.patch[bi], b.sub[bi], qc, b.radius[bi], cur, nullptr, sdf_out_dev + (q0 - q_begin),
                           nullptr, nullptr, s);
        if (rc) return fail(rc);
        if (ci + nbuf < nchunks)
            if ((rc = produce(ci + nbuf))) return fail(rc);
    }
    if (q_out_dev)
        P2S_HIP_CHECK(hipMemcpyAsync(q_out_dev, b.q + (size_t)q_begin * 3, (size_t)nq * 12, hipMemcpyDeviceToDevice, s));
    m->counters.queries += nq;
    // buffers are freed below: both streams must be done with them
    P2S_HIP_CHECK(hipStreamSynchronize(s));
    if (sa != s) P2S_HIP_CHECK(hipStreamSynchronize(sa));
    p2s_prof_collect(m);
    rc = p2s_rng_check(r, s);
    if (rc) return fail(rc);
    if (n_done) *n_done = nq;
    return fail(P2S_OK);
}
