// p2s_infer_shape: the batch loop of the reference's points_to_surf_eval for ONE shape in
// reconstruction mode (source/points_to_surf_eval.py:358-404), entirely on the device:
//   query grid -> per chunk { kNN patch + radius, MT19937 sub-sample + gather, encoders + decoder } -> sdf
#include "p2s_common.h"
#include "p2s_internal.h"
#include <algorithm>
#include <cstring>

namespace {

struct PipeBuffers {
    float *q = nullptr;        // [Q][3]
    float *patch = nullptr;    // [C][k][3]
    float *radius = nullptr;   // [C]
    int32_t *sub_ids = nullptr;// [C][n]
    float *sub = nullptr;      // [C][n][3]
};

void free_pipe(PipeBuffers &b) {
    if (b.q) (void)hipFree(b.q);
    if (b.patch) (void)hipFree(b.patch);
    if (b.radius) (void)hipFree(b.radius);
    if (b.sub_ids) (void)hipFree(b.sub_ids);
    if (b.sub) (void)hipFree(b.sub);
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
    if (m->cfg.use_point_stn) {
        // p2s_vanilla draws the sub-sample with legacy choice(p, replace=False): ids come from the host
        p2s_set_error("p2s_infer_shape: model needs the distance-weighted sub-sample (host ids); "
                      "use p2s_knn_patch + p2s_gather_points + p2s_encode_decode");
        return P2S_EINVAL;
    }
    P2S_HIP_CHECK(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    const int k = m->cfg.points_per_patch, n = m->cfg.sub_sample_size;
    if (chunk <= 0) chunk = m->max_chunk;
    chunk = std::min(chunk, m->max_chunk);

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
        free_pipe(b);
        return code;
    };
    const int64_t nq = q_end - q_begin;
    if (n_done) *n_done = 0;
    if (Q == 0 || nq == 0) return fail(P2S_OK);
    const int C = (int)std::min<int64_t>(chunk, nq);
    if (hipMalloc(&b.q, (size_t)Q * 12) != hipSuccess || hipMalloc(&b.patch, (size_t)C * k * 12) != hipSuccess ||
        hipMalloc(&b.radius, (size_t)C * 4) != hipSuccess || hipMalloc(&b.sub_ids, (size_t)C * n * 4) != hipSuccess ||
        hipMalloc(&b.sub, (size_t)C * n * 12) != hipSuccess) {
        p2s_set_error("p2s_infer_shape: hipMalloc of pipeline buffers failed");
        return fail(P2S_ENOMEM);
    }
    rc = p2s_query_grid(c, res, eps, b.q, Q, &Q, stream);
    if (rc) return fail(rc);
    p2s_prof_span(m, ST_GRID, eg0, p2s_prof_mark(m, s));
    rc = p2s_model_reserve(m, C);
    if (rc) return fail(rc);

    for (int64_t q0 = q_begin; q0 < q_end; q0 += C) {
        const int cur = (int)std::min<int64_t>(C, q_end - q0);
        const float *qc = b.q + (size_t)q0 * 3;
        const int ek0 = p2s_prof_mark(m, s);
        rc = p2s_knn_patch(c, qc, cur, k, nullptr, b.patch, b.radius, stream);
        if (rc) return fail(rc);
        const int ek1 = p2s_prof_mark(m, s);
        rc = p2s_subsample_uniform(r, c, cur, n, b.sub_ids, b.sub, stream);
        if (rc) return fail(rc);
        const int es1 = p2s_prof_mark(m, s);
        p2s_prof_span(m, ST_KNN, ek0, ek1);
        p2s_prof_span(m, ST_SUB, ek1, es1);
        rc = p2s_run_chunk(m, b.patch, b.sub, qc, b.radius, cur, nullptr, sdf_out_dev + (q0 - q_begin), nullptr,
                           nullptr, s);
        if (rc) return fail(rc);
    }
    if (q_out_dev)
        P2S_HIP_CHECK(hipMemcpyAsync(q_out_dev, b.q + (size_t)q_begin * 3, (size_t)nq * 12, hipMemcpyDeviceToDevice, s));
    m->counters.queries += nq;
    // buffers are freed below: the stream must be done with them
    P2S_HIP_CHECK(hipStreamSynchronize(s));
    p2s_prof_collect(m);
    if (n_done) *n_done = nq;
    return fail(P2S_OK);
}
