// The batch loop of the reference's points_to_surf_eval for ONE shape (source/points_to_surf_eval.py:358-404),
// entirely on the device:
//   p2s_infer_shape    reconstruction pass: query grid -> per chunk { kNN patch + radius, sub-sample + gather,
//                      encoders + decoder } -> sdf
//   p2s_infer_queries  GT-query evaluation pass (full_eval.py:31-33): the same for caller-provided query points, with
//                      the per-query random rotation of source/data_loader.py:381-393 (second RandomState ->
//                      rand(3) -> rotation matrix -> float64 transform of patch, sub-sample and query point)
//
// Two HIP streams: the data path of chunks i+1, i+2 (latency-bound select/gather work that needs a
// handful of CUs; the MT19937 recurrence is serial) runs on an auxiliary stream while the MFMA-bound
// encoders of chunk i own the rest of the chip.  Double-buffered; events order buffer reuse.  The buffers live on
// the model handle (grown on demand, reused across shapes, released by p2s_model_destroy): the per-shape path
// allocates nothing, and every error path leaves through fail(), which drains both streams first.
#include "p2s_common.h"
#include "p2s_internal.h"
#include <algorithm>
#include <cstring>
#include <cstdlib>

void p2s_pipe_free(p2s_model_s *m) {
    PipeBuffers &b = m->pipe;
    for (int i = 0; i < 2; ++i) {
        if (b.patch[i]) (void)hipFree(b.patch[i]);
        if (b.radius[i]) (void)hipFree(b.radius[i]);
        if (b.sub_ids[i]) (void)hipFree(b.sub_ids[i]);
        if (b.sub[i]) (void)hipFree(b.sub[i]);
        if (b.qrot[i]) (void)hipFree(b.qrot[i]);
        if (b.knn_ids[i]) (void)hipFree(b.knn_ids[i]);
        if (b.perm[i]) (void)hipFree(b.perm[i]);
        if (b.rot[i]) (void)hipFree(b.rot[i]);
        if (b.ready[i]) (void)hipEventDestroy(b.ready[i]);
        if (b.freed[i]) (void)hipEventDestroy(b.freed[i]);
        if (b.done[i]) (void)hipEventDestroy(b.done[i]);
        if (b.ball_ready[i]) (void)hipEventDestroy(b.ball_ready[i]);
    }
    if (b.grid) (void)hipEventDestroy(b.grid);
    b = PipeBuffers();
}

namespace {

int pipe_reserve(p2s_model_s *m, int C, int k, int n, bool small) {
    PipeBuffers &b = m->pipe;
    if (b.cap_chunk >= C && b.cap_k == k && b.cap_n == n && (!small || b.cap_small)) return P2S_OK;
    C = std::max(C, b.cap_chunk);
    small = small || b.cap_small;
    P2S_HIP_CHECK(hipDeviceSynchronize());
    p2s_pipe_free(m);
    bool ok = hipEventCreateWithFlags(&b.grid, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 2 && ok; ++i) {
        ok = hipMalloc(&b.patch[i], (size_t)C * k * 12) == hipSuccess && hipMalloc(&b.radius[i], (size_t)C * 4) == hipSuccess &&
             hipMalloc(&b.sub_ids[i], (size_t)C * n * 4) == hipSuccess && hipMalloc(&b.sub[i], (size_t)C * n * 12) == hipSuccess &&
             hipMalloc(&b.qrot[i], (size_t)C * 12) == hipSuccess && hipMalloc(&b.rot[i], (size_t)C * 72) == hipSuccess &&
             (!small || (hipMalloc(&b.knn_ids[i], (size_t)C * k * 4) == hipSuccess &&
                         hipMalloc(&b.perm[i], (size_t)C * n * 4) == hipSuccess)) &&
             hipEventCreateWithFlags(&b.ready[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&b.freed[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&b.done[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&b.ball_ready[i], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        (void)hipGetLastError();
        p2s_pipe_free(m);
        p2s_set_error("p2s pipeline: allocation of the chunk buffers failed (chunk %d)", C);
        return P2S_ENOMEM;
    }
    b.cap_chunk = C;
    b.cap_k = k;
    b.cap_n = n;
    b.cap_small = small ? 1 : 0;
    return P2S_OK;
}

// ---------------------------------------------------------------------------------------------------------
// GT-query pass: random rotation per query (reference source/data_loader.py:381-393).
//   rand3 = self.rng.rand(3)                       numpy legacy: (a >> 5, b >> 6) -> (a * 2^26 + b) / 2^53 per double
//   M = trimesh.transformations.random_rotation_matrix(rand3)   = quaternion_matrix(random_quaternion(rand3))
// (trimesh = Gohlke's transformations.py; restated in oracle/trimesh_restated.py).  All float64, no contraction.
// ---------------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void p2s_rand_rot_kernel(const uint32_t *__restrict__ words, const long long *__restrict__ meta,
                                                           long long cap_words, long long n, double *__restrict__ rot,
                                                           long long *__restrict__ err) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long c0 = meta ? meta[0] : 0;   // meta == NULL: the words of query i lie at words[6 i]
    if (meta && c0 + 6 * n > cap_words) {
        if (i == 0) err[0] = 2;          // random words exhausted (host-side accounting makes this unreachable)
        return;
    }
    const uint32_t *w = words + c0 + 6 * i;
    double rnd[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double a = (double)(w[2 * j] >> 5), b = (double)(w[2 * j + 1] >> 6);
        rnd[j] = (a * 67108864.0 + b) / 9007199254740992.0;
    }
    const double r1 = sqrt(1.0 - rnd[0]), r2 = sqrt(rnd[0]);
    const double pi2 = 3.141592653589793 * 2.0;
    const double t1 = pi2 * rnd[1], t2 = pi2 * rnd[2];
    double q[4] = {cos(t2) * r2, sin(t1) * r1, cos(t1) * r1, sin(t2) * r2};
    const double nn = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3];
    double *R = rot + 9 * i;
    if (nn < 2.220446049250313e-16 * 4.0) {
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
        return;
    }
    const double sc = sqrt(2.0 / nn);
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] *= sc;
    const double q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
    const double q12 = q[1] * q[2], q13 = q[1] * q[3], q23 = q[2] * q[3];
    const double q10 = q[1] * q[0], q20 = q[2] * q[0], q30 = q[3] * q[0];
    R[0] = 1.0 - q22 - q33; R[1] = q12 - q30;       R[2] = q13 + q20;
    R[3] = q12 + q30;       R[4] = 1.0 - q11 - q33; R[5] = q23 - q10;
    R[6] = q13 - q20;       R[7] = q23 + q10;       R[8] = 1.0 - q11 - q22;
}

__global__ void p2s_advance_cursor_kernel(long long *meta, long long words) { meta[0] += words; }

// trimesh.transformations.transform_points(points, M).astype(np.float32): float64 homogeneous product, row c of
// the result = sum_k M[c][k] * [x y z 1][k]  (M[c][3] = 0).  One thread per point.
__global__ __launch_bounds__(256) void p2s_rotate_points_kernel(const double *__restrict__ rot, const float *__restrict__ in,
                                                                float *__restrict__ out, int P, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const double *R = rot + 9 * (i / P);
    const double x = in[3 * i + 0], y = in[3 * i + 1], z = in[3 * i + 2];
    // identity shortcut of transform_points (|M - I|.max() < 1e-8 returns the points unchanged)
    double dev = 0.0;
#pragma unroll
    for (int j = 0; j < 9; ++j) dev = fmax(dev, fabs(R[j] - ((j % 4 == 0) ? 1.0 : 0.0)));
    if (dev < 1e-8) {
        out[3 * i + 0] = (float)x; out[3 * i + 1] = (float)y; out[3 * i + 2] = (float)z;
        return;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double acc = R[3 * c + 0] * x;
        acc = __builtin_fma(R[3 * c + 1], y, acc);
        acc = __builtin_fma(R[3 * c + 2], z, acc);
        out[3 * i + c] = (float)acc;
    }
}

}  // namespace

int p2s_rotations_from_words(const uint32_t *six_dev, long long n, double *rot_out_dev, hipStream_t s) {
    if (n <= 0) return P2S_OK;
    hipLaunchKernelGGL(p2s_rand_rot_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, six_dev, (const long long *)nullptr,
                       6 * n, n, rot_out_dev, (long long *)nullptr);
    P2S_LAUNCH_CHECK("p2s_rand_rot_kernel");
    return P2S_OK;
}

extern "C" int p2s_random_rotations(p2s_rng_t r, int64_t n, double *rot_out_dev, void *stream) {
    if (!r || n < 0 || (n > 0 && !rot_out_dev)) {
        p2s_set_error("p2s_random_rotations: bad argument");
        return P2S_EINVAL;
    }
    if (r->levels_max == 0) {
        p2s_set_error("p2s_random_rotations: needs the jump-ahead tables (p2s_rng_set_jump_tables)");
        return P2S_EINVAL;
    }
    if (n == 0) return P2S_OK;
    P2S_HIP_CHECK(hipSetDevice(r->device));
    hipStream_t s = (hipStream_t)stream;
    const long long cap = p2s_rng_session_words(r);
    const long long per = std::max<long long>(1, std::min<long long>(n, (cap - 1024) / 6));
    for (int64_t done = 0; done < n;) {
        const long long cur = std::min<long long>(per, n - done);
        const int rc = p2s_rng_session_raw(r, 6 * cur, s);
        if (rc) return rc;
        long long *meta = p2s_rng_raw_meta(r);
        hipLaunchKernelGGL(p2s_rand_rot_kernel, dim3((unsigned)((cur + 255) / 256)), dim3(256), 0, s, r->tmp, meta, cap, cur,
                           rot_out_dev + (size_t)done * 9, meta + 1);
        hipLaunchKernelGGL(p2s_advance_cursor_kernel, dim3(1), dim3(1), 0, s, meta, 6 * cur);
        P2S_LAUNCH_CHECK("p2s_rand_rot_kernel");
        done += cur;
    }
    return P2S_OK;
}

extern "C" int p2s_rotate_points(const double *rot_dev, const float *pts_in_dev, int points_per_item, int64_t n_items,
                                 float *pts_out_dev, void *stream) {
    if (!rot_dev || !pts_in_dev || !pts_out_dev || points_per_item < 1 || n_items < 0) {
        p2s_set_error("p2s_rotate_points: bad argument");
        return P2S_EINVAL;
    }
    const long long total = (long long)n_items * points_per_item;
    if (total == 0) return P2S_OK;
    hipLaunchKernelGGL(p2s_rotate_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       rot_dev, pts_in_dev, pts_out_dev, points_per_item, total);
    P2S_LAUNCH_CHECK("p2s_rotate_points_kernel");
    return P2S_OK;
}

extern "C" int p2s_model_capture_logits(p2s_model_t m, float *logits_out_dev, int64_t capacity_queries) {
    if (!m || capacity_queries < 0 || (capacity_queries > 0 && !logits_out_dev)) return P2S_EINVAL;
    m->logits_capture = capacity_queries > 0 ? logits_out_dev : nullptr;
    m->logits_capacity = capacity_queries;
    return P2S_OK;
}

extern "C" int p2s_debug_fault_chunk(p2s_model_t m, int chunk_index) {
    if (!m) return P2S_EINVAL;
    m->fault_chunk = chunk_index;
    return P2S_OK;
}

// queries q_all[q_begin, q_end) through the double-buffered pipeline.  r_rot != NULL: GT-query pass (rotation).
// r_patch: fixed-radius models only (the generator of the patch choice; in the GT-query pass the same handle as r_rot).
// one-shot logits capture (p2s_model_capture_logits): taken off the model at the top of the call that consumes it, whatever
// the call's outcome
struct LogitsCapture {
    float *dev;
    int64_t room;
    explicit LogitsCapture(p2s_model_s *m) : dev(m ? m->logits_capture : nullptr), room(m ? m->logits_capacity : 0) {
        if (m) {
            m->logits_capture = nullptr;
            m->logits_capacity = 0;
        }
    }
};

static int run_pipeline(p2s_model_s *m, p2s_cloud_s *c, p2s_rng_s *r, p2s_rng_s *r_rot, p2s_rng_s *r_patch, const float *q_all,
                        int64_t q_begin, int64_t q_end, int chunk, float *sdf_out_dev, hipStream_t s, const LogitsCapture &cap) {
    const bool weighted = m->cfg.weighted_subsample != 0;   // p2s_vanilla: choice(p, replace=False) per query
    const double ball_r = m->cfg.patch_radius;
    const bool ball = ball_r > 0.0;                         // patch = points within a fixed radius (p2s_ball.hip)
    const int k = m->cfg.points_per_patch, n = m->cfg.sub_sample_size;
    // default chunk: 8192 queries, except the distance-weighted sub-sample next to the fp32 encoders: 4096 (its generator works
    // in batches of 4096 queries; fp32, test shape at 256^3, r05: 2048 / 4096 / 6144 / 8192 queries: 110.8 / 112.1 / 111.6 /
    // 112.2 k queries/s -- the smaller one for its table memory; fp16 pair: 1024 / 2048 / 4096 / 8192: 250.6 / 261.4 / 261.9 /
    // 266.8 k: until r04 the auxiliary stream's chain kernel waited milliseconds for a drained CU and 2048 was best)
    if (chunk <= 0) chunk = (weighted && !m->cfg.encoder_bf16) ? std::min(m->max_chunk, 4096) : m->max_chunk;
    chunk = std::min(chunk, m->max_chunk);
    if (m->overlap && !m->aux) {
        // high queue priority: the data-path kernels are tiny next to the encoder kernel and must not queue
        // behind its ~8k workgroups for a free CU slot.  (Giving the stream its own CUs with a CU mask was
        // measured slower: the masked compute stream lost far more than the masked-off CUs.)
        int lo = 0, hi = 0;
        P2S_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        P2S_HIP_CHECK(hipStreamCreateWithPriority(&m->aux, hipStreamNonBlocking, hi));
    }
    hipStream_t sa = m->overlap ? m->aux : s;
    // (kNN + gather of chunk i run on the compute stream, in front of its encoders: on a third stream under the encoders of
    //  chunk i - 1 they were measured to cost the encoder launches exactly what they cost alone -- r03: 177.6 vs 177.3 k
    //  queries/s -- so the simpler order stays and the encoder launch durations (roofline) are those of the kernel alone)
    // fixed radius: the walk along the first generator's stream is ONE wave for tens of ms per chunk; on the auxiliary
    // stream it would sit in front of (or behind) the sub-sample's all-CU kernels, on its own stream it runs beside them
    const bool ball_own = m->cfg.patch_radius > 0.0 && m->overlap;
    if (ball_own && !m->ball) {
        int lo = 0, hi = 0;
        P2S_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        P2S_HIP_CHECK(hipStreamCreateWithPriority(&m->ball, hipStreamNonBlocking, hi));
    }
    // the handle remembers only the CALLER's stream.  The model-owned streams (aux / prep / ball) are drained by this
    // function on every exit (fail(), and the guard below as a backstop): a cloud handle must not keep, let alone
    // synchronise, a stream of a model that may be destroyed before it
    p2s_cloud_note_stream(c, s);
    const int64_t nq = q_end - q_begin;
    float *const logits_cap = cap.dev;
    const int64_t logits_room = cap.room;
    if (logits_cap && logits_room < nq) {
        p2s_set_error("p2s pipeline: logits capture buffer holds %lld queries, the call processes %lld", (long long)logits_room, (long long)nq);
        return P2S_ECAPACITY;
    }
    if (nq <= 0) return P2S_OK;
    struct QuietGuard {
        p2s_cloud_s *c;
        p2s_model_s *m;
        bool drained = false;
        ~QuietGuard() {
            if (!drained) {        // (the model's own streams: an error here is a real fault and stays pending)
                if (m->aux) (void)hipStreamSynchronize(m->aux);
                if (m->ball) (void)hipStreamSynchronize(m->ball);
            }
            --c->foreign_streams_quiet;
        }
    } quiet{c, m};
    ++c->foreign_streams_quiet;
    const int C = (int)std::min<int64_t>(chunk, nq);
    // clouds with fewer points than the sub-sample: shuffle + pad, and shape.pts permuted under the kd-tree
    // (reference source/base/utils.py:221-226; p2s_subsample_shuffle_pad)
    const bool small = c->d.n < n;
    if (ball && (small || !r_patch || (r_rot && r_rot != r_patch))) {
        p2s_set_error("p2s pipeline: a fixed-radius model needs the generator of the patch choice (the rotation generator of the "
                      "GT-query pass) and a cloud with at least sub_sample_size points");
        return P2S_EINVAL;
    }
    int rc = pipe_reserve(m, C, k, n, small);
    if (rc) return rc;
    rc = p2s_model_reserve(m, C);
    if (rc) return rc;
    PipeBuffers &b = m->pipe;
    const int nbuf = (nq > C) ? 2 : 1;
    // every exit after the first launch: both streams drained, so the caller may free / reuse its buffers and the
    // model-owned chunk buffers are idle again
    auto fail = [&](int code) {
        quiet.drained = true;
        const hipError_t e1 = hipStreamSynchronize(s);
        hipError_t e2 = (sa != s) ? hipStreamSynchronize(sa) : hipSuccess;
        const hipError_t e3 = m->ball ? hipStreamSynchronize(m->ball) : hipSuccess;
        if (e2 == hipSuccess) e2 = e3;
        if (code == P2S_OK && (e1 != hipSuccess || e2 != hipSuccess)) {
            p2s_set_error("p2s pipeline: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
            return (int)P2S_EHIP;
        }
        return code;
    };
#define PIPE_HIP(expr)                                                                                     \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess) {                                                                            \
            p2s_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);      \
            return fail(P2S_EHIP);                                                                         \
        }                                                                                                  \
    } while (0)
    hipStream_t sbl = ball_own ? m->ball : sa;        // stream of the fixed-radius patch work
    if (sa != s) {
        PIPE_HIP(hipEventRecord(b.grid, s));
        PIPE_HIP(hipStreamWaitEvent(sa, b.grid, 0));
        if (sbl != sa) PIPE_HIP(hipStreamWaitEvent(sbl, b.grid, 0));
    }

    // fixed radius: the number of points in every query's ball, on the host too (batch sizes and the random words each
    // batch may need follow from it) -- the one blocking call of this mode, before the pipeline starts
    int32_t *ball_cd = nullptr;
    const int32_t *ball_ch = nullptr;
    if (ball) {
        const int eb0 = p2s_prof_mark(m, s);
        if ((rc = p2s_ball_counts_to_host(r_patch, c, q_all + (size_t)q_begin * 3, nq, ball_r, &ball_cd, &ball_ch, s))) return fail(rc);
        p2s_prof_span(m, ST_KNN, eb0, p2s_prof_mark(m, s));
    }
    const bool use_done = ball && sbl != s;

    const int64_t nchunks = (nq + C - 1) / C;
    auto produce = [&](int64_t ci) -> int {       // sub-sample ids of chunk ci on the aux stream
        const int bi = (int)(ci % nbuf);
        const int64_t q0 = q_begin + ci * C;
        const int cur = (int)std::min<int64_t>(C, q_end - q0);
        if (ci >= nbuf && sa != s) PIPE_HIP(hipStreamWaitEvent(sa, b.freed[bi], 0));
        const int e0 = p2s_prof_mark(m, sa);
        int rc2;
        if (small)
            rc2 = p2s_subsample_shuffle_pad(r, c, cur, n, b.perm[bi], b.sub_ids[bi], sa);
        else if (m->cfg.fixed_subsample)      // rng.seed(42) before every draw (reference source/base/utils.py:210-211)
            rc2 = p2s_subsample_fixed(r, c, weighted ? q_all + (size_t)q0 * 3 : nullptr, cur, n, 42u, b.sub_ids[bi], nullptr, sa);
        else
            rc2 = weighted ? p2s_subsample_weighted(r, c, q_all + (size_t)q0 * 3, cur, n, b.sub_ids[bi], nullptr, sa)
                           : p2s_subsample_uniform(r, c, cur, n, b.sub_ids[bi], nullptr, sa);
        if (rc2) return fail(rc2);
        p2s_prof_span(m, ST_SUB, e0, p2s_prof_mark(m, sa));
        if (sa != s) PIPE_HIP(hipEventRecord(b.ready[bi], sa));
        if (ball) {
            // patch choice (+ the rotation of the GT-query pass): a serial walk along the first generator's stream, under
            // the encoders of the chunks before; the encoders of chunk ci - nbuf read the patch buffer it fills
            if (ci >= nbuf && sbl != s) PIPE_HIP(hipStreamWaitEvent(sbl, b.done[bi], 0));
            const int eb0 = p2s_prof_mark(m, sbl);
            rc2 = p2s_ball_patch_counted(r_patch, c, q_all + (size_t)q0 * 3, ball_cd + (q0 - q_begin), ball_ch + (q0 - q_begin), cur,
                                         ball_r, k, r_rot ? 6 : 0, nullptr, b.patch[bi], b.radius[bi], r_rot ? b.rot[bi] : nullptr, sbl);
            if (rc2) return fail(rc2);
            p2s_prof_span(m, ST_KNN, eb0, p2s_prof_mark(m, sbl));
            if (sbl != s) PIPE_HIP(hipEventRecord(b.ball_ready[bi], sbl));
        }
        return P2S_OK;
    };

    // kNN patch + radius and the gathered sub-sample of chunk ci into buffer ci % nbuf (compute stream)
    auto prepare = [&](int64_t ci) -> int {
        const int bi = (int)(ci % nbuf);
        const int64_t q0 = q_begin + ci * C;
        const int cur = (int)std::min<int64_t>(C, q_end - q0);
        const float *qc = q_all + (size_t)q0 * 3;
        const int ek0 = p2s_prof_mark(m, s);
        int rc2 = P2S_OK;
        if (!ball) {
            rc2 = small ? p2s_knn_patch(c, qc, cur, k, b.knn_ids[bi], nullptr, nullptr, s)
                        : p2s_knn_patch_set(c, qc, cur, k, b.patch[bi], b.radius[bi], s);
            if (rc2) return fail(rc2);
            p2s_prof_span(m, ST_KNN, ek0, p2s_prof_mark(m, s));
        }
        if (sa != s) PIPE_HIP(hipStreamWaitEvent(s, b.ready[bi], 0));
        if (ball && sbl != s) PIPE_HIP(hipStreamWaitEvent(s, b.ball_ready[bi], 0));
        if (small) {       // the patch is gathered from the array as the queries before this one left it
            rc2 = p2s_patch_from_ids(c, b.knn_ids[bi], b.perm[bi], qc, cur, k, b.patch[bi], b.radius[bi], s);
            if (rc2) return fail(rc2);
        }
        rc2 = p2s_gather_points(c, b.sub_ids[bi], (int64_t)cur * n, b.sub[bi], s);
        if (rc2) return fail(rc2);
        // the producer only writes sub_ids: free for chunk ci + nbuf as soon as the gather has read them
        if (sa != s) PIPE_HIP(hipEventRecord(b.freed[bi], s));
        return P2S_OK;
    };

    // prologue: up to nbuf chunks of ids in flight
    for (int64_t ci = 0; ci < std::min<int64_t>(nbuf, nchunks); ++ci)
        if ((rc = produce(ci))) return rc;
    for (int64_t ci = 0; ci < nchunks; ++ci) {
        const int bi = (int)(ci % nbuf);
        const int64_t q0 = q_begin + ci * C;
        const int cur = (int)std::min<int64_t>(C, q_end - q0);
        const float *qc = q_all + (size_t)q0 * 3;
        if (m->fault_chunk == (int)ci) {
            m->fault_chunk = -1;
            p2s_set_error("p2s pipeline: injected fault before chunk %lld (p2s_debug_fault_chunk)", (long long)ci);
            return fail(P2S_EHIP);
        }
        if ((rc = prepare(ci))) return rc;
        if (r_rot) {
            // data_loader.py:381-393: rotate sub-sample (model space), patch (patch space) and the query point
            if (!ball && (rc = p2s_random_rotations(r_rot, cur, b.rot[bi], s))) return fail(rc);
            if ((rc = p2s_rotate_points(b.rot[bi], b.sub[bi], n, cur, b.sub[bi], s))) return fail(rc);
            if ((rc = p2s_rotate_points(b.rot[bi], b.patch[bi], k, cur, b.patch[bi], s))) return fail(rc);
            if ((rc = p2s_rotate_points(b.rot[bi], qc, 1, cur, b.qrot[bi], s))) return fail(rc);
            qc = b.qrot[bi];
        }
        rc = p2s_run_chunk(m, b.patch[bi], b.sub[bi], qc, b.radius[bi], cur,
                           logits_cap ? logits_cap + (size_t)(q0 - q_begin) * m->cfg.output_dim : nullptr,
                           sdf_out_dev + (q0 - q_begin), nullptr, nullptr, s, q0 - q_begin);
        if (rc) return fail(rc);
        if (use_done) PIPE_HIP(hipEventRecord(b.done[bi], s));
        if (ci + nbuf < nchunks) {
            if ((rc = produce(ci + nbuf))) return rc;
        }
    }
#undef PIPE_HIP
    m->counters.queries += nq;
    if ((rc = fail(P2S_OK))) return rc;
    // fp16 pair encoder: the queries it flagged (activations beyond the half range) through the fp32 kernels, now
    return p2s_model_fallback_finish(m, logits_cap, sdf_out_dev, s);
}

extern "C" int p2s_infer_shape(p2s_model_t m, p2s_cloud_t c, p2s_rng_t r, int res, int eps, int64_t q_begin,
                               int64_t q_end, int chunk, float *sdf_out_dev, float *q_out_dev, int64_t *n_done,
                               void *stream) {
    if (m && m->cfg.patch_radius > 0.0) {
        p2s_set_error("p2s_infer_shape: fixed-radius model (patch_radius %g): use p2s_infer_shape_ball with the generator of "
                      "the patch choice", m->cfg.patch_radius);
        return P2S_EINVAL;
    }
    return p2s_infer_shape_ball(m, c, r, nullptr, res, eps, q_begin, q_end, chunk, sdf_out_dev, q_out_dev, n_done, stream);
}

extern "C" int p2s_infer_shape_ball(p2s_model_t m, p2s_cloud_t c, p2s_rng_t r, p2s_rng_t r_patch, int res, int eps,
                                    int64_t q_begin, int64_t q_end, int chunk, float *sdf_out_dev, float *q_out_dev,
                                    int64_t *n_done, void *stream) {
    const LogitsCapture cap(m);
    if (!m || !c || !r || !sdf_out_dev || r_patch == r) {
        p2s_set_error("p2s_infer_shape: null argument (or one generator handle passed twice)");
        return P2S_EINVAL;
    }
    if (m->cfg.patch_radius <= 0.0) r_patch = nullptr;
    P2S_HIP_CHECK(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    p2s_prof_reset(m);
    const float *q_all = nullptr;
    long long Q = 0;
    const int eg0 = p2s_prof_mark(m, s);
    int rc = p2s_cloud_grid(c, res, eps, &q_all, &Q, s);
    if (rc) return rc;
    p2s_prof_span(m, ST_GRID, eg0, p2s_prof_mark(m, s));
    if (q_end < 0) q_end = Q;
    if (q_begin < 0 || q_begin > q_end || q_end > Q) {
        p2s_set_error("p2s_infer_shape: query range [%lld,%lld) outside [0,%lld]", (long long)q_begin,
                      (long long)q_end, (long long)Q);
        return P2S_EINVAL;
    }
    if (n_done) *n_done = 0;
    const int64_t nq = q_end - q_begin;
    if (nq == 0) return P2S_OK;
    rc = run_pipeline(m, c, r, nullptr, r_patch, q_all, q_begin, q_end, chunk, sdf_out_dev, s, cap);
    if (rc) return rc;
    if (q_out_dev) {
        P2S_HIP_CHECK(hipMemcpyAsync(q_out_dev, q_all + (size_t)q_begin * 3, (size_t)nq * 12, hipMemcpyDeviceToDevice, s));
        P2S_HIP_CHECK(hipStreamSynchronize(s));
    }
    p2s_prof_collect(m);
    rc = p2s_rng_check(r, s);
    if (rc) return rc;
    if (r_patch && (rc = p2s_rng_check(r_patch, s))) return rc;
    if (n_done) *n_done = nq;
    return P2S_OK;
}

extern "C" int p2s_infer_queries(p2s_model_t m, p2s_cloud_t c, p2s_rng_t r_sub, p2s_rng_t r_rot, const float *q_dev,
                                 int64_t n_queries, int chunk, float *sdf_out_dev, void *stream) {
    const LogitsCapture cap(m);
    if (!m || !c || !r_sub || (n_queries > 0 && (!q_dev || !sdf_out_dev)) || n_queries < 0 || r_rot == r_sub) {
        p2s_set_error("p2s_infer_queries: bad argument (the rotation generator must be a second handle)");
        return P2S_EINVAL;
    }
    P2S_HIP_CHECK(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    p2s_prof_reset(m);
    if (n_queries == 0) return P2S_OK;
    int rc = run_pipeline(m, c, r_sub, r_rot, m->cfg.patch_radius > 0.0 ? r_rot : nullptr, q_dev, 0, n_queries, chunk, sdf_out_dev, s, cap);
    if (rc) return rc;
    p2s_prof_collect(m);
    if ((rc = p2s_rng_check(r_sub, s))) return rc;
    if (r_rot && (rc = p2s_rng_check(r_rot, s))) return rc;
    return P2S_OK;
}
