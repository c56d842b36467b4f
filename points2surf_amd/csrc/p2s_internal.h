// Handle definitions shared by the API translation units.
#pragma once
#include "p2s_common.h"
#include <vector>

// per-shape pipeline buffers (p2s_pipeline.hip): owned by the model handle, grown on demand, reused across
// shapes, released by p2s_model_destroy -- no allocation and no leak on the per-shape path
struct PipeBuffers {
    float *patch[2] = {};           // [C][k][3]
    float *radius[2] = {};          // [C]
    int32_t *sub_ids[2] = {};       // [C][n]
    float *sub[2] = {};             // [C][n][3]
    float *qrot[2] = {};            // [C][3]   rotated query points (GT-query pass)
    double *rot[2] = {};            // [C][9]   per-query rotation (GT-query pass)
    hipEvent_t ready[2] = {};       // data path of the buffer finished (aux stream)
    hipEvent_t freed[2] = {};       // the gather has read the sub-sample ids of the buffer (compute stream)
    hipEvent_t done[2] = {};        // encoders finished reading the buffer (main stream)
    hipEvent_t ball_ready[2] = {};  // fixed-radius patch of the buffer finished (ball stream)
    hipEvent_t grid = nullptr;
    int32_t *knn_ids[2] = {};       // [C][k]   } only for clouds with fewer points than the sub-sample
    int32_t *perm[2] = {};          // [C][N]   } (shape.pts is shuffled in place by every query)
    int cap_chunk = 0, cap_k = 0, cap_n = 0, cap_small = 0;
};

struct p2s_model_s {
    p2s_model_cfg cfg;
    p2s_weight_offsets offs;
    int device = 0;
    float *blob = nullptr;
    size_t n_floats = 0;
    // bf16 encoder (cfg.encoder_bf16): bf16 B fragments of the per-point layers, converted once at creation
    unsigned short *blob_h = nullptr;      // [pieces][h_total] (cfg.encoder_bf16 = number of bf16 pieces per operand)
    size_t h_total = 0;
    size_t h_w0b[2] = {}, h_s1[2] = {}, h_s2[2] = {}, h_s3[2] = {}, h_m2[2] = {}, h_m3[2] = {}, h_qc2 = 0, h_qc3 = 0;
    // fp16 pair mode: the STN / QSTN head layers as 16-bit fragments too (p2s_gemm_f16_kernel); the decoder stays fp32
    size_t h_sf1[2] = {}, h_sf2[2] = {}, h_sf3[2] = {}, h_qf1 = 0, h_qf2 = 0;
    bool heads_f16 = false;
    // fp16 pair mode: queries with an activation beyond the half range are flagged by the 16-bit kernels, collected per
    // chunk (inputs copied aside) and re-run through the fp32 kernels at the end of the same call (p2s_model_fallback_finish)
    struct Fallback {
        int *flags = nullptr;              // [max_chunk] per query of the chunk in flight
        int *count = nullptr;              // queries collected since the last finish (may run past cap: overflow)
        float *patch = nullptr, *sub = nullptr, *query = nullptr, *radius = nullptr;   // [cap] inputs of the collected queries
        long long *index = nullptr;        // [cap] position in the call's output arrays
        float *sdf = nullptr, *logits = nullptr;                                       // [cap] results of the fp32 run
        int cap = 0;
    } fb;
    // one-shot capture of the decoder logits of the next pipeline call (p2s_model_capture_logits; the drop-in's tie report)
    float *logits_capture = nullptr;
    int64_t logits_capacity = 0;
    float *ws = nullptr;      // per-chunk workspace, grown on demand
    int ws_chunk = 0;
    int max_chunk = 8192;     // queries per internal batch (r03: 4096 -> 8192: 180.4 -> 182.0 k queries/s, the launch boundaries
                              // of a chunk -- four drains of the chip -- weigh half as much; 12288: 182.1 k)
    // profiling: HIP events recorded on the launch stream, no host synchronisation until collect
    bool profiling = false;
    std::vector<hipEvent_t> evpool;
    int ev_used = 0;
    struct Span { int stage, a, b; };
    std::vector<Span> spans;
    p2s_counters counters = {};
    // auxiliary stream: the data path (kNN, sub-sample) of chunk i+1, i+2 overlaps the encoders of chunk i
    hipStream_t aux = nullptr;     // high-priority stream of the sub-sample generator
    hipStream_t ball = nullptr;    // fixed-radius models: the serial walk along the first generator's stream (one wave) -- its own
                                   // stream so that it runs beside the sub-sample kernels of the auxiliary stream, not behind them
    bool overlap = true;
    PipeBuffers pipe;
    int fault_chunk = -1;          // test hook (p2s_debug_fault_chunk): fail with P2S_EHIP before this chunk
};
void p2s_pipe_free(p2s_model_s *m);
// cfg.encoder_bf16: 0 fp32, 1 bf16, 2 / 3 split bf16, 4 fp16 pair (2 pieces, two accumulators)
inline int p2s_enc_pieces(const p2s_model_cfg &c) { return c.encoder_bf16 == 4 ? 2 : c.encoder_bf16; }
inline int p2s_enc_f16(const p2s_model_cfg &c) { return c.encoder_bf16 == 4 ? 1 : 0; }
// fp16 pair mode, at the end of every call: the queries the 16-bit kernels flagged (an activation beyond the half range) run
// through the fp32 kernels and their results replace the poisoned ones in logits_out [.][output_dim] / sdf_out (either
// may be null).  Synchronises `s` in that mode.  More flagged queries than the side buffers hold (16384 per call), or a
// call without either output (p2s_encode_features): P2S_EINVAL.
int p2s_model_fallback_finish(p2s_model_s *m, float *logits_out, float *sdf_out, hipStream_t s);

enum P2SStage { ST_CHAIN_STN = 0, ST_HEAD, ST_CHAIN_MAIN, ST_DECODER, ST_KNN, ST_SUB, ST_GRID, ST_CHAIN_QSTN };
int p2s_prof_mark(p2s_model_s *m, hipStream_t s);                 // event index or -1
void p2s_prof_span(p2s_model_s *m, int stage, int a, int b);
void p2s_prof_reset(p2s_model_s *m);
void p2s_prof_collect(p2s_model_s *m);                            // synchronises the last event

int p2s_model_reserve(p2s_model_s *m, int chunk);
// index0: position of the chunk's first query in the CALL's output arrays (what the fp32 fallback scatters to)
int p2s_run_chunk(p2s_model_s *m, const float *patch, const float *sub, const float *query, const float *radius,
                  int C, float *logits_out, float *sdf_out, float *feat_local_out, float *feat_global_out,
                  hipStream_t s, long long index0 = 0);

// ---------------------------------------------------------------------------------------------
// cloud / rng handles (p2s_cloud.hip, p2s_rng.hip)
// ---------------------------------------------------------------------------------------------
struct CloudDev {
    const float *pts;        // [n][3] original order (owned copy)
    const float4 *spts;      // [n] sorted by cell: xyz + original id (bit cast)
    const int *cell_start;   // [G^3 + 1]
    const int *sat;          // [(G+1)^3] inclusive 3-D prefix sums of the per-cell counts
    float lo[3];
    float inv_cell;
    int G;
    int n;
};

struct p2s_cloud_s {
    int device = 0;
    CloudDev d = {};
    // one block of the device's cache (p2s_pool_alloc) holds pts, spts, cell_start, sat, totals + the build scratch
    char *arena = nullptr;
    float *pts = nullptr;
    float4 *spts = nullptr;
    int *cell_start = nullptr;
    int *sat = nullptr;
    // streams that may have work on this handle's memory in flight: drained before its blocks return to the cache
    hipStream_t streams[4] = {};
    int n_streams = 0;
    bool many_streams = false;
    // > 0 while a per-shape pipeline runs on this handle: the model-owned auxiliary streams it uses are NOT noted --
    // run_pipeline drains them on every exit, and a handle must never synchronise a stream it does not own (the model
    // may be destroyed before the cloud)
    int foreign_streams_quiet = 0;
    // query-grid scratch (grown on demand)
    uint32_t *occ = nullptr;
    size_t occ_words = 0;
    int *blk_cnt = nullptr;
    size_t blk_cap = 0;
    long long *totals = nullptr;   // [2] device: total count, error flag
    int *shuffle_perm = nullptr;   // clouds with fewer points than the sub-sample: current row order of shape.pts
    // the last query grid stays on the handle: the pipeline and the callers that size their outputs share it
    float *qcache = nullptr;
    int qc_res = 0, qc_eps = 0;
    long long qc_n = -1, qc_cap = 0;
    hipEvent_t grid_ev = nullptr;  // recorded behind the compaction of the cached grid on grid_stream
    hipStream_t grid_stream = nullptr;
    // summation plan of np.sum(float32[n]) for the weighted sub-sample (p2s_wchoice.hip), built on first use
    int *wc_plan = nullptr;        // device: leaves [L][3], ops [O][3], level offsets [levels+1]
    int wc_leaves = 0, wc_ops_at = 0, wc_lvl_at = 0, wc_levels = 0, wc_root = 0, wc_nodes = 0;
    // fixed-radius patches (p2s_ball.hip), built on first use: the cloud in scipy's cKDTree(pts, 1000).indices order
    char *kd_blob = nullptr;
    float4 *kd_pts = nullptr;      // [n] xyz + original id
    int *kd_leaf = nullptr;        // [kd_leaves + 1] leaf ranges
    float *kd_box = nullptr;       // [kd_leaves][6] lo, hi of each leaf's points
    int kd_leaves = 0;
};

struct p2s_rng_s {
    int device = 0;
    uint32_t *state = nullptr;     // [624] mt + [1] pos
    // parallel generation (GF(2) jump-ahead), optional: tables uploaded by p2s_rng_set_jump_tables
    int levels_max = 0;            // jump tables uploaded (sessions use 2^levels_max streams)
    int levels_alloc = 0;          // tmp / blk_cum sized for 2^levels_alloc streams
    int blocks_per_stream = 0;
    // session (p2s_rng.hip): 0 closed, 1 randint values, 2 raw words
    int sess_mode = 0;
    uint32_t sess_rng = 0, sess_mask = 0;
    long long sess_cursor = 0;     // mode 1: values handed out; mode 2: conservative word estimate
    long long sess_limit = 0;
    uint16_t *jump_sup = nullptr;  // concatenated supports of the jump polynomials
    int jump_off[16] = {};         // offset / count per level
    int jump_cnt[16] = {};
    uint32_t *streams = nullptr;   // [S][624] block states
    uint32_t *tmp = nullptr;       // [S][B*624] accepted values per stream
    int *blk_cum = nullptr;        // [S][B] cumulative accepted count per block
    long long *meta = nullptr;     // [S] offsets + locate record + sticky error flag + raw-request record
    // weighted sub-sample workspace (p2s_wchoice.hip), grown on demand
    double *wc_S = nullptr;        // [C][n]   exact prefix sums of the probabilities
    void *wc_T = nullptr;          // [C][K]   guide records of the cdf (16 B each)
    double *wc_sc = nullptr;       // [C]      per-query scalars (stot, word offset, pmax, dmax + sum, mu)
    void *wc_spec = nullptr;       // offsets pass: ctl, window origins, verdicts [SP_B][SP_W], tentative path, saved windows, scratch
    unsigned short *wc_J = nullptr; // [2 SP_B][SP_W] ruler of jump tables of the offsets chain (p2s_wchoice.hip)
    size_t wc_cap_q = 0, wc_cap_n = 0, wc_cap_k = 0;
    // fixed-radius patches (p2s_ball.hip): hit counts of a shape's queries (device + pinned host), batch work space
    int32_t *ball_counts_dev = nullptr, *ball_counts_host = nullptr;
    size_t ball_counts_cap = 0;
    void *ball_ws = nullptr;
    size_t ball_ws_bytes = 0;
};

// per-device cache of device-memory blocks for the cloud handles (p2s_cloud.hip)
void *p2s_pool_alloc(int device, size_t bytes);
void p2s_pool_free(int device, void *p);
void p2s_cloud_note_stream(p2s_cloud_s *c, hipStream_t s);


// query grid of (res, eps), computed once per cloud handle and kept on the device (p2s_cloud.hip); *q is owned by
// the handle and valid until the next call with other parameters; stream-ordered on `s` (synchronises it once to
// learn the count)
int p2s_cloud_grid(p2s_cloud_s *c, int res, int eps, const float **q, long long *n, hipStream_t s);

// serial generator (p2s_cloud.hip) and parallel generator (p2s_rng.hip)
int p2s_rng_serial_randint(p2s_rng_s *r, uint32_t rng, uint32_t mask, long long target, int32_t *out, hipStream_t s);
// sessions: one large generated segment of the stream that many calls draw from (p2s_rng.hip)
long long p2s_rng_session_words(const p2s_rng_s *r);           // raw words a session holds
long long *p2s_rng_raw_meta(p2s_rng_s *r);                     // device: [0] word cursor of the raw session, [1] sticky error
int p2s_rng_session_close(p2s_rng_s *r, hipStream_t s);        // advance the generator to the cursor; no-op if closed
int p2s_rng_session_randint(p2s_rng_s *r, uint32_t rng, uint32_t mask, long long target, int32_t *out, hipStream_t s);
int p2s_rng_session_raw(p2s_rng_s *r, long long need_words, hipStream_t s);
void p2s_wc_free_rng(p2s_rng_s *r);               // p2s_wchoice.hip workspace
// fixed-radius patches (p2s_ball.hip)
void p2s_ball_free_rng(p2s_rng_s *r);
int p2s_cloud_kd_prepare(p2s_cloud_s *c);
int p2s_ball_counts_to_host(p2s_rng_s *r, p2s_cloud_s *c, const float *q_dev, int64_t nq, double radius, int32_t **count_dev,
                            const int32_t **count_host, hipStream_t s);
// patch_out_dev == NULL: advance the generator only
int p2s_ball_patch_counted(p2s_rng_s *r, p2s_cloud_s *c, const float *q_dev, const int32_t *count_dev, const int32_t *count_host,
                           int64_t nq, double radius, int k, int tail_words, int32_t *ids_out_dev, float *patch_out_dev,
                           float *radius_out_dev, double *rot_out_dev, hipStream_t s);
// rand(3) -> rotation matrix for n queries whose 6 words lie contiguously at six_dev[6 * i] (p2s_pipeline.hip)
int p2s_rotations_from_words(const uint32_t *six_dev, long long n, double *rot_out_dev, hipStream_t s);
void p2s_mt_seed_host(uint32_t seed, uint32_t st[625]);                 // init_genrand
int p2s_rng_reseed(p2s_rng_s *r, uint32_t seed, hipStream_t s);         // rng.seed(seed): closes any session
int p2s_wc_subsample_fixed(p2s_rng_s *r, p2s_cloud_s *c, const float *q_dev, int64_t nq, int n_sel, uint32_t seed,
                           int32_t *ids_out_dev, float *pts_out_dev, hipStream_t s);
