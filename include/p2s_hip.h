/*
 * p2s_hip.h -- C ABI of the MI355X-native Points2Surf SDF-inference engine (libp2s_hip.so).
 *
 * The reference (ErlerPhilipp/points2surf) is pure Python and has no FFI; its boundary for
 * this path is the Python API
 *     source/points_to_surf_eval.py:297-404   points_to_surf_eval(eval_opt)
 *     source/points_to_surf_model.py:237-352  PointsToSurfModel(**kw).forward(dict) -> [B,2]
 * Each entry point below names the reference call it replaces underneath that API.  The
 * host-side mirror of the Python API (points2surf_amd/dropin/source/...) binds these symbols
 * with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions: plain C types only; every function returns 0 on success or a negative
 * P2S_E* code (message via p2s_last_error(), thread-local).  All "dev" pointers are device
 * (HBM) pointers owned by the caller (e.g. PyTorch-ROCm tensor storage).  `stream` is a
 * hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous on `stream`
 * unless documented otherwise; handles are owned by the library until *_destroy.
 */
#ifndef P2S_HIP_H
#define P2S_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P2S_ABI_VERSION 5

#define P2S_OK            0
#define P2S_EINVAL       -1   /* bad argument / unsupported configuration */
#define P2S_EHIP         -2   /* HIP runtime error (see p2s_last_error) */
#define P2S_ENOMEM       -3
#define P2S_ECAPACITY    -4   /* caller-provided output buffer too small, or an input beyond a documented size limit */
#define P2S_ENODEVICE    -5   /* no gfx950 device visible */
#define P2S_EIO          -6   /* host file could not be opened / written / closed (the p2s_write_* functions) */

typedef struct p2s_model_s *p2s_model_t;
typedef struct p2s_cloud_s *p2s_cloud_t;
typedef struct p2s_rng_s   *p2s_rng_t;

int         p2s_abi_version(void);
const char *p2s_last_error(void);
/* number of visible HIP devices (0 if none); never fails */
int         p2s_device_count(void);
/* The library caches device memory per device: the blocks of destroyed cloud handles (so that a handle per shape costs
 * no hipMalloc / hipFree) and the scratch of the volume / iso-surface stages (~2 GB after a 512^3 call).  This gives
 * all of it back to HIP.  Waits for a running p2s_sdf_volume / p2s_marching_cubes on `device`. */
int         p2s_release_scratch(int device);

/* ------------------------------------------------------------------------------------------
 * Model  (replaces make_regressor: PointsToSurfModel(...).cuda(); load_state_dict(); eval(),
 *         reference source/points_to_surf_eval.py:150-171)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t net_size;            /* 1024 (train --net_size)                                    */
    int32_t points_per_patch;    /* 300  (train --points_per_patch)                            */
    int32_t sub_sample_size;     /* 1000 (train --sub_sample_size)                             */
    int32_t output_dim;          /* 2: [|d| logit, sign logit] (outputs imp_surf_magnitude, imp_surf_sign);
                                    1: the signed-distance logit (output imp_surf, p2s_regression)  */
    int32_t use_point_stn;       /* QSTN present (p2s_vanilla and most ablation models)        */
    int32_t shared_transformer;  /* 1: one QSTN over cat(patch, sub-sample) (p2s_vanilla);
                                    0: the QSTN of feat_global, over the sub-sample only; its
                                       rotation also turns the patch (p2s_uniform, *_kNN ...)  */
    int32_t weighted_subsample;  /* 0: ids = randint (train --uniform_subsample 1, p2s_max);
                                    1: distance-weighted choice without replacement (p2s_vanilla) */
    int32_t encoder_bf16;        /* 0 (default): exact fp32.  1: per-point encoder layers on bf16 MFMA (fp32 accumulate; first
                                    layer, STN/QSTN heads, fold and decoder stay fp32) -- outside the 1e-4 contract.
                                    2 / 3: split precision, every operand as 2 / 3 bf16 pieces (3 / 6 bf16 MFMAs per
                                    product, 16 / 24 mantissa bits).  4: fp16 PAIR per operand, x = h0 + h1 * 2^-11 with
                                    the residual scaled into the normal range and a second accumulator (3 fp16 MFMAs per
                                    product, 22 mantissa bits): as exact as fp32 on the reference's goldens at 2.5x its
                                    throughput.  A query with an activation beyond the half range (6e4) is re-run through
                                    the fp32 kernels inside the same call (p2s_counters.fallback_queries counts them; more
                                    than 16384 per call: P2S_EINVAL) -- safe for an arbitrary checkpoint.  See DESIGN.md  */
    int32_t fixed_subsample;     /* train --fixed_subsample 1 (ablation): the generator is re-seeded with 42 before every
                                    query's draw (reference source/base/utils.py:210-211)                       */
    int32_t single_transformer;  /* train --single_transformer 1 (p2s_shared_encoder): ONE encoder over cat(patch,
                                    sub-sample); enc[0] = enc[1] = feat_local_global.*, the QSTN is its stn1 and sees
                                    all points, d1l / d1g = the two halves of fc1_local_global (1024 -> 1024)  */
    double  patch_radius;        /* train --patch_radius: 0 = the points_per_patch nearest neighbours, radius = their largest
                                    distance; > 0 (p2s_{small,medium,large}_radius: 0.05 / 0.1 / 0.2) = all points within
                                    this distance, a random points_per_patch of them if there are more, padded with the
                                    query point if fewer (reference source/base/point_cloud.py:177-191).  The float64
                                    value of the reference's Python float: the ball test is r * r in float64          */
    int32_t sym_sum;             /* train --sym_op sum (reference source/points_to_surf_model.py:170-175,211-214; set by no
                                    experiment script): the pool of PointNetfeat is a SUM over the points instead of the
                                    max.  The STN / QSTN trunks keep their max-pool in the reference too (:47, :106)  */
    int32_t reserved[3];
} p2s_model_cfg;

/* Offsets (in floats) into the weight blob.  The blob holds BatchNorm-folded fp32 weights,
 * GEMM operands pre-packed in MFMA B-fragment order ([N/32][K/8][64 lanes][4]); it is produced
 * by points2surf_amd/weights.py from a reference state_dict.  enc[0] = feat_local (kNN patch),
 * enc[1] = feat_global (sub-sample).  See DESIGN.md "weight blob". */
typedef struct {
    uint64_t w0a, b0a;           /* conv0a+bn0a: [3][64] plain, [64]                           */
    uint64_t w0b, b0b;           /* conv0b+bn0b: packed K=64 N=64                              */
    uint64_t s1, sb1;            /* stn2.conv1+bn1 packed 64x64                                */
    uint64_t s2, sb2;            /* stn2.conv2+bn2 packed 64x128                               */
    uint64_t s3, sb3;            /* stn2.conv3+bn3 packed 128x1024                             */
    uint64_t sf1, sfb1;          /* stn2.fc1+bn4 packed 1024x512                               */
    uint64_t sf2, sfb2;          /* stn2.fc2+bn5 packed 512x256                                */
    uint64_t sf3, sfb3;          /* stn2.fc3 (+identity in bias) packed 256x4096               */
    uint64_t m1t, mb1;           /* conv1+bn1, transposed & packed as B operand of the fold    */
    uint64_t m2, mb2;            /* conv2+bn2 packed 64x128                                    */
    uint64_t m3, mb3;            /* conv3+bn3 packed 128x1024                                  */
} p2s_encoder_offsets;

typedef struct {
    uint64_t c1, cb1;            /* conv1+bn1: [3][64] plain                                   */
    uint64_t c2, cb2;            /* conv2+bn2 packed 64x128                                    */
    uint64_t c3, cb3;            /* conv3+bn3 packed 128x1024                                  */
    uint64_t f1, fb1;            /* fc1+bn4 packed 1024x512                                    */
    uint64_t f2, fb2;            /* fc2+bn5 packed 512x256                                     */
    uint64_t f3, fb3;            /* fc3 (+[1,0,0,0] in bias): plain [256][4], [4]              */
} p2s_qstn_offsets;

typedef struct {
    p2s_encoder_offsets enc[2];
    p2s_qstn_offsets    qstn;    /* valid iff cfg.use_point_stn: point_stn.* (shared) or
                                    feat_global.stn1.* (not shared)                            */
    uint64_t d1l, db1l;          /* fc1_local+bn1_local   packed 1024x512                      */
    uint64_t d1g, db1g;          /* fc1_global+bn1_global packed 1024x512                      */
    uint64_t d2, db2;            /* fc2+bn2 packed 1024x256                                    */
    uint64_t d3, db3;            /* fc3+bn3 packed 256x128                                     */
    uint64_t d4, db4;            /* fc4: plain [128][2], [2]                                   */
} p2s_weight_offsets;

int p2s_model_create(const p2s_model_cfg *cfg, const float *blob_host, size_t n_floats,
                     const p2s_weight_offsets *offs, int device, p2s_model_t *out);
int p2s_model_destroy(p2s_model_t m);

/* a8 + a9: PointsToSurfModel.forward (reference source/points_to_surf_model.py:296-352) followed
 * by post_process (source/points_to_surf_eval.py:174-196, source/sdf_nn.py:11-21) and the
 * magnitude*sign / NaN->1 of save_evaluation (:263-273, :205-207).
 *   patch_ps_dev [B][points_per_patch][3]   kNN patch in patch space
 *   sub_ms_dev   [B][sub_sample_size][3]    global sub-sample in MODEL space (NOT translated; the
 *                                           kernel subtracts the query point; input is not modified)
 *   query_dev    [B][3], radius_dev [B] (may be NULL iff sdf_out_dev is NULL)
 *   logits_out_dev [B][output_dim] (may be NULL), sdf_out_dev [B] (may be NULL)
 * Work buffers are owned by the model and grown on demand (not thread-safe per model). */
int p2s_encode_decode(p2s_model_t m, const float *patch_ps_dev, const float *sub_ms_dev,
                      const float *query_dev, const float *radius_dev, int B,
                      float *logits_out_dev, float *sdf_out_dev, void *stream);

/* stage-wise access for parity tests: encoder features of both branches
 * (feat_local / feat_global outputs, reference points_to_surf_model.py:333,341), [B][net_size] each */
int p2s_encode_features(p2s_model_t m, const float *patch_ps_dev, const float *sub_ms_dev,
                        const float *query_dev, int B, float *feat_local_dev, float *feat_global_dev,
                        void *stream);

/* ------------------------------------------------------------------------------------------
 * Cloud  (replaces load_shape: cKDTree(pts, leaf_size=1000), reference source/data_loader.py:16-68)
 * ------------------------------------------------------------------------------------------ */
/* The neighbour index -- uniform cell grid G^3 over the bounding box, points counting-sorted by cell (stable: original
 * order inside a cell), 3-D summed-area table of the cell counts -- is built ON THE DEVICE, stream-ordered on `stream`:
 * bounding box + non-finite check, one 32-byte read-back (the call's only blocking operation; non-finite coordinates
 * are rejected with P2S_EINVAL like the reference's kd-tree build would fail), cell histogram, three scan passes,
 * scatter, in-cell ranking.  The handle owns a copy of the points; its memory comes from a per-device block cache
 * (see p2s_release_scratch), so creating a handle per shape allocates nothing once the cache is warm.
 * p2s_cloud_destroy drains the streams the handle was used on before its blocks are recycled. */
int p2s_cloud_create(const float *pts_dev, int n_points, int device, void *stream, p2s_cloud_t *out);
int p2s_cloud_destroy(p2s_cloud_t c);
int p2s_cloud_num_points(p2s_cloud_t c);
/* test / diagnostic read-back of the index (host buffers, any may be NULL): *G_host = cells per axis; geom_host[4] =
 * bounding-box minimum x, y, z and 1 / cell size; cell_start_host [G^3 + 1]; sat_host [(G + 1)^3] (inclusive prefix
 * sums, zero borders); sorted_host [n][4] = x, y, z, original index (int32 bit pattern).  Synchronises `stream`. */
int p2s_cloud_index_export(p2s_cloud_t c, int32_t *G_host, float *geom_host, int32_t *cell_start_host,
                           int32_t *sat_host, float *sorted_host, void *stream);

/* a1: get_voxel_centers_grid_smaller_pc (reference source/sdf.py:46-79).  Writes up to `capacity`
 * query points (C order of the voxel index) to q_out_dev [capacity][3]; *n_queries receives the
 * full count (synchronises `stream`).  Returns P2S_ECAPACITY if capacity is too small (count is
 * still reported) -- call once with capacity 0 to size the buffer. */
int p2s_query_grid(p2s_cloud_t c, int grid_resolution, int epsilon, float *q_out_dev,
                   int64_t capacity, int64_t *n_queries, void *stream);

/* a4 + a5: get_patch_kdtree (kdtree.query(k), fp64 ranking; reference source/base/point_cloud.py:170-175)
 * + get_patch_radii / model_space_to_patch_space (source/base/utils.py:62-69,80-88;
 * source/data_loader.py:341-350).  ids sorted by ascending distance.
 *   ids_out_dev [Q][k] int32 (may be NULL), patch_ps_out_dev [Q][k][3] (may be NULL),
 *   radius_out_dev [Q] (may be NULL) */
int p2s_knn_patch(p2s_cloud_t c, const float *query_dev, int64_t n_queries, int k,
                  int32_t *ids_out_dev, float *patch_ps_out_dev, float *radius_out_dev, void *stream);
/* the same k nearest points as a SET: patch rows in an arbitrary (deterministic) order, no ids -- what the per-shape
 * pipeline uses: the encoders max-pool over the patch, so the order of its points changes no bit of the result, and the
 * k smallest distances are selected by bisection on their bit patterns instead of a sort (~3x faster).  Ties at the
 * k-th distance are broken by id exactly like p2s_knn_patch.  radius_out_dev as above. */
int p2s_knn_patch_set(p2s_cloud_t c, const float *query_dev, int64_t n_queries, int k,
                      float *patch_ps_out_dev, float *radius_out_dev, void *stream);

/* ------------------------------------------------------------------------------------------
 * a6: global sub-sample (reference source/base/utils.py:196-227 with the dataset-wide
 *     np.random.RandomState of source/data_loader.py:274-277)
 * ------------------------------------------------------------------------------------------ */
int p2s_rng_create(uint32_t seed, int device, p2s_rng_t *out);   /* RandomState(seed) */
int p2s_rng_destroy(p2s_rng_t r);
/* copy the 624-word MT19937 state + position out / in (host memory); for sharding + tests */
int p2s_rng_get_state(p2s_rng_t r, uint32_t *mt624_host, int32_t *pos_host, void *stream);
int p2s_rng_set_state(p2s_rng_t r, const uint32_t *mt624_host, int32_t pos, void *stream);

/* Tables for parallel generation of the stream (GF(2) jump-ahead, tools/mt_jump.py ->
 * points2surf_amd/mt_jump_tables.npz): supports of t^(B*2^m*624) mod phi(t), m = 0..levels-1, concatenated
 * (uint16 exponents), counts_host[m] entries each.  Large requests are then served from a *session*: 2^levels
 * streams of B blocks generated at once (1.3 GB for 512 x 1024 blocks), from which consecutive calls take their
 * values; the 624-word state is advanced when the session ends (any call that needs the state, or a request with
 * another modulus).  Results are bit-identical to the serial generator.  Required by p2s_subsample_weighted. */
int p2s_rng_set_jump_tables(p2s_rng_t r, const uint16_t *supports_host, const int32_t *counts_host, int levels,
                            int blocks_per_stream);
/* sticky error of the parallel generator (0 = none); synchronises `stream` */
int p2s_rng_check(p2s_rng_t r, void *stream);

/* uniform mode (p2s_max, uniform_subsample=1): ids = rng.randint(0, N, n) per query, consumed in
 * query order from one continuous stream.  ids_out_dev [Q][n] int32; NULL = only advance the stream past
 * these queries (query-range sharding), pts_out_dev [Q][n][3] gathered points in model space (may be NULL). */
int p2s_subsample_uniform(p2s_rng_t r, p2s_cloud_t c, int64_t n_queries, int n,
                          int32_t *ids_out_dev, float *pts_out_dev, void *stream);
/* distance-weighted mode (p2s_vanilla, uniform_subsample=0; reference source/base/utils.py:200-219):
 * per query p = clip(1 - 1.5 d/max(d), 0.05, 1) / sum (float32, numpy's summation order) and
 * ids = rng.choice(N, n, replace=False, p=p), consumed in query order from the same continuous stream --
 * bit-identical to numpy's legacy RandomState.  Needs the jump tables (p2s_rng_set_jump_tables).
 * q_dev [Q][3] query points (model space), ids_out_dev [Q][n] int32 (NULL = advance only), pts_out_dev as above.
 * Errors found on the device (degenerate distances) are reported by p2s_rng_check.  n <= 1024; clouds of more than
 * 475,040 points are refused with P2S_ECAPACITY before the stream is touched (LDS bitmap of the in-place algorithm). */
int p2s_subsample_weighted(p2s_rng_t r, p2s_cloud_t c, const float *q_dev, int64_t n_queries, int n,
                           int32_t *ids_out_dev, float *pts_out_dev, void *stream);
/* fixed mode (train --fixed_subsample 1; reference source/base/utils.py:210-211): ``rng.seed(seed)`` before EVERY
 * query's draw, seed = 42 in the reference.  q_dev = NULL: uniform draw (every query gets the same ids);
 * q_dev [Q][3]: distance-weighted choice (same random words, per-query probabilities; needs the jump tables).
 * Afterwards the generator is where numpy's is: seeded + the last query's consumption. */
int p2s_subsample_fixed(p2s_rng_t r, p2s_cloud_t c, const float *q_dev, int64_t n_queries, int n, uint32_t seed,
                        int32_t *ids_out_dev, float *pts_out_dev, void *stream);
/* clouds with FEWER points than the sub-sample size (reference source/base/utils.py:221-226): rng.shuffle of the
 * point array (numpy legacy shuffle: rk_interval per row) + zero padding.  The reference shuffles shape.pts IN PLACE
 * under the kd-tree, so every query permutes the array later patches are gathered from; the cloud handle keeps that
 * permutation.  perm_before_dev [Q][N] (may be NULL): row -> original point id BEFORE query i's shuffle (what the
 * patch gather of query i sees); ids_out_dev [Q][n] (may be NULL): the shuffled ids, -1 = zero padding.
 * p2s_subsample_uniform / _weighted / _fixed route here automatically for such clouds. */
int p2s_subsample_shuffle_pad(p2s_rng_t r, p2s_cloud_t c, int64_t n_queries, int n, int32_t *perm_before_dev,
                              int32_t *ids_out_dev, void *stream);
/* a5 from explicit kNN ids (rows looked up through perm_before_dev, NULL = identity): radius + patch space */
int p2s_patch_from_ids(p2s_cloud_t c, const int32_t *ids_dev, const int32_t *perm_before_dev, const float *query_dev,
                       int64_t n_queries, int k, float *patch_ps_out_dev, float *radius_out_dev, void *stream);
/* a4, fixed-radius branch: get_patch_kdtree with patch_radius > 0 (reference source/base/point_cloud.py:177-191,
 * source/data_loader.py:335-350) -- see points2surf_amd/csrc/p2s_ball.hip.
 * p2s_kd_order_host: HOST function (no device needed): the index array of scipy.spatial.cKDTree(pts, leafsize) --
 * query_ball_point returns its hits in this order.  order_out [n]; leaf_start_out [n_leaves + 1] (may be NULL).
 * p2s_ball_count: number of points within `radius` of every query (= len(query_ball_point)).
 * p2s_ball_patch: for queries in order, the patch ids (ids_out_dev [Q][k], may be NULL; padding = 0), the patch in patch
 * space (patch_out_dev [Q][k][3]; NULL = only advance the generator) and radius_out_dev [Q] (may be NULL) = 1: the scale
 * of the distance output, which fixed-radius models do not rescale (source/points_to_surf_eval.py:180,188).  r = the
 * data set's FIRST RandomState (self.rng): a query with more than k points in its ball consumes the words of
 * permutation(count).  with_rotation: every query then also draws the rand(3) of the GT-query pass
 * (data_loader.py:384) and rot_out_dev [Q][9] receives its rotation matrix.  Synchronises `stream` once (hit counts). */
int p2s_kd_order_host(const float *pts_host, int64_t n, int leafsize, int32_t *order_out, int32_t *leaf_start_out,
                      int64_t leaf_cap, int32_t *n_leaves_out);
int p2s_ball_count(p2s_cloud_t c, const float *query_dev, int64_t n_queries, double radius, int32_t *count_out_dev,
                   void *stream);
int p2s_ball_patch(p2s_rng_t r, p2s_cloud_t c, const float *query_dev, int64_t n_queries, double radius,
                   int points_per_patch, int with_rotation, int32_t *ids_out_dev, float *patch_out_dev,
                   float *radius_out_dev, double *rot_out_dev, void *stream);
/* given-ids mode (ids produced elsewhere; id < 0 = zero point) */
int p2s_gather_points(p2s_cloud_t c, const int32_t *ids_dev, int64_t n_ids, float *pts_out_dev,
                      void *stream);

/* ------------------------------------------------------------------------------------------
 * Fused per-shape pipeline: what the batch loop of points_to_surf_eval does for one shape in
 * reconstruction mode (reference source/points_to_surf_eval.py:358-404), queries [q_begin, q_end)
 * of the shape's query list (pass 0, -1 for all).  The RNG stream is consumed for exactly the
 * processed queries.  sdf_out_dev [q_end-q_begin]; q_out_dev [q_end-q_begin][3] (may be NULL).
 * chunk = queries per internal batch (0 = default).
 * ------------------------------------------------------------------------------------------ */
int p2s_infer_shape(p2s_model_t m, p2s_cloud_t c, p2s_rng_t r, int grid_resolution, int epsilon,
                    int64_t q_begin, int64_t q_end, int chunk, float *sdf_out_dev, float *q_out_dev,
                    int64_t *n_done, void *stream);
/* the same for a fixed-radius model (cfg.patch_radius > 0): r_patch = the data set's FIRST RandomState (self.rng), which
 * the patch choice draws from (reference source/data_loader.py:335-338); p2s_infer_shape refuses such a model.  With
 * cfg.patch_radius == 0, r_patch is ignored (may be NULL). */
int p2s_infer_shape_ball(p2s_model_t m, p2s_cloud_t c, p2s_rng_t r, p2s_rng_t r_patch, int grid_resolution, int epsilon,
                         int64_t q_begin, int64_t q_end, int chunk, float *sdf_out_dev, float *q_out_dev,
                         int64_t *n_done, void *stream);

/* ------------------------------------------------------------------------------------------
 * GT-query evaluation pass: the batch loop of points_to_surf_eval with reconstruction=False, the pass
 * full_eval.py:31-33 runs first when <indir>/05_query_dist exists (reference source/data_loader.py:365-393).
 * Query points are given (05_query_pts/<shape>.ply.npy), and every query draws a random rotation
 *     rand_rot = trimesh.transformations.random_rotation_matrix(self.rng.rand(3))     (data_loader.py:384)
 * from the dataset's FIRST RandomState (self.rng, :272; the sub-sample uses the second one, :277), applied in
 * float64 to the sub-sample (model space), the patch (patch space) and the query point, each cast back to float32
 * (:385-393).  r_rot = NULL: no rotation (plain inference at given query points).  sdf_out_dev [n_queries].
 * Fixed-radius models (cfg.patch_radius > 0) need r_rot: the same generator makes every query's patch choice, right
 * before its rotation.  Synchronises `stream`.
 * ------------------------------------------------------------------------------------------ */
int p2s_infer_queries(p2s_model_t m, p2s_cloud_t c, p2s_rng_t r_sub, p2s_rng_t r_rot, const float *q_dev,
                      int64_t n_queries, int chunk, float *sdf_out_dev, void *stream);
/* the two building blocks, for stage-wise parity tests:
 *   n rotations from the stream of r (6 raw words each): rot_out_dev [n][9] float64 row-major upper-left 3x3 of
 *   random_rotation_matrix(r.rand(3)); needs the jump tables */
int p2s_random_rotations(p2s_rng_t r, int64_t n, double *rot_out_dev, void *stream);
/*   trimesh.transformations.transform_points(pts, M).astype(float32) per item: pts_in_dev / pts_out_dev
 *   [n_items][points_per_item][3] (may alias), rot_dev [n_items][9] float64 */
int p2s_rotate_points(const double *rot_dev, const float *pts_in_dev, int points_per_item, int64_t n_items,
                      float *pts_out_dev, void *stream);
/* One-shot capture of the decoder's raw logits: the NEXT p2s_infer_shape / p2s_infer_shape_ball / p2s_infer_queries call on
 * this model also writes logits_out_dev [processed queries][output_dim] (column output_dim - 1 = the sign logit, or the one
 * signed-distance logit of the regression model) -- what post_process (reference source/points_to_surf_eval.py:174-196)
 * starts from.  The drop-in's tie report (P2S_TIE_REPORT) lists the queries whose sign logit lies within fp32 noise of the
 * reference's decision ``logit >= 0`` (source/sdf_nn.py:16-21).  capacity_queries < the queries of that call: the call
 * returns P2S_ECAPACITY.  capacity_queries = 0 cancels. */
int p2s_model_capture_logits(p2s_model_t m, float *logits_out_dev, int64_t capacity_queries);
/* test hook: the next pipeline call on this model fails with P2S_EHIP before chunk `chunk_index` (-1 = off) */
int p2s_debug_fault_chunk(p2s_model_t m, int chunk_index);

/* ------------------------------------------------------------------------------------------
 * "next" row: the consumer of the SDF samples.  add_samples_to_volume + propagate_sign
 * (+ the clamp that follows) of reference source/sdf.py:82-178,199-201: scatter the samples into a dense
 * grid_res^3 volume, set the border faces to -1 (outside), propagate signs into the unknown voxels with a
 * sigma^3 box filter (edge replication) thresholded at certainty_threshold until the number of unknown
 * voxels stops falling.  vol_out_dev: [grid_res]^3 float32 in C order -- every value of the reference's
 * float64 volume (float32 SDF samples, -1, 0, +1) is exactly representable.  One sample per voxel is
 * expected (what the query grid produces).  *iterations (host, may be NULL) = number of sweeps.
 * Synchronises `stream`.
 * ------------------------------------------------------------------------------------------ */
int p2s_sdf_volume(const float *query_dev, const float *sdf_dev, int64_t n, int grid_res, int sigma,
                   float certainty_threshold, int clamp, int device, float *vol_out_dev, int32_t *iterations,
                   void *stream);

/* ------------------------------------------------------------------------------------------
 * "next" row: iso-surface of the clamped volume at level 0 -- the step the reference hands to
 * scikit-image, `measure.marching_cubes_lewiner(volume, 0)` (reference source/sdf.py:211-215) -- followed by the
 * vertex transform ((v + 0.5) / res - 0.5) * 2 (:223, float32 arithmetic like numpy; model_space != 0) and
 * trimesh.repair.fix_inversion (:224-225, fix_inversion != 0: faces are flipped when the signed volume is negative;
 * *inverted reports it).
 * The algorithm is the one scikit-image 0.18.3 runs: Lewiner et al. 2003 -- case / face-test / interior-test look-up
 * tables, tunnel tilings, centre vertex --, "inside" = value > 0, vertices interpolated with the weights
 * 1 / (eps + |value|), every face reversed (gradient_direction='descent').  The mesh equals scikit-image's vertex
 * position for vertex position and triangle for triangle (pinned on the reference's volumes); only the
 * ORDER of vertices and faces is this library's (grid points / cells in C order).
 * vol_dev [res]^3 float32, C order; verts_out_dev [cap_verts][3] float32 (array-index coordinates, or model space),
 * faces_out_dev [cap_faces][3] int32.  *n_verts / *n_faces always receive the full counts; P2S_ECAPACITY if a capacity
 * is too small (call with capacities 0 to size the buffers).  Synchronises `stream`.
 * ------------------------------------------------------------------------------------------ */
int p2s_marching_cubes(const float *vol_dev, int grid_res, float *verts_out_dev, int64_t cap_verts,
                       int32_t *faces_out_dev, int64_t cap_faces, int64_t *n_verts, int64_t *n_faces,
                       int model_space, int fix_inversion, int *inverted, int device, void *stream);
/* single-cube diagnostic, host arithmetic (the decision code the kernels run): values8 = the cube's values minus the
 * level in Lewiner's corner order (0 (0,0,0) 1 (1,0,0) 2 (1,1,0) 3 (0,1,0) 4 (0,0,1) 5 (1,0,1) 6 (1,1,1) 7 (0,1,1) as
 * x, y, z).  *row = the selected row of the tiling table (-1: none), edges36 = its cube edge ids, 3 per triangle,
 * 12 = the centre vertex, -1 padded */
int p2s_mc_cell(const float *values8, int32_t *row, int32_t *n_tri, int32_t *edges36);

/* ------------------------------------------------------------------------------------------
 * "next" row: mesh metrics (reference source/base/evaluation.py:222-305): even surface sampling
 * (trimesh.sample.sample_surface_even), directed Hausdorff and Chamfer distances between the sample sets.
 * ------------------------------------------------------------------------------------------ */
/* RandomState.random_sample(n): n float64 in [0, 1) from the stream of r (2 words each); needs the jump tables */
int p2s_rng_random_sample(p2s_rng_t r, int64_t n, double *out_dev, void *stream);
/* trimesh.sample.sample_surface: u_dev [3 n] float64 uniform deviates laid out as the reference draws them (n face
 * picks, then n x 2 lengths); pts_out_dev [n][3] float32, face_out_dev [n] (may be NULL); *area_host = mesh area */
int p2s_mesh_sample_surface(const float *verts_dev, const int32_t *faces_dev, int64_t n_faces, const double *u_dev,
                            int64_t n_samples, float *pts_out_dev, int32_t *face_out_dev, double *area_host,
                            int device, void *stream);
/* trimesh.points.remove_close + the [:count] cut of sample_surface_even: of every pair of points within `radius` the
 * one with more close neighbours is dropped (ties: the first); the survivors, in order, at most `limit` */
int p2s_points_remove_close(const float *pts_dev, int64_t m, double radius, int64_t limit, float *pts_out_dev,
                            int64_t *n_out, int device, void *stream);
/* nearest-neighbour distance (float64, exact) of every query point to the cloud `target`; *max_host = directed
 * Hausdorff distance, *sum_host = the Chamfer term; dist_out_dev [n] float64 may be NULL.  Synchronises. */
int p2s_nn_distance_stats(p2s_cloud_t target, const float *query_dev, int64_t n, double *dist_out_dev, double *max_host,
                          double *sum_host, void *stream);

/* ------------------------------------------------------------------------------------------
 * "next" row: the per-shape text / debug files of save_evaluation and implicit_surface_to_mesh, written
 * by native HOST code (no device is touched; all pointers are host pointers).  Byte-identical to what the reference's
 * numpy / Python calls write.
 * p2s_write_txt_f32:       np.savetxt(path, sdf) (reference source/points_to_surf_eval.py:210): '%.18e' per line.
 * p2s_write_query_vis_ply: sdf.visualize_query_points (source/sdf.py:269-285) in the drop-in's PLY layout
 *                          (points2surf_amd/ply.py): float32 xyz + uchar rgba per query point.
 * p2s_write_coff_samples:  mesh_io.write_off(file, query_pts_ms, [], colors_vertex=...) with the colours of
 *                          source/sdf.py:203-209 (source/base/mesh_io.py:75-140): str() of every number.
 * A file that cannot be opened, written, flushed or closed: P2S_EIO with errno's text (never a silently truncated file).
 * ------------------------------------------------------------------------------------------ */
int p2s_write_txt_f32(const char *path, const float *values_host, int64_t n);
int p2s_write_query_vis_ply(const char *path, const float *query_host, const float *dist_host, int64_t n);
int p2s_write_coff_samples(const char *path, const float *query_host, const float *dist_host, int64_t n);

/* per-stage counters of the last p2s_encode_decode / p2s_infer_shape on this model
 * (HIP-event milliseconds on the launch stream; valid after the stream is synchronised) */
typedef struct {
    double  ms_chain_stn, ms_stn_head, ms_chain_main, ms_decoder, ms_knn, ms_subsample, ms_grid;
    int64_t queries;
    int64_t launches_chain;      /* number of point-chain kernel launches (2 per chunk; 3 with a QSTN) */
    double  ms_chain_qstn;       /* QSTN trunk launch (models with use_point_stn); its head layers count under ms_stn_head */
    int64_t fallback_queries;    /* fp16 pair encoder: queries of the call re-run through the fp32 kernels (activation > 6e4) */
    double  reserved[6];
} p2s_counters;
int p2s_set_profiling(p2s_model_t m, int enabled);
int p2s_get_counters(p2s_model_t m, p2s_counters *out);

#ifdef __cplusplus
}
#endif
#endif /* P2S_HIP_H */
