// Data path of the SDF inference on the device (gfx950): everything the reference does on CPU worker
// processes per query (source/data_loader.py:322-421) runs here from the HBM-resident cloud.
//
//   a1  query grid        source/sdf.py:46-79           voxel bitmask -> box dilation -> ordered compaction
//   a2  cell index        source/data_loader.py:40-42   uniform cell grid + 3-D summed-area table
//                                                       (replaces cKDTree(leaf_size=1000))
//   a4  kNN (fp64 rank)   source/base/point_cloud.py:170-175
//   a5  radius / patch    source/base/utils.py:62-69,80-88; source/data_loader.py:341-350 (fp32, no FMA)
//   a6  uniform subsample source/base/utils.py:196-227 + numpy legacy MT19937 masked rejection
//
// These stages are integer / select / gather work bound by L2 + LDS latency (the cloud, <= 1.8 MB, is
// L2/MALL resident); they are deliberately NOT reshaped into GEMMs.
#include "p2s_common.h"
#include "p2s_internal.h"
#include <vector>
#include <cstdlib>
#include <mutex>

// a5 must reproduce numpy's fp32 results bit for bit: no FMA contraction anywhere in this file, and
// sqrtf / operator/ (correctly rounded under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt);
// NOT __fsqrt_rn, which HIP maps to the approximate native sqrt.
#pragma clang fp contract(off)

#include <cmath>
#include <cstring>
#include <algorithm>

namespace {

__host__ __device__ inline int cell_coord(float x, float lo, float inv, int G) {
    // monotone non-decreasing in x (needed for the conservative range computation below)
    float f = floorf((x - lo) * inv);
    int c = (f < 0.f) ? 0 : (f > (float)(G - 1) ? G - 1 : (int)f);
    return c;
}

#include "p2s_cloud_grid.inl"
#include "p2s_cloud_knn.inl"
#include "p2s_cloud_serial_rng.inl"
#include "p2s_cloud_index.inl"

// ---------------------------------------------------------------------------------------------
// Device-memory cache of the cloud handles.  The drop-in creates one cloud handle per shape: hipMalloc / hipFree per
// shape are blocking calls (and hipFree drains the device).  Freed blocks are kept per device and handed to the next
// handle; sizes are rounded to 1/8 octave so that clouds of similar size reuse each other's blocks.
// ---------------------------------------------------------------------------------------------
struct PoolBlock {
    void *p;
    size_t bytes;
};
struct DevPool {
    std::mutex mu;
    std::vector<PoolBlock> free_list, used;
    size_t cached = 0;
};
DevPool g_pool[P2S_MAX_DEVICES];
constexpr size_t POOL_MAX_CACHED = (size_t)4 << 30;      // per device; beyond that freed blocks go back to HIP

size_t pool_round(size_t bytes) {
    bytes = std::max<size_t>(bytes, 256);
    size_t p2 = 256;
    while (p2 < bytes) p2 <<= 1;          // smallest power of two >= bytes
    const size_t step = p2 / 16;          // 1/8 octave of the octave below
    return (bytes + step - 1) / step * step;
}

}  // namespace

void *p2s_pool_alloc(int device, size_t bytes) {
    if (device < 0 || device >= P2S_MAX_DEVICES) return nullptr;
    const size_t want = pool_round(bytes);
    DevPool &pl = g_pool[device];
    std::lock_guard<std::mutex> g(pl.mu);
    int best = -1;
    for (int i = 0; i < (int)pl.free_list.size(); ++i)
        if (pl.free_list[i].bytes >= want && pl.free_list[i].bytes <= 2 * want &&
            (best < 0 || pl.free_list[i].bytes < pl.free_list[best].bytes))
            best = i;
    PoolBlock b;
    if (best >= 0) {
        b = pl.free_list[best];
        pl.free_list.erase(pl.free_list.begin() + best);
        pl.cached -= b.bytes;
    } else {
        b.p = nullptr;
        b.bytes = want;
        if (hipMalloc(&b.p, want) != hipSuccess) {
            (void)hipGetLastError();
            // give the cache back to HIP and retry once
            for (auto &f : pl.free_list) (void)hipFree(f.p);
            pl.free_list.clear();
            pl.cached = 0;
            if (hipMalloc(&b.p, want) != hipSuccess) {
                (void)hipGetLastError();
                return nullptr;
            }
        }
    }
    pl.used.push_back(b);
    return b.p;
}

// the caller guarantees that no work touching the block is in flight
void p2s_pool_free(int device, void *p) {
    if (!p || device < 0 || device >= P2S_MAX_DEVICES) return;
    DevPool &pl = g_pool[device];
    std::lock_guard<std::mutex> g(pl.mu);
    for (size_t i = 0; i < pl.used.size(); ++i) {
        if (pl.used[i].p != p) continue;
        const PoolBlock b = pl.used[i];
        pl.used.erase(pl.used.begin() + i);
        if (pl.cached + b.bytes > POOL_MAX_CACHED) {
            (void)hipFree(b.p);
        } else {
            pl.free_list.push_back(b);
            pl.cached += b.bytes;
        }
        return;
    }
    (void)hipFree(p);       // not ours (cannot happen): do not leak
}

void p2s_cloud_pool_release(int device) {
    if (device < 0 || device >= P2S_MAX_DEVICES) return;
    DevPool &pl = g_pool[device];
    std::lock_guard<std::mutex> g(pl.mu);
    for (auto &f : pl.free_list) (void)hipFree(f.p);
    pl.free_list.clear();
    pl.cached = 0;
}

void p2s_cloud_note_stream(p2s_cloud_s *c, hipStream_t s) {
    if (c->foreign_streams_quiet > 0) return;      // inside run_pipeline: its streams are drained before it returns
    for (int i = 0; i < c->n_streams; ++i)
        if (c->streams[i] == s) return;
    if (c->n_streams < 4) c->streams[c->n_streams++] = s;
    else c->many_streams = true;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
extern "C" {

int p2s_cloud_create(const float *pts_dev, int n, int device, void *stream, p2s_cloud_t *out) {
    if (!pts_dev || n < 1 || !out) {
        p2s_set_error("p2s_cloud_create: bad argument (n=%d)", n);
        return P2S_EINVAL;
    }
    if (p2s_device_count() <= device || device < 0 || device >= P2S_MAX_DEVICES) {
        p2s_set_error("p2s_cloud_create: no HIP device %d", device);
        return P2S_ENODEVICE;
    }
    P2S_HIP_CHECK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    // uniform cell grid over the bounding box; ~10 points per occupied cell for surface-like clouds
    int G = (int)std::ceil(std::sqrt((double)n / 32.0));
    G = std::max(4, std::min(G, 128));
    const size_t ncell = (size_t)G * G * G;
    const int G1 = G + 1;
    const size_t nsat = (size_t)G1 * G1 * G1;
    // one arena: pts | spts | tmp | cid | cnt / fill | cell_start | sat | bbox record (256-byte aligned pieces)
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_pts = 0, o_spts = o_pts + al((size_t)n * 12), o_tmp = o_spts + al((size_t)n * 16),
                 o_cid = o_tmp + al((size_t)n * 16), o_cnt = o_cid + al((size_t)n * 4), o_start = o_cnt + al((ncell + 1) * 4),
                 o_sat = o_start + al((ncell + 1) * 4), o_rec = o_sat + al(nsat * 4), o_tot = o_rec + 256, total = o_tot + 256;
    p2s_cloud_s *c = new p2s_cloud_s();
    c->device = device;
    p2s_cloud_note_stream(c, s);
    c->arena = (char *)p2s_pool_alloc(device, total);
    if (!c->arena) {
        p2s_set_error("p2s_cloud_create: device allocation of %zu bytes failed", total);
        delete c;
        return P2S_ENOMEM;
    }
    auto fail = [&](int code) {
        (void)hipStreamSynchronize(s);
        p2s_cloud_destroy(c);
        return code;
    };
#define CLOUD_HIP(expr)                                                                                    \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess) {                                                                            \
            p2s_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);      \
            return fail(P2S_EHIP);                                                                         \
        }                                                                                                  \
    } while (0)
    c->pts = (float *)(c->arena + o_pts);
    c->spts = (float4 *)(c->arena + o_spts);
    float4 *tmp = (float4 *)(c->arena + o_tmp);
    int *cid = (int *)(c->arena + o_cid);
    int *cnt = (int *)(c->arena + o_cnt);
    c->cell_start = (int *)(c->arena + o_start);
    c->sat = (int *)(c->arena + o_sat);
    uint32_t *rec = (uint32_t *)(c->arena + o_rec);
    c->totals = (long long *)(c->arena + o_tot);
    // the handle owns a copy of the points (the caller's tensor may go away)
    CLOUD_HIP(hipMemcpyAsync(c->pts, pts_dev, (size_t)n * 12, hipMemcpyDeviceToDevice, s));
    const uint32_t rec0[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0xffffffffu, 0u};
    CLOUD_HIP(hipMemcpyAsync(rec, rec0, sizeof(rec0), hipMemcpyHostToDevice, s));
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(p2s_bbox_kernel, dim3(std::min(nb, 1024u)), dim3(256), 0, s, c->pts, n, rec);
    CLOUD_HIP(hipGetLastError());
    // scratch that does not depend on the box is cleared while the record travels
    CLOUD_HIP(hipMemsetAsync(cnt, 0, (ncell + 1) * 4, s));
    CLOUD_HIP(hipMemsetAsync(c->sat, 0, nsat * 4, s));
    uint32_t h[8];
    CLOUD_HIP(hipMemcpyAsync(h, rec, sizeof(h), hipMemcpyDeviceToHost, s));
    CLOUD_HIP(hipStreamSynchronize(s));            // the one blocking call: bounding box + non-finite check
    if (h[6] != 0xffffffffu) {
        p2s_set_error("p2s_cloud_create: non-finite coordinate at point %u", h[6]);
        return fail(P2S_EINVAL);
    }
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = f32_from_ordered(h[a]);
        hi[a] = f32_from_ordered(h[3 + a]);
    }
    float ext = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
    if (!(ext > 0.f)) ext = 1.0f;
    const float cell = ext / (float)G * 1.0001f;
    const float inv = 1.0f / cell;
    CellGeom g;
    for (int a = 0; a < 3; ++a) g.lo[a] = lo[a];
    g.inv = inv;
    g.G = G;
    hipLaunchKernelGGL(p2s_cell_hist_kernel, dim3(nb), dim3(256), 0, s, c->pts, n, g, cid, cnt);
    hipLaunchKernelGGL(p2s_sat_z_kernel, dim3((unsigned)((G * G + 3) / 4)), dim3(256), 0, s, cnt, G, c->sat);
    hipLaunchKernelGGL(p2s_sat_axis_kernel, dim3((unsigned)((G * G + 255) / 256)), dim3(256), 0, s, c->sat, G, 1);
    hipLaunchKernelGGL(p2s_sat_axis_kernel, dim3((unsigned)((G * G + 255) / 256)), dim3(256), 0, s, c->sat, G, 0);
    hipLaunchKernelGGL(p2s_cell_start_kernel, dim3((unsigned)((ncell + 1 + 255) / 256)), dim3(256), 0, s, c->sat, G, n,
                       c->cell_start, cnt);
    hipLaunchKernelGGL(p2s_cell_scatter_kernel, dim3(nb), dim3(256), 0, s, c->pts, cid, n, cnt, tmp);
    hipLaunchKernelGGL(p2s_cell_rank_kernel, dim3(nb), dim3(256), 0, s, tmp, c->cell_start, n, g, c->spts);
    CLOUD_HIP(hipGetLastError());
#undef CLOUD_HIP
    c->d.pts = c->pts;
    c->d.spts = c->spts;
    c->d.cell_start = c->cell_start;
    c->d.sat = c->sat;
    for (int a = 0; a < 3; ++a) c->d.lo[a] = lo[a];
    c->d.inv_cell = inv;
    c->d.G = G;
    c->d.n = n;
    *out = c;
    return P2S_OK;
}

// Drain the streams a handle has noted.  Only the complaint about a caller's stream that no longer exists is dropped; a
// genuine asynchronous fault (of this or of unrelated work) is recorded and stays pending for the next launch check.
static void drain_noted_streams(p2s_cloud_s *c, const char *who) {
    hipError_t fault = hipSuccess;
    bool stale = false;
    auto look = [&](hipError_t e) {
        if (e == hipSuccess) return;
        if (e == hipErrorInvalidHandle || e == hipErrorInvalidResourceHandle || e == hipErrorContextIsDestroyed) stale = true;
        else fault = e;
    };
    if (c->many_streams) look(hipDeviceSynchronize());
    else
        for (int i = 0; i < c->n_streams; ++i) look(hipStreamSynchronize(c->streams[i]));
    if (fault != hipSuccess) p2s_set_error("%s: asynchronous HIP error while draining the handle's streams: %s", who, hipGetErrorString(fault));
    else if (stale) (void)hipGetLastError();
}

int p2s_cloud_destroy(p2s_cloud_t c) {
    if (!c) return P2S_OK;
    (void)hipSetDevice(c->device);
    // the blocks go back to the cache: nothing may still be reading them
    drain_noted_streams(c, "p2s_cloud_destroy");
    if (c->grid_ev) (void)hipEventDestroy(c->grid_ev);
    p2s_pool_free(c->device, c->arena);
    p2s_pool_free(c->device, c->occ);
    p2s_pool_free(c->device, c->blk_cnt);
    p2s_pool_free(c->device, c->qcache);
    p2s_pool_free(c->device, c->shuffle_perm);
    p2s_pool_free(c->device, c->wc_plan);
    p2s_pool_free(c->device, c->kd_blob);
    delete c;
    return P2S_OK;
}

/* test / diagnostic access to the index: any output may be NULL.  geom_host: lo[3], inv_cell; sizes from *G_host:
 * cell_start (G^3 + 1) int32, sat (G + 1)^3 int32, sorted n x (x, y, z, original id as int bits).  Synchronises. */
int p2s_cloud_index_export(p2s_cloud_t c, int32_t *G_host, float *geom_host, int32_t *cell_start_host, int32_t *sat_host,
                           float *sorted_host, void *stream) {
    if (!c) {
        p2s_set_error("p2s_cloud_index_export: null handle");
        return P2S_EINVAL;
    }
    P2S_HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    p2s_cloud_note_stream(c, s);
    const int G = c->d.G;
    if (G_host) *G_host = G;
    if (geom_host) {
        for (int a = 0; a < 3; ++a) geom_host[a] = c->d.lo[a];
        geom_host[3] = c->d.inv_cell;
    }
    if (cell_start_host)
        P2S_HIP_CHECK(hipMemcpyAsync(cell_start_host, c->cell_start, ((size_t)G * G * G + 1) * 4, hipMemcpyDeviceToHost, s));
    if (sat_host)
        P2S_HIP_CHECK(hipMemcpyAsync(sat_host, c->sat, (size_t)(G + 1) * (G + 1) * (G + 1) * 4, hipMemcpyDeviceToHost, s));
    if (sorted_host) P2S_HIP_CHECK(hipMemcpyAsync(sorted_host, c->spts, (size_t)c->d.n * 16, hipMemcpyDeviceToHost, s));
    P2S_HIP_CHECK(hipStreamSynchronize(s));
    return P2S_OK;
}

int p2s_cloud_num_points(p2s_cloud_t c) { return c ? c->d.n : P2S_EINVAL; }

int p2s_query_grid(p2s_cloud_t c, int res, int eps, float *q_out_dev, int64_t capacity, int64_t *n_queries,
                   void *stream) {
    if (!c || !n_queries || (capacity > 0 && !q_out_dev)) {
        p2s_set_error("p2s_query_grid: bad argument (res=%d eps=%d)", res, eps);
        return P2S_EINVAL;
    }
    const float *q = nullptr;
    long long n = 0;
    const int rc = p2s_cloud_grid(c, res, eps, &q, &n, (hipStream_t)stream);
    if (rc) return rc;
    *n_queries = n;
    if (capacity <= 0) return n > 0 ? P2S_ECAPACITY : P2S_OK;
    if (n > capacity) {
        p2s_set_error("p2s_query_grid: capacity %lld < %lld queries", (long long)capacity, n);
        return P2S_ECAPACITY;
    }
    if (n > 0)
        P2S_HIP_CHECK(hipMemcpyAsync(q_out_dev, q, (size_t)n * 12, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return P2S_OK;
}

}  // extern "C"

int p2s_cloud_grid(p2s_cloud_s *c, int res, int eps, const float **q_out, long long *n_out, hipStream_t s) {
    if (!c || res < 2 || res > 1024 || eps < 1 || eps > 16) {
        p2s_set_error("p2s_query_grid: bad argument (res=%d eps=%d)", res, eps);
        return P2S_EINVAL;
    }
    if (c->qc_n >= 0 && c->qc_res == res && c->qc_eps == eps) {
        // the compaction was queued on grid_stream and not waited for: other streams order themselves behind it
        if (c->grid_ev && s != c->grid_stream) P2S_HIP_CHECK(hipStreamWaitEvent(s, c->grid_ev, 0));
        p2s_cloud_note_stream(c, s);
        *q_out = c->qcache;
        *n_out = c->qc_n;
        return P2S_OK;
    }
    P2S_HIP_CHECK(hipSetDevice(c->device));
    c->qc_n = -1;
    const size_t vox = (size_t)res * res * res;
    const size_t words = (vox + 31) / 32;
    const long long rm = res - 1;
    const long long nblk = (rm * rm * rm + 255) / 256;
    p2s_cloud_note_stream(c, s);
    // scratch from the device's block cache (p2s_pool_alloc): no hipMalloc on the per-shape path once it is warm.
    // Replaced blocks may still be read by work queued on the handle's streams: drain them first (rare path)
    auto drain = [&]() { drain_noted_streams(c, "p2s_query_grid"); };
    if (words > c->occ_words) {
        if (c->occ) {
            drain();
            p2s_pool_free(c->device, c->occ);
        }
        c->occ_words = 0;
        c->occ = (uint32_t *)p2s_pool_alloc(c->device, words * 4);
        if (!c->occ) {
            p2s_set_error("p2s_query_grid: device allocation of %zu bytes failed", words * 4);
            return P2S_ENOMEM;
        }
        c->occ_words = words;
    }
    if ((size_t)nblk > c->blk_cap) {
        if (c->blk_cnt) {
            drain();
            p2s_pool_free(c->device, c->blk_cnt);
        }
        c->blk_cap = 0;
        // counts (int) followed by offsets (long long)
        c->blk_cnt = (int *)p2s_pool_alloc(c->device, (size_t)nblk * 4 + (size_t)nblk * 8 + 64);
        if (!c->blk_cnt) {
            p2s_set_error("p2s_query_grid: device allocation (block scan) failed");
            return P2S_ENOMEM;
        }
        c->blk_cap = (size_t)nblk;
    }
    long long *blk_off = reinterpret_cast<long long *>(reinterpret_cast<char *>(c->blk_cnt) +
                                                       ((c->blk_cap * 4 + 63) / 64) * 64);
    P2S_HIP_CHECK(hipMemsetAsync(c->occ, 0, words * 4, s));
    P2S_HIP_CHECK(hipMemsetAsync(c->totals, 0, 16, s));
    hipLaunchKernelGGL(p2s_voxelize_kernel, dim3((c->d.n + 255) / 256), dim3(256), 0, s, c->d.pts, c->d.n, res, c->occ,
                       c->totals);
    P2S_LAUNCH_CHECK("p2s_voxelize_kernel");
    GridOffsets go;
    go.n = eps;
    for (int j = 0; j < eps; ++j) go.o[j] = eps / 2 - j;   // scipy.ndimage.convolve, origin 0
    hipLaunchKernelGGL(p2s_grid_compact_kernel<0>, dim3((unsigned)nblk), dim3(256), 0, s, c->occ, res, go, c->blk_cnt,
                       blk_off, (float *)nullptr, 0LL);
    P2S_LAUNCH_CHECK("p2s_grid_compact_kernel<0>");
    hipLaunchKernelGGL(p2s_scan_blocks_kernel, dim3(1), dim3(1024), 0, s, c->blk_cnt, nblk, blk_off, c->totals);
    P2S_LAUNCH_CHECK("p2s_scan_blocks_kernel");
    long long host_tot[2] = {0, 0};
    P2S_HIP_CHECK(hipMemcpyAsync(host_tot, c->totals, 16, hipMemcpyDeviceToHost, s));
    P2S_HIP_CHECK(hipStreamSynchronize(s));
    if (host_tot[1]) {
        p2s_set_error("p2s_query_grid: point outside the [-1,1) volume (IndexError in the reference)");
        return P2S_EINVAL;
    }
    const long long n = host_tot[0];
    if (n > c->qc_cap) {
        if (c->qcache) {
            drain();
            p2s_pool_free(c->device, c->qcache);
        }
        c->qc_cap = 0;
        c->qcache = (float *)p2s_pool_alloc(c->device, (size_t)n * 12);
        if (!c->qcache) {
            p2s_set_error("p2s_query_grid: device allocation of %lld query points failed", n);
            return P2S_ENOMEM;
        }
        c->qc_cap = n;
    }
    if (n > 0) {
        // stream-ordered on `s`; consumers on other streams order themselves behind it with an event (run_pipeline)
        hipLaunchKernelGGL(p2s_grid_compact_kernel<1>, dim3((unsigned)nblk), dim3(256), 0, s, c->occ, res, go, c->blk_cnt,
                           blk_off, c->qcache, n);
        P2S_LAUNCH_CHECK("p2s_grid_compact_kernel<1>");
    }
    if (!c->grid_ev) P2S_HIP_CHECK(hipEventCreateWithFlags(&c->grid_ev, hipEventDisableTiming));
    P2S_HIP_CHECK(hipEventRecord(c->grid_ev, s));
    c->grid_stream = s;
    c->qc_res = res;
    c->qc_eps = eps;
    c->qc_n = n;
    *q_out = c->qcache;
    *n_out = n;
    return P2S_OK;
}

extern "C" {

int p2s_knn_patch(p2s_cloud_t c, const float *query_dev, int64_t nq, int k, int32_t *ids_out_dev,
                  float *patch_ps_out_dev, float *radius_out_dev, void *stream) {
    if (!c || !query_dev || nq < 0 || k < 1) {
        p2s_set_error("p2s_knn_patch: bad argument");
        return P2S_EINVAL;
    }
    if (k > c->d.n) {
        // the reference fails too: cKDTree returns index N -> IndexError
        p2s_set_error("p2s_knn_patch: cloud has %d points < k=%d", c->d.n, k);
        return P2S_EINVAL;
    }
    if (k > KNN_CAP - 128) {
        p2s_set_error("p2s_knn_patch: k=%d exceeds the supported maximum %d", k, KNN_CAP - 128);
        return P2S_EINVAL;
    }
    if (nq == 0) return P2S_OK;
    P2S_HIP_CHECK(hipSetDevice(c->device));
    p2s_cloud_note_stream(c, (hipStream_t)stream);
    const unsigned grid = (unsigned)std::min<int64_t>(nq, 256 * 64);
    hipLaunchKernelGGL(p2s_knn_kernel<true>, dim3(grid), dim3(64), 0, (hipStream_t)stream, c->d, query_dev, (long long)nq, k,
                       ids_out_dev, patch_ps_out_dev, radius_out_dev);
    P2S_LAUNCH_CHECK("p2s_knn_kernel");
    return P2S_OK;
}

// the pipeline's variant: the same k nearest points, patch rows in arbitrary (deterministic) order, no ids
int p2s_knn_patch_set(p2s_cloud_t c, const float *query_dev, int64_t nq, int k, float *patch_ps_out_dev,
                      float *radius_out_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!c || !query_dev || nq < 0 || k < 1 || k > c->d.n || k > KNN_CAP - 128) {
        p2s_set_error("p2s_knn_patch_set: bad argument (k=%d)", k);
        return P2S_EINVAL;
    }
    if (nq == 0) return P2S_OK;
    P2S_HIP_CHECK(hipSetDevice(c->device));
    p2s_cloud_note_stream(c, stream);
    const unsigned grid = (unsigned)std::min<int64_t>(nq, 256 * 64);
    hipLaunchKernelGGL(p2s_knn_kernel<false>, dim3(grid), dim3(64), 0, stream, c->d, query_dev, (long long)nq, k,
                       (int *)nullptr, patch_ps_out_dev, radius_out_dev);
    P2S_LAUNCH_CHECK("p2s_knn_kernel");
    return P2S_OK;
}

int p2s_gather_points(p2s_cloud_t c, const int32_t *ids_dev, int64_t n_ids, float *pts_out_dev, void *stream) {
    if (!c || !ids_dev || !pts_out_dev || n_ids < 0) {
        p2s_set_error("p2s_gather_points: bad argument");
        return P2S_EINVAL;
    }
    if (n_ids == 0) return P2S_OK;
    P2S_HIP_CHECK(hipSetDevice(c->device));
    p2s_cloud_note_stream(c, (hipStream_t)stream);
    hipLaunchKernelGGL(p2s_gather_kernel, dim3((unsigned)((n_ids + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       c->d.pts, ids_dev, (long long)n_ids, c->d.n, pts_out_dev);
    P2S_LAUNCH_CHECK("p2s_gather_kernel");
    return P2S_OK;
}

// ---- RNG -----------------------------------------------------------------------------------------
}  // extern "C"

// init_genrand(seed): numpy legacy seeding with an integer seed -> mt[624] + position 624
void p2s_mt_seed_host(uint32_t seed, uint32_t st[625]) {
    st[0] = seed;
    for (int i = 1; i < 624; ++i) st[i] = 1812433253u * (st[i - 1] ^ (st[i - 1] >> 30)) + (uint32_t)i;
    st[624] = 624;
}

int p2s_rng_reseed(p2s_rng_s *r, uint32_t seed, hipStream_t s) {
    int rc = p2s_rng_session_close(r, s);
    if (rc) return rc;
    uint32_t st[625];
    p2s_mt_seed_host(seed, st);
    P2S_HIP_CHECK(hipMemcpyAsync(r->state, st, sizeof(st), hipMemcpyHostToDevice, s));
    P2S_HIP_CHECK(hipStreamSynchronize(s));        // st lives on this stack frame
    return P2S_OK;
}

namespace {
__global__ void p2s_replicate_rows_kernel(int32_t *__restrict__ ids, int n, long long rows) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (rows - 1) * n) ids[n + i] = ids[i % n];
}
}  // namespace

int p2s_rng_serial_randint(p2s_rng_s *r, uint32_t rng, uint32_t mask, long long target, int32_t *out, hipStream_t s) {
    // The recurrence is serial (one workgroup) and pure latency: co-resident MFMA-saturated encoder
    // workgroups slow it ~3x.  Requesting most of a CU's LDS keeps any 50 KB encoder workgroup off its CU
    // (the workgroup is placed when CUs drain at an encoder-kernel boundary); cost: 1 of 256 CUs.
    constexpr int hog = 120 * 1024;
    {   // per device (a process may drive several); the call is cheap
        (void)hipFuncSetAttribute((const void *)p2s_mt_randint_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    }
    // tiny requests (tests, tails) do not need a CU of their own
    const int lds = target >= 100000 ? hog : 0;
    hipLaunchKernelGGL(p2s_mt_randint_kernel, dim3(1), dim3(128), lds, s, r->state, rng, mask, target, out);
    P2S_LAUNCH_CHECK("p2s_mt_randint_kernel");
    return P2S_OK;
}

extern "C" {

int p2s_rng_create(uint32_t seed, int device, p2s_rng_t *out) {
    if (!out) return P2S_EINVAL;
    if (p2s_device_count() <= device || device < 0) {
        p2s_set_error("p2s_rng_create: no HIP device %d", device);
        return P2S_ENODEVICE;
    }
    P2S_HIP_CHECK(hipSetDevice(device));
    uint32_t st[625];
    p2s_mt_seed_host(seed, st);
    p2s_rng_s *r = new p2s_rng_s();
    r->device = device;
    if (hipMalloc(&r->state, sizeof(st)) != hipSuccess) {
        delete r;
        p2s_set_error("p2s_rng_create: hipMalloc failed");
        return P2S_ENOMEM;
    }
    if (hipMemcpy(r->state, st, sizeof(st), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(r->state);
        delete r;
        p2s_set_error("p2s_rng_create: hipMemcpy failed");
        return P2S_EHIP;
    }
    *out = r;
    return P2S_OK;
}

int p2s_rng_destroy(p2s_rng_t r) {
    if (!r) return P2S_OK;
    (void)hipSetDevice(r->device);
    if (r->state) (void)hipFree(r->state);
    if (r->jump_sup) (void)hipFree(r->jump_sup);
    if (r->streams) (void)hipFree(r->streams);
    if (r->tmp) (void)hipFree(r->tmp);
    if (r->blk_cum) (void)hipFree(r->blk_cum);
    if (r->meta) (void)hipFree(r->meta);
    p2s_wc_free_rng(r);
    p2s_ball_free_rng(r);
    delete r;
    return P2S_OK;
}

int p2s_rng_get_state(p2s_rng_t r, uint32_t *mt624_host, int32_t *pos_host, void *stream) {
    if (!r || !mt624_host || !pos_host) return P2S_EINVAL;
    P2S_HIP_CHECK(hipSetDevice(r->device));
    {
        const int rc = p2s_rng_session_close(r, (hipStream_t)stream);    // the state lags behind an open session
        if (rc) return rc;
    }
    uint32_t st[625];
    P2S_HIP_CHECK(hipMemcpyAsync(st, r->state, sizeof(st), hipMemcpyDeviceToHost, (hipStream_t)stream));
    P2S_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    memcpy(mt624_host, st, 624 * 4);
    *pos_host = (int32_t)st[624];
    return P2S_OK;
}

int p2s_rng_set_state(p2s_rng_t r, const uint32_t *mt624_host, int32_t pos, void *stream) {
    if (!r || !mt624_host || pos < 0 || pos > 624) return P2S_EINVAL;
    P2S_HIP_CHECK(hipSetDevice(r->device));
    {
        const int rc = p2s_rng_session_close(r, (hipStream_t)stream);
        if (rc) return rc;
    }
    uint32_t st[625];
    memcpy(st, mt624_host, 624 * 4);
    st[624] = (uint32_t)pos;
    P2S_HIP_CHECK(hipMemcpyAsync(r->state, st, sizeof(st), hipMemcpyHostToDevice, (hipStream_t)stream));
    P2S_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return P2S_OK;
}

int p2s_subsample_shuffle_pad(p2s_rng_t r, p2s_cloud_t c, int64_t nq, int n, int32_t *perm_before_dev, int32_t *ids_out_dev,
                              void *stream) {
    if (!r || !c || nq < 0 || n < 1) {
        p2s_set_error("p2s_subsample_shuffle_pad: bad argument");
        return P2S_EINVAL;
    }
    if (c->d.n >= n || c->d.n > 1024) {
        p2s_set_error("p2s_subsample_shuffle_pad: only for clouds with fewer points (%d) than the sub-sample size (%d <= 1024)",
                      c->d.n, n);
        return P2S_EINVAL;
    }
    if (nq == 0) return P2S_OK;
    P2S_HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    p2s_cloud_note_stream(c, s);
    int rc = p2s_rng_session_close(r, s);
    if (rc) return rc;
    if (!c->shuffle_perm) {
        std::vector<int> id((size_t)c->d.n);
        for (int i = 0; i < c->d.n; ++i) id[i] = i;
        c->shuffle_perm = (int *)p2s_pool_alloc(c->device, (size_t)c->d.n * 4);
        if (!c->shuffle_perm) {
            p2s_set_error("p2s_subsample_shuffle_pad: device allocation failed");
            return P2S_ENOMEM;
        }
        P2S_HIP_CHECK(hipMemcpy(c->shuffle_perm, id.data(), id.size() * 4, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(p2s_shuffle_pad_kernel, dim3(1), dim3(64), 0, s, r->state, c->shuffle_perm, c->d.n, (long long)nq, n,
                       perm_before_dev, ids_out_dev);
    P2S_LAUNCH_CHECK("p2s_shuffle_pad_kernel");
    return P2S_OK;
}

int p2s_patch_from_ids(p2s_cloud_t c, const int32_t *ids_dev, const int32_t *perm_before_dev, const float *query_dev,
                       int64_t nq, int k, float *patch_ps_out_dev, float *radius_out_dev, void *stream) {
    if (!c || !ids_dev || !query_dev || nq < 0 || k < 1) {
        p2s_set_error("p2s_patch_from_ids: bad argument");
        return P2S_EINVAL;
    }
    if (nq == 0) return P2S_OK;
    P2S_HIP_CHECK(hipSetDevice(c->device));
    p2s_cloud_note_stream(c, (hipStream_t)stream);
    const unsigned grid = (unsigned)std::min<int64_t>(nq, 256 * 64);
    hipLaunchKernelGGL(p2s_patch_from_ids_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, c->d.pts, ids_dev,
                       perm_before_dev, c->d.n, query_dev, (long long)nq, k, patch_ps_out_dev, radius_out_dev);
    P2S_LAUNCH_CHECK("p2s_patch_from_ids_kernel");
    return P2S_OK;
}

int p2s_subsample_fixed(p2s_rng_t r, p2s_cloud_t c, const float *q_dev, int64_t nq, int n, uint32_t seed,
                        int32_t *ids_out_dev, float *pts_out_dev, void *stream) {
    if (!r || !c || nq < 0 || n < 1 || !ids_out_dev) {
        p2s_set_error("p2s_subsample_fixed: bad argument (ids_out_dev is required)");
        return P2S_EINVAL;
    }
    if (c->d.n < n) {        // the reseed sits inside the N >= n branch of the reference: small clouds shuffle + pad
        const int rc = p2s_subsample_shuffle_pad(r, c, nq, n, nullptr, ids_out_dev, stream);
        if (rc || !pts_out_dev) return rc;
        return p2s_gather_points(c, ids_out_dev, (int64_t)nq * n, pts_out_dev, stream);
    }
    if (q_dev) return p2s_wc_subsample_fixed(r, c, q_dev, nq, n, seed, ids_out_dev, pts_out_dev, (hipStream_t)stream);
    if (nq == 0) return P2S_OK;
    P2S_HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    // every query draws the same n values from the freshly seeded generator: draw them once, copy the row
    int rc = p2s_rng_reseed(r, seed, s);
    if (rc) return rc;
    const uint32_t rng = (uint32_t)(c->d.n - 1);
    if (rng == 0) {
        P2S_HIP_CHECK(hipMemsetAsync(ids_out_dev, 0, (size_t)nq * n * 4, s));
    } else {
        uint32_t mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        if ((rc = p2s_rng_serial_randint(r, rng, mask, n, ids_out_dev, s))) return rc;
        if (nq > 1) {
            const long long tot = (long long)(nq - 1) * n;
            hipLaunchKernelGGL(p2s_replicate_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, ids_out_dev, n,
                               (long long)nq);
            P2S_LAUNCH_CHECK("p2s_replicate_rows_kernel");
        }
    }
    if (pts_out_dev) return p2s_gather_points(c, ids_out_dev, (int64_t)nq * n, pts_out_dev, stream);
    return P2S_OK;
}

int p2s_subsample_uniform(p2s_rng_t r, p2s_cloud_t c, int64_t nq, int n, int32_t *ids_out_dev, float *pts_out_dev,
                          void *stream) {
    if (!r || !c || nq < 0 || n < 1 || (!ids_out_dev && pts_out_dev)) {
        p2s_set_error("p2s_subsample_uniform: bad argument (pts_out_dev needs ids_out_dev)");
        return P2S_EINVAL;
    }
    if (c->d.n < n) {        // reference source/base/utils.py:221-226: shuffle (in place) + zero padding
        const int rc = p2s_subsample_shuffle_pad(r, c, nq, n, nullptr, ids_out_dev, stream);
        if (rc || !pts_out_dev) return rc;
        return p2s_gather_points(c, ids_out_dev, (int64_t)nq * n, pts_out_dev, stream);
    }
    if (nq == 0) return P2S_OK;
    P2S_HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const uint32_t rng = (uint32_t)(c->d.n - 1);
    const long long target = (long long)nq * n;
    if (rng == 0) {
        if (ids_out_dev) P2S_HIP_CHECK(hipMemsetAsync(ids_out_dev, 0, (size_t)target * 4, s));   // numpy consumes no randomness
    } else {
        uint32_t mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        constexpr long long par_min = 400000;
        // large requests: values come from a session (2^levels_max jump-ahead streams generated once, many calls
        // take consecutive ranges); small ones outside a matching session: the serial kernel
        const bool in_session = r->sess_mode == 1 && r->sess_rng == rng && r->sess_mask == mask;
        int rc;
        if (r->levels_max > 0 && (in_session || target >= par_min)) {
            rc = p2s_rng_session_randint(r, rng, mask, target, ids_out_dev, s);
        } else {
            rc = p2s_rng_session_close(r, s);
            if (rc) return rc;
            rc = p2s_rng_serial_randint(r, rng, mask, target, ids_out_dev, s);
        }
        if (rc) return rc;
    }
    if (pts_out_dev) return p2s_gather_points(c, ids_out_dev, target, pts_out_dev, stream);
    return P2S_OK;
}

}  // extern "C"
