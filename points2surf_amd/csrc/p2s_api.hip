// C-ABI glue: error reporting, model handle, workspace, the encoder/decoder pipeline.
#include "p2s_common.h"
#include "p2s_internal.h"
#include <vector>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <mutex>

static thread_local char g_err[512] = "";

void p2s_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Process-wide grow-only scratch buffer per device (volume / iso-surface stages).  The caller holds the device's
// scratch lock for its whole call (P2sScratchLock), so concurrent host threads serialise on it instead of each keeping
// ~2 GB of HBM alive until the process exits; p2s_release_scratch() gives the memory back.
namespace {
struct DevScratch {
    std::mutex mu;
    void *p = nullptr;
    size_t cap = 0;
};
DevScratch g_scratch[P2S_MAX_DEVICES];
}  // namespace

P2sScratchLock::P2sScratchLock(int device) : dev_(device) {
    if (dev_ >= 0 && dev_ < P2S_MAX_DEVICES) g_scratch[dev_].mu.lock();
}
P2sScratchLock::~P2sScratchLock() {
    if (dev_ >= 0 && dev_ < P2S_MAX_DEVICES) g_scratch[dev_].mu.unlock();
}
void *P2sScratchLock::get(size_t bytes) {
    if (dev_ < 0 || dev_ >= P2S_MAX_DEVICES) return nullptr;
    DevScratch &sl = g_scratch[dev_];
    if (bytes <= sl.cap) return sl.p;
    if (sl.p) (void)hipFree(sl.p);
    sl.p = nullptr;
    sl.cap = 0;
    const size_t want = bytes + bytes / 8;
    if (hipMalloc(&sl.p, want) != hipSuccess) {
        (void)hipGetLastError();
        sl.p = nullptr;
        return nullptr;
    }
    sl.cap = want;
    return sl.p;
}

extern "C" {

int p2s_abi_version(void) { return P2S_ABI_VERSION; }
const char *p2s_last_error(void) { return g_err; }

int p2s_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int p2s_release_scratch(int device) {
    if (device < 0 || device >= P2S_MAX_DEVICES || device >= p2s_device_count()) {
        p2s_set_error("p2s_release_scratch: no HIP device %d", device);
        return P2S_ENODEVICE;
    }
    P2S_HIP_CHECK(hipSetDevice(device));
    {
        std::lock_guard<std::mutex> g(g_scratch[device].mu);     // waits for a running volume / iso-surface call
        if (g_scratch[device].p) (void)hipFree(g_scratch[device].p);
        g_scratch[device].p = nullptr;
        g_scratch[device].cap = 0;
    }
    p2s_cloud_pool_release(device);
    return P2S_OK;
}

int p2s_model_create(const p2s_model_cfg *cfg, const float *blob_host, size_t n_floats,
                     const p2s_weight_offsets *offs, int device, p2s_model_t *out) {
    if (!cfg || !blob_host || !offs || !out || n_floats == 0) {
        p2s_set_error("p2s_model_create: null argument");
        return P2S_EINVAL;
    }
    if (cfg->net_size != 1024 || (cfg->output_dim != 2 && cfg->output_dim != 1) || cfg->points_per_patch < 1 || cfg->sub_sample_size < 1) {
        p2s_set_error("p2s_model_create: unsupported cfg (net_size=%d output_dim=%d)", cfg->net_size, cfg->output_dim);
        return P2S_EINVAL;
    }
    if (p2s_device_count() <= device || device < 0) {
        p2s_set_error("p2s_model_create: no HIP device %d", device);
        return P2S_ENODEVICE;
    }
    P2S_HIP_CHECK(hipSetDevice(device));
    p2s_model_s *m = new p2s_model_s();
    m->cfg = *cfg;
    m->offs = *offs;
    m->device = device;
    m->n_floats = n_floats;
    hipError_t e = hipMalloc(&m->blob, n_floats * sizeof(float));
    if (e != hipSuccess) {
        delete m;
        p2s_set_error("hipMalloc(weights) failed: %s", hipGetErrorString(e));
        return P2S_ENOMEM;
    }
    e = hipMemcpy(m->blob, blob_host, n_floats * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(m->blob);
        delete m;
        p2s_set_error("hipMemcpy(weights) failed: %s", hipGetErrorString(e));
        return P2S_EHIP;
    }
    if (cfg->encoder_bf16) {
        // bf16 fragments of the per-point MFMA layers, packed on the device from the fp32 fragments
        struct Item { size_t *dst; uint64_t src; int K, N; };
        std::vector<Item> items;
        for (int e = 0; e < 2; ++e) {
            const p2s_encoder_offsets &eo = offs->enc[e];
            items.push_back({&m->h_w0b[e], eo.w0b, 64, 64});
            items.push_back({&m->h_s1[e], eo.s1, 64, 64});
            items.push_back({&m->h_s2[e], eo.s2, 64, 128});
            items.push_back({&m->h_s3[e], eo.s3, 128, 1024});
            items.push_back({&m->h_m2[e], eo.m2, 64, 128});
            items.push_back({&m->h_m3[e], eo.m3, 128, 1024});
        }
        if (cfg->use_point_stn) {
            items.push_back({&m->h_qc2, offs->qstn.c2, 64, 128});
            items.push_back({&m->h_qc3, offs->qstn.c3, 128, 1024});
        }
        // fp16 pair mode: the encoder-side head layers (STN fc1..fc3, QSTN fc1 / fc2) run on fp16-pair MFMAs too
        m->heads_f16 = cfg->encoder_bf16 == 4;
        if (m->heads_f16) {
            for (int e = 0; e < 2; ++e) {
                const p2s_encoder_offsets &eo = offs->enc[e];
                items.push_back({&m->h_sf1[e], eo.sf1, 1024, 512});
                items.push_back({&m->h_sf2[e], eo.sf2, 512, 256});
                items.push_back({&m->h_sf3[e], eo.sf3, 256, 4096});
            }
            if (cfg->use_point_stn) {
                items.push_back({&m->h_qf1, offs->qstn.f1, 1024, 512});
                items.push_back({&m->h_qf2, offs->qstn.f2, 512, 256});
            }
        }
        size_t total = 0;
        for (auto &it : items) {
            *it.dst = total;
            total += (size_t)it.K * it.N;
        }
        if (cfg->encoder_bf16 < 1 || cfg->encoder_bf16 > 4) {
            (void)hipFree(m->blob);
            delete m;
            p2s_set_error("p2s_model_create: encoder_bf16 = %d (0 fp32, 1 bf16, 2 / 3 split bf16, 4 fp16 pair)", cfg->encoder_bf16);
            return P2S_EINVAL;
        }
        const int ns = p2s_enc_pieces(*cfg);           // 1 plain bf16, 2 / 3 split bf16 pieces, 2 for the fp16 pair
        const int f16 = p2s_enc_f16(*cfg);
        m->h_total = total;
        if (hipMalloc(&m->blob_h, total * 2 * ns) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(m->blob);
            delete m;
            p2s_set_error("hipMalloc(bf16 weights) failed");
            return P2S_ENOMEM;
        }
        int *wflag = nullptr;       // fp16 pair: raised by the packing kernel when a BN-folded weight does not fit the half range
        if (f16 && (hipMalloc(&wflag, 4) != hipSuccess || hipMemset(wflag, 0, 4) != hipSuccess)) {
            (void)hipGetLastError();
            p2s_model_destroy(m);
            p2s_set_error("hipMalloc(weight range flag) failed");
            return P2S_ENOMEM;
        }
        for (int piece = 0; piece < ns; ++piece)
            for (auto &it : items) {
                const int rc = p2s_launch_pack_bf16(m->blob + it.src, m->blob_h + (size_t)piece * total + *it.dst, it.K, it.N, 0, 0,
                                                    1, piece, f16, nullptr, wflag);
                if (rc) {
                    if (wflag) (void)hipFree(wflag);
                    p2s_model_destroy(m);
                    return rc;
                }
            }
        if (f16) {
            int h = 0;
            const hipError_t e2 = hipMemcpy(&h, wflag, 4, hipMemcpyDeviceToHost);
            (void)hipFree(wflag);
            if (e2 != hipSuccess || h) {
                p2s_model_destroy(m);
                if (e2 != hipSuccess) {
                    p2s_set_error("hipMemcpy(weight range flag) failed: %s", hipGetErrorString(e2));
                    return P2S_EHIP;
                }
                p2s_set_error("fp16 pair encoder (encoder_bf16 = 4): a BatchNorm-folded weight of this checkpoint does not fit the half "
                              "range (|w| > 6e4, or non-finite) -- use encoder_bf16 = 3 (the same accuracy) or 0 for this model");
                return P2S_EINVAL;
            }
            // side buffers of the fp32 fallback: inputs + results of up to 16384 flagged queries per call (256 MB at k = 300,
            // n = 1000; touched only when a query is flagged)
            p2s_model_s::Fallback &fb = m->fb;
            fb.cap = 16384;
            const size_t k3 = (size_t)cfg->points_per_patch * 3, n3 = (size_t)cfg->sub_sample_size * 3, cap = fb.cap;
            bool ok = hipMalloc(&fb.flags, (size_t)m->max_chunk * 4) == hipSuccess && hipMalloc(&fb.count, 4) == hipSuccess &&
                      hipMalloc(&fb.patch, cap * k3 * 4) == hipSuccess && hipMalloc(&fb.sub, cap * n3 * 4) == hipSuccess &&
                      hipMalloc(&fb.query, cap * 12) == hipSuccess && hipMalloc(&fb.radius, cap * 4) == hipSuccess &&
                      hipMalloc(&fb.index, cap * 8) == hipSuccess && hipMalloc(&fb.sdf, cap * 4) == hipSuccess &&
                      hipMalloc(&fb.logits, cap * 8) == hipSuccess;
            ok = ok && hipMemset(fb.flags, 0, (size_t)m->max_chunk * 4) == hipSuccess && hipMemset(fb.count, 0, 4) == hipSuccess &&
                 hipMemset(fb.radius, 0, cap * 4) == hipSuccess;
            if (!ok) {
                (void)hipGetLastError();
                p2s_model_destroy(m);
                p2s_set_error("hipMalloc(fp32 fallback buffers of the fp16 pair mode) failed");
                return P2S_ENOMEM;
            }
        }
        P2S_HIP_CHECK(hipDeviceSynchronize());
    }
    *out = m;
    return P2S_OK;
}

int p2s_model_destroy(p2s_model_t m) {
    if (!m) return P2S_OK;
    (void)hipSetDevice(m->device);
    p2s_pipe_free(m);
    if (m->ws) (void)hipFree(m->ws);
    if (m->blob) (void)hipFree(m->blob);
    if (m->blob_h) (void)hipFree(m->blob_h);
    for (void *p : {(void *)m->fb.flags, (void *)m->fb.count, (void *)m->fb.patch, (void *)m->fb.sub, (void *)m->fb.query,
                    (void *)m->fb.radius, (void *)m->fb.index, (void *)m->fb.sdf, (void *)m->fb.logits})
        if (p) (void)hipFree(p);
    for (auto &ev : m->evpool)
        if (ev) (void)hipEventDestroy(ev);
    if (m->aux) (void)hipStreamDestroy(m->aux);
    if (m->ball) (void)hipStreamDestroy(m->ball);
    delete m;
    return P2S_OK;
}

int p2s_set_profiling(p2s_model_t m, int enabled) {
    if (!m) return P2S_EINVAL;
    m->profiling = enabled != 0;
    return P2S_OK;
}

int p2s_get_counters(p2s_model_t m, p2s_counters *out) {
    if (!m || !out) return P2S_EINVAL;
    *out = m->counters;
    return P2S_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------

static size_t ws_floats_per_query(const p2s_model_s *m) {
    size_t n = 2 * 1024 + 2 * 512 + 2 * 256 + 2 * 4096 + 2 * 4096 + 2 * 1024 + 1024 + 256 + 128;
    if (m->cfg.use_point_stn) n += 2 * 1024 + 512 + 256 + 16;
    if (m->cfg.encoder_bf16) n += 4096 * (size_t)p2s_enc_pieces(m->cfg);   // W1' of both encoders as 16-bit fragments, per piece
    return n;
}

int p2s_model_reserve(p2s_model_s *m, int chunk) {
    if (chunk <= m->ws_chunk) return P2S_OK;
    if (m->ws) {
        P2S_HIP_CHECK(hipDeviceSynchronize());
        (void)hipFree(m->ws);
        m->ws = nullptr;
        m->ws_chunk = 0;
    }
    const size_t n = ws_floats_per_query(m) * (size_t)chunk;
    hipError_t e = hipMalloc(&m->ws, n * sizeof(float));
    if (e != hipSuccess) {
        p2s_set_error("hipMalloc(workspace %zu MB) failed: %s", n * 4 >> 20, hipGetErrorString(e));
        return P2S_ENOMEM;
    }
    m->ws_chunk = chunk;
    return P2S_OK;
}

namespace {

struct Ws {
    float *g_stn, *h1, *h2, *T, *w1p, *feat, *d1, *d2, *d3, *qg, *qg2, *qh1, *qh2, *rot;
    unsigned short *w1h;
};

Ws carve(const p2s_model_s *m, int C) {
    Ws w;
    float *p = m->ws;
    auto take = [&](size_t n) { float *r = p; p += n; return r; };
    w.g_stn = take((size_t)2 * C * 1024);
    w.h1 = take((size_t)2 * C * 512);
    w.h2 = take((size_t)2 * C * 256);
    w.T = take((size_t)2 * C * 4096);
    w.w1p = take((size_t)2 * C * 4096);
    w.feat = take((size_t)2 * C * 1024);
    w.d1 = take((size_t)C * 1024);
    w.d2 = take((size_t)C * 256);
    w.d3 = take((size_t)C * 128);
    w.qg = w.qg2 = w.qh1 = w.qh2 = w.rot = nullptr;
    if (m->cfg.use_point_stn) {
        w.qg = take((size_t)C * 1024);
        w.qg2 = take((size_t)C * 1024);
        w.qh1 = take((size_t)C * 512);
        w.qh2 = take((size_t)C * 256);
        w.rot = take((size_t)C * 16);
    }
    w.w1h = m->cfg.encoder_bf16 ? reinterpret_cast<unsigned short *>(take((size_t)C * 4096 * p2s_enc_pieces(m->cfg))) : nullptr;
    return w;
}

}  // namespace

// fp16 pair mode: one workgroup per query of the chunk; a flagged query (ChainArgs.bad_items / GemmArgs.bad_rows) takes the
// next slot of the side buffers and its network inputs are copied there
__global__ __launch_bounds__(256) void p2s_fb_collect_kernel(int *__restrict__ flags, int *__restrict__ count, int cap,
                                                             const float *__restrict__ patch, const float *__restrict__ sub,
                                                             const float *__restrict__ query, const float *__restrict__ radius,
                                                             int k3, int n3, long long index0, float *__restrict__ fpatch,
                                                             float *__restrict__ fsub, float *__restrict__ fquery,
                                                             float *__restrict__ fradius, long long *__restrict__ findex) {
    __shared__ int s_slot;
    const int q = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int slot = -1;
        if (flags[q]) {
            flags[q] = 0;
            slot = atomicAdd(count, 1);
        }
        s_slot = slot;
    }
    __syncthreads();
    const int slot = s_slot;
    if (slot < 0 || slot >= cap) return;
    for (int i = tid; i < k3; i += 256) fpatch[(size_t)slot * k3 + i] = patch[(size_t)q * k3 + i];
    for (int i = tid; i < n3; i += 256) fsub[(size_t)slot * n3 + i] = sub[(size_t)q * n3 + i];
    if (tid < 3) fquery[(size_t)slot * 3 + tid] = query[(size_t)q * 3 + tid];
    if (tid == 3) {
        if (radius) fradius[slot] = radius[q];
        findex[slot] = index0 + q;
    }
}

__global__ void p2s_fb_scatter_kernel(const long long *__restrict__ index, const float *__restrict__ sdf,
                                      const float *__restrict__ logits, int n, int od, float *__restrict__ sdf_out,
                                      float *__restrict__ logits_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long d = index[i];
    if (sdf_out) sdf_out[d] = sdf[i];
    if (logits_out)
        for (int j = 0; j < od; ++j) logits_out[d * od + j] = logits[(size_t)i * od + j];
}

// One chunk (C <= ws_chunk queries) through encoders (+ decoder if want_decode).
int p2s_run_chunk(p2s_model_s *m, const float *patch, const float *sub, const float *query, const float *radius,
                  int C, float *logits_out, float *sdf_out, float *feat_local_out, float *feat_global_out,
                  hipStream_t s, long long index0) {
    const p2s_weight_offsets &o = m->offs;
    const float *W = m->blob;
    const int PL = m->cfg.points_per_patch, PG = m->cfg.sub_sample_size;
    Ws w = carve(m, C);
    int rc;
    const bool bf16 = m->cfg.encoder_bf16 != 0;
    const int ev0 = p2s_prof_mark(m, s);
    int evq1 = -1, evq2 = -1;                         // QSTN models: behind the QSTN trunk launch / behind its head layers

    const float *rot = nullptr;
    if (m->cfg.use_point_stn) {
        // shared QSTN over cat(patch, sub-sample - q): reference points_to_surf_model.py:325-331, :100-131; without
        // shared_transformer the QSTN belongs to feat_global and sees the sub-sample alone (:283-284, :177-185), its
        // rotation is applied to the sub-sample and to the patch (:337-339) -- the same kernels, other points
        ChainArgs a;
        memset(&a, 0, sizeof(a));
        a.ns = p2s_enc_pieces(m->cfg);
        a.f16 = p2s_enc_f16(m->cfg);
        a.bad_items = m->fb.flags;
        a.piece_stride = (long long)m->h_total;
        a.w1_piece_stride = (long long)2 * C * 4096;
        ChainBranch &b = a.br[0];
        b.ptsA = patch; b.ptsB = sub; b.center = query; b.rot = nullptr;
        const bool qstn_shared = m->cfg.shared_transformer != 0 || m->cfg.single_transformer != 0;
        b.w0a = W + o.qstn.c1; b.b0a = W + o.qstn.cb1;
        b.w0b = b.b0b = nullptr; b.w1 = W; b.b1 = nullptr; b.w1_item_stride = 0;
        b.w2 = W + o.qstn.c2; b.b2 = W + o.qstn.cb2;
        b.w3 = W + o.qstn.c3; b.b3 = W + o.qstn.cb3;
        // the shared QSTN's 1300 points run as TWO workgroups per query -- sub-sample (1000) and patch (300), like the
        // encoder passes -- and the head takes max(pool, pool): 4096 equal 1300-point workgroups filled the 768
        // workgroup slots in 5.33 rounds (105 TFLOP/s), 8192 unequal ones pack like the encoder passes (141 TFLOP/s)
        b.ptsA = nullptr; b.out = w.qg; b.P = PG; b.P1 = 0; b.n_items = C; b.relu_out = 1; b.short_chain = 1;
        if (bf16) {
            b.w2 = reinterpret_cast<const float *>(m->blob_h + m->h_qc2);
            b.w3 = reinterpret_cast<const float *>(m->blob_h + m->h_qc3);
        }
        a.br[1] = b;
        if (qstn_shared) {
            ChainBranch &p = a.br[1];
            p.ptsA = patch; p.ptsB = nullptr; p.center = nullptr; p.P = PL; p.P1 = PL; p.out = w.qg2;
        } else {
            a.br[1].n_items = 0;
        }
        if ((rc = bf16 ? p2s_launch_chain_bf16(a, s) : p2s_launch_chain(a, s))) return rc;
        evq1 = p2s_prof_mark(m, s);
        m->counters.launches_chain += 1;
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.A = w.qg; g.A2 = qstn_shared ? w.qg2 : nullptr; g.a2_z = 0; g.lda = 1024; g.a_z = 0; g.W[0] = g.W[1] = W + o.qstn.f1; g.bias[0] = g.bias[1] = W + o.qstn.fb1;
        g.C = w.qh1; g.ldc = 512; g.c_z = 0; g.M = C; g.N = 512; g.K = 1024; g.Z = 1; g.relu = 1;
        if (m->heads_f16) {
            g.Wh[0] = g.Wh[1] = m->blob_h + m->h_qf1; g.wh_piece = (long long)m->h_total; g.bad_rows = m->fb.flags;
        }
        if ((rc = p2s_launch_gemm(g, s))) return rc;
        g.A = w.qh1; g.A2 = nullptr; g.lda = 512; g.W[0] = g.W[1] = W + o.qstn.f2; g.bias[0] = g.bias[1] = W + o.qstn.fb2;
        g.C = w.qh2; g.ldc = 256; g.N = 256; g.K = 512;
        if (m->heads_f16) g.Wh[0] = g.Wh[1] = m->blob_h + m->h_qf2;
        if ((rc = p2s_launch_gemm(g, s))) return rc;
        if ((rc = p2s_launch_qstn_tail(w.qh2, W + o.qstn.f3, W + o.qstn.fb3, w.rot, C, 256, s))) return rc;
        rot = w.rot;
        evq2 = p2s_prof_mark(m, s);
    }

    // ---- pass 1: stem + STN trunk + max-pool, both encoders (global items first: longest first) ----
    ChainArgs a;
    memset(&a, 0, sizeof(a));
    a.ns = p2s_enc_pieces(m->cfg);
    a.f16 = p2s_enc_f16(m->cfg);
    a.bad_items = m->fb.flags;
    a.piece_stride = (long long)m->h_total;
    a.w1_piece_stride = (long long)2 * C * 4096;
    for (int slot = 0; slot < 2; ++slot) {
        const int e = 1 - slot;   // slot 0 = feat_global (e=1), slot 1 = feat_local (e=0)
        const p2s_encoder_offsets &eo = o.enc[e];
        ChainBranch &b = a.br[slot];
        if (e == 1) { b.ptsA = nullptr; b.ptsB = sub; b.center = query; b.P = PG; b.P1 = 0; }
        else        { b.ptsA = patch; b.ptsB = nullptr; b.center = nullptr; b.P = PL; b.P1 = PL; }
        b.rot = rot;
        b.w0a = W + eo.w0a; b.b0a = W + eo.b0a; b.w0b = W + eo.w0b; b.b0b = W + eo.b0b;
        b.w1 = W + eo.s1; b.b1 = W + eo.sb1; b.w1_item_stride = 0;
        b.w2 = W + eo.s2; b.b2 = W + eo.sb2; b.w3 = W + eo.s3; b.b3 = W + eo.sb3;
        b.out = w.g_stn + (size_t)e * C * 1024;
        b.n_items = C; b.relu_out = 1; b.short_chain = 0;
        if (bf16) {
            b.w0b = reinterpret_cast<const float *>(m->blob_h + m->h_w0b[e]);
            b.w1 = reinterpret_cast<const float *>(m->blob_h + m->h_s1[e]);
            b.w2 = reinterpret_cast<const float *>(m->blob_h + m->h_s2[e]);
            b.w3 = reinterpret_cast<const float *>(m->blob_h + m->h_s3[e]);
        }
    }
    if ((rc = bf16 ? p2s_launch_chain_bf16(a, s) : p2s_launch_chain(a, s))) return rc;
    const int ev1 = p2s_prof_mark(m, s);

    // ---- STN head: 1024 -> 512 -> 256 -> 4096 (+I), then W1' = W1 . trans2 -------------------------
    {
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.M = C; g.Z = 2; g.relu = 1;
        g.A = w.g_stn; g.lda = 1024; g.a_z = (long long)C * 1024;
        if (m->cfg.single_transformer) {      // one encoder over both point sets: its pool = max of the two branches' pools
            g.A2 = w.g_stn + (size_t)C * 1024;
            g.a2_z = -(long long)C * 1024;
        }
        g.W[0] = W + o.enc[0].sf1; g.W[1] = W + o.enc[1].sf1; g.bias[0] = W + o.enc[0].sfb1; g.bias[1] = W + o.enc[1].sfb1;
        g.C = w.h1; g.ldc = 512; g.c_z = (long long)C * 512; g.N = 512; g.K = 1024;
        if (m->heads_f16) {
            g.Wh[0] = m->blob_h + m->h_sf1[0]; g.Wh[1] = m->blob_h + m->h_sf1[1];
            g.wh_piece = (long long)m->h_total; g.bad_rows = m->fb.flags;
        }
        if ((rc = p2s_launch_gemm(g, s))) return rc;
        g.A = w.h1; g.A2 = nullptr; g.lda = 512; g.a_z = (long long)C * 512;
        g.W[0] = W + o.enc[0].sf2; g.W[1] = W + o.enc[1].sf2; g.bias[0] = W + o.enc[0].sfb2; g.bias[1] = W + o.enc[1].sfb2;
        g.C = w.h2; g.ldc = 256; g.c_z = (long long)C * 256; g.N = 256; g.K = 512;
        if (m->heads_f16) { g.Wh[0] = m->blob_h + m->h_sf2[0]; g.Wh[1] = m->blob_h + m->h_sf2[1]; }
        if ((rc = p2s_launch_gemm(g, s))) return rc;
        g.A = w.h2; g.lda = 256; g.a_z = (long long)C * 256; g.relu = 0;
        g.W[0] = W + o.enc[0].sf3; g.W[1] = W + o.enc[1].sf3; g.bias[0] = W + o.enc[0].sfb3; g.bias[1] = W + o.enc[1].sfb3;
        g.C = w.T; g.ldc = 4096; g.c_z = (long long)C * 4096; g.N = 4096; g.K = 256;
        if (m->heads_f16) { g.Wh[0] = m->blob_h + m->h_sf3[0]; g.Wh[1] = m->blob_h + m->h_sf3[1]; }
        if ((rc = p2s_launch_gemm(g, s))) return rc;
        FoldArgs f;
        memset(&f, 0, sizeof(f));
        for (int e = 0; e < 2; ++e) {
            f.T[e] = w.T + (size_t)e * C * 4096;
            f.m1t[e] = W + o.enc[e].m1t;
            f.out[e] = w.w1p + (size_t)e * C * 4096;
            f.outh[e] = bf16 ? w.w1h + (size_t)e * C * 4096 : nullptr;      // 16-bit modes: pieces written by the fold itself
        }
        f.h_piece_stride = (long long)2 * C * 4096;
        f.ns = p2s_enc_pieces(m->cfg);
        f.f16 = p2s_enc_f16(m->cfg);
        f.bad_items = f.f16 ? m->fb.flags : nullptr;
        f.n_items = C;
        if ((rc = p2s_launch_fold(f, s))) return rc;
    }
    const int ev2 = p2s_prof_mark(m, s);

    // ---- pass 2: stem (recomputed) + transformed conv1 + conv2 + conv3 + max-pool -------------------
    for (int slot = 0; slot < 2; ++slot) {
        const int e = 1 - slot;
        const p2s_encoder_offsets &eo = o.enc[e];
        ChainBranch &b = a.br[slot];
        b.w1 = w.w1p + (size_t)e * C * 4096; b.b1 = W + eo.mb1; b.w1_item_stride = 4096;
        b.w2 = W + eo.m2; b.b2 = W + eo.mb2; b.w3 = W + eo.m3; b.b3 = W + eo.mb3;
        b.out = w.feat + (size_t)e * C * 1024;
        b.relu_out = 0;
        b.pool_sum = m->cfg.sym_sum ? 1 : 0;        // sym_op='sum': PointNetfeat's pool only (the STN / QSTN trunks keep the max)
        if (bf16) {
            b.w1 = reinterpret_cast<const float *>(w.w1h + (size_t)e * C * 4096);
            b.w2 = reinterpret_cast<const float *>(m->blob_h + m->h_m2[e]);
            b.w3 = reinterpret_cast<const float *>(m->blob_h + m->h_m3[e]);
        }
    }
    if ((rc = bf16 ? p2s_launch_chain_bf16(a, s) : p2s_launch_chain(a, s))) return rc;
    const int ev3 = p2s_prof_mark(m, s);
    m->counters.launches_chain += 2;

    if (feat_local_out) P2S_HIP_CHECK(hipMemcpyAsync(feat_local_out, w.feat, (size_t)C * 1024 * 4, hipMemcpyDeviceToDevice, s));
    if (feat_global_out)
        P2S_HIP_CHECK(hipMemcpyAsync(feat_global_out, w.feat + (size_t)C * 1024, (size_t)C * 1024 * 4, hipMemcpyDeviceToDevice, s));

    if (logits_out || sdf_out) {
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.M = C; g.relu = 1;
        // fc1_local | fc1_global -> cat (local first): reference points_to_surf_model.py:335,343,346
        g.Z = 2; g.A = w.feat; g.lda = 1024; g.a_z = (long long)C * 1024;
        if (m->cfg.single_transformer) {      // fc1_local_global reads the ONE pooled feature; d1l / d1g are its column halves
            g.A2 = w.feat + (size_t)C * 1024;
            g.a2_z = -(long long)C * 1024;
            g.a2_add = m->cfg.sym_sum ? 1 : 0;        // sym_op='sum': the pool over both point sets = the sum of the two sums
        }
        g.W[0] = W + o.d1l; g.W[1] = W + o.d1g; g.bias[0] = W + o.db1l; g.bias[1] = W + o.db1g;
        g.C = w.d1; g.ldc = 1024; g.c_z = 512; g.N = 512; g.K = 1024;
        if ((rc = p2s_launch_gemm(g, s))) return rc;
        g.Z = 1; g.A = w.d1; g.A2 = nullptr; g.lda = 1024; g.a_z = 0; g.W[0] = g.W[1] = W + o.d2; g.bias[0] = g.bias[1] = W + o.db2;
        g.C = w.d2; g.ldc = 256; g.c_z = 0; g.N = 256; g.K = 1024;
        if ((rc = p2s_launch_gemm(g, s))) return rc;
        g.A = w.d2; g.lda = 256; g.W[0] = g.W[1] = W + o.d3; g.bias[0] = g.bias[1] = W + o.db3;
        g.C = w.d3; g.ldc = 128; g.N = 128; g.K = 256;
        if ((rc = p2s_launch_gemm(g, s))) return rc;
        if ((rc = p2s_launch_decoder_tail(w.d3, W + o.d4, W + o.db4, radius, logits_out, sdf_out, C, 128, m->cfg.output_dim, s))) return rc;
    }
    const int ev4 = p2s_prof_mark(m, s);
    if (evq1 >= 0 && evq2 >= 0) {
        p2s_prof_span(m, ST_CHAIN_QSTN, ev0, evq1);
        p2s_prof_span(m, ST_HEAD, evq1, evq2);
        p2s_prof_span(m, ST_CHAIN_STN, evq2, ev1);
    } else {
        p2s_prof_span(m, ST_CHAIN_STN, ev0, ev1);
    }
    p2s_prof_span(m, ST_HEAD, ev1, ev2);
    p2s_prof_span(m, ST_CHAIN_MAIN, ev2, ev3);
    p2s_prof_span(m, ST_DECODER, ev3, ev4);
    if (m->cfg.encoder_bf16 == 4 && m->fb.flags) {
        // fp16 pair mode: the inputs of the queries the 16-bit kernels flagged are put aside (the chunk buffers are reused
        // two chunks on); one workgroup per query, all but the flagged ones return at once
        hipLaunchKernelGGL(p2s_fb_collect_kernel, dim3(C), dim3(256), 0, s, m->fb.flags, m->fb.count, m->fb.cap, patch, sub, query,
                           radius, PL * 3, PG * 3, index0, m->fb.patch, m->fb.sub, m->fb.query, m->fb.radius, m->fb.index);
        P2S_LAUNCH_CHECK("p2s_fb_collect_kernel");
    }
    return P2S_OK;
}

int p2s_model_fallback_finish(p2s_model_s *m, float *logits_out, float *sdf_out, hipStream_t s) {
    p2s_model_s::Fallback &fb = m->fb;
    if (!fb.count) return P2S_OK;
    int h = 0;
    P2S_HIP_CHECK(hipMemcpyAsync(&h, fb.count, 4, hipMemcpyDeviceToHost, s));
    P2S_HIP_CHECK(hipStreamSynchronize(s));
    if (!h) return P2S_OK;
    P2S_HIP_CHECK(hipMemsetAsync(fb.count, 0, 4, s));
    if (h > fb.cap || (!logits_out && !sdf_out)) {
        p2s_set_error("fp16 pair encoder (encoder_bf16 = 4): %d queries of this call have activations beyond the half range (> 6e4)%s "
                      "-- use encoder_bf16 = 3 or 0 for this model", h,
                      h > fb.cap ? ", more than the fp32 fallback takes per call (16384)" : " and the call has no output the fp32 fallback could repair");
        return P2S_EINVAL;
    }
    m->counters.fallback_queries += h;
    // the same queries through the fp32 kernels (the fp32 fragments of every layer are resident in any mode)
    const p2s_model_cfg saved = m->cfg;
    const bool saved_heads = m->heads_f16;
    m->cfg.encoder_bf16 = 0;
    m->heads_f16 = false;
    const size_t k3 = (size_t)m->cfg.points_per_patch * 3, n3 = (size_t)m->cfg.sub_sample_size * 3;
    const int od = m->cfg.output_dim;
    int rc = P2S_OK;
    for (int i0 = 0; i0 < h && rc == P2S_OK; i0 += m->ws_chunk) {
        const int C = std::min(m->ws_chunk, h - i0);
        rc = p2s_run_chunk(m, fb.patch + i0 * k3, fb.sub + i0 * n3, fb.query + (size_t)i0 * 3, sdf_out ? fb.radius + i0 : nullptr, C,
                           fb.logits + (size_t)i0 * od, sdf_out ? fb.sdf + i0 : nullptr, nullptr, nullptr, s, 0);
    }
    m->cfg = saved;
    m->heads_f16 = saved_heads;
    if (rc) return rc;
    hipLaunchKernelGGL(p2s_fb_scatter_kernel, dim3((h + 255) / 256), dim3(256), 0, s, fb.index, fb.sdf, fb.logits, h, od, sdf_out, logits_out);
    P2S_LAUNCH_CHECK("p2s_fb_scatter_kernel");
    P2S_HIP_CHECK(hipStreamSynchronize(s));
    return P2S_OK;
}

int p2s_prof_mark(p2s_model_s *m, hipStream_t s) {
    if (!m->profiling) return -1;
    if (m->ev_used >= (int)m->evpool.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return -1;
        m->evpool.push_back(e);
    }
    const int i = m->ev_used++;
    if (hipEventRecord(m->evpool[i], s) != hipSuccess) return -1;
    return i;
}

void p2s_prof_span(p2s_model_s *m, int stage, int a, int b) {
    if (a >= 0 && b >= 0) m->spans.push_back({stage, a, b});
}

void p2s_prof_reset(p2s_model_s *m) {
    m->ev_used = 0;
    m->spans.clear();
    memset(&m->counters, 0, sizeof(m->counters));
}

void p2s_prof_collect(p2s_model_s *m) {
    if (!m->profiling || m->ev_used == 0) return;
    (void)hipEventSynchronize(m->evpool[m->ev_used - 1]);
    for (const auto &sp : m->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, m->evpool[sp.a], m->evpool[sp.b]) != hipSuccess) {
            (void)hipGetLastError();          // an event that was never reached: only this complaint is dropped
            continue;
        }
        double *dst = nullptr;
        switch (sp.stage) {
            case ST_CHAIN_STN: dst = &m->counters.ms_chain_stn; break;
            case ST_HEAD: dst = &m->counters.ms_stn_head; break;
            case ST_CHAIN_MAIN: dst = &m->counters.ms_chain_main; break;
            case ST_DECODER: dst = &m->counters.ms_decoder; break;
            case ST_KNN: dst = &m->counters.ms_knn; break;
            case ST_SUB: dst = &m->counters.ms_subsample; break;
            case ST_GRID: dst = &m->counters.ms_grid; break;
            case ST_CHAIN_QSTN: dst = &m->counters.ms_chain_qstn; break;
        }
        if (dst) *dst += ms;
    }
    m->ev_used = 0;
    m->spans.clear();
}

static int run_batched(p2s_model_s *m, const float *patch, const float *sub, const float *query, const float *radius,
                       int B, float *logits, float *sdf, float *fl, float *fg, hipStream_t s) {
    if (!m || B < 0 || !patch || !sub || !query) {
        p2s_set_error("encode: null argument");
        return P2S_EINVAL;
    }
    if (sdf && !radius) {
        p2s_set_error("encode: sdf_out requested without radius");
        return P2S_EINVAL;
    }
    P2S_HIP_CHECK(hipSetDevice(m->device));
    const int chunk = std::min(B, m->max_chunk);
    int rc = p2s_model_reserve(m, chunk);
    if (rc) return rc;
    const int PL = m->cfg.points_per_patch, PG = m->cfg.sub_sample_size;
    p2s_prof_reset(m);
    for (int q0 = 0; q0 < B; q0 += chunk) {
        const int C = std::min(chunk, B - q0);
        rc = p2s_run_chunk(m, patch + (size_t)q0 * PL * 3, sub + (size_t)q0 * PG * 3, query + (size_t)q0 * 3,
                           radius ? radius + q0 : nullptr, C, logits ? logits + (size_t)q0 * m->cfg.output_dim : nullptr,
                           sdf ? sdf + q0 : nullptr, fl ? fl + (size_t)q0 * 1024 : nullptr,
                           fg ? fg + (size_t)q0 * 1024 : nullptr, s, q0);
        if (rc) return rc;
    }
    p2s_prof_collect(m);
    m->counters.queries += B;
    // fp16 pair encoder: queries with activations beyond the half range are repaired by the fp32 kernels now (this
    // synchronises `s` in that mode only)
    return p2s_model_fallback_finish(m, logits, sdf, s);
}

extern "C" {

int p2s_encode_decode(p2s_model_t m, const float *patch_ps_dev, const float *sub_ms_dev, const float *query_dev,
                      const float *radius_dev, int B, float *logits_out_dev, float *sdf_out_dev, void *stream) {
    if (!logits_out_dev && !sdf_out_dev) {
        p2s_set_error("p2s_encode_decode: no output requested");
        return P2S_EINVAL;
    }
    return run_batched(m, patch_ps_dev, sub_ms_dev, query_dev, radius_dev, B, logits_out_dev, sdf_out_dev, nullptr,
                       nullptr, (hipStream_t)stream);
}

int p2s_encode_features(p2s_model_t m, const float *patch_ps_dev, const float *sub_ms_dev, const float *query_dev,
                        int B, float *feat_local_dev, float *feat_global_dev, void *stream) {
    if (!feat_local_dev && !feat_global_dev) {
        p2s_set_error("p2s_encode_features: no output requested");
        return P2S_EINVAL;
    }
    return run_batched(m, patch_ps_dev, sub_ms_dev, query_dev, nullptr, B, nullptr, nullptr, feat_local_dev,
                       feat_global_dev, (hipStream_t)stream);
}

}  // extern "C"
