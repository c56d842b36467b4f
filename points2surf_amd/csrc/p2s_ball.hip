// Fixed-radius patches: the patch_radius > 0 branch of the reference's patch selection
//   source/base/point_cloud.py:170-191   get_patch_kdtree: ids = kdtree.query_ball_point(query, r); more than
//       points_per_patch of them: ids[rng.choice(np.arange(count), points_per_patch, replace=False)]; fewer: -1 padding
//   source/data_loader.py:335-350        padding ids become 0 and their points the query point (the patch-space origin);
//       patch space = (p - q) / r in float32; the predicted distance is not rescaled (points_to_surf_eval.py:180,188)
//   experiments/train_p2s_{small,medium,large}_radius.sh:25   r = 0.05 / 0.1 / 0.2 with 300 points per patch
//
// ORDER.  For ONE query point scipy's query_ball_point does not sort: every leaf appends its hits in the order of the
// tree's index array and the "less" child is visited before the "greater" one whatever the pruning decided, so the
// result is cKDTree(pts, leafsize=1000).indices filtered by  d2 <= r * r  (float64, ((dx^2 + dy^2) + dz^2)).  The index
// array is the product of scipy's build (ckdtree/src/build.cxx): per node the dimension of largest extent, the median
// by std::nth_element over the indices compared by coordinate alone, a std::partition around that value, the sliding
// step when a side stays empty.  Introselect's permutation is inherently sequential, so the order is built on the HOST
// (p2s_kd_order_host; 5 ms for 87k points, once per cloud, only for fixed-radius models) -- checked against scipy itself
// in tests/test_kd_order.py -- and everything per query runs on the device: ball counts, hit lists in tree order, the
// serial dependence of the random stream (ball_chain_kernel) and the numpy-legacy shuffle (ball_patch_kernel).
//
// RANDOM STREAM.  The choice draws from the data set's FIRST generator (self.rng, source/data_loader.py:272,336), the
// one that also makes the rotations of the GT-query pass (:384) -- not from the sub-sample's (rng_global_sample).
// choice(arange(c), k, replace=False) = permutation(c)[:k] = the legacy shuffle: for i = c-1 .. 1: j = rk_interval(i)
// (32-bit words, masked rejection), swap(a[i], a[j]).  How many words a query consumes depends on the words: the start
// of query q+1 is known only behind query q.  ball_chain_kernel walks that chain with ONE wave, resolving 64 words at
// a time: a word is accepted iff (w & mask) <= i - (accepted words before it), a fixed point reached from the left
// and detected when an evaluation changes nothing (1-2 evaluations; model: tests/radius_model.py).  With every start
// known, ball_patch_kernel (one wave per query, all CUs) repeats the walk, applies the swaps to the query's hit list
// and writes the patch.
#include "p2s_common.h"
#include "p2s_internal.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#pragma clang fp contract(off)

namespace {

// ---------------------------------------------------------------------------------------------------------------
// host: index array of scipy.spatial.cKDTree(data, leafsize) (balanced_tree = compact_nodes = True, the defaults)
// ---------------------------------------------------------------------------------------------------------------
struct KdBuild {
    const double *data;
    int *ind;
    int leaf;
    std::vector<int> *leaf_start;

    void node(int s, int e) {
        if (e - s <= leaf) {
            leaf_start->push_back(s);
            return;
        }
        double mx[3], mn[3];
        for (int a = 0; a < 3; ++a) mx[a] = mn[a] = data[(size_t)ind[s] * 3 + a];
        for (int j = s + 1; j < e; ++j)
            for (int a = 0; a < 3; ++a) {
                const double t = data[(size_t)ind[j] * 3 + a];
                mx[a] = mx[a] > t ? mx[a] : t;
                mn[a] = mn[a] < t ? mn[a] : t;
            }
        int d = 0;
        double size = 0.0;
        for (int a = 0; a < 3; ++a)
            if (mx[a] - mn[a] > size) {
                d = a;
                size = mx[a] - mn[a];
            }
        if (mx[d] == mn[d]) {                        // all points identical: a leaf however large
            leaf_start->push_back(s);
            return;
        }
        const double *dat = data;
        auto less = [dat, d](int a, int b) { return dat[(size_t)a * 3 + d] < dat[(size_t)b * 3 + d]; };
        std::nth_element(ind + s, ind + s + (e - s) / 2, ind + e, less);
        double split = data[(size_t)ind[s + (e - s) / 2] * 3 + d];
        auto part = [&](double pivot) {
            return (int)(std::partition(ind + s, ind + e, [dat, d, pivot](int a) { return dat[(size_t)a * 3 + d] < pivot; }) - ind);
        };
        int p = part(split);
        if (p == s) {                                // nothing below the median value: everything equal to the minimum goes left
            const int mi = *std::min_element(ind + s, ind + e, less);
            split = std::nextafter(data[(size_t)mi * 3 + d], HUGE_VAL);
            p = part(split);
        } else if (p == e) {
            const int ma = *std::max_element(ind + s, ind + e, less);
            split = data[(size_t)ma * 3 + d];
            p = part(split);
        }
        node(s, p);
        node(p, e);
    }
};

// ---------------------------------------------------------------------------------------------------------------
// device: hits of the ball in tree order
// ---------------------------------------------------------------------------------------------------------------
struct BallTree {
    const float4 *pts;       // [n] in index-array order: xyz + original id (bit cast)
    const int *leaf;         // [L + 1]
    const float *box;        // [L][6] lo, hi of the leaf's points
    int L;
};

// one wave per query.  LIST: append the ids to list[q * stride ...] in order; else count only.
template <bool LIST>
__global__ __launch_bounds__(256) void ball_scan_kernel(BallTree t, const float *__restrict__ q, long long nq, double r2,
                                                        int *__restrict__ count, int *__restrict__ lists,
                                                        const long long *__restrict__ off) {
    const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (w >= nq) return;
    const double qx = q[3 * w + 0], qy = q[3 * w + 1], qz = q[3 * w + 2];
    const uint64_t lt = (1ull << lane) - 1ull;
    int *out = LIST ? lists + off[w] : nullptr;
    int cnt = 0;
    for (int l = 0; l < t.L; ++l) {
        const float *b = t.box + 6 * l;
        const double ex = fmax(0.0, fmax((double)b[0] - qx, qx - (double)b[3]));
        const double ey = fmax(0.0, fmax((double)b[1] - qy, qy - (double)b[4]));
        const double ez = fmax(0.0, fmax((double)b[2] - qz, qz - (double)b[5]));
        // pruning only (every point is tested exactly below): a generous margin keeps it conservative
        if (ex * ex + ey * ey + ez * ez > r2 * 1.000001 + 1e-30) continue;
        const int s = t.leaf[l], e = t.leaf[l + 1];
        for (int i0 = s; i0 < e; i0 += 64) {
            const int i = i0 + lane;
            const float4 p = t.pts[i < e ? i : e - 1];
            const double dx = (double)p.x - qx, dy = (double)p.y - qy, dz = (double)p.z - qz;
            const double d2 = (dx * dx + dy * dy) + dz * dz;
            const bool in = i < e && d2 <= r2;
            const uint64_t m = __ballot(in);
            if (LIST && in) out[cnt + __popcll(m & lt)] = __float_as_int(p.w);
            cnt += __popcll(m);
        }
    }
    if (!LIST && lane == 0) count[w] = cnt;
}

__global__ __launch_bounds__(1024) void ball_offsets_kernel(const int *__restrict__ count, int nq, long long *__restrict__ off) {
    // exclusive prefix sums of <= 16384 counts, one workgroup
    __shared__ long long part[1024];
    const int tid = threadIdx.x;
    const int per = (nq + 1023) / 1024;
    long long s = 0;
    for (int k = 0; k < per; ++k) {
        const int i = tid * per + k;
        s += i < nq ? count[i] : 0;
    }
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const long long v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    long long run = tid ? part[tid - 1] : 0;
    for (int k = 0; k < per; ++k) {
        const int i = tid * per + k;
        if (i < nq) {
            off[i] = run;
            run += count[i];
        }
    }
    if (tid == 1023) off[nq] = part[1023];
}

// ---------------------------------------------------------------------------------------------------------------
// the legacy shuffle (one wave): 64 W words are fetched at a time (word 64 u + lane in slot u), the slots are resolved
// one after the other.  Word j of a slot belongs to step i - A_j (A_j = accepted words before it): it is accepted iff
//     (w_j & mask(i - A_j)) <= i - A_j,      mask(t) = smallest 2^b - 1 >= t,
// a fixed point in the accept flags that is reached from the left (word 0 is right after one evaluation) and detected
// when an evaluation changes nothing.  The first evaluation starts from the expected counts A_j ~ j (i + 1) / 2^bits.
// Model with evaluation counts: tests/radius_model.py (n = 3047: 70 slots, 1.2 evaluations each; resolving all 256
// words at once needs 2.2 evaluations of FOUR slots each -- twice the instructions).
// src.prepare(p) / src.word(p, u) = raw word p + 64 u + lane of the stream; put(flag, t, j): flagged lanes carry step
// t (< steps) of this slot, whose swap is (a[i - t], a[j]); flush(i, steps) closes the slot.  Returns the position
// behind the shuffle.  i, pos and every count are wave-uniform and live in scalar registers as long as n does (callers
// pass a readlane / scalar-load value); a lone wave issues ONE instruction, vector or scalar, per 4 cycles, so the
// cost of the walk is its instruction count.
// ---------------------------------------------------------------------------------------------------------------
template <int W, class Src, class Put, class Flush>
__device__ __forceinline__ long long ball_shuffle_walk(Src &&src, long long pos, int n, Put &&put, Flush &&flush) {
    const int lane = threadIdx.x & 63;
    int i = n - 1;
    while (i >= 1) {
        uint32_t w[W];
        src.prepare(pos);
#pragma unroll
        for (int u = 0; u < W; ++u) w[u] = src.word(pos, u);
        long long p = pos;
#pragma unroll
        for (int u = 0; u < W; ++u) {
            if (i < 1) break;                                   // the shuffle ended in an earlier slot
            const int bits = 32 - __builtin_clz((uint32_t)i);
            const int thr0 = i - (int)(((uint32_t)lane * (uint32_t)(i + 1)) >> bits);      // expected step of this word
            const uint32_t m0 = 0xffffffffu >> __builtin_clz((uint32_t)(thr0 > 1 ? thr0 : 1));
            uint64_t b = __builtin_amdgcn_ballot_w64((int)(w[u] & m0) <= thr0);
            int before, v;
            for (;;) {
                // words behind the last step (thr < 1) may come out "accepted" (thr = 0: an even word): they lie behind
                // the cut below and change nothing before it -- the flags are still a function of the flags to their left
                before = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                const int thr = i - before;
                const uint32_t m = 0xffffffffu >> __builtin_clz((uint32_t)(thr > 1 ? thr : 1));
                v = (int)(w[u] & m);
                const uint64_t nb = __builtin_amdgcn_ballot_w64(v <= thr);
                if (nb == b) break;
                b = nb;
            }
            const int total = __popcll(b);
            int consumed = 64, steps = total;
            if (total >= i) {                                   // the shuffle ends inside this slot: stop behind step i
                steps = i;
                consumed = __ffsll((long long)(b & __builtin_amdgcn_ballot_w64(before == i - 1)));
            }
            put(((b >> lane) & 1ull) != 0 && before < steps, before, v);
            flush(i, steps);
            p += consumed;
            i -= steps;
        }
        pos = p;
    }
    return pos;
}

constexpr int BALL_W = 4;                            // words per lane and block

constexpr int BC_RING = 8192;                        // words staged in LDS by the chain wave

// ONE wave: where every query's shuffle starts (spos) and, with tail_words, where its trailing words sit (tpos: the
// rand(3) of the GT-query pass, source/data_loader.py:384).  meta[0] = word cursor (in/out), meta[1] = sticky error.
__global__ __launch_bounds__(64) void ball_chain_kernel(const uint32_t *__restrict__ words, long long cap_words, long long alloc_words,
                                                        const int *__restrict__ count, int nq, int k, int tail_words,
                                                        long long *__restrict__ spos, long long *__restrict__ tpos,
                                                        long long *__restrict__ meta) {
    extern __shared__ __attribute__((aligned(16))) uint32_t bc_ring[];
    const int lane = threadIdx.x;
    if (meta[1] != 0) return;
    long long pos = meta[0];
    long long w_hi = pos & ~3LL;                     // words below w_hi are resident at ring[w & (RING - 1)]
    uint4 pre[4];                                    // the next 1024 words, on their way
    auto issue = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long w = w_hi + 256 * u + 4 * lane;
            pre[u] = (w + 4 <= alloc_words) ? *(const uint4 *)(words + w) : make_uint4(0, 0, 0, 0);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) *(uint4 *)(bc_ring + ((w_hi + 256 * u + 4 * lane) & (BC_RING - 1))) = pre[u];
        w_hi += 1024;
        issue();
        __syncthreads();
    };
    issue();
    bool overflow = false;
    struct RingSrc {
        decltype(commit) &commit_;
        long long &w_hi_;
        bool &overflow_;
        long long cap_;
        const uint32_t *ring_;
        int lane_;
        __device__ __forceinline__ void prepare(long long p) {
            while (p + 64 * BALL_W > w_hi_) commit_();
            overflow_ |= p + 64 * BALL_W > cap_;
        }
        __device__ __forceinline__ uint32_t word(long long p, int u) const {
            return ring_[((uint32_t)p + 64u * (uint32_t)u + (uint32_t)lane_) & (BC_RING - 1)];
        }
    } src{commit, w_hi, overflow, cap_words, bc_ring, lane};
    auto no_put = [](bool, int, int) {};
    auto no_flush = [](int, int) {};
    __builtin_amdgcn_s_setprio(3);
    for (int q0 = 0; q0 < nq; q0 += 64) {
        const int mine = q0 + lane < nq ? count[q0 + lane] : 0;
        long long my_s = 0, my_t = 0;
        const int m = nq - q0 < 64 ? nq - q0 : 64;
        for (int u = 0; u < m; ++u) {
            const int c = __builtin_amdgcn_readlane(mine, u);      // scalar: the whole walk stays on the scalar unit
            if (lane == u) my_s = pos;
            if (c > k) pos = ball_shuffle_walk<BALL_W>(src, pos, c, no_put, no_flush);
            if (lane == u) my_t = pos;
            pos += tail_words;
        }
        if (lane < m) {
            spos[q0 + lane] = my_s;
            if (tpos) tpos[q0 + lane] = my_t;
        }
    }
    if (lane == 0) {
        meta[0] = pos;
        if (overflow || pos > cap_words) meta[1] = 2;          // random words exhausted (the host's reservation was too small)
    }
}

constexpr int BP_CAP = 8192;                         // hit lists up to this length are shuffled in LDS

// one wave per query: shuffle of its hit list (count > k), ids + patch-space points + radius
__global__ __launch_bounds__(64) void ball_patch_kernel(const uint32_t *__restrict__ words, long long cap_words,
                                                        const float *__restrict__ pts, const float *__restrict__ q,
                                                        const int *__restrict__ count, const long long *__restrict__ off,
                                                        int *__restrict__ lists, const long long *__restrict__ spos, int k,
                                                        float radius, int32_t *__restrict__ ids_out,
                                                        float *__restrict__ patch_out, float *__restrict__ radius_out,
                                                        const long long *__restrict__ meta) {
    __shared__ int lds_list[BP_CAP];
    __shared__ int jbuf[64];
    const int w = blockIdx.x, lane = threadIdx.x;
    if (meta[1] != 0) return;
    const int c = count[w];
    int *arr = lists + off[w];
    if (c > k) {
        if (c <= BP_CAP) {
            for (int i = lane; i < c; i += 64) lds_list[i] = arr[i];
            arr = lds_list;
            __syncthreads();
        }
        struct GlobalSrc {
            const uint32_t *words_;
            long long cap_;
            int lane_;
            __device__ __forceinline__ void prepare(long long) const {}
            __device__ __forceinline__ uint32_t word(long long p, int u) const {
                const long long w = p + 64 * u + lane_;
                return w < cap_ ? words_[w] : 0u;
            }
        } src{words, cap_words, lane};
        auto put = [&](bool flag, int t, int j) {
            if (flag) jbuf[t] = j;
        };
        auto flush = [&](int i, int steps) {
            __syncthreads();
            if (lane == 0)
                for (int u = 0; u < steps; ++u) {
                    const int a = i - u, b = jbuf[u];
                    const int va = arr[a], vb = arr[b];
                    arr[a] = vb;
                    arr[b] = va;
                }
            __syncthreads();
        };
        (void)ball_shuffle_walk<BALL_W>(src, spos[w], c, put, flush);
    }
    const float qx = q[3 * (size_t)w + 0], qy = q[3 * (size_t)w + 1], qz = q[3 * (size_t)w + 2];
    for (int t = lane; t < k; t += 64) {
        const bool pad = t >= c;                     // only when c < k: -1 ids -> id 0, point = the query point
        const int id = pad ? 0 : arr[t];
        float px = qx, py = qy, pz = qz;
        if (!pad) {
            px = pts[3 * (size_t)id + 0];
            py = pts[3 * (size_t)id + 1];
            pz = pts[3 * (size_t)id + 2];
        }
        if (ids_out) ids_out[(size_t)w * k + t] = id;
        float *o = patch_out + ((size_t)w * k + t) * 3;
        o[0] = (px - qx) / radius;
        o[1] = (py - qy) / radius;
        o[2] = (pz - qz) / radius;
    }
    // the distance output of a fixed-radius model is NOT rescaled (reference source/points_to_surf_eval.py:180,188:
    // `if not fixed_radius: pred *= patch_radius`): the per-query scale the decoder tail multiplies with is 1
    if (lane == 0 && radius_out) radius_out[w] = 1.0f;
}

// rand(3) -> rotation matrices from per-query word positions (GT-query pass of a fixed-radius model)
__global__ __launch_bounds__(256) void ball_rot_words_kernel(const uint32_t *__restrict__ words, const long long *__restrict__ tpos,
                                                             long long n, uint32_t *__restrict__ six, long long alloc_words,
                                                             const long long *__restrict__ meta) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // a sticky generator error (meta[1] != 0): the chain kernel returned before writing tpos -- nothing to read
    const long long t = meta[1] != 0 ? -1 : tpos[i];
    const bool ok = t >= 0 && t + 6 <= alloc_words;
#pragma unroll
    for (int j = 0; j < 6; ++j) six[6 * i + j] = ok ? words[t + j] : 0u;
}

int ball_ws_reserve(p2s_rng_s *r, size_t bytes) {
    if (bytes <= r->ball_ws_bytes) return P2S_OK;
    if (r->ball_ws) {
        P2S_HIP_CHECK(hipDeviceSynchronize());
        (void)hipFree(r->ball_ws);
        r->ball_ws = nullptr;
        r->ball_ws_bytes = 0;
    }
    bytes += bytes / 4;
    if (hipMalloc(&r->ball_ws, bytes) != hipSuccess) {
        (void)hipGetLastError();
        p2s_set_error("fixed-radius patches: hipMalloc of %zu bytes of work space failed", bytes);
        return P2S_ENOMEM;
    }
    r->ball_ws_bytes = bytes;
    return P2S_OK;
}

}  // namespace

void p2s_ball_free_rng(p2s_rng_s *r) {
    if (r->ball_ws) (void)hipFree(r->ball_ws);
    if (r->ball_counts_host) (void)hipHostFree(r->ball_counts_host);
    if (r->ball_counts_dev) (void)hipFree(r->ball_counts_dev);
}

extern "C" int p2s_kd_order_host(const float *pts_host, int64_t n, int leafsize, int32_t *order_out, int32_t *leaf_start_out,
                                 int64_t leaf_cap, int32_t *n_leaves_out) {
    if (!pts_host || n < 1 || n > 0x7fffffff / 4 || leafsize < 1 || !order_out) {
        p2s_set_error("p2s_kd_order_host: bad argument");
        return P2S_EINVAL;
    }
    std::vector<double> data((size_t)n * 3);
    for (size_t i = 0; i < (size_t)n * 3; ++i) data[i] = (double)pts_host[i];
    for (int i = 0; i < (int)n; ++i) order_out[i] = i;
    std::vector<int> leaves;
    KdBuild kb{data.data(), order_out, leafsize, &leaves};
    kb.node(0, (int)n);
    if (n_leaves_out) *n_leaves_out = (int32_t)leaves.size();
    if (leaf_start_out) {
        if ((int64_t)leaves.size() + 1 > leaf_cap) {
            p2s_set_error("p2s_kd_order_host: %zu leaves do not fit the caller's array (%lld)", leaves.size(), (long long)leaf_cap);
            return P2S_ECAPACITY;
        }
        for (size_t i = 0; i < leaves.size(); ++i) leaf_start_out[i] = leaves[i];
        leaf_start_out[leaves.size()] = (int32_t)n;
    }
    return P2S_OK;
}

// tree order + leaf boxes of the handle's cloud, on first use (blocking: D2H of the points, host build, upload)
int p2s_cloud_kd_prepare(p2s_cloud_s *c) {
    if (c->kd_blob) return P2S_OK;
    P2S_HIP_CHECK(hipSetDevice(c->device));
    const int n = c->d.n;
    for (int i = 0; i < c->n_streams; ++i) P2S_HIP_CHECK(hipStreamSynchronize(c->streams[i]));
    if (c->many_streams) P2S_HIP_CHECK(hipDeviceSynchronize());
    std::vector<float> h((size_t)n * 3);
    P2S_HIP_CHECK(hipMemcpy(h.data(), c->pts, (size_t)n * 12, hipMemcpyDeviceToHost));
    std::vector<int32_t> order(n), leaf((size_t)n + 2);
    int32_t L = 0;
    int rc = p2s_kd_order_host(h.data(), n, 1000, order.data(), leaf.data(), (int64_t)n + 2, &L);   // data_loader.py:40-42
    if (rc) return rc;
    std::vector<float> kp((size_t)n * 4), box((size_t)L * 6);
    for (int i = 0; i < n; ++i) {
        const int id = order[i];
        kp[4 * (size_t)i + 0] = h[3 * (size_t)id + 0];
        kp[4 * (size_t)i + 1] = h[3 * (size_t)id + 1];
        kp[4 * (size_t)i + 2] = h[3 * (size_t)id + 2];
        std::memcpy(&kp[4 * (size_t)i + 3], &id, 4);
    }
    for (int l = 0; l < L; ++l) {
        float *b = &box[6 * (size_t)l];
        for (int a = 0; a < 3; ++a) b[a] = b[3 + a] = kp[4 * (size_t)leaf[l] + a];
        for (int i = leaf[l] + 1; i < leaf[l + 1]; ++i)
            for (int a = 0; a < 3; ++a) {
                b[a] = std::min(b[a], kp[4 * (size_t)i + a]);
                b[3 + a] = std::max(b[3 + a], kp[4 * (size_t)i + a]);
            }
    }
    const size_t o_leaf = (size_t)n * 16, o_box = o_leaf + (((size_t)L + 1) * 4 + 15) / 16 * 16, total = o_box + (size_t)L * 24;
    char *blob = (char *)p2s_pool_alloc(c->device, total);
    if (!blob) {
        p2s_set_error("fixed-radius patches: device allocation of the tree order failed (%zu bytes)", total);
        return P2S_ENOMEM;
    }
    hipError_t e = hipMemcpy(blob, kp.data(), (size_t)n * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(blob + o_leaf, leaf.data(), ((size_t)L + 1) * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(blob + o_box, box.data(), (size_t)L * 24, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        p2s_pool_free(c->device, blob);
        p2s_set_error("fixed-radius patches: upload of the tree order failed: %s", hipGetErrorString(e));
        return P2S_EHIP;
    }
    c->kd_blob = blob;
    c->kd_pts = (float4 *)blob;
    c->kd_leaf = (int *)(blob + o_leaf);
    c->kd_box = (float *)(blob + o_box);
    c->kd_leaves = L;
    return P2S_OK;
}

static BallTree ball_tree(const p2s_cloud_s *c) { return BallTree{c->kd_pts, c->kd_leaf, c->kd_box, c->kd_leaves}; }

extern "C" int p2s_ball_count(p2s_cloud_t c, const float *q_dev, int64_t nq, double radius, int32_t *count_out_dev, void *stream) {
    if (!c || nq < 0 || (nq > 0 && (!q_dev || !count_out_dev)) || !(radius > 0.0)) {
        p2s_set_error("p2s_ball_count: bad argument");
        return P2S_EINVAL;
    }
    if (nq == 0) return P2S_OK;
    int rc = p2s_cloud_kd_prepare(c);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    p2s_cloud_note_stream(c, s);
    hipLaunchKernelGGL(ball_scan_kernel<false>, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, ball_tree(c), q_dev, (long long)nq,
                       radius * radius, count_out_dev, (int *)nullptr, (const long long *)nullptr);
    P2S_LAUNCH_CHECK("ball_scan_kernel");
    return P2S_OK;
}

// queries [0, nq) with their hit counts known on the host: lists, chain, shuffle, patch.  One batch = one launch set.
int p2s_ball_patch_counted(p2s_rng_s *r, p2s_cloud_s *c, const float *q_dev, const int32_t *count_dev, const int32_t *count_host,
                           int64_t nq, double radius, int k, int tail_words, int32_t *ids_out_dev, float *patch_out_dev,
                           float *radius_out_dev, double *rot_out_dev, hipStream_t s) {
    if (r->levels_max == 0) {
        p2s_set_error("fixed-radius patches: the generator needs the jump-ahead tables (p2s_rng_set_jump_tables)");
        return P2S_EINVAL;
    }
    const long long cap = p2s_rng_session_words(r);
    (void)hipFuncSetAttribute((const void *)ball_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    for (int64_t done = 0; done < nq;) {
        // a batch: <= 16384 queries, hit lists <= 1 GiB, random words within one session
        long long hits = 0, need = 8192;
        int cur = 0;
        while (done + cur < nq && cur < 16384) {
            const long long cq = patch_out_dev ? count_host[done + cur] : 0;      // no hit lists when only advancing
            const long long cw = count_host[done + cur];
            const long long wq = (cw > k ? (long long)(2.2 * (double)cw) + 320 : 0) + tail_words;
            if (cur > 0 && (hits + cq > (1LL << 28) || need + wq > cap)) break;
            hits += cq;
            need += wq;
            ++cur;
        }
        if (need > cap) {
            p2s_set_error("fixed-radius patches: one query with %d points in its ball needs more random words than a session holds",
                          (int)count_host[done]);
            return P2S_ECAPACITY;
        }
        const size_t o_off = 0, o_spos = ((size_t)cur + 1) * 8, o_tpos = o_spos + (size_t)cur * 8, o_six = o_tpos + (size_t)cur * 8,
                     o_list = o_six + (size_t)cur * 24, total = o_list + (size_t)std::max<long long>(hits, 1) * 4;
        int rc = ball_ws_reserve(r, total);
        if (rc) return rc;
        char *ws = (char *)r->ball_ws;
        long long *off = (long long *)(ws + o_off), *spos = (long long *)(ws + o_spos), *tpos = (long long *)(ws + o_tpos);
        uint32_t *six = (uint32_t *)(ws + o_six);
        int *lists = (int *)(ws + o_list);
        rc = p2s_rng_session_raw(r, need, s);
        if (rc) return rc;
        long long *meta = p2s_rng_raw_meta(r);
        const float *qb = q_dev + (size_t)done * 3;
        const int32_t *cb = count_dev + done;
        if (patch_out_dev) {
        hipLaunchKernelGGL(ball_offsets_kernel, dim3(1), dim3(1024), 0, s, cb, cur, off);
        hipLaunchKernelGGL(ball_scan_kernel<true>, dim3((unsigned)((cur + 3) / 4)), dim3(256), 0, s, ball_tree(c), qb, (long long)cur,
                           radius * radius, (int *)nullptr, lists, off);
        }
        // one latency-bound wave next to MFMA-saturated encoders: claim most of a CU's LDS so that it gets a CU of its own
        const size_t lds = cur >= 64 ? (size_t)120 * 1024 : (size_t)BC_RING * 4;
        hipLaunchKernelGGL(ball_chain_kernel, dim3(1), dim3(64), lds, s, r->tmp, cap, cap + 624, cb, cur, k, tail_words, spos,
                           tail_words ? tpos : (long long *)nullptr, meta);
        if (patch_out_dev)
        hipLaunchKernelGGL(ball_patch_kernel, dim3(cur), dim3(64), 0, s, r->tmp, cap, c->pts, qb, cb, off, lists, spos, k, (float)radius,
                           ids_out_dev ? ids_out_dev + (size_t)done * k : (int32_t *)nullptr, patch_out_dev + (size_t)done * k * 3,
                           radius_out_dev ? radius_out_dev + done : (float *)nullptr, meta);
        P2S_LAUNCH_CHECK("fixed-radius patch kernels");
        if (rot_out_dev && tail_words == 6) {
            hipLaunchKernelGGL(ball_rot_words_kernel, dim3((unsigned)((cur + 255) / 256)), dim3(256), 0, s, r->tmp, tpos, (long long)cur, six,
                               cap + 624, meta);
            P2S_LAUNCH_CHECK("ball_rot_words_kernel");
            rc = p2s_rotations_from_words(six, cur, rot_out_dev + (size_t)done * 9, s);
            if (rc) return rc;
        }
        done += cur;
    }
    return P2S_OK;
}

// hit counts of nq queries -> device array + pinned host copy on the generator handle (blocking)
int p2s_ball_counts_to_host(p2s_rng_s *r, p2s_cloud_s *c, const float *q_dev, int64_t nq, double radius, int32_t **count_dev,
                            const int32_t **count_host, hipStream_t s) {
    if ((size_t)nq > r->ball_counts_cap) {
        P2S_HIP_CHECK(hipDeviceSynchronize());
        if (r->ball_counts_host) (void)hipHostFree(r->ball_counts_host);
        if (r->ball_counts_dev) (void)hipFree(r->ball_counts_dev);
        r->ball_counts_host = nullptr;
        r->ball_counts_dev = nullptr;
        r->ball_counts_cap = 0;
        const size_t cap = (size_t)nq + (size_t)nq / 4 + 1024;
        if (hipHostMalloc((void **)&r->ball_counts_host, cap * 4, hipHostMallocDefault) != hipSuccess ||
            hipMalloc((void **)&r->ball_counts_dev, cap * 4) != hipSuccess) {
            (void)hipGetLastError();
            p2s_set_error("fixed-radius patches: allocation of %zu hit counts failed", cap);
            return P2S_ENOMEM;
        }
        r->ball_counts_cap = cap;
    }
    int rc = p2s_ball_count(c, q_dev, nq, radius, r->ball_counts_dev, (void *)s);
    if (rc) return rc;
    P2S_HIP_CHECK(hipMemcpyAsync(r->ball_counts_host, r->ball_counts_dev, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
    P2S_HIP_CHECK(hipStreamSynchronize(s));
    *count_dev = r->ball_counts_dev;
    *count_host = r->ball_counts_host;
    return P2S_OK;
}

extern "C" int p2s_ball_patch(p2s_rng_t r, p2s_cloud_t c, const float *q_dev, int64_t nq, double radius, int points_per_patch,
                              int with_rotation, int32_t *ids_out_dev, float *patch_out_dev, float *radius_out_dev,
                              double *rot_out_dev, void *stream) {
    if (!r || !c || nq < 0 || (nq > 0 && !q_dev) || !(radius > 0.0) || points_per_patch < 1 ||
        points_per_patch > 4096 || (with_rotation && !rot_out_dev) || (ids_out_dev && !patch_out_dev)) {
        p2s_set_error("p2s_ball_patch: bad argument (ids_out_dev needs patch_out_dev)");
        return P2S_EINVAL;
    }
    if (nq == 0) return P2S_OK;
    P2S_HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    p2s_cloud_note_stream(c, s);
    int32_t *cd = nullptr;
    const int32_t *ch = nullptr;
    int rc = p2s_ball_counts_to_host(r, c, q_dev, nq, radius, &cd, &ch, s);
    if (rc) return rc;
    return p2s_ball_patch_counted(r, c, q_dev, cd, ch, nq, radius, points_per_patch, with_rotation ? 6 : 0, ids_out_dev, patch_out_dev,
                                  radius_out_dev, rot_out_dev, s);
}
