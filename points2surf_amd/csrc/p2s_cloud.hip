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

// ---------------------------------------------------------------------------------------------
// a1: query grid
// ---------------------------------------------------------------------------------------------
__global__ void p2s_voxelize_kernel(const float *__restrict__ pts, int n, int res, uint32_t *__restrict__ occ,
                                    long long *__restrict__ totals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int v[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // fp32 exactly as numpy: floor(((p + 1.0) / 2.0) * res)   (source/sdf.py:73-75)
        const float t = (pts[3 * i + a] + 1.0f) / 2.0f;
        v[a] = (int)floorf(t * (float)res);
    }
    if (v[0] < 0 || v[1] < 0 || v[2] < 0 || v[0] >= res || v[1] >= res || v[2] >= res) {
        totals[1] = 1;   // numpy would raise IndexError (or wrap a negative index)
        return;
    }
    const long long lin = ((long long)v[0] * res + v[1]) * res + v[2];
    atomicOr(&occ[lin >> 5], 1u << (lin & 31));
}

struct GridOffsets {
    int n;
    int o[16];
};

__device__ __forceinline__ bool near_surface(const uint32_t *__restrict__ occ, int res, int x, int y, int z,
                                             const GridOffsets &go) {
    // box filter of a 0/1 volume with edge replication == OR over the index-clamped neighbourhood
    for (int ix = 0; ix < go.n; ++ix) {
        const int xx = min(max(x + go.o[ix], 0), res - 1);
        for (int iy = 0; iy < go.n; ++iy) {
            const int yy = min(max(y + go.o[iy], 0), res - 1);
            const long long row = ((long long)xx * res + yy) * res;
            for (int iz = 0; iz < go.n; ++iz) {
                const int zz = min(max(z + go.o[iz], 0), res - 1);
                const long long lin = row + zz;
                if ((occ[lin >> 5] >> (lin & 31)) & 1u) return true;
            }
        }
    }
    return false;
}

// pass 0: per-block counts; pass 1: ordered write using the scanned block offsets
template <int PASS>
__global__ __launch_bounds__(256) void p2s_grid_compact_kernel(const uint32_t *__restrict__ occ, int res,
                                                               GridOffsets go, int *__restrict__ blk_cnt,
                                                               const long long *__restrict__ blk_off,
                                                               float *__restrict__ q_out, long long capacity) {
    __shared__ int wsum[4];
    const int rm = res - 1;                                   // the reference drops the last slab: [:-1,:-1,:-1]
    const long long total = (long long)rm * rm * rm;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    bool flag = false;
    int x = 0, y = 0, z = 0;
    if (i < total) {
        z = (int)(i % rm);
        const long long t = i / rm;
        y = (int)(t % rm);
        x = (int)(t / rm);
        flag = near_surface(occ, res, x, y, z, go);
    }
    const unsigned long long m = __ballot(flag);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    if (PASS == 0) {
        if (threadIdx.x == 0) blk_cnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        return;
    }
    if (!flag) return;
    int before = __popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) before += wsum[w];
    const long long dst = blk_off[blockIdx.x] + before;
    if (dst >= capacity) return;
    // centre: float32(((idx + 0.5) / res) * 2 - 1) evaluated in float64 (source/sdf.py:78-79,70)
    q_out[3 * dst + 0] = (float)((((double)x + 0.5) / (double)res) * 2.0 - 1.0);
    q_out[3 * dst + 1] = (float)((((double)y + 0.5) / (double)res) * 2.0 - 1.0);
    q_out[3 * dst + 2] = (float)((((double)z + 0.5) / (double)res) * 2.0 - 1.0);
}

// exclusive scan of the block counts (one workgroup, sequential over chunks of 1024)
__global__ __launch_bounds__(1024) void p2s_scan_blocks_kernel(const int *__restrict__ cnt, long long nblk,
                                                               long long *__restrict__ off,
                                                               long long *__restrict__ totals) {
    __shared__ long long part[16];
    __shared__ long long carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (long long base = 0; base < nblk; base += 1024) {
        const long long i = base + threadIdx.x;
        const long long v = (i < nblk) ? cnt[i] : 0;
        long long s = v;
        for (int d = 1; d < 64; d <<= 1) {
            const long long t = __shfl_up(s, d);
            if (lane >= d) s += t;
        }
        if (lane == 63) part[wave] = s;
        __syncthreads();
        long long wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += part[w];
        if (i < nblk) off[i] = carry + wbase + s - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += wbase + s;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[0] = carry;
}

// ---------------------------------------------------------------------------------------------
// a4/a5: exact k-nearest neighbours (float64 ranking) + radius + patch space, one wave per query
// ---------------------------------------------------------------------------------------------
constexpr int KNN_CAP = 2048;

__device__ __forceinline__ int sat_at(const CloudDev &c, int x, int y, int z) {
    const int G1 = c.G + 1;
    return c.sat[(x * G1 + y) * G1 + z];
}
// number of points in cells [lo, hi] (inclusive)
__device__ __forceinline__ int box_count(const CloudDev &c, const int lo[3], const int hi[3]) {
    const int x0 = lo[0], y0 = lo[1], z0 = lo[2], x1 = hi[0] + 1, y1 = hi[1] + 1, z1 = hi[2] + 1;
    return sat_at(c, x1, y1, z1) - sat_at(c, x0, y1, z1) - sat_at(c, x1, y0, z1) - sat_at(c, x1, y1, z0) +
           sat_at(c, x0, y0, z1) + sat_at(c, x0, y1, z0) + sat_at(c, x1, y0, z0) - sat_at(c, x0, y0, z0);
}

// single-wave bitonic sort of (key, id) pairs in LDS; n2 = power of two
__device__ void wave_bitonic_sort(unsigned long long *keys, int *ids, int n2, int lane) {
    for (int size = 2; size <= n2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = lane; i < (n2 >> 1); i += 64) {
                const int a = 2 * i - (i & (stride - 1));
                const int b = a + stride;
                const bool up = (a & size) == 0;
                const unsigned long long ka = keys[a], kb = keys[b];
                const int ia = ids[a], ib = ids[b];
                const bool gt = (ka > kb) || (ka == kb && ia > ib);
                if (gt == up) {
                    keys[a] = kb; keys[b] = ka;
                    ids[a] = ib; ids[b] = ia;
                }
            }
        }
    }
    __syncthreads();
}

struct KnnList {
    unsigned long long *keys;
    int *ids;
    int len;
    double thr2;
    int k;
};

// sort the list; keep the k smallest; tighten the acceptance threshold
__device__ void knn_prune(KnnList &L, int lane) {
    int n2 = 64;
    while (n2 < L.len) n2 <<= 1;
    __syncthreads();
    for (int i = L.len + lane; i < n2; i += 64) {
        L.keys[i] = ~0ull;
        L.ids[i] = 0x7fffffff;
    }
    wave_bitonic_sort(L.keys, L.ids, n2, lane);
    if (L.len >= L.k) {
        L.len = L.k;
        L.thr2 = __longlong_as_double((long long)L.keys[L.k - 1]);
    }
}

// The k smallest (distance, id) pairs of the list as a SET, moved to its front in list order -- no sort.  What the
// pipeline needs: the encoders max-pool over the patch, so the order of its points changes no bit of the result; only
// the API that hands out ids keeps the sorted order (knn_prune).  Bisection on the 64-bit distance patterns for a value
// that separates the k-th from the (k+1)-th smallest: about log2(len) + 2 steps of len / 64 LDS reads each, where the
// bitonic network costs ~45 barriers and 1440 LDS accesses for 512 entries.  A tie at the k-th distance (duplicate
// points) falls back to the sort, whose id tie-break is the reference's.  thr2 = the separating value (an upper bound of
// the k-th distance of everything scanned so far).
__device__ void knn_select(KnnList &L, int lane) {
    __syncthreads();
    const int len = L.len, k = L.k;
    if (len < k) return;
    unsigned long long kmin = ~0ull, kmax = 0ull;
    for (int i = lane; i < len; i += 64) {
        const unsigned long long v = L.keys[i];
        kmin = v < kmin ? v : kmin;
        kmax = v > kmax ? v : kmax;
    }
    for (int d = 32; d > 0; d >>= 1) {
        const unsigned long long a = __shfl_xor(kmin, d), b = __shfl_xor(kmax, d);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
    }
    unsigned long long T = kmax;
    if (len > k) {
        // invariant: count(<= lo) < k <= count(<= hi)
        unsigned long long lo = kmin - 1ull, hi = kmax;         // kmin >= 0 as a pattern; kmin - 1 wraps only for d2 = +0.0
        bool found = false;
        if (kmin == 0ull) {                                      // (a query on top of a point): count(<= 0) may already be >= k
            int c0 = 0;
            for (int i = lane; i < len; i += 64) c0 += L.keys[i] == 0ull ? 1 : 0;
            for (int d = 32; d > 0; d >>= 1) c0 += __shfl_xor(c0, d);
            if (c0 >= k) {
                hi = 0ull;
                lo = 0ull;
                found = c0 == k;
                T = 0ull;
            } else {
                lo = 0ull;
            }
        }
        while (!found && hi - lo > 1ull) {
            const unsigned long long mid = lo + ((hi - lo) >> 1);
            int cnt = 0;
            for (int i = lane; i < len; i += 64) cnt += L.keys[i] <= mid ? 1 : 0;
            for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d);
            if (cnt == k) {
                T = mid;
                found = true;
            } else if (cnt > k) {
                hi = mid;
            } else {
                lo = mid;
            }
        }
        if (!found) {                                            // several entries share the k-th distance
            knn_prune(L, lane);
            return;
        }
        // ordered in-place compaction of the entries <= T (a chunk is read before anything is written over it)
        int out = 0;
        for (int base = 0; base < len; base += 64) {
            const int i = base + lane;
            unsigned long long v = 0ull;
            int id = 0;
            bool keep = false;
            if (i < len) {
                v = L.keys[i];
                id = L.ids[i];
                keep = v <= T;
            }
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const int o = out + __popcll(m & ((1ull << lane) - 1ull));
                L.keys[o] = v;
                L.ids[o] = id;
            }
            out += __popcll(m);
        }
        L.len = k;
    }
    L.thr2 = __longlong_as_double((long long)T);
    __syncthreads();
}

// scan cells [lo, hi]; if has_ex, cells inside [exlo, exhi] were scanned before and are skipped.
// The box is a set of z-contiguous cell runs, one per (x, y) column (two where the column crosses the excluded box).
// Walking them one after the other costs two dependent L2 round trips per run for ~10 points (r02: 49 runs per query,
// most lanes idle).  Instead: up to 64 runs at a time, one LANE per run fetches its point range, a wave scan turns the
// run lengths into offsets, and all 64 lanes then walk the FLATTENED point list of the batch (the run of a flat index
// is found by a 6-step binary search over the offsets in LDS).  The order candidates enter the list in does not matter:
// every consumer sorts by (distance, id).
__device__ void knn_scan(const CloudDev &c, KnnList &L, const int lo[3], const int hi[3], bool has_ex,
                         const int exlo[3], const int exhi[3], double qx, double qy, double qz, int lane,
                         int *run_start, int *run_off) {
    const int G = c.G;
    const int nx = hi[0] - lo[0] + 1, ny = hi[1] - lo[1] + 1;
    const int ncol = nx * ny;
    // run index r -> column r >> 1, segment r & 1 (segment 1 only exists for columns inside the xy-exclusion)
    for (int r0 = 0; r0 < 2 * ncol; r0 += 64) {
        const int r = r0 + lane;
        int start = 0, cnt = 0;
        if (r < 2 * ncol) {
            const int col = r >> 1, seg = r & 1;
            const int cx = lo[0] + col / ny, cy = lo[1] + col % ny;
            const bool inside_xy = has_ex && cx >= exlo[0] && cx <= exhi[0] && cy >= exlo[1] && cy <= exhi[1];
            int za = lo[2], zb = hi[2];
            if (inside_xy) {
                if (seg == 0) zb = exlo[2] - 1;
                else za = exhi[2] + 1;
            } else if (seg == 1) {
                zb = za - 1;                                  // no second segment
            }
            if (za <= zb) {
                const int rowbase = (cx * G + cy) * G;
                start = c.cell_start[rowbase + za];
                cnt = c.cell_start[rowbase + zb + 1] - start;
            }
        }
        // exclusive scan of the run lengths
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        const int total = __shfl(incl, 63);
        __syncthreads();                                      // the previous batch's readers are done
        run_start[lane] = start;
        run_off[lane] = incl - cnt;
        __syncthreads();
        for (int f0 = 0; f0 < total; f0 += 64) {
            const int f = f0 + lane;
            bool keep = false;
            unsigned long long key = 0;
            int id = 0;
            if (f < total) {
                int a = 0;                                    // last run whose offset is <= f (empty runs share offsets:
#pragma unroll                                                //  the search lands on the last of them, the non-empty one)
                for (int step = 32; step >= 1; step >>= 1)
                    if (a + step < 64 && run_off[a + step] <= f) a += step;
                const float4 p = c.spts[run_start[a] + (f - run_off[a])];
                const double dx = qx - (double)p.x, dy = qy - (double)p.y, dz = qz - (double)p.z;
                const double d2 = dx * dx + dy * dy + dz * dz;
                keep = d2 <= L.thr2;
                key = (unsigned long long)__double_as_longlong(d2);
                id = __float_as_int(p.w);
            }
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const int off = L.len + __popcll(m & ((1ull << lane) - 1ull));
                L.keys[off] = key;
                L.ids[off] = id;
            }
            L.len += __popcll(m);
            if (L.len > KNN_CAP - 64) knn_prune(L, lane);
        }
    }
}

// SORTED: ids / patch rows in ascending distance (the API's contract).  !SORTED (the per-shape pipeline): the same k
// points in list order, selected without sorting (knn_select).
template <bool SORTED>
__global__ __launch_bounds__(64) void p2s_knn_kernel(CloudDev c, const float *__restrict__ queries, long long nq,
                                                     int k, int *__restrict__ ids_out,
                                                     float *__restrict__ patch_out,
                                                     float *__restrict__ radius_out) {
    __shared__ unsigned long long keys[KNN_CAP];
    __shared__ int lids[KNN_CAP];
    __shared__ int run_start[64], run_off[64];
    const int lane = threadIdx.x;
    const int G = c.G;
    for (long long qi = blockIdx.x; qi < nq; qi += gridDim.x) {
        const float qxf = queries[3 * qi + 0], qyf = queries[3 * qi + 1], qzf = queries[3 * qi + 2];
        const double qx = qxf, qy = qyf, qz = qzf;
        const int cq[3] = {cell_coord(qxf, c.lo[0], c.inv_cell, G), cell_coord(qyf, c.lo[1], c.inv_cell, G),
                           cell_coord(qzf, c.lo[2], c.inv_cell, G)};
        // smallest cube of cells around the query's cell that holds >= k points (O(1) per try via the SAT)
        int lo[3], hi[3];
        for (int rho = 0;; ++rho) {
            bool all = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                lo[a] = max(cq[a] - rho, 0);
                hi[a] = min(cq[a] + rho, G - 1);
                all = all && lo[a] == 0 && hi[a] == G - 1;
            }
            if (all || box_count(c, lo, hi) >= k) break;
        }
        KnnList L{keys, lids, 0, INFINITY, k};
        __syncthreads();
        knn_scan(c, L, lo, hi, false, lo, hi, qx, qy, qz, lane, run_start, run_off);
        // exact k-th distance among the cube's points (or a value just above it): an upper bound of the true one
        if (SORTED) knn_prune(L, lane);
        else knn_select(L, lane);
        // every point within sqrt(thr2) of q lies in cells [lo2, hi2] (conservative: radius rounded up, and
        // cell_coord is monotone)
        const float r = (float)sqrt(L.thr2) * 1.00001f + 1e-30f;
        int lo2[3], hi2[3];
        const float qf[3] = {qxf, qyf, qzf};
        bool grow = false;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo2[a] = min(lo[a], cell_coord(qf[a] - r, c.lo[a], c.inv_cell, G));
            hi2[a] = max(hi[a], cell_coord(qf[a] + r, c.lo[a], c.inv_cell, G));
            grow = grow || lo2[a] != lo[a] || hi2[a] != hi[a];
        }
        if (grow) {
            const int before = L.len;
            knn_scan(c, L, lo2, hi2, true, lo, hi, qx, qy, qz, lane, run_start, run_off);
            if (L.len != before) {
                if (SORTED) knn_prune(L, lane);
                else knn_select(L, lane);
            }
        }
        __syncthreads();
        // ---- outputs: ids (ascending distance), r = max ||q - p||_2 (fp32, numpy op order), (p - q) / r ----
        float smax = 0.0f;
        for (int j = lane; j < k; j += 64) {
            int id = lids[j];
            if ((unsigned)id >= (unsigned)c.n) id = 0;      // a non-finite query finds no candidates: stay inside the cloud
            if (ids_out) ids_out[qi * k + j] = id;
            const float dx = qxf - c.pts[3 * id + 0];
            const float dy = qyf - c.pts[3 * id + 1];
            const float dz = qzf - c.pts[3 * id + 2];
            const float s = (dx * dx + dy * dy) + dz * dz;      // contraction is off: three roundings + two
            smax = fmaxf(smax, s);
        }
        for (int d = 32; d > 0; d >>= 1) smax = fmaxf(smax, __shfl_xor(smax, d));
        const float rad = sqrtf(smax);   // sqrt is monotone: max_i sqrt(s_i) == sqrt(max_i s_i)
        if (radius_out && lane == 0) radius_out[qi] = rad;
        if (patch_out) {
            for (int j = lane; j < k; j += 64) {
                int id = lids[j];
                if ((unsigned)id >= (unsigned)c.n) id = 0;
                float *dst = patch_out + (qi * k + j) * 3;
                dst[0] = (c.pts[3 * id + 0] - qxf) / rad;
                dst[1] = (c.pts[3 * id + 1] - qyf) / rad;
                dst[2] = (c.pts[3 * id + 2] - qzf) / rad;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// a6: numpy legacy RandomState.randint(0, N, size) on the device: MT19937 + masked rejection +
// ordered compaction.  One workgroup walks the stream block by block (the recurrence is serial
// across 624-word blocks; inside a block it has three internally parallel phases).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mt_mix(uint32_t u, uint32_t v) {
    const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// Out-of-place twist of one 624-word block by ONE wave.  With lane j-mapping j = 64*it + lane the three
// dependent phases of the recurrence chain through the lane's own registers
//   new[j] -> new[227+j] -> new[454+j]
// so every LDS read is from the old block (independent, issued back to back): no dependent LDS round trip.
__device__ __forceinline__ void mt_twist_wave(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int lane) {
    uint32_t v1[4], v2[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int j = 64 * it + lane;
        v1[it] = 0;
        if (j < 227) {
            v1[it] = src[j + 397] ^ mt_mix(src[j], src[j + 1]);
            dst[j] = v1[it];
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int j = 64 * it + lane;
        v2[it] = 0;
        if (j < 227) {
            v2[it] = v1[it] ^ mt_mix(src[227 + j], src[228 + j]);
            dst[227 + j] = v2[it];
        }
    }
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int j = 64 * it + lane;
        if (j < 169) dst[454 + j] = v2[it] ^ mt_mix(src[454 + j], src[455 + j]);
    }
    // new[623] = new[396] ^ mix(old[623], new[0]);  new[396] = v2 of j = 169 (it 2, lane 41), new[0] = v1 of j = 0
    const uint32_t n396 = __builtin_amdgcn_readlane(v2[2], 41);
    const uint32_t n0 = __builtin_amdgcn_readlane(v1[0], 0);
    if (lane == 0) dst[623] = n396 ^ mt_mix(src[623], n0);
}

// Two-wave pipeline: wave 0 twists block b+1 (out of place) while wave 1 tempers / mask-rejects /
// compacts block b into the output.  One workgroup barrier per 624-word block.  The waves run at raised
// priority: the kernel shares its CU with MFMA-saturated encoder waves and is pure latency.
__global__ __launch_bounds__(128) void p2s_mt_randint_kernel(uint32_t *__restrict__ state, uint32_t rng,
                                                             uint32_t mask, long long target,
                                                             int32_t *__restrict__ out) {
    __shared__ uint32_t st[2][624];
    __shared__ uint32_t stage[640];
    __shared__ int s_done[2];
    __shared__ int s_pos;
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 624; i += 128) st[0][i] = state[i];
    int pos = (int)state[624];
    if (tid == 0) {
        s_done[0] = 0;
        s_done[1] = 0;
        s_pos = 624;
    }
    __syncthreads();
    int cur = 0, start = pos;
    if (pos >= 624) {                       // numpy: "needs twist before the first draw"
        if (wave == 0) mt_twist_wave(st[0], st[1], lane);
        __syncthreads();
        cur = 1;
        start = 0;
    }
    long long produced = 0;                 // meaningful in the consumer wave only
    for (int iter = 0;; ++iter) {
        if (wave == 0) {
            mt_twist_wave(st[cur], st[cur ^ 1], lane);
        } else {
            // lane l owns words 10l .. 10l+9 (contiguous -> ordered compaction by an exclusive lane scan).
            // Branch-free: all 10 LDS reads are issued back to back; accepted words are compacted through
            // an LDS staging buffer and leave as coalesced 256-byte stores.
            const uint32_t *src = st[cur];
            uint32_t w[10];
            unsigned okmask = 0;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int idx = 10 * lane + j;
                w[j] = src[idx < 624 ? idx : 623];
            }
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int idx = 10 * lane + j;
                w[j] = mt_temper(w[j]) & mask;
                const unsigned ok = (idx < 624) & (idx >= start) & (w[j] <= rng);
                okmask |= ok << j;
            }
            const int c = __popc(okmask);
            // exclusive prefix of c (<= 10, 4 bits) over the lanes without any LDS traffic: one ballot +
            // mbcnt per bit plane (a shuffle scan would be six dependent ds_bpermute round trips)
            int excl = 0, total = 0;
#pragma unroll
            for (int bit = 0; bit < 4; ++bit) {
                const unsigned long long m = __ballot((c >> bit) & 1);
                excl += (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)) << bit;
                total += __popcll(m) << bit;
            }
            const long long need = target - produced;          // > 0
            int r = excl;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                if (okmask & (1u << j)) stage[r] = w[j];
                r += (okmask >> j) & 1u;
            }
            const int lim = (total < need) ? total : (int)need;
            {
                uint32_t sv[10];
#pragma unroll
                for (int it = 0; it < 10; ++it) sv[it] = stage[64 * it + lane];      // batched LDS reads
#pragma unroll
                for (int it = 0; it < 10; ++it)
                    if (out && 64 * it + lane < lim) out[produced + 64 * it + lane] = (int32_t)sv[it];
            }
            if (total >= need) {
                // the stream resumes after the word holding the need-th accepted value
                if (excl < need && need <= excl + c) {
                    int left = (int)need - excl;
                    int pos_end = 0;
#pragma unroll
                    for (int j = 0; j < 10; ++j) {
                        if ((okmask >> j) & 1u) {
                            if (--left == 0) pos_end = 10 * lane + j + 1;
                        }
                    }
                    s_pos = pos_end;
                }
                if (lane == 0) s_done[iter & 1] = 1;
                produced = target;
            } else {
                produced += total;
            }
        }
        // LDS-only synchronisation: __syncthreads() would add s_waitcnt vmcnt(0) and stall every block on
        // the completion of its (fire-and-forget) id stores
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s_done[iter & 1]) break;        // parity-indexed: the consumer may already be one block ahead
        cur ^= 1;
        start = 0;
    }
    // the block in which the target was reached stays the current block (the twister's look-ahead went to
    // the other buffer)
    for (int i = tid; i < 624; i += 128) state[i] = st[cur][i];
    if (tid == 0) state[624] = (uint32_t)s_pos;
}

__global__ void p2s_gather_kernel(const float *__restrict__ pts, const int32_t *__restrict__ ids, long long n,
                                  int n_points, float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int id = ids[i];
    if (id < 0) {            // zero padding of the N < sub_sample_size branch (reference source/base/utils.py:225-226)
        out[3 * i + 0] = out[3 * i + 1] = out[3 * i + 2] = 0.0f;
        return;
    }
    id = min(id, n_points - 1);
    out[3 * i + 0] = pts[3 * id + 0];
    out[3 * i + 1] = pts[3 * id + 1];
    out[3 * i + 2] = pts[3 * id + 2];
}

// ---------------------------------------------------------------------------------------------
// a6, clouds with FEWER points than the sub-sample size (reference source/base/utils.py:221-226):
//     pts_shuffled = pts_ms[:, :3]; rng.shuffle(pts_shuffled); pad with zeros
// The view is shuffled IN PLACE: shape.pts itself is permuted by every query, under the kd-tree (which holds its own
// float64 copy), so the patch of a later query gathers pts[knn ids] from the permuted array (Appendix A of
// SURVEY.md).  Reproduced literally: `perm` (device, persistent per cloud) maps the row of shape.pts to the original
// point; per query the state before the shuffle goes to perm_before (patch gather), the state after it is the
// sub-sample (+ -1 padding).  numpy legacy shuffle of a 2-D array: for i = n-1 .. 1: j = rk_interval(i) (masked
// rejection on 32-bit words); swap rows i, j.  One wave: lane 0 walks, all lanes twist.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void p2s_shuffle_pad_kernel(uint32_t *__restrict__ state, int *__restrict__ perm, int n,
                                                             long long nq, int n_sel, int *__restrict__ perm_before,
                                                             int *__restrict__ ids_out) {
    __shared__ uint32_t st[2][624];
    __shared__ int sp[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 624; i += 64) st[0][i] = state[i];
    for (int i = lane; i < n; i += 64) sp[i] = perm[i];
    int pos = (int)state[624];
    int cur = 0;
    __syncthreads();
    for (long long q = 0; q < nq; ++q) {
        if (perm_before)
            for (int i = lane; i < n; i += 64) perm_before[q * n + i] = sp[i];
        int i = n - 1;
        while (i >= 1) {                                  // uniform: i and pos are broadcast from lane 0
            if (pos >= 624) {
                mt_twist_wave(st[cur], st[cur ^ 1], lane);
                cur ^= 1;
                pos = 0;
                __syncthreads();
            }
            if (lane == 0) {
                // consume words of the current block until it is exhausted or the shuffle is done
                while (i >= 1 && pos < 624) {
                    uint32_t mask = (uint32_t)i;
                    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
                    const uint32_t v = mt_temper(st[cur][pos++]) & mask;
                    if (v <= (uint32_t)i) {
                        if ((int)v != i) {
                            const int t = sp[v];
                            sp[v] = sp[i];
                            sp[i] = t;
                        }
                        --i;
                    }
                }
            }
            i = __shfl(i, 0);
            pos = __shfl(pos, 0);
            __syncthreads();
        }
        if (ids_out) {
            for (int k = lane; k < n_sel; k += 64) ids_out[q * n_sel + k] = k < n ? sp[k] : -1;
        }
    }
    for (int i = lane; i < 624; i += 64) state[i] = st[cur][i];
    for (int i = lane; i < n; i += 64) perm[i] = sp[i];
    if (lane == 0) state[624] = (uint32_t)pos;
}

// patch from explicit kNN ids, rows looked up through the current permutation of shape.pts (NULL = identity):
// r = max ||q - p||_2 and (p - q) / r exactly as p2s_knn_kernel computes them
__global__ __launch_bounds__(64) void p2s_patch_from_ids_kernel(const float *__restrict__ pts, const int *__restrict__ ids,
                                                                const int *__restrict__ perm_before, int n,
                                                                const float *__restrict__ queries, long long nq, int k,
                                                                float *__restrict__ patch_out, float *__restrict__ radius_out) {
    const int lane = threadIdx.x;
    for (long long qi = blockIdx.x; qi < nq; qi += gridDim.x) {
        const float qxf = queries[3 * qi + 0], qyf = queries[3 * qi + 1], qzf = queries[3 * qi + 2];
        float smax = 0.0f;
        for (int j = lane; j < k; j += 64) {
            int id = ids[qi * k + j];
            if (perm_before) id = perm_before[qi * n + id];
            const float dx = qxf - pts[3 * id + 0], dy = qyf - pts[3 * id + 1], dz = qzf - pts[3 * id + 2];
            smax = fmaxf(smax, (dx * dx + dy * dy) + dz * dz);
        }
        for (int d = 32; d > 0; d >>= 1) smax = fmaxf(smax, __shfl_xor(smax, d));
        const float rad = sqrtf(smax);
        if (radius_out && lane == 0) radius_out[qi] = rad;
        if (patch_out) {
            for (int j = lane; j < k; j += 64) {
                int id = ids[qi * k + j];
                if (perm_before) id = perm_before[qi * n + id];
                float *dst = patch_out + (qi * k + j) * 3;
                dst[0] = (pts[3 * id + 0] - qxf) / rad;
                dst[1] = (pts[3 * id + 1] - qyf) / rad;
                dst[2] = (pts[3 * id + 2] - qzf) / rad;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// a2: cell index built ON THE DEVICE (replaces cKDTree(pts, leaf_size=1000), source/data_loader.py:40-42)
//   bbox + finite check -> [one 32-byte read-back: the only blocking call] -> cell ids + histogram -> 3-D summed-area
//   table (three axis scans) -> cell_start derived from the SAT -> scatter -> in-cell rank by original id
// The result is the STABLE counting sort of the points by cell (original order inside a cell), i.e. independent of the
// order the atomics retire in: bit-identical to oracle/cloud_index_oracle.py.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f32_ordered(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ inline float f32_from_ordered(uint32_t o) {
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// rec[0..2] = min (ordered encoding), rec[3..5] = max, rec[6] = lowest index of a non-finite point (or 0xffffffff)
__global__ __launch_bounds__(256) void p2s_bbox_kernel(const float *__restrict__ pts, int n, uint32_t *__restrict__ rec) {
    __shared__ uint32_t red[4][7];
    uint32_t lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u}, bad = 0xffffffffu;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = pts[3 * (size_t)i + a];
            if (!(v == v) || isinf(v)) bad = min(bad, (uint32_t)i);
            const uint32_t o = f32_ordered(v);
            lo[a] = min(lo[a], o);
            hi[a] = max(hi[a], o);
        }
    }
    for (int d = 32; d > 0; d >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = min(lo[a], (uint32_t)__shfl_xor((int)lo[a], d));
            hi[a] = max(hi[a], (uint32_t)__shfl_xor((int)hi[a], d));
        }
        bad = min(bad, (uint32_t)__shfl_xor((int)bad, d));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        for (int a = 0; a < 3; ++a) {
            red[wave][a] = lo[a];
            red[wave][3 + a] = hi[a];
        }
        red[wave][6] = bad;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const int j = threadIdx.x;
        uint32_t v = red[0][j];
        for (int w = 1; w < 4; ++w) v = (j >= 3 && j < 6) ? max(v, red[w][j]) : min(v, red[w][j]);
        if (j >= 3 && j < 6) atomicMax(&rec[j], v);
        else atomicMin(&rec[j], v);
    }
}

struct CellGeom {
    float lo[3];
    float inv;
    int G;
};

__global__ __launch_bounds__(256) void p2s_cell_hist_kernel(const float *__restrict__ pts, int n, CellGeom g,
                                                            int *__restrict__ cid, int *__restrict__ cnt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int cx = cell_coord(pts[3 * (size_t)i + 0], g.lo[0], g.inv, g.G);
    const int cy = cell_coord(pts[3 * (size_t)i + 1], g.lo[1], g.inv, g.G);
    const int cz = cell_coord(pts[3 * (size_t)i + 2], g.lo[2], g.inv, g.G);
    const int c = (cx * g.G + cy) * g.G + cz;
    cid[i] = c;
    atomicAdd(&cnt[c], 1);
}

// SAT pass 1 (z): one wave per (x, y) row of the count grid; inclusive scan along z into sat[x+1][y+1][1..G]
__global__ __launch_bounds__(256) void p2s_sat_z_kernel(const int *__restrict__ cnt, int G, int *__restrict__ sat) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= G * G) return;
    const int x = row / G, y = row % G;
    const int G1 = G + 1;
    int carry = 0;
    for (int z0 = 0; z0 < G; z0 += 64) {
        const int z = z0 + lane;
        const int v = z < G ? cnt[(size_t)row * G + z] : 0;
        int sacc = v;
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(sacc, d);
            if (lane >= d) sacc += t;
        }
        if (z < G) sat[((size_t)(x + 1) * G1 + (y + 1)) * G1 + z + 1] = carry + sacc;
        carry += __shfl(sacc, 63);
    }
}
// SAT passes 2 / 3: running sums along y (stride G1) / x (stride G1^2); one thread per line, consecutive threads =
// consecutive z -> coalesced
__global__ __launch_bounds__(256) void p2s_sat_axis_kernel(int *__restrict__ sat, int G, int axis) {
    const int G1 = G + 1;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= G * G) return;
    const int z = t % G + 1, o = t / G + 1;            // o = x (axis 1: scan y) or y (axis 0: scan x)
    size_t base, stride;
    if (axis == 1) {
        base = ((size_t)o * G1) * G1 + z;
        stride = G1;
    } else {
        base = ((size_t)o) * G1 + z;
        stride = (size_t)G1 * G1;
    }
    int run = 0;
    for (int j = 1; j <= G; ++j) {
        run += sat[base + j * stride];
        sat[base + j * stride] = run;
    }
}
// cell_start[c] = number of points in cells with a smaller linear index = three box counts of the SAT
__global__ __launch_bounds__(256) void p2s_cell_start_kernel(const int *__restrict__ sat, int G, int n,
                                                             int *__restrict__ cell_start, int *__restrict__ fill) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int ncell = G * G * G;
    if (c > ncell) return;
    if (c == ncell) {
        cell_start[c] = n;
        return;
    }
    const int G1 = G + 1;
    const int z = c % G, y = (c / G) % G, x = c / (G * G);
    auto S = [&](int a, int b, int d) { return sat[((size_t)a * G1 + b) * G1 + d]; };
    const int before = S(x, G, G) + (S(x + 1, y, G) - S(x, y, G)) +
                       (S(x + 1, y + 1, z) - S(x, y + 1, z) - S(x + 1, y, z) + S(x, y, z));
    cell_start[c] = before;
    fill[c] = before;
}
__global__ __launch_bounds__(256) void p2s_cell_scatter_kernel(const float *__restrict__ pts, const int *__restrict__ cid, int n,
                                                               int *__restrict__ fill, float4 *__restrict__ tmp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int dst = atomicAdd(&fill[cid[i]], 1);
    float4 v;
    v.x = pts[3 * (size_t)i + 0];
    v.y = pts[3 * (size_t)i + 1];
    v.z = pts[3 * (size_t)i + 2];
    v.w = __int_as_float(i);
    tmp[dst] = v;
}
// the atomics above place a cell's points in arbitrary order: rank every point inside its cell by original id
__global__ __launch_bounds__(256) void p2s_cell_rank_kernel(const float4 *__restrict__ tmp, const int *__restrict__ cell_start,
                                                            int n, CellGeom g, float4 *__restrict__ spts) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const float4 v = tmp[p];
    const int c = (cell_coord(v.x, g.lo[0], g.inv, g.G) * g.G + cell_coord(v.y, g.lo[1], g.inv, g.G)) * g.G +
                  cell_coord(v.z, g.lo[2], g.inv, g.G);
    const int s0 = cell_start[c], e0 = cell_start[c + 1];
    const int id = __float_as_int(v.w);
    int rank = 0;
    for (int j = s0; j < e0; ++j) rank += (__float_as_int(tmp[j].w) < id) ? 1 : 0;
    spts[s0 + rank] = v;
}

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
