// a4 / a5 (included by p2s_cloud.hip inside its anonymous namespace): exact k nearest neighbours over the cell index, radius, patch space.
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
