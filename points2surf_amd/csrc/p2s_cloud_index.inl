// a2 (included by p2s_cloud.hip inside its anonymous namespace): the neighbour index built on the device.
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
