// a1 (included by p2s_cloud.hip inside its anonymous namespace): the query grid -- voxelise, box-dilate, ordered compaction.
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
