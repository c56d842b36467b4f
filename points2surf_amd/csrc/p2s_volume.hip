// "next" row f-1: SDF samples -> dense volume -> iterative sign propagation, on the device.
//   add_samples_to_volume   reference source/sdf.py:82-111
//   propagate_sign          reference source/sdf.py:114-178  (scipy.ndimage.convolve(ones(sigma^3), mode='nearest'))
//   clamp to [-1, 1]        reference source/sdf.py:199-201
// The reference runs this on one CPU core per shape (149 s at 256^3); it is a pure stencil: HBM/L2-bound.
// Working set per sweep: int8 sign volume + int8/int16 partial sums (separable box filter along z, y, x).
#include "p2s_common.h"
#include <vector>
#include <utility>

#pragma clang fp contract(off)

namespace {

struct VolOffsets {
    int n;
    int o[16];
};

// vol[voxel(q_i)] = sdf_i ; voxel index in fp32 exactly as numpy (source/sdf.py:73-75)
__global__ void vol_scatter_kernel(const float *__restrict__ q, const float *__restrict__ sdf, long long n, int res,
                                   float *__restrict__ vol, int *__restrict__ err) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int v[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float t = (q[3 * i + a] + 1.0f) / 2.0f;
        v[a] = (int)floorf(t * (float)res);
    }
    if (v[0] < 0 || v[1] < 0 || v[2] < 0 || v[0] >= res || v[1] >= res || v[2] >= res) {
        *err = 1;
        return;
    }
    vol[((long long)v[0] * res + v[1]) * res + v[2]] = sdf[i];
}

__device__ __forceinline__ int sgn_of(float x) { return (x > 0.0f) - (x < 0.0f); }

// counters: 64 shards per quantity ([0..64) zeros of s / next, [64..128) zeros of new), summed on the host.
// One atomic per workgroup on shard blockIdx % 64: every wave hitting ONE address serialised at ~12 ns per
// atomic and made the counting kernels 20-300x slower than their memory traffic.
constexpr int NSHARD = 64;
__device__ __forceinline__ void block_count2(int a, int b, unsigned long long *__restrict__ counts) {
    __shared__ int red[2][4];
    for (int d = 32; d > 0; d >>= 1) {
        a += __shfl_xor(a, d);
        b += __shfl_xor(b, d);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = a; red[1][wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        int sa = 0, sb = 0;
        for (int w = 0; w < nw; ++w) { sa += red[0][w]; sb += red[1][w]; }
        const int shard = blockIdx.x & (NSHARD - 1);
        if (sa) atomicAdd(&counts[shard], (unsigned long long)sa);
        if (sb) atomicAdd(&counts[NSHARD + shard], (unsigned long long)sb);
    }
}

// s = sign(vol), unk0 = (s == 0); counts[0] += #unknown
__global__ void vol_sign_init_kernel(const float *__restrict__ vol, long long nvox, signed char *__restrict__ s,
                                     unsigned char *__restrict__ unk0, unsigned long long *__restrict__ counts) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int z = 0;
    if (i < nvox) {
        const int sg = sgn_of(vol[i]);
        s[i] = (signed char)sg;
        unk0[i] = sg == 0;
        z = sg == 0;
    }
    block_count2(z, 0, counts);
}

// separable box sums with edge replication.  AXIS 2 = z (fastest), 1 = y, 0 = x
template <typename TIn, typename TOut, int AXIS>
__global__ void vol_boxsum_kernel(const TIn *__restrict__ in, TOut *__restrict__ out, int res, VolOffsets off) {
    const long long nvox = (long long)res * res * res;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvox) return;
    const int z = (int)(i % res);
    const long long t = i / res;
    const int y = (int)(t % res);
    const int x = (int)(t / res);
    const int c = AXIS == 2 ? z : (AXIS == 1 ? y : x);
    const long long stride = AXIS == 2 ? 1 : (AXIS == 1 ? res : (long long)res * res);
    const long long base = i - (long long)c * stride;
    int acc = 0;
    for (int j = 0; j < off.n; ++j) {
        const int cc = min(max(c + off.o[j], 0), res - 1);
        acc += (int)in[base + (long long)cc * stride];
    }
    out[i] = (TOut)acc;
}

// last pass (along x) fused with: threshold -> sign -> count zeros (over ALL voxels, as the reference does)
__global__ void vol_boxsum_x_sign_kernel(const short *__restrict__ in, signed char *__restrict__ newsgn, int res,
                                         VolOffsets off, float thr, unsigned long long *__restrict__ counts) {
    const long long nvox = (long long)res * res * res;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int zero = 0;
    if (i < nvox) {
        const long long plane = (long long)res * res;
        const int x = (int)(i / plane);
        const long long base = i - (long long)x * plane;
        int acc = 0;
        for (int j = 0; j < off.n; ++j) {
            const int xx = min(max(x + off.o[j], 0), res - 1);
            acc += (int)in[base + (long long)xx * plane];
        }
        const int sg = (fabsf((float)acc) < thr) ? 0 : ((acc > 0) - (acc < 0));
        newsgn[i] = (signed char)sg;
        zero = sg == 0;
    }
    block_count2(0, zero, counts);
}

// accepted sweep: s[unknown_initially] = new[unknown_initially]; counts[0] += #zeros of the updated s
__global__ void vol_apply_kernel(signed char *__restrict__ s, const signed char *__restrict__ newsgn,
                                 const unsigned char *__restrict__ unk0, long long nvox,
                                 unsigned long long *__restrict__ counts) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int zero = 0;
    if (i < nvox) {
        signed char v = s[i];
        if (unk0[i]) {
            v = newsgn[i];
            s[i] = v;
        }
        zero = v == 0;
    }
    block_count2(zero, 0, counts);
}

// borders := -1 ; remaining zeros := propagated sign ; optional clamp to [-1, 1]
__global__ void vol_compose_kernel(float *__restrict__ vol, const signed char *__restrict__ s, int res, int clamp) {
    const long long nvox = (long long)res * res * res;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvox) return;
    const int z = (int)(i % res);
    const long long t = i / res;
    const int y = (int)(t % res);
    const int x = (int)(t / res);
    float v = vol[i];
    if (x == 0 || y == 0 || z == 0 || x == res - 1 || y == res - 1 || z == res - 1) v = -1.0f;
    if (v == 0.0f) v = (float)s[i];
    if (clamp) v = fminf(fmaxf(v, -1.0f), 1.0f);
    vol[i] = v;
}

// ---------------------------------------------------------------------------------------------
// fast path (grid_res % 16 == 0, sigma <= 5): 16 voxels per thread as one 16-byte vector along z, per-byte
// (SWAR) int8 arithmetic, |partial sums| <= 5 / 25 / 125 fit int8.  The last pass writes the NEXT sign volume
// speculatively (ping-pong) and counts both "zeros of new" and "zeros of next", so a sweep is three launches
// and one host read-back; nothing is copied when a sweep is rejected.
// ---------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned swar_add8(unsigned a, unsigned b) {   // per-byte modular add
    return ((a & 0x7f7f7f7fu) + (b & 0x7f7f7f7fu)) ^ ((a ^ b) & 0x80808080u);
}
__device__ __forceinline__ u32x4 swar_add8(u32x4 a, u32x4 b) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = swar_add8(a[i], b[i]);
    return r;
}
__device__ __forceinline__ int byte_of(const u32x4 &v, int k) { return (int)(signed char)((v[k >> 2] >> (8 * (k & 3))) & 0xffu); }

// z pass: out[z] = sum_j in[clamp(z + o_j)]; thread = (row, 16-byte segment)
__global__ __launch_bounds__(256) void vol16_z_kernel(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, int res,
                                                      long long nvec, VolOffsets off) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    const int segs = res >> 4;
    const int seg = (int)(i % segs);
    const u32x4 cur = in[i];
    signed char w[48];
    const u32x4 prev = seg > 0 ? in[i - 1] : cur;
    const u32x4 next = seg < segs - 1 ? in[i + 1] : cur;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        w[k] = seg > 0 ? (signed char)byte_of(prev, k) : (signed char)byte_of(cur, 0);          // edge replication
        w[16 + k] = (signed char)byte_of(cur, k);
        w[32 + k] = seg < segs - 1 ? (signed char)byte_of(next, k) : (signed char)byte_of(cur, 15);
    }
    int m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0;
    for (int j = 0; j < off.n; ++j) {
        m0 += off.o[j] == -2; m1 += off.o[j] == -1; m2 += off.o[j] == 0; m3 += off.o[j] == 1; m4 += off.o[j] == 2;
    }
    u32x4 o = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        // static window of +-2 with 0/1 taps (runtime-indexed register arrays would go to scratch)
        const int acc = m0 * w[14 + k] + m1 * w[15 + k] + m2 * w[16 + k] + m3 * w[17 + k] + m4 * w[18 + k];
        o[k >> 2] |= ((unsigned)(acc & 0xff)) << (8 * (k & 3));
    }
    out[i] = o;
}

// y pass (stride = one row of vectors)
__global__ __launch_bounds__(256) void vol16_y_kernel(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, int res,
                                                      long long nvec, VolOffsets off) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    const int segs = res >> 4;
    const int y = (int)((i / segs) % res);
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (int j = 0; j < off.n; ++j) {
        const int yy = min(max(y + off.o[j], 0), res - 1);
        acc = swar_add8(acc, in[i + (long long)(yy - y) * segs]);
    }
    out[i] = acc;
}

// x pass + threshold + sign + speculative update + both zero counts
__global__ __launch_bounds__(256) void vol16_x_kernel(const u32x4 *__restrict__ in, const u32x4 *__restrict__ s_cur,
                                                      const u32x4 *__restrict__ unk0, u32x4 *__restrict__ s_next,
                                                      int res, long long nvec, VolOffsets off, float thr,
                                                      unsigned long long *__restrict__ counts) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    int z_new = 0, z_next = 0;
    if (i < nvec) {
        const long long plane = (long long)(res >> 4) * res;
        const int x = (int)(i / plane);
        u32x4 acc = {0u, 0u, 0u, 0u};
        for (int j = 0; j < off.n; ++j) {
            const int xx = min(max(x + off.o[j], 0), res - 1);
            acc = swar_add8(acc, in[i + (long long)(xx - x) * plane]);
        }
        const u32x4 sc = s_cur[i], uk = unk0[i];
        u32x4 o = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int a = byte_of(acc, k);
            const int nw = (fabsf((float)a) < thr) ? 0 : ((a > 0) - (a < 0));
            const int nx = byte_of(uk, k) ? nw : byte_of(sc, k);
            z_new += nw == 0;
            z_next += nx == 0;
            o[k >> 2] |= ((unsigned)(nx & 0xff)) << (8 * (k & 3));
        }
        s_next[i] = o;
    }
    block_count2(z_next, z_new, counts);
}

}  // namespace

extern "C" int p2s_sdf_volume(const float *query_dev, const float *sdf_dev, int64_t n, int grid_res, int sigma,
                              float certainty_threshold, int clamp, int device, float *vol_out_dev,
                              int32_t *iterations, void *stream) {
    if (((!query_dev || !sdf_dev) && n > 0) || !vol_out_dev || n < 0 || grid_res < 2 || grid_res > 1024 || sigma < 1 || sigma > 15) {
        p2s_set_error("p2s_sdf_volume: bad argument (res=%d sigma=%d)", grid_res, sigma);
        return P2S_EINVAL;
    }
    if (p2s_device_count() <= device || device < 0) {
        p2s_set_error("p2s_sdf_volume: no HIP device %d", device);
        return P2S_ENODEVICE;
    }
    P2S_HIP_CHECK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    const long long nvox = (long long)grid_res * grid_res * grid_res;
    // scratch: sign (1) + unknown_initially (1) + new sign (1) + z sums (1) + zy sums (2) bytes per voxel
    char *scratch = nullptr;
    unsigned long long *counts = nullptr;   // [0..64) zeros of s, [64..128) zeros of new, [128] error flag (as int)
    if (hipMalloc(&scratch, (size_t)nvox * 7) != hipSuccess || hipMalloc(&counts, (2 * NSHARD + 2) * 8) != hipSuccess) {
        if (scratch) (void)hipFree(scratch);
        p2s_set_error("p2s_sdf_volume: hipMalloc(%lld bytes) failed", nvox * 6);
        (void)hipGetLastError();
        return P2S_ENOMEM;
    }
    signed char *sg = (signed char *)scratch;
    unsigned char *unk0 = (unsigned char *)(scratch + nvox);
    signed char *newsg = (signed char *)(scratch + 2 * nvox);
    signed char *t1 = (signed char *)(scratch + 3 * nvox);
    short *t2 = (short *)(scratch + 4 * nvox);
    signed char *sg2 = (signed char *)(scratch + 6 * nvox);   // ping-pong partner of sg (fast path)
    auto cleanup = [&](int code) {
        (void)hipStreamSynchronize(s);
        (void)hipFree(scratch);
        (void)hipFree(counts);
        return code;
    };
    const unsigned grid = (unsigned)((nvox + 255) / 256);
    VolOffsets off;
    off.n = sigma;
    for (int j = 0; j < sigma; ++j) off.o[j] = sigma / 2 - j;   // scipy.ndimage.convolve, origin 0

    if (hipMemsetAsync(vol_out_dev, 0, (size_t)nvox * 4, s) != hipSuccess ||
        hipMemsetAsync(counts, 0, (2 * NSHARD + 2) * 8, s) != hipSuccess) {
        p2s_set_error("p2s_sdf_volume: memset failed");
        return cleanup(P2S_EHIP);
    }
    if (n > 0) {
        hipLaunchKernelGGL(vol_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, query_dev, sdf_dev,
                           (long long)n, grid_res, vol_out_dev, (int *)(counts + 2 * NSHARD));
    }
    hipLaunchKernelGGL(vol_sign_init_kernel, dim3(grid), dim3(256), 0, s, vol_out_dev, nvox, sg, unk0, counts);
    unsigned long long hc[2 * NSHARD + 2];
    unsigned long long h[2] = {0, 0};
    auto read_counts = [&]() -> bool {
        if (hipMemcpyAsync(hc, counts, sizeof(hc), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
            return false;
        h[0] = h[1] = 0;
        for (int i = 0; i < NSHARD; ++i) { h[0] += hc[i]; h[1] += hc[NSHARD + i]; }
        return true;
    };
    if (!read_counts()) {
        p2s_set_error("p2s_sdf_volume: %s", hipGetErrorString(hipGetLastError()));
        return cleanup(P2S_EHIP);
    }
    if ((int)hc[2 * NSHARD]) {
        p2s_set_error("p2s_sdf_volume: query point outside the [-1,1) volume");
        return cleanup(P2S_EINVAL);
    }
    unsigned long long unknown_before = h[0];
    int iters = 0;
    const bool fast = (grid_res % 16 == 0) && sigma <= 5;
    signed char *s_final = sg;
    if (fast) {
        const long long nvec = nvox / 16;
        const unsigned gv = (unsigned)((nvec + 255) / 256);
        signed char *cur = sg, *nxt = sg2;
        while (unknown_before != 0) {
            (void)hipMemsetAsync(counts, 0, 2 * NSHARD * 8, s);
            hipLaunchKernelGGL(vol16_z_kernel, dim3(gv), dim3(256), 0, s, (const u32x4 *)cur, (u32x4 *)t1, grid_res, nvec, off);
            hipLaunchKernelGGL(vol16_y_kernel, dim3(gv), dim3(256), 0, s, (const u32x4 *)t1, (u32x4 *)t2, grid_res, nvec, off);
            hipLaunchKernelGGL(vol16_x_kernel, dim3(gv), dim3(256), 0, s, (const u32x4 *)t2, (const u32x4 *)cur,
                               (const u32x4 *)unk0, (u32x4 *)nxt, grid_res, nvec, off, certainty_threshold, counts);
            if (!read_counts()) {
                p2s_set_error("p2s_sdf_volume: %s", hipGetErrorString(hipGetLastError()));
                return cleanup(P2S_EHIP);
            }
            ++iters;
            if (h[1] >= unknown_before) break;   // no progress: the speculative volume is discarded
            std::swap(cur, nxt);
            unknown_before = h[0];
        }
        s_final = cur;
    } else {
    while (unknown_before != 0) {
        (void)hipMemsetAsync(counts, 0, 2 * NSHARD * 8, s);
        hipLaunchKernelGGL((vol_boxsum_kernel<signed char, signed char, 2>), dim3(grid), dim3(256), 0, s, sg, t1, grid_res, off);
        hipLaunchKernelGGL((vol_boxsum_kernel<signed char, short, 1>), dim3(grid), dim3(256), 0, s, t1, t2, grid_res, off);
        hipLaunchKernelGGL(vol_boxsum_x_sign_kernel, dim3(grid), dim3(256), 0, s, t2, newsg, grid_res, off,
                           certainty_threshold, counts);
        if (!read_counts()) {
            p2s_set_error("p2s_sdf_volume: %s", hipGetErrorString(hipGetLastError()));
            return cleanup(P2S_EHIP);
        }
        ++iters;
        const unsigned long long unknown_after = h[1];
        if (unknown_after >= unknown_before) break;   // no progress: some voxels are caught in a tie
        hipLaunchKernelGGL(vol_apply_kernel, dim3(grid), dim3(256), 0, s, sg, newsg, unk0, nvox, counts);
        if (!read_counts()) {
            p2s_set_error("p2s_sdf_volume: %s", hipGetErrorString(hipGetLastError()));
            return cleanup(P2S_EHIP);
        }
        unknown_before = h[0];
    }
    }
    hipLaunchKernelGGL(vol_compose_kernel, dim3(grid), dim3(256), 0, s, vol_out_dev, s_final, grid_res, clamp);
    P2S_LAUNCH_CHECK("vol_compose_kernel");
    if (iterations) *iterations = iters;
    return cleanup(P2S_OK);
}
