// "next" row f-1: SDF samples -> dense volume -> iterative sign propagation, on the device.
//   add_samples_to_volume   reference source/sdf.py:82-111
//   propagate_sign          reference source/sdf.py:114-178  (scipy.ndimage.convolve(ones(sigma^3), mode='nearest'))
//   clamp to [-1, 1]        reference source/sdf.py:199-201
// The reference runs this on one CPU core per shape (149 s at 256^3); it is a pure stencil: HBM/L2-bound.
// Working set per sweep: int8 sign volume + int8/int16 partial sums (separable box filter along z, y, x).
#include "p2s_common.h"
#include <vector>
#include <utility>
#include <algorithm>
#include <cstring>
#include <cstdlib>

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
// fast path (grid_res % 16 == 0, sigma <= 5): ONE fused kernel per sweep, no host round trip per sweep.
//
// State: one byte per voxel -- bits 0-1 the propagated sign (two's complement: 0, 1, 3 = -1), bit 2 "unknown
// initially" (the only voxels a sweep may change) -- in two ping-pong buffers: a sweep reads each voxel's byte once
// and writes one byte (the algorithmic 2 B / voxel / sweep; the separable three-pass version moved 7).
// A workgroup owns a tile of 8 x 16 x 64 voxels: the tile + a halo of 2 (coordinates clamped = scipy's
// mode='nearest') is staged in LDS, the sigma^3 box sum is evaluated separably inside LDS with per-byte SWAR adds
// (|partial sums| <= 5 / 25 / 125 fit int8), then threshold -> new sign -> speculative next state -> both zero
// counts (block-reduced, one atomic pair per workgroup on a per-sweep counter).
// Termination on the device: sweep k decides from the counters of sweeps k-1 and k-2 (complete: kernel boundary)
// exactly as the reference's loop does (source/sdf.py:156-176) -- every workgroup evaluates the same pure function
// of those counters, workgroup 0 records the verdict.  Sweeps are launched in batches; a finished run makes the
// remaining launches of its batch exit at once; the host looks at the verdict once per batch.
// ---------------------------------------------------------------------------------------------
#ifndef P2S_VT_Y
#define P2S_VT_Y 16
#endif
constexpr int VT_X = 8, VT_Y = P2S_VT_Y, VT_Z = 64, VT_H = 2;     // VT_Y <= 16 (one thread per (y, z dword) of 256)
constexpr int VA_X = VT_X + 2 * VT_H, VA_Y = VT_Y + 2 * VT_H;      // halo'd rows
constexpr int VA_ZD = VT_Z / 4 + 2, VA_ZS = VA_ZD + 1;             // dwords per halo'd row (+1: odd stride, no bank conflicts)
constexpr int VB_ZS = VT_Z / 4 + 1;

// counters are sharded (one atomic per workgroup on shard = workgroup % 64): thousands of workgroups hitting ONE
// address serialise at ~12 ns per atomic -- 50 us per sweep at 256^3, more than the sweep itself
struct VolState {
    unsigned long long cnt[4][2][NSHARD];   // per sweep (mod 4): [0] zeros of the speculative next state, [1] zeros of `new`
    unsigned long long unknown0[NSHARD];    // zeros of the initial sign field
    int done, final_buf, iters, err;
};
__device__ __forceinline__ unsigned long long wave_sum64(const unsigned long long *p, int lane) {
    unsigned long long v = p[lane];
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

__device__ __forceinline__ unsigned swar_add8(unsigned a, unsigned b) {   // per-byte modular add
    return ((a & 0x7f7f7f7fu) + (b & 0x7f7f7f7fu)) ^ ((a ^ b) & 0x80808080u);
}
__device__ __forceinline__ unsigned sign_bytes(unsigned raw) {            // 2-bit two's complement -> int8 per byte
    return swar_add8((raw & 0x03030303u) ^ 0x02020202u, 0xfefefefeu);
}

// byte = sign(vol) & 3 | (sign == 0) << 2 ; unknown0 = #zeros.  4 voxels per thread (nvox is a multiple of 4096)
__global__ void vol_state_init_kernel(const float *__restrict__ vol, long long nvox, unsigned char *__restrict__ st,
                                      VolState *__restrict__ vs) {
    __shared__ int red[4];
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    int z = 0;
    if (i < nvox) {
        const float4 v = *(const float4 *)(vol + i);
        const float f[4] = {v.x, v.y, v.z, v.w};
        unsigned o = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int sg = sgn_of(f[b]);
            o |= (unsigned)((sg & 3) | ((sg == 0) << 2)) << (8 * b);
            z += sg == 0;
        }
        *(unsigned *)(st + i) = o;
    }
    for (int d = 32; d > 0; d >>= 1) z += __shfl_xor(z, d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = z;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = red[0] + red[1] + red[2] + red[3];
        if (t) atomicAdd(&vs->unknown0[blockIdx.x & (NSHARD - 1)], (unsigned long long)t);
    }
}

// sweep k: buf[k & 1] -> buf[(k + 1) & 1]
__global__ __launch_bounds__(256) void vol_sweep_kernel(unsigned char *__restrict__ buf0, unsigned char *__restrict__ buf1,
                                                        int res, int k, unsigned taps, float thr,
                                                        VolState *__restrict__ vs, unsigned char *__restrict__ act,
                                                        int *__restrict__ tcnt) {
    __shared__ unsigned A[VA_X * VA_Y * VA_ZS];
    __shared__ unsigned B[VA_X * VA_Y * VB_ZS];
    unsigned *C = A;      // the zy sums overlay the staged tile (its interior bytes are kept in registers): 34.6 KB, 4 workgroups / CU
    static_assert(VA_X * VT_Y * VB_ZS <= VA_X * VA_Y * VA_ZS, "C must fit into A");
    __shared__ int red[2][4];
    __shared__ unsigned long long s_dec[3];
    __shared__ int s_changed;
    const int tid = threadIdx.x;
    // XCD-aware tile order: workgroup i runs on XCD i % 8 (own L2).  Giving every XCD one contiguous slab of tiles
    // keeps the halo re-reads and the two 64-byte halves of every 128-byte line (adjacent z tiles) in ONE L2; the
    // naive order fetched each line from HBM into two L2s.
    const int wg = blockIdx.x;
    const int tz_n = (res + VT_Z - 1) / VT_Z, ty_n = (res + VT_Y - 1) / VT_Y, tx_n = (res + VT_X - 1) / VT_X;
    const int n_tiles = tz_n * ty_n * tx_n, slab = (n_tiles + 7) >> 3;
    const int tile = (wg & 7) * slab + (wg >> 3);
    const bool has_tile = tile < n_tiles && (wg >> 3) < slab;
    const int tile_c = has_tile ? tile : 0;
    const int tz_i = tile_c % tz_n, ty_i = (tile_c / tz_n) % ty_n, tx_i = tile_c / (tz_n * ty_n);
    const unsigned char *__restrict__ in = (k & 1) ? buf1 : buf0;
    unsigned char *__restrict__ out = (k & 1) ? buf0 : buf1;
    const int z0 = tz_i * VT_Z, y0 = ty_i * VT_Y, x0 = tx_i * VT_X;
    const int rowd = res >> 2;                                      // dwords per z row
    // ---- active tiles only.  A tile whose state bytes did not change in a sweep, and whose 26 neighbours did not
    //      either, would reproduce its previous output: it is skipped, its two buffers already agree and its zero
    //      counts stay in the running totals.  After the first sweeps only the tiles along the advancing front
    //      remain (the flags: sweep k reads act[k % 3], sets act[(k + 1) % 3], clears act[(k + 2) % 3]).
    const bool active = has_tile && act[(k % 3) * n_tiles + tile] != 0;
    if (has_tile && tid == 0) act[((k + 2) % 3) * n_tiles + tile] = 0;
    if (tid == 0) s_changed = 0;
    if (!active && wg != 0) return;
    // ---- stage the halo'd tile (raw state bytes); x, y clamped, z replicated at the volume faces.  All loads of a
    //      thread are issued before anything waits: a rolled loop would pay the memory latency 17 times, and the loop
    //      control below (dependent loads of the counters) would add its own round trip in front of them.  If the
    //      run turns out to be over the loads were for nothing (both buffers stay valid memory).
    constexpr int TOT = VA_X * VA_Y * VA_ZD, PER = (TOT + 255) / 256;
    unsigned w[PER];
#pragma unroll
    for (int it = 0; it < PER; ++it) {
        const int idx = tid + 256 * it;
        const int row = idx / VA_ZD, d = idx - row * VA_ZD;
        const int ax = row / VA_Y, ay = row - ax * VA_Y;
        const int x = min(max(x0 - VT_H + ax, 0), res - 1), y = min(max(y0 - VT_H + ay, 0), res - 1);
        const int z = z0 - 4 + 4 * d;
        const unsigned *rowp = (const unsigned *)(in + ((long long)x * res + y) * res);
        const int zi = z < 0 ? 0 : (z >= res ? rowd - 1 : z >> 2);
        w[it] = (active && idx < TOT) ? rowp[zi] : 0u;
    }
    // ---- the reference's loop control (source/sdf.py:156-176) as a function of the finished sweeps' counters
    const int done = vs->done;
    if (tid < 64) {                        // wave 0 sums the shards
        const unsigned long long before = (k <= 1) ? wave_sum64(vs->unknown0, tid) : wave_sum64(vs->cnt[(k - 2) & 3][0], tid);
        const unsigned long long zn = k >= 1 ? wave_sum64(vs->cnt[(k - 1) & 3][1], tid) : 0ull;
        const unsigned long long zx = k >= 1 ? wave_sum64(vs->cnt[(k - 1) & 3][0], tid) : 0ull;
        if (tid == 0) {
            s_dec[0] = before;
            s_dec[1] = zn;
            s_dec[2] = zx;
        }
    }
#pragma unroll
    for (int it = 0; it < PER; ++it) {
        const int idx = tid + 256 * it;
        const int row = idx / VA_ZD, d = idx - row * VA_ZD;
        const int z = z0 - 4 + 4 * d;
        unsigned v = w[it];
        if (z < 0) v = (v & 0xffu) * 0x01010101u;
        else if (z >= res) v = (v >> 24) * 0x01010101u;
        if (idx < TOT) A[row * VA_ZS + d] = v;
    }
    __syncthreads();
    if (done) return;
    {
        bool stop = false;
        int fin = 0;
        if (k == 0) {
            stop = s_dec[0] == 0;
            fin = 0;
        } else {
            const unsigned long long before = s_dec[0];
            const unsigned long long z_new = s_dec[1], z_next = s_dec[2];
            if (z_new >= before) {            // no progress: the speculative state of sweep k-1 is discarded
                stop = true;
                fin = (k - 1) & 1;
            } else if (z_next == 0) {         // everything known
                stop = true;
                fin = k & 1;
            }
        }
        if (stop) {
            if (wg == 0 && tid == 0) {
                vs->final_buf = fin;
                vs->iters = k;
                __threadfence();
                vs->done = 1;
            }
            return;
        }
        if (wg == 0 && tid < 2 * NSHARD)      // counters of the next sweep (nobody reads or adds to them now)
            vs->cnt[(k + 1) & 3][tid >> 6][tid & (NSHARD - 1)] = 0;
        if (wg == 0 && tid == 0 && k >= 1) {  // running totals: this sweep's = the previous sweep's + the deltas of the active tiles
            atomicAdd(&vs->cnt[k & 3][0][0], s_dec[2]);
            atomicAdd(&vs->cnt[k & 3][1][0], s_dec[1]);
        }
    }
    if (!active) return;
    // this lane's interior state bytes (its 8 output dwords, thread = (y, z dword)): read now, A is recycled below
    unsigned raw_in[VT_X];
    {
        const int y = tid / (VT_Z / 4), zd = tid - y * (VT_Z / 4);
#pragma unroll
        for (int x = 0; x < VT_X; ++x) raw_in[x] = A[((x + VT_H) * VA_Y + (y + VT_H)) * VA_ZS + zd + 1];
    }
    // All sums are kept BIASED: sign + 1 in {0, 1, 2} per byte, so the z / zy / zyx sums are <= 10 / 50 / 250 -- plain
    // 32-bit adds and subtracts never carry between bytes (the carry-safe SWAR add costs 7 ops, this costs 1), and the
    // y and x passes slide their window (out[y] = out[y-1] + entering - leaving).  tap range: offsets [t_lo, t_hi].
    const int t_lo = __ffs(taps) - 1 - 2, t_hi = 31 - __clz(taps) - 2, nt = t_hi - t_lo + 1;
    // ---- z sums: one halo'd row per thread
    if (tid < VA_X * VA_Y) {
        unsigned w[VA_ZD];
#pragma unroll
        for (int d = 0; d < VA_ZD; ++d) w[d] = ((A[tid * VA_ZS + d] & 0x03030303u) + 0x01010101u) & 0x03030303u;
#pragma unroll
        for (int d = 0; d < VT_Z / 4; ++d) {
            const unsigned lo = w[d], mi = w[d + 1], hi = w[d + 2];
            unsigned acc = 0;
            if (taps & 1u) acc += __builtin_amdgcn_alignbyte(mi, lo, 2);      // offset -2: bytes z-2 .. z+1
            if (taps & 2u) acc += __builtin_amdgcn_alignbyte(mi, lo, 3);      // -1
            if (taps & 4u) acc += mi;                                         //  0
            if (taps & 8u) acc += __builtin_amdgcn_alignbyte(hi, mi, 1);      // +1
            if (taps & 16u) acc += __builtin_amdgcn_alignbyte(hi, mi, 2);     // +2
            B[tid * VB_ZS + d] = acc;
        }
    }
    __syncthreads();
    // ---- y sums: thread = (halo'd x, z dword), sliding window over the halo'd column
    if (tid < VA_X * (VT_Z / 4)) {
        const int ax = tid / (VT_Z / 4), zd = tid - ax * (VT_Z / 4);
        unsigned v[VA_Y];
#pragma unroll
        for (int ay = 0; ay < VA_Y; ++ay) v[ay] = B[(ax * VA_Y + ay) * VB_ZS + zd];
        unsigned acc = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j)
            if ((taps >> j) & 1u) acc += v[j];
        C[(ax * VT_Y) * VB_ZS + zd] = acc;
#pragma unroll
        for (int y = 1; y < VT_Y; ++y) {
            // window [y + 2 + t_lo, y + 2 + t_hi]: uniform selects instead of runtime-indexed registers
            unsigned in_v = 0, out_v = 0;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                in_v = (t_hi + 2 == j) ? v[y + j] : in_v;
                out_v = (t_lo + 2 == j) ? v[y - 1 + j] : out_v;
            }
            acc += in_v - out_v;
            C[(ax * VT_Y + y) * VB_ZS + zd] = acc;
        }
    }
    __syncthreads();
    // ---- x sums, threshold, speculative update, counts: thread = (y, z dword)
    int z_new = 0, z_next = 0;
    bool changed = false;
    {
        const int y = tid / (VT_Z / 4), zd = tid - y * (VT_Z / 4);
        unsigned v[VA_X];
#pragma unroll
        for (int ax = 0; ax < VA_X; ++ax) v[ax] = C[(ax * VT_Y + y) * VB_ZS + zd];
        const int gy = y0 + y, gz = z0 + 4 * zd;
        const bool inside_yz = y < VT_Y && gy < res && gz < res;
        // |a| < thr -> 0 for the integer a = acc - bias:  a >= T  <=>  acc >= bias + T ;  a <= -T  <=>  !(acc >= bias - T + 1)
        const int bias = nt * nt * nt;
        const int T = thr > 1.0f ? (int)ceilf(thr) : 1;
        const unsigned c_pos = (unsigned)min(bias + T, 256) * 0x00010001u;
        const unsigned c_neg = (unsigned)max(bias - T + 1, 0) * 0x00010001u;
        unsigned acc = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j)
            if ((taps >> j) & 1u) acc += v[j];
#pragma unroll
        for (int x = 0; x < VT_X; ++x) {
            if (x > 0) {
                unsigned in_v = 0, out_v = 0;
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    in_v = (t_hi + 2 == j) ? v[x + j] : in_v;
                    out_v = (t_lo + 2 == j) ? v[x - 1 + j] : out_v;
                }
                acc += in_v - out_v;
            }
            const int gx = x0 + x;
            if (inside_yz && gx < res) {
                const unsigned raw = raw_in[x];
                // per-byte compares on the even / odd bytes as 16-bit lanes: bit 8 of (x | 0x100) - c is x >= c
                const unsigned ev = acc & 0x00ff00ffu, od = (acc >> 8) & 0x00ff00ffu;
                const unsigned pe = (((ev | 0x01000100u) - c_pos) >> 8) & 0x00010001u, po = (((od | 0x01000100u) - c_pos) >> 8) & 0x00010001u;
                const unsigned ne = (~(((ev | 0x01000100u) - c_neg) >> 8)) & 0x00010001u, no = (~(((od | 0x01000100u) - c_neg) >> 8)) & 0x00010001u;
                const unsigned pos = pe | (po << 8), neg = ne | (no << 8);          // 0x01 per byte
                const unsigned code = pos | (neg * 3u);                             // new sign, 2-bit two's complement
                const unsigned um = ((raw >> 2) & 0x01010101u) * 0xffu;             // 0xff where unknown initially
                const unsigned o = raw ^ ((raw ^ (code | 0x04040404u)) & um);
                z_new += 4 - __popc(pos | neg);
                z_next += 4 - __popc((o | (o >> 1)) & 0x01010101u);
                changed |= o != raw;
                *(unsigned *)(out + ((long long)gx * res + gy) * res + gz) = o;
            }
        }
    }
    for (int d = 32; d > 0; d >>= 1) {
        z_new += __shfl_xor(z_new, d);
        z_next += __shfl_xor(z_next, d);
    }
    if ((tid & 63) == 0) {
        red[0][tid >> 6] = z_next;
        red[1][tid >> 6] = z_new;
    }
    if (changed) s_changed = 1;
    __syncthreads();
    if (tid == 0) {
        // delta against this tile's counts of its previous evaluation (64-bit wrap-around = signed add)
        const int a = red[0][0] + red[0][1] + red[0][2] + red[0][3], b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const long long da = (long long)a - tcnt[2 * tile], db = (long long)b - tcnt[2 * tile + 1];
        tcnt[2 * tile] = a;
        tcnt[2 * tile + 1] = b;
        if (da) atomicAdd(&vs->cnt[k & 3][0][wg & (NSHARD - 1)], (unsigned long long)da);
        if (db) atomicAdd(&vs->cnt[k & 3][1][wg & (NSHARD - 1)], (unsigned long long)db);
    }
    if (s_changed && tid < 27) {              // the state changed here: this tile and its neighbours run in the next sweep
        const int nz = tz_i + tid % 3 - 1, ny = ty_i + (tid / 3) % 3 - 1, nx = tx_i + tid / 9 - 1;
        if (nz >= 0 && nz < tz_n && ny >= 0 && ny < ty_n && nx >= 0 && nx < tx_n)
            act[((k + 1) % 3) * n_tiles + (nx * ty_n + ny) * tz_n + nz] = 1;
    }
}

// borders := -1 ; remaining zeros := propagated sign of the final state ; optional clamp to [-1, 1]
__global__ void vol_compose_state_kernel(float *__restrict__ vol, const unsigned char *__restrict__ buf0,
                                         const unsigned char *__restrict__ buf1, const VolState *__restrict__ vs, int res,
                                         int clamp) {
    const long long nvox = (long long)res * res * res;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvox) return;
    const unsigned char *st = vs->final_buf ? buf1 : buf0;
    const int z = (int)(i % res);
    const long long t = i / res;
    const int y = (int)(t % res);
    const int x = (int)(t / res);
    float v = vol[i];
    if (x == 0 || y == 0 || z == 0 || x == res - 1 || y == res - 1 || z == res - 1) v = -1.0f;
    if (v == 0.0f) {
        const int sb = st[i] & 3;
        v = (float)(sb == 3 ? -1 : sb);
    }
    if (clamp) v = fminf(fmaxf(v, -1.0f), 1.0f);
    vol[i] = v;
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
    const bool fast = (grid_res % 16 == 0) && sigma <= 5 && !getenv("P2S_VOLUME_GENERIC");
    // scratch, fast path: two state bytes per voxel (ping-pong) + the sweep state.  Generic path: sign (1) +
    // unknown_initially (1) + new sign (1) + z sums (1) + zy sums (2) bytes per voxel
    const size_t scratch_bytes = (size_t)nvox * (fast ? 2 : 7) + sizeof(VolState) + 256 + (size_t)(nvox / 512 + 64) * 12;
    const size_t counts_at = (scratch_bytes + 63) & ~(size_t)63;
    char *scratch = (char *)p2s_scratch(device, counts_at + (2 * NSHARD + 2) * 8);
    if (!scratch) {
        p2s_set_error("p2s_sdf_volume: hipMalloc(%zu bytes) failed", counts_at);
        return P2S_ENOMEM;
    }
    unsigned long long *counts = (unsigned long long *)(scratch + counts_at);   // [0..64) zeros of s, [64..128) zeros of new, [128] error flag
    auto cleanup = [&](int code) {
        (void)hipStreamSynchronize(s);       // the scratch buffer is idle again when we return
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
    int iters = 0;
    if (fast) {
        unsigned char *buf0 = (unsigned char *)scratch, *buf1 = buf0 + nvox;
        VolState *vs = (VolState *)(((uintptr_t)(buf1 + nvox) + 63) & ~(uintptr_t)63);
        if (hipMemsetAsync(vs, 0, sizeof(VolState), s) != hipSuccess) {
            p2s_set_error("p2s_sdf_volume: memset failed");
            return cleanup(P2S_EHIP);
        }
        hipLaunchKernelGGL(vol_state_init_kernel, dim3((grid + 3) / 4), dim3(256), 0, s, vol_out_dev, nvox, buf0, vs);
        unsigned taps = 0;
        for (int j = 0; j < sigma; ++j) taps |= 1u << (off.o[j] + 2);
        const int n_tiles = ((grid_res + VT_Z - 1) / VT_Z) * ((grid_res + VT_Y - 1) / VT_Y) * ((grid_res + VT_X - 1) / VT_X);
        const dim3 tg((unsigned)(((n_tiles + 7) / 8) * 8));
        // per tile: counts of its last evaluation (2 ints) and three generations of "active" flags
        int *tcnt = (int *)(vs + 1);
        unsigned char *act = (unsigned char *)(tcnt + 2 * (size_t)n_tiles);
        if (hipMemsetAsync(tcnt, 0, (size_t)n_tiles * 8, s) != hipSuccess || hipMemsetAsync(act, 1, (size_t)n_tiles, s) != hipSuccess ||
            hipMemsetAsync(act + n_tiles, 0, (size_t)n_tiles * 2, s) != hipSuccess) {
            p2s_set_error("p2s_sdf_volume: memset failed");
            return cleanup(P2S_EHIP);
        }
        // the number of sweeps is data dependent (the front advances ~2 voxels per sweep): batches without a host
        // round trip; launches behind the final sweep exit at once
        const int batch = getenv("P2S_VOLUME_BATCH") ? std::max(1, atoi(getenv("P2S_VOLUME_BATCH"))) : 16;
        // The verdict of batch j is copied out behind it and looked at only after batch j + 1 has been queued: the host
        // round trip (~70 us) is off the critical path; a finished run makes the extra batch exit at once.
        static thread_local VolState *pinned = nullptr;
        static thread_local hipEvent_t look[2] = {nullptr, nullptr};
        if (!pinned) {
            if (hipHostMalloc((void **)&pinned, 2 * sizeof(VolState), hipHostMallocDefault) != hipSuccess ||
                hipEventCreateWithFlags(&look[0], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&look[1], hipEventDisableTiming) != hipSuccess) {
                pinned = nullptr;
                p2s_set_error("p2s_sdf_volume: pinned verdict buffer: %s", hipGetErrorString(hipGetLastError()));
                return cleanup(P2S_ENOMEM);
            }
        }
        VolState host_vs;
        memset(&host_vs, 0, sizeof(host_vs));
        int k = 0;
        const int k_max = 64 * grid_res + 64;            // far beyond any possible run; guards the host loop only
        bool flag_checked = false;
        for (int j = 0; k < k_max; ++j) {
            for (int t = 0; t < batch; ++t, ++k)
                hipLaunchKernelGGL(vol_sweep_kernel, tg, dim3(256), 0, s, buf0, buf1, grid_res, k, taps, certainty_threshold, vs, act, tcnt);
            if (hipMemcpyAsync(&pinned[j & 1], vs, sizeof(VolState), hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipEventRecord(look[j & 1], s) != hipSuccess) {
                p2s_set_error("p2s_sdf_volume: %s", hipGetErrorString(hipGetLastError()));
                return cleanup(P2S_EHIP);
            }
            if (j == 0) continue;                        // look at batch j - 1 now that batch j is queued
            if (hipEventSynchronize(look[(j - 1) & 1]) != hipSuccess) {
                p2s_set_error("p2s_sdf_volume: %s", hipGetErrorString(hipGetLastError()));
                return cleanup(P2S_EHIP);
            }
            if (!flag_checked) {                         // scatter error flag, once
                flag_checked = true;
                int flag = 0;
                (void)hipMemcpy(&flag, counts + 2 * NSHARD, 4, hipMemcpyDeviceToHost);
                if (flag) {
                    p2s_set_error("p2s_sdf_volume: query point outside the [-1,1) volume");
                    return cleanup(P2S_EINVAL);
                }
            }
            host_vs = pinned[(j - 1) & 1];
            if (host_vs.done) break;
        }
        if (!host_vs.done) {                             // the verdict may sit in the batch queued last
            if (hipStreamSynchronize(s) == hipSuccess) {
                (void)hipMemcpy(&host_vs, vs, sizeof(VolState), hipMemcpyDeviceToHost);
                if (!flag_checked) {
                    int flag = 0;
                    (void)hipMemcpy(&flag, counts + 2 * NSHARD, 4, hipMemcpyDeviceToHost);
                    if (flag) {
                        p2s_set_error("p2s_sdf_volume: query point outside the [-1,1) volume");
                        return cleanup(P2S_EINVAL);
                    }
                }
            }
        }
        if (!host_vs.done) {
            p2s_set_error("p2s_sdf_volume: sign propagation did not terminate within %d sweeps", k_max);
            return cleanup(P2S_EHIP);
        }
        iters = host_vs.iters;
        hipLaunchKernelGGL(vol_compose_state_kernel, dim3(grid), dim3(256), 0, s, vol_out_dev, buf0, buf1, vs, grid_res, clamp);
        P2S_LAUNCH_CHECK("vol_compose_state_kernel");
        if (iterations) *iterations = iters;
        return cleanup(P2S_OK);
    }
    signed char *sg = (signed char *)scratch;
    unsigned char *unk0 = (unsigned char *)(scratch + nvox);
    signed char *newsg = (signed char *)(scratch + 2 * nvox);
    signed char *t1 = (signed char *)(scratch + 3 * nvox);
    short *t2 = (short *)(scratch + 4 * nvox);
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
    signed char *s_final = sg;
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
    hipLaunchKernelGGL(vol_compose_kernel, dim3(grid), dim3(256), 0, s, vol_out_dev, s_final, grid_res, clamp);
    P2S_LAUNCH_CHECK("vol_compose_kernel");
    if (iterations) *iterations = iters;
    return cleanup(P2S_OK);
}
