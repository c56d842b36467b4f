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
    int lcount[3][8];                        // entries of the active-tile list of generation k % 3, per XCD slab
};
// host-visible mailbox (pinned, coherent): the host reads it while the sweeps run -- no copy, no event in the stream
struct VolMail {
    int progress;      // sweep index the device has reached (workgroup 0 writes it at the start of every sweep)
    int done, final_buf, iters;
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

// geometry of one tile (workgroup-uniform)
struct VolTile {
    int tile, tx_i, ty_i, tz_i, x0, y0, z0;
};
__device__ __forceinline__ VolTile vol_tile(int tile, int tz_n, int ty_n) {
    VolTile t;
    t.tile = tile;
    t.tz_i = tile % tz_n;
    t.ty_i = (tile / tz_n) % ty_n;
    t.tx_i = tile / (tz_n * ty_n);
    t.z0 = t.tz_i * VT_Z;
    t.y0 = t.ty_i * VT_Y;
    t.x0 = t.tx_i * VT_X;
    return t;
}
// staging map: thread = (row r0 of HALF a halo'd x plane, z dword d = tid % 18), 180 of the 256 threads; iteration
// `it` moves half h = it & 1 of plane ax = it >> 1.  The x term of the address is workgroup-uniform (scalar offset of
// the buffer load), the y / z term takes two registers per thread, the LDS address is base + a compile-time constant,
// the face replication is fixed per thread: ONE vector instruction per element on the global side, three on the LDS
// side.  (The map idx = tid + 256 it cost 27: two constant divisions per element, twice -- a quarter of the kernel.)
static_assert(VA_Y % 2 == 0, "half planes");
constexpr int VS_ROWS = VA_Y / 2, VS_THREADS = VS_ROWS * VA_ZD, VS_PER = 2 * VA_X;
static_assert(VS_THREADS <= 256, "staging map");

// the halo'd tile (raw state bytes) into registers; x, y clamped, z clamped to the first / last dword of the row (the
// face bytes are replicated when the registers go to LDS).  All loads of a thread are issued back to back, through a
// buffer descriptor (32-bit offsets: 64-bit pointers for the loads in flight would cost two VGPRs each).
__device__ __forceinline__ void vol_tile_load(unsigned (&w)[VS_PER], __amdgpu_buffer_rsrc_t rsrc, const VolTile &t, int res, int tid) {
    const int r0 = tid / VA_ZD, d = tid - r0 * VA_ZD;
    const int z = t.z0 - 4 + 4 * d;
    const int zb = z < 0 ? 0 : (z >= res ? res - 4 : z);
    int voff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) voff[h] = min(max(t.y0 - VT_H + r0 + h * VS_ROWS, 0), res - 1) * res + zb;
    if (tid < VS_THREADS) {
#pragma unroll
        for (int it = 0; it < VS_PER; ++it) {
            const int x = min(max(t.x0 - VT_H + (it >> 1), 0), res - 1);
            w[it] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[it & 1], x * res * res, 0);
        }
    }
}
__device__ __forceinline__ void vol_tile_stage(const unsigned (&w)[VS_PER], unsigned *__restrict__ A, const VolTile &t, int res, int tid) {
    const int r0 = tid / VA_ZD, d = tid - r0 * VA_ZD;
    const int z = t.z0 - 4 + 4 * d;
    const unsigned sel = z < 0 ? 0x00000000u : (z >= res ? 0x03030303u : 0x03020100u);   // v_perm: replicate byte 0 / byte 3 / identity
    unsigned *a = A + r0 * VA_ZS + d;
    if (tid < VS_THREADS) {
#pragma unroll
        for (int it = 0; it < VS_PER; ++it)
            a[((it >> 1) * VA_Y + (it & 1) * VS_ROWS) * VA_ZS] = __builtin_amdgcn_perm(w[it], w[it], sel);
    }
}

// per-byte compare constants for v_lerp_u8: byte of lerp(x, K, R) = (x + K + R) >> 1, its bit 7 is  x >= c  for the
// biased sums x <= 250:  c = 0 -> always (K = 255, R = 1) ; 1 <= c <= 255 -> K = 255 - c, R = 1 ; c >= 256 -> never (K = R = 0)
__device__ __forceinline__ void lerp_ge_consts(int c, unsigned &K, unsigned &R) {
    const int kk = c <= 0 ? 255 : (c >= 256 ? 0 : 255 - c);
    K = (unsigned)kk * 0x01010101u;
    R = c >= 256 ? 0u : 0x01010101u;
}

// generation 0 of the active-tile lists: every tile, in order
__global__ void vol_list_init_kernel(int *__restrict__ tlist, VolState *__restrict__ vs, int n_tiles, int slab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_tiles) tlist[(size_t)(i / slab) * slab + i % slab] = i;
    if (i < 8) vs->lcount[0][i] = min(max(n_tiles - i * slab, 0), slab);
}

// Tiles a workgroup activated for the next sweep -> the per-XCD lists: one atomicAdd per workgroup and slab touched
// (an append per tile would put ~3000 atomics per sweep on 8 addresses at 512^3).  All 256 threads call it.
constexpr int VOL_NEW_CAP = 512;
__device__ __forceinline__ void vol_list_flush(int *s_new, int *s_new_n, int *lcount_next, int *list_next, int slab, int tid) {
    const int n = *s_new_n;
    __syncthreads();
    for (int b = 0; b < n; b += 256) {
        const int e = b + tid;
        const int nt = e < n ? s_new[e] : -1;
        const int x = nt >= 0 ? nt / slab : -1;
#pragma unroll 1
        for (int xv = 0; xv < 8; ++xv) {
            const unsigned long long mask = __ballot(x == xv);
            if (mask == 0) continue;                                        // wave-uniform
            const int lane = tid & 63;
            int base = 0;
            if (lane == __ffsll((long long)mask) - 1) base = atomicAdd(&lcount_next[xv], __popcll(mask));
            base = __shfl(base, __ffsll((long long)mask) - 1);
            if (x == xv) list_next[(size_t)xv * slab + base + __popcll(mask & ((1ull << lane) - 1ull))] = nt;
        }
    }
    __syncthreads();
    if (tid == 0) *s_new_n = 0;
    __syncthreads();
}

// sweep k: buf[k & 1] -> buf[(k + 1) & 1]; taps = the offsets [T_LO, T_HI] of scipy's convolve (compile time: the five
// sigmas are five kernels; with run-time taps every window step cost ten selects).  PERSISTENT workgroups: <= 4 per
// CU (LDS), each walks the active tiles of its share; the staging loads of the NEXT tile are in flight while the
// current one is summed (their registers are free as soon as the tile sits in LDS), and the dispatcher starts <= 1024
// workgroups per sweep instead of one per tile.  The kernel is VALU-bound (~600 instructions per wave and tile).
template <int T_LO, int T_HI>
__global__ __launch_bounds__(256, 4) void vol_sweep_kernel(unsigned char *__restrict__ buf0, unsigned char *__restrict__ buf1,
                                                           int res, int k, float thr,
                                                           VolState *__restrict__ vs, int *__restrict__ act,
                                                           int *__restrict__ tcnt, int *__restrict__ tlist, VolMail *__restrict__ mail) {
    static_assert(T_LO >= -2 && T_HI <= 2 && T_LO <= T_HI, "taps");
    constexpr int NT = T_HI - T_LO + 1;
    __shared__ unsigned A[VA_X * VA_Y * VA_ZS];
    __shared__ unsigned B[VA_X * VA_Y * VB_ZS];
    unsigned *C = A;      // the zy sums overlay the staged tile (its interior bytes are kept in registers): 34.6 KB, 4 workgroups / CU
    static_assert(VA_X * VT_Y * VB_ZS <= VA_X * VA_Y * VA_ZS, "C must fit into A");
    __shared__ int red[2][4];
    __shared__ unsigned long long s_dec[3];
    __shared__ unsigned s_changed[2];      // 27 bits: which of the 3x3x3 tiles around this one (itself = bit 13) see a change
    __shared__ int s_list[256], s_n, s_done;
    __shared__ int s_new[VOL_NEW_CAP], s_new_n;      // tiles this workgroup activated for the next sweep (flushed in bulk)
    const int tid = threadIdx.x;
    // XCD-aware tile order: workgroup i runs on XCD i % 8 (own L2).  Every XCD owns one contiguous slab of tiles: the
    // halo re-reads and the two 64-byte halves of every 128-byte line (adjacent z tiles) stay in ONE L2.  Inside the
    // slab the workgroups of an XCD interleave (tile = first + m * workgroups): the front is spatially coherent, its
    // tiles spread over all of them.
    const int wg = blockIdx.x, xcd = wg & 7, wl = wg >> 3, wgs = gridDim.x >> 3;
    const int tz_n = (res + VT_Z - 1) / VT_Z, ty_n = (res + VT_Y - 1) / VT_Y, tx_n = (res + VT_X - 1) / VT_X;
    const int n_tiles = tz_n * ty_n * tx_n, slab = (n_tiles + 7) >> 3;
    const int slab_end = min((xcd + 1) * slab, n_tiles);
    const int first = xcd * slab + wl;
    const unsigned char *__restrict__ in = (k & 1) ? buf1 : buf0;
    unsigned char *__restrict__ out = (k & 1) ? buf0 : buf1;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(in), 0, res * res * res, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, res * res * res, 0x00020000);
    // ---- wave 0: the reference's loop control (source/sdf.py:156-176) from the finished sweeps' counters, and this
    //      workgroup's share of the ACTIVE tiles.  A tile must be re-evaluated iff a byte inside its halo'd box changed in
    //      the previous sweep; every other tile would reproduce its output: it is skipped, its two buffers already agree
    //      and its zero counts stay in the running totals.  The active tiles of a sweep are a COMPACT list per XCD slab
    //      (generation k % 3: written by sweep k - 1 through the dedupe flags act[k % 3], read here, recycled by sweep
    //      k + 1): workgroup w of the slab takes entries w, w + W, w + 2W ... -- equal shares by construction (with one
    //      flag per tile and static shares the spatially coherent front left most workgroups idle and a few with 8 tiles).
    unsigned w[VS_PER];
    VolTile cur;
    const int gen = k % 3, gen_next = (k + 1) % 3, gen_free = (k + 2) % 3;
    const int *__restrict__ my_list = tlist + ((size_t)gen * 8 + xcd) * slab;
    if (tid < 64) {
        const int done = vs->done;
        const int cnt = vs->lcount[gen][xcd];
        const unsigned long long before = (k <= 1) ? wave_sum64(vs->unknown0, tid) : wave_sum64(vs->cnt[(k - 2) & 3][0], tid);
        const unsigned long long zn = k >= 1 ? wave_sum64(vs->cnt[(k - 1) & 3][1], tid) : 0ull;
        const unsigned long long zx = k >= 1 ? wave_sum64(vs->cnt[(k - 1) & 3][0], tid) : 0ull;
        const int n = wl < cnt ? (cnt - wl + wgs - 1) / wgs : 0;          // <= 256 (host check)
        for (int m = tid; m < n; m += 64) s_list[m] = my_list[wl + m * wgs];
        // the dedupe flags of generation k + 2 (last used by sweep k - 1's appends): this workgroup's static share
        for (int tile = first + tid * wgs; tile < slab_end; tile += 64 * wgs) act[(size_t)gen_free * n_tiles + tile] = 0;
        if (tid == 0) {
            s_dec[0] = before;
            s_dec[1] = zn;
            s_dec[2] = zx;
            s_n = n;
            s_done = done;
            s_changed[0] = s_changed[1] = 0;
            s_new_n = 0;
        }
    }
    __syncthreads();
    if (s_done) return;
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
                if (mail) {
                    mail->final_buf = fin;
                    mail->iters = k;
                    __threadfence_system();
                    __hip_atomic_store(&mail->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            return;
        }
        if (wg == 0 && tid == 0 && mail) __hip_atomic_store(&mail->progress, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (wg == 0 && tid < 2 * NSHARD)      // counters of the next sweep (nobody reads or adds to them now)
            vs->cnt[(k + 1) & 3][tid >> 6][tid & (NSHARD - 1)] = 0;
        if (wg == 0 && tid < 8) vs->lcount[gen_free][tid] = 0;      // the list sweep k + 1 will append to
        if (wg == 0 && tid == 0 && k >= 1) {  // running totals: this sweep's = the previous sweep's + the deltas of the active tiles
            atomicAdd(&vs->cnt[k & 3][0][0], s_dec[2]);
            atomicAdd(&vs->cnt[k & 3][1][0], s_dec[1]);
        }
    }
    const int n_act = s_n;
    if (n_act == 0) return;
    cur = vol_tile(s_list[0], tz_n, ty_n);
    vol_tile_load(w, rsrc, cur, res, tid);
    // All sums are kept BIASED: sign + 1 in {0, 1, 2} per byte, so the z / zy / zyx sums are <= 10 / 50 / 250 -- plain
    // 32-bit adds and subtracts never carry between bytes, and the y and x passes slide their window
    // (out[y] = out[y-1] + entering - leaving).
    // |a| < thr -> 0 for the integer a = acc - bias:  a >= T  <=>  acc >= bias + T ;  a <= -T  <=>  !(acc >= bias - T + 1)
    constexpr int bias = NT * NT * NT;
    const int T = thr > 1.0f ? (int)ceilf(fminf(thr, 1024.0f)) : 1;
    unsigned k_pos, r_pos, k_neg, r_neg;
    lerp_ge_consts(bias + T, k_pos, r_pos);
    lerp_ge_consts(bias - T + 1, k_neg, r_neg);
    const int ty = tid / (VT_Z / 4), tzd = tid - ty * (VT_Z / 4);         // thread = (y, z dword) of the tile
    long long sum_da = 0, sum_db = 0;         // thread 0: deltas of this workgroup's tiles against their previous counts
    for (int i = 0; i < n_act; ++i) {
        const VolTile t = cur;
        vol_tile_stage(w, A, t, res, tid);    // registers -> LDS, z faces replicated
        __syncthreads();
        if (s_new_n > VOL_NEW_CAP - 27)       // stable here (appends happen behind the barrier at the end of a tile): uniform
            vol_list_flush(s_new, &s_new_n, vs->lcount[gen_next], tlist + (size_t)gen_next * 8 * slab, slab, tid);
        if (tid == 0) s_changed[(i + 1) & 1] = 0;
        if (i + 1 < n_act) {                  // next tile: loads in flight during the three passes below
            cur = vol_tile(s_list[i + 1], tz_n, ty_n);
            vol_tile_load(w, rsrc, cur, res, tid);
        }
        // this lane's interior state bytes (its 8 output dwords): read now, A is recycled below
        unsigned raw_in[VT_X];
#pragma unroll
        for (int x = 0; x < VT_X; ++x) raw_in[x] = A[((x + VT_H) * VA_Y + (ty + VT_H)) * VA_ZS + tzd + 1];
        // ---- z sums: one halo'd row per thread
        if (tid < VA_X * VA_Y) {
            unsigned r[VA_ZD];
#pragma unroll
            for (int d = 0; d < VA_ZD; ++d) r[d] = ((A[tid * VA_ZS + d] & 0x03030303u) + 0x01010101u) & 0x03030303u;
#pragma unroll
            for (int d = 0; d < VT_Z / 4; ++d) {
                const unsigned lo = r[d], mi = r[d + 1], hi = r[d + 2];
                unsigned acc = mi;                                                            // offset 0
                if (T_LO <= -2) acc += __builtin_amdgcn_alignbyte(mi, lo, 2);                 // -2: bytes z-2 .. z+1
                if (T_LO <= -1) acc += __builtin_amdgcn_alignbyte(mi, lo, 3);                 // -1
                if (T_HI >= 1) acc += __builtin_amdgcn_alignbyte(hi, mi, 1);                  // +1
                if (T_HI >= 2) acc += __builtin_amdgcn_alignbyte(hi, mi, 2);                  // +2
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
            for (int j = T_LO + 2; j <= T_HI + 2; ++j) acc += v[j];
            C[(ax * VT_Y) * VB_ZS + zd] = acc;
#pragma unroll
            for (int y = 1; y < VT_Y; ++y) {              // window [y + 2 + T_LO, y + 2 + T_HI]
                acc += v[y + 2 + T_HI] - v[y + 1 + T_LO];
                C[(ax * VT_Y + y) * VB_ZS + zd] = acc;
            }
        }
        __syncthreads();
        // ---- x sums, threshold, speculative update, counts
        int known_new = 0, known_next = 0;
        unsigned d_all = 0, d_xlo = 0, d_xhi = 0;        // changed state bits: anywhere / within VT_H of the -x / +x face
        const int gy = t.y0 + ty, gz = t.z0 + 4 * tzd;
        const bool inside = gy < res && gz < res;         // x: res is a multiple of 16 >= VT_X
        {
            unsigned v[VA_X];
#pragma unroll
            for (int ax = 0; ax < VA_X; ++ax) v[ax] = C[(ax * VT_Y + ty) * VB_ZS + tzd];
            unsigned acc = 0;
#pragma unroll
            for (int j = T_LO + 2; j <= T_HI + 2; ++j) acc += v[j];
            const int voff = gy * res + gz;
            if (inside) {
#pragma unroll
                for (int x = 0; x < VT_X; ++x) {
                    if (x > 0) acc += v[x + 2 + T_HI] - v[x + 1 + T_LO];
                    const unsigned raw = raw_in[x];
                    // bit 7 of every byte: sum >= bias + T (-> +1) ; sum < bias - T + 1 (-> -1)
                    const unsigned P = __builtin_amdgcn_lerp(acc, k_pos, r_pos) & 0x80808080u;
                    const unsigned N = ~__builtin_amdgcn_lerp(acc, k_neg, r_neg) & 0x80808080u;
                    const unsigned PN = P | N;
                    const unsigned code = (PN >> 7) | (N >> 6);                         // new sign, 2-bit two's complement
                    const unsigned u1 = (raw >> 2) & 0x01010101u;                       // unknown initially
                    const unsigned m3 = u1 | (u1 << 1);
                    const unsigned o = (raw & ~m3) | (code & m3);                       // v_bfi
                    known_new += __popc(PN);
                    known_next += __popc((o | (o >> 1)) & 0x01010101u);
                    d_all |= o ^ raw;
                    if (x < VT_H) d_xlo |= o ^ raw;
                    if (x >= VT_X - VT_H) d_xhi |= o ^ raw;
                    __builtin_amdgcn_raw_buffer_store_b32(o, rsrc_out, voff, (t.x0 + x) * res * res, 0);
                }
            }
        }
        int z_new = inside ? 4 * VT_X - known_new : 0, z_next = inside ? 4 * VT_X - known_next : 0;
        for (int d = 32; d > 0; d >>= 1) {
            z_new += __shfl_xor(z_new, d);
            z_next += __shfl_xor(z_next, d);
        }
        if ((tid & 63) == 0) {
            red[0][tid >> 6] = z_next;
            red[1][tid >> 6] = z_new;
        }
        // ---- who has to run in the next sweep.  A tile must be re-evaluated iff a byte inside its halo'd box changed:
        //      this tile if anything changed here, the neighbour in direction (dx, dy, dz) only if something changed
        //      within VT_H voxels of the face / edge / corner it touches (a flag per whole tile kept ~2/3 of the volume
        //      active while the front passed: the 27-neighbourhood of an 8 x 16 x 64 tile spans 24 x 48 x 192 voxels).
        //      Bit (dz+1) + 3 (dy+1) + 9 (dx+1).  The z band is the first / last two BYTES of the row's first / last dword.
        {
            const unsigned z_lo = tzd == 0 ? 0x0000ffffu : 0u, z_hi = tzd == VT_Z / 4 - 1 ? 0xffff0000u : 0u;
            const unsigned py = (ty < VT_H ? 1u : 0u) | 8u | (ty >= VT_Y - VT_H ? 64u : 0u);      // dy = -1 / 0 / +1 at bits 0 / 3 / 6
            unsigned m = 0;
            if (d_all) {
                const unsigned mz_lo = ((d_xlo & z_lo) ? 1u : 0u) | (d_xlo ? 2u : 0u) | ((d_xlo & z_hi) ? 4u : 0u);
                const unsigned mz_al = ((d_all & z_lo) ? 1u : 0u) | 2u | ((d_all & z_hi) ? 4u : 0u);
                const unsigned mz_hi = ((d_xhi & z_lo) ? 1u : 0u) | (d_xhi ? 2u : 0u) | ((d_xhi & z_hi) ? 4u : 0u);
                m = (py * mz_lo) | ((py * mz_al) << 9) | ((py * mz_hi) << 18);                     // 3-bit slots: no carries
            }
            for (int d = 32; d > 0; d >>= 1) m |= __shfl_xor(m, d);
            if ((tid & 63) == 0 && m) atomicOr(&s_changed[i & 1], m);
        }
        __syncthreads();                      // also: every read of C (= A) is done before the next tile is staged
        if (tid == 0) {
            // delta against this tile's counts of its previous evaluation
            const int a = red[0][0] + red[0][1] + red[0][2] + red[0][3], b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
            sum_da += (long long)a - tcnt[2 * t.tile];
            sum_db += (long long)b - tcnt[2 * t.tile + 1];
            tcnt[2 * t.tile] = a;
            tcnt[2 * t.tile + 1] = b;
        }
        if (tid < 27 && ((s_changed[i & 1] >> tid) & 1u)) {
            const int nz = t.tz_i + tid % 3 - 1, ny = t.ty_i + (tid / 3) % 3 - 1, nx = t.tx_i + tid / 9 - 1;
            if (nz >= 0 && nz < tz_n && ny >= 0 && ny < ty_n && nx >= 0 && nx < tx_n) {
                const int nt = (nx * ty_n + ny) * tz_n + nz;
                // first to flag it for the next sweep -> it goes onto the list (exactly once)
                if (atomicExch(&act[(size_t)gen_next * n_tiles + nt], 1) == 0) s_new[atomicAdd(&s_new_n, 1)] = nt;
            }
        }
    }
    __syncthreads();
    vol_list_flush(s_new, &s_new_n, vs->lcount[gen_next], tlist + (size_t)gen_next * 8 * slab, slab, tid);
    if (tid == 0) {                           // 64-bit wrap-around = signed add
        if (sum_da) atomicAdd(&vs->cnt[k & 3][0][wg & (NSHARD - 1)], (unsigned long long)sum_da);
        if (sum_db) atomicAdd(&vs->cnt[k & 3][1][wg & (NSHARD - 1)], (unsigned long long)sum_db);
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
    if (p2s_device_count() <= device || device < 0 || device >= P2S_MAX_DEVICES) {
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
    P2sScratchLock scratch_lock(device);     // also guards the per-device pinned mailboxes below
    char *scratch = (char *)scratch_lock.get(counts_at + (2 * NSHARD + 2) * 8);
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
        const int n_tiles = ((grid_res + VT_Z - 1) / VT_Z) * ((grid_res + VT_Y - 1) / VT_Y) * ((grid_res + VT_X - 1) / VT_X);
        // persistent workgroups: 4 fit a CU (LDS) -> 1024 fill the chip; fewer tiles than that: one workgroup per tile
        constexpr int wgs_max = 1024;
        const int slab_tiles = (n_tiles + 7) / 8;
        if ((slab_tiles + wgs_max / 8 - 1) / (wgs_max / 8) > 256) {
            p2s_set_error("p2s_sdf_volume: %d tiles exceed the per-workgroup tile list", n_tiles);
            return cleanup(P2S_EINVAL);
        }
        const dim3 tg((unsigned)std::min(slab_tiles * 8, wgs_max));
        // per tile: counts of its last evaluation (2 ints) and three generations of "active" flags
        int *tcnt = (int *)(vs + 1);
        int *act = tcnt + 2 * (size_t)n_tiles;                    // [3][n_tiles] dedupe flags of the list appends
        int *tlist = act + 3 * (size_t)n_tiles;                   // [3][8][slab] active tiles per generation and XCD slab
        const int slab_n = (n_tiles + 7) / 8;
        if (hipMemsetAsync(tcnt, 0, (size_t)n_tiles * (8 + 12), s) != hipSuccess) {
            p2s_set_error("p2s_sdf_volume: memset failed");
            return cleanup(P2S_EHIP);
        }
        hipLaunchKernelGGL(vol_list_init_kernel, dim3((unsigned)((std::max(n_tiles, 8) + 255) / 256)), dim3(256), 0, s, tlist, vs, n_tiles, slab_n);
        // The number of sweeps is data dependent (the front advances ~2 voxels per sweep).  The device reports its progress
        // and its verdict into a pinned host mailbox; the host keeps only a few launches ahead of it and stops launching
        // when the verdict is there: no copy and no event between the sweeps (a verdict copied out behind batches of 16
        // launches cost six 10-us gaps and ~17 exit-at-once launches of 4.8 us: 9 % of the run at 256^3).
        constexpr int ahead = 4;
        // one mailbox per device (allocated with that device current, portable: visible to every context), used
        // under the device's scratch lock
        static VolMail *mail_h_dev[P2S_MAX_DEVICES] = {};
        static VolMail *mail_d_dev[P2S_MAX_DEVICES] = {};
        VolMail *&mail_h = mail_h_dev[device];
        VolMail *&mail_d = mail_d_dev[device];
        const bool use_mail = !getenv("P2S_VOLUME_NO_MAILBOX");     // test hook: the copy-per-batch protocol below
        if (!mail_h && use_mail) {
            if (hipHostMalloc((void **)&mail_h, sizeof(VolMail), hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess ||
                hipHostGetDevicePointer((void **)&mail_d, mail_h, 0) != hipSuccess) {
                (void)hipGetLastError();
                mail_h = mail_d = nullptr;               // fall back to the copy-per-batch protocol below
            }
        }
        VolState host_vs;
        memset(&host_vs, 0, sizeof(host_vs));
        int k = 0;
        const int k_max = 64 * grid_res + 64;            // far beyond any possible run; guards the host loop only
        auto launch = [&](int kk, VolMail *m) {
            auto go = [&](auto kern) { hipLaunchKernelGGL(kern, tg, dim3(256), 0, s, buf0, buf1, grid_res, kk, certainty_threshold, vs, act, tcnt, tlist, m); };
            switch (sigma) {                  // offsets sigma / 2 - j of scipy's convolve (origin 0)
            case 1: go(vol_sweep_kernel<0, 0>); break;
            case 2: go(vol_sweep_kernel<0, 1>); break;
            case 3: go(vol_sweep_kernel<-1, 1>); break;
            case 4: go(vol_sweep_kernel<-1, 2>); break;
            default: go(vol_sweep_kernel<-2, 2>); break;
            }
        };
        auto check_scatter_flag = [&]() -> bool {
            int flag = 0;
            (void)hipMemcpy(&flag, counts + 2 * NSHARD, 4, hipMemcpyDeviceToHost);
            if (flag) p2s_set_error("p2s_sdf_volume: query point outside the [-1,1) volume");
            return flag == 0;
        };
        bool flag_checked = false;
        if (mail_h && use_mail) {
            volatile VolMail *mv = mail_h;               // (no kernel of this call touches the mailbox before the first sweep)
            mv->progress = 0;
            mv->final_buf = 0;
            mv->iters = 0;
            mv->done = 0;
            __sync_synchronize();
            long long spins = 0;
            while (!mv->done && k < k_max) {
                if (k - mv->progress >= ahead) {
                    if (++spins > 200000000LL) break;     // no progress visible for ~a second: continue with the protocol below
                    continue;
                }
                spins = 0;
                launch(k, mail_d);
                ++k;
            }
            if (mv->done) {
                if (hipStreamSynchronize(s) != hipSuccess) {
                    p2s_set_error("p2s_sdf_volume: %s", hipGetErrorString(hipGetLastError()));
                    return cleanup(P2S_EHIP);
                }
                flag_checked = true;
                if (!check_scatter_flag()) return cleanup(P2S_EINVAL);
                (void)hipMemcpy(&host_vs, vs, sizeof(VolState), hipMemcpyDeviceToHost);
            }
        }
        if (!host_vs.done) {
            // copy-per-batch protocol (no mapped host memory, or the mailbox stayed silent): the verdict of batch j is
            // copied out behind it and looked at only after batch j + 1 has been queued
            constexpr int batch = 16;
            static VolState *pinned_dev[P2S_MAX_DEVICES] = {};
            static hipEvent_t look_dev[P2S_MAX_DEVICES][2] = {};
            VolState *&pinned = pinned_dev[device];
            hipEvent_t *look = look_dev[device];
            if (!pinned) {
                if (hipHostMalloc((void **)&pinned, 2 * sizeof(VolState), hipHostMallocPortable) != hipSuccess ||
                    hipEventCreateWithFlags(&look[0], hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&look[1], hipEventDisableTiming) != hipSuccess) {
                    pinned = nullptr;
                    p2s_set_error("p2s_sdf_volume: pinned verdict buffer: %s", hipGetErrorString(hipGetLastError()));
                    return cleanup(P2S_ENOMEM);
                }
            }
            for (int j = 0; k < k_max; ++j) {
                for (int t = 0; t < batch; ++t, ++k) launch(k, nullptr);
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
                    if (!check_scatter_flag()) return cleanup(P2S_EINVAL);
                }
                host_vs = pinned[(j - 1) & 1];
                if (host_vs.done) break;
            }
            if (!host_vs.done) {                             // the verdict may sit in the batch queued last
                if (hipStreamSynchronize(s) == hipSuccess) {
                    (void)hipMemcpy(&host_vs, vs, sizeof(VolState), hipMemcpyDeviceToHost);
                    if (!flag_checked && !check_scatter_flag()) return cleanup(P2S_EINVAL);
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
