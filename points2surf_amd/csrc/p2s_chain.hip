// Fused PointNet point-wise MLP chain + symmetric max-pool for gfx950 (CDNA4).
//
// Replaces, per (query, encoder) item, the reference's eager op sequence
//   conv0a,bn0a,relu / conv0b,bn0b,relu            (source/points_to_surf_model.py:190-191)
//   [bmm(trans2, x)] conv1,bn1,relu / conv2,bn2,relu / conv3,bn3[,relu] / MaxPool1d(P)
//                                                   (:41-49 STN trunk, :195-212 PointNetfeat)
// with ONE kernel in which no per-point activation ever reaches HBM.
//
// Design (MI355X-first, not a translation of the ATen ops):
//  * one 256-thread workgroup (4 waves, one per SIMD) per item; 3 workgroups per CU (50 KB LDS
//    each) so that one workgroup's MFMA stream covers another's epilogue / staging phases;
//  * points are processed in tiles of 64; activations of a tile live in LDS as [point][channel]
//    (row stride K+4 floats -> conflict-free ds_read_b128 of 4 consecutive channels);
//  * every layer with K >= 64 runs on v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD) with
//    points = MFMA rows (A operand, from LDS) and output channels = MFMA columns (B operand);
//    weights are BatchNorm-folded on the host and pre-packed in B-fragment order
//    [N/32][K/8][64 lanes][4] so one coalesced global_load_dwordx4 (L2-resident, 1 KB per wave)
//    feeds 4 MFMAs per accumulator -- weights never pass through LDS;
//    the k index inside a group of 8 is permuted (lane-half kk owns k = 8g+4kk+t) identically for
//    A and B, which is what makes 16-byte operand loads possible with the 32x32x2 layout;
//  * with points on rows, the max-pool over points is 15 in-lane v_max over the accumulator
//    registers + one cross-half exchange per 32x32 tile: the [P x 1024] activation of conv3
//    (87.7 % of all FLOPs) is reduced in registers and never stored anywhere;
//  * bias (and ReLU for STN trunks) commute with the max and are applied once per item.
// Padding rows of the last tile replicate the item's last point (max is idempotent).
#include "p2s_common.h"

namespace {

constexpr int MT = 64;    // points per tile
constexpr int SA = 68;    // LDS row stride (floats) of a 64-channel activation tile
constexpr int SB = 132;   // LDS row stride (floats) of a 128-channel activation tile

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 ldg4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ f32x4 lds4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

// C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ void store_tile_bias_relu(const f32x16 &acc, float *dst, int stride, int row0,
                                                     int col0, const float *__restrict__ bias, int lane) {
    const int col = col0 + (lane & 31);
    const float b = bias[col];
    const int rbase = row0 + 4 * (lane >> 5);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = rbase + (reg & 3) + 8 * (reg >> 2);
        dst[row * stride + col] = fmaxf(acc[reg] + b, 0.0f);
    }
}

// one K=64 layer: RT row tiles (rows row0 + 32 r) x one 32-column tile
template <int RT>
__device__ __forceinline__ void layer_k64(const float *src, int ss, int row0, const float *__restrict__ wp,
                                          int lane, f32x16 (&acc)[RT]) {
    const int arow = row0 + (lane & 31);
    const int koff = 4 * (lane >> 5);
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
        const f32x4 b = ldg4(wp + (kg * 64 + lane) * 4);
        f32x4 a[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) a[r] = lds4(src + (arow + 32 * r) * ss + 8 * kg + koff);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < RT; ++r) acc[r] = mfma32(a[r][t], b[t], acc[r]);
    }
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.0f;
    return z;
}

__device__ __forceinline__ float tile_colmax(const f32x16 &a, const f32x16 &b) {
    float m = fmaxf(a[0], b[0]);
#pragma unroll
    for (int i = 1; i < 16; ++i) m = fmaxf(m, fmaxf(a[i], b[i]));
    return m;
}

__global__ __launch_bounds__(256, 3) void p2s_chain_kernel(ChainArgs args) {
    __shared__ __attribute__((aligned(16))) float smem[MT * SA + MT * SB];
    float *bufA = smem;
    float *bufB = smem + MT * SA;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int item = blockIdx.x;
    int bsel = 0;
    if (item >= args.br[0].n_items) {
        item -= args.br[0].n_items;
        bsel = 1;
    }
    const ChainBranch &br = args.br[bsel];
    const int P = br.P, P1 = br.P1;
    const float *__restrict__ w0a = br.w0a;
    const float *__restrict__ b0a = br.b0a;
    const float *__restrict__ w1 = br.w1 + (long long)item * br.w1_item_stride;
    const float *__restrict__ w3 = br.w3;
    const bool short_chain = br.short_chain != 0;

    float cx = 0.f, cy = 0.f, cz = 0.f;
    if (br.center) {
        cx = br.center[item * 3 + 0];
        cy = br.center[item * 3 + 1];
        cz = br.center[item * 3 + 2];
    }
    float R[9];
    const bool has_rot = br.rot != nullptr;
    if (has_rot) {
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = br.rot[item * 9 + i];
    }

    float rmax[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rmax[i] = -INFINITY;
    // torch's conv/ReLU/MaxPool propagate NaN (a non-finite coordinate poisons every channel of the
    // item); v_max_f32 does not.  Track non-finite inputs and poison the pooled output instead.
    bool bad = false;

    const int ntiles = (P + MT - 1) / MT;
    for (int tile = 0; tile < ntiles; ++tile) {
        // ---- load this lane's point (all 4 waves load the same 64 points; L1/L2 hits) ------------
        int p = tile * MT + lane;
        if (p >= P) p = P - 1;
        float x0, x1, x2;
        if (p < P1) {
            const float *src = br.ptsA + ((long long)item * P1 + p) * 3;
            x0 = src[0]; x1 = src[1]; x2 = src[2];
        } else {
            const float *src = br.ptsB + ((long long)item * (P - P1) + (p - P1)) * 3;
            x0 = src[0] - cx; x1 = src[1] - cy; x2 = src[2] - cz;
        }
        if (has_rot) {
            const float y0 = R[0] * x0 + R[1] * x1 + R[2] * x2;
            const float y1 = R[3] * x0 + R[4] * x1 + R[5] * x2;
            const float y2 = R[6] * x0 + R[7] * x1 + R[8] * x2;
            x0 = y0; x1 = y1; x2 = y2;
        }
        bad = bad || !(fabsf(x0) <= 3.0e38f) || !(fabsf(x1) <= 3.0e38f) || !(fabsf(x2) <= 3.0e38f);
        // ---- first layer (K = 3) on the VALU: wave w produces channels [16w, 16w+16) -------------
        {
            float *dst = bufA + lane * SA + 16 * wave;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int o = 16 * wave + 4 * c4 + c;   // wave-uniform -> scalar loads
                    float s = b0a[o];
                    s = fmaf(w0a[o], x0, s);
                    s = fmaf(w0a[64 + o], x1, s);
                    s = fmaf(w0a[128 + o], x2, s);
                    v[c] = fmaxf(s, 0.0f);
                }
                *reinterpret_cast<f32x4 *>(dst + 4 * c4) = v;
            }
        }
        __syncthreads();   // bufA ready; every wave is past its conv3 reads of bufB (previous tile)

        if (!short_chain) {
            // ---- conv0b: bufA[64x64] -> bufB[:, 0:64] ; wave = (row tile, col tile) ---------------
            {
                const int rt = wave >> 1, nt = wave & 1;
                f32x16 acc[1] = {zero16()};
                layer_k64<1>(bufA, SA, 32 * rt, br.w0b + nt * 8 * 256, lane, acc);
                store_tile_bias_relu(acc[0], bufB, SB, 32 * rt, 32 * nt, br.b0b, lane);
            }
            __syncthreads();
            // ---- conv1 (STN: shared weights; main: per-item W1' = W1 . trans2): bufB -> bufA ---------
            {
                const int rt = wave >> 1, nt = wave & 1;
                f32x16 acc[1] = {zero16()};
                layer_k64<1>(bufB, SB, 32 * rt, w1 + nt * 8 * 256, lane, acc);
                store_tile_bias_relu(acc[0], bufA, SA, 32 * rt, 32 * nt, br.b1, lane);
            }
            __syncthreads();
        }
        // ---- conv2: bufA[64x64] -> bufB[64x128]; wave w = column tile w, both row tiles --------------
        {
            f32x16 acc[2] = {zero16(), zero16()};
            layer_k64<2>(bufA, SA, 0, br.w2 + wave * 8 * 256, lane, acc);
            store_tile_bias_relu(acc[0], bufB, SB, 0, 32 * wave, br.b2, lane);
            store_tile_bias_relu(acc[1], bufB, SB, 32, 32 * wave, br.b2, lane);
        }
        __syncthreads();
        // ---- conv3 (K = 128, N = 1024) + running max over points -----------------------------------
        // wave w owns channels [256w, 256w+256) = 8 column tiles, processed as 4 pairs with a
        // 2 (row tiles) x 2 (column tiles) register block: 64 accumulator registers.
        {
            const float *a0p = bufB + (lane & 31) * SB + 4 * (lane >> 5);
            const float *a1p = a0p + 32 * SB;
#pragma unroll 1
            for (int pr = 0; pr < 4; ++pr) {
                const float *wb0 = w3 + (long long)((8 * wave + 2 * pr) * 16) * 256 + lane * 4;
                const float *wb1 = wb0 + 16 * 256;
                f32x16 c00 = zero16(), c01 = zero16(), c10 = zero16(), c11 = zero16();
                // explicit ping-pong pipeline: the operands of the next k-group are in flight while the
                // 16 MFMAs (1024 cycles) of the current one issue (2x unrolled, two register sets).
#define P2S_MFMA16(A0, A1, B0, B1)                                   \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                  \
        c00 = mfma32(A0[t], B0[t], c00);                             \
        c01 = mfma32(A0[t], B1[t], c01);                             \
        c10 = mfma32(A1[t], B0[t], c10);                             \
        c11 = mfma32(A1[t], B1[t], c11);                             \
    }
                f32x4 bA0 = ldg4(wb0), bA1 = ldg4(wb1);
                f32x4 aA0 = lds4(a0p), aA1 = lds4(a1p);
#pragma unroll 1
                for (int kg = 0; kg < 16; kg += 2) {
                    f32x4 bB0 = ldg4(wb0 + (kg + 1) * 256);
                    f32x4 bB1 = ldg4(wb1 + (kg + 1) * 256);
                    f32x4 aB0 = lds4(a0p + 8 * (kg + 1));
                    f32x4 aB1 = lds4(a1p + 8 * (kg + 1));
                    P2S_MFMA16(aA0, aA1, bA0, bA1)
                    const int kn = (kg + 2 < 16) ? kg + 2 : kg;   // last: harmless re-load
                    bA0 = ldg4(wb0 + kn * 256);
                    bA1 = ldg4(wb1 + kn * 256);
                    aA0 = lds4(a0p + 8 * kn);
                    aA1 = lds4(a1p + 8 * kn);
                    P2S_MFMA16(aB0, aB1, bB0, bB1)
                    // pin the issue order: next operands first, then the MFMA block that hides them
                    // (the default scheduler sinks every load to just before its first use)
                    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // 2 VMEM reads
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS reads
                    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);  // 16 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                }
#undef P2S_MFMA16
                float m0 = tile_colmax(c00, c10);
                float m1 = tile_colmax(c01, c11);
                m0 = fmaxf(m0, __shfl_xor(m0, 32));
                m1 = fmaxf(m1, __shfl_xor(m1, 32));
                // static register indexing (runtime-indexed arrays would go to scratch)
                if (pr == 0) { rmax[0] = fmaxf(rmax[0], m0); rmax[1] = fmaxf(rmax[1], m1); }
                else if (pr == 1) { rmax[2] = fmaxf(rmax[2], m0); rmax[3] = fmaxf(rmax[3], m1); }
                else if (pr == 2) { rmax[4] = fmaxf(rmax[4], m0); rmax[5] = fmaxf(rmax[5], m1); }
                else { rmax[6] = fmaxf(rmax[6], m0); rmax[7] = fmaxf(rmax[7], m1); }
            }
        }
        // next tile's first layer writes bufA, whose last readers (conv2) are behind the barrier above
    }

    // ---- pooled affine epilogue: out = [relu](max + bias) ------------------------------------------
    const bool any_bad = __ballot(bad) != 0ull;
    if (lane < 32) {
        float *out = br.out + (long long)item * 1024 + 256 * wave + lane;
        const float *b3 = br.b3 + 256 * wave + lane;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            float v = rmax[t] + b3[32 * t];
            if (br.relu_out) v = fmaxf(v, 0.0f);
            if (any_bad) v = __builtin_nanf("");
            out[32 * t] = v;
        }
    }
}

// W1'[o][j] = sum_i W1f[o][i] * T[i][j], emitted directly in packed B-fragment order.
// Computed transposed (rows = j, cols = o) so that the MFMA C layout (row = 8g + 4*half + t)
// coincides with the packed layout (k = 8 kg + 4 kk + t): each lane stores whole float4s.
__global__ __launch_bounds__(256) void p2s_fold_kernel(FoldArgs args) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e = blockIdx.y;
    const int item = blockIdx.x;
    const float *__restrict__ T = args.T[e] + (long long)item * 4096;
    const float *__restrict__ wp = args.m1t[e];
    float *__restrict__ out = args.out[e] + (long long)item * 4096;
    const int jt = wave >> 1, ot = wave & 1;
    const int kk = lane >> 5;
    f32x16 acc = zero16();
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
        const f32x4 b = ldg4(wp + ((ot * 8 + kg) * 64 + lane) * 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float a = T[(8 * kg + 4 * kk + t) * 64 + 32 * jt + (lane & 31)];
            acc = mfma32(a, b[t], acc);
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 v = {acc[4 * g + 0], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        *reinterpret_cast<f32x4 *>(out + ((ot * 8 + 4 * jt + g) * 64 + lane) * 4) = v;
    }
}

}  // namespace

int p2s_launch_chain(const ChainArgs &args, hipStream_t stream) {
    const int n = args.br[0].n_items + args.br[1].n_items;
    if (n <= 0) return P2S_OK;
    hipLaunchKernelGGL(p2s_chain_kernel, dim3(n), dim3(256), 0, stream, args);
    P2S_LAUNCH_CHECK("p2s_chain_kernel");
    return P2S_OK;
}

int p2s_launch_fold(const FoldArgs &args, hipStream_t stream) {
    if (args.n_items <= 0) return P2S_OK;
    hipLaunchKernelGGL(p2s_fold_kernel, dim3(args.n_items, 2), dim3(256), 0, stream, args);
    P2S_LAUNCH_CHECK("p2s_fold_kernel");
    return P2S_OK;
}
