// bf16 variant of the fused point-wise MLP chain + max-pool (BASELINE.json configs[3]: "bf16 encoder + fp32 decoder").
//
// Same work split as p2s_chain_kernel (one 256-lane workgroup per (item, encoder), 64-point tiles in LDS as
// [point][channel], points on the MFMA rows, max-pool in registers), but the per-point layers run on
// v_mfma_f32_32x32x16_bf16: activations are rounded to bf16 (nearest even) when they are written to LDS, weights are
// bf16 B fragments, accumulation / bias / ReLU / max stay fp32.  The first layer (K = 3), the STN / QSTN heads, the
// fold W1' = W1 . trans2 and the decoder stay fp32 (the fold result is re-packed to bf16 fragments per item).
//
// Fragment layout (A from LDS, B from the packed weights): lane l holds 8 consecutive k of row / column l & 31,
// k = 16 * kb + 8 * (l >> 5) + [0, 8)  -- one 16-byte load each.  Any permutation of k inside a fragment cancels as long
// as A and B use the same one.  C / D: column l & 31, rows (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), as for 32x32x2.
//
// conv3 (128 -> 1024) dominates: a wave keeps the A fragments of both row blocks for all 8 k-blocks in registers
// (64 VGPRs) and streams its 8 column tiles of weights from L2 (64 KB per wave and tile).  Measured ~40 % of the bf16
// MFMA peak; 128-point tiles (-DP2S_BF16_MT=128: half the weight stream, 2 workgroups/CU) are no faster.
//
// Split precision (template parameter NS = 2 or 3; cfg.encoder_bf16 = 2 / 3): every activation and weight is carried as
// NS bf16 pieces  x = x0 + x1 (+ x2),  x_p = bf16(residual),  and a product is the sum of the piece products with
// p + q < NS (3 MFMAs for NS = 2, 6 for NS = 3), all into the same fp32 accumulator: 16 / 24 mantissa bits per operand
// at 1/16 of the fp32 MFMA cost per piece product.  Arithmetic model + measured deviation: DESIGN.md "bf16 modes".
//
// fp16 pair mode (template parameter F16; cfg.encoder_bf16 = 4, "fp16x2"): every operand as TWO fp16 numbers,
//   x = h0 + h1 * 2^-11,   h0 = fp16(x),   h1 = fp16((x - h0) * 2^11)     (22 mantissa bits; the residual is scaled so
// that it never goes subnormal), and a product = h0 h0' into one fp32 accumulator, (h0 h1' + h1 h0') into a second one
// that enters with the factor 2^-11 at the end -- THREE v_mfma_f32_32x32x16_f16 per product instead of the six bf16
// ones of the three-piece mode, for the same deviation from fp32 (relative feature error 5e-7 vs 4e-7 in the CPU model;
// two bf16 pieces: 1.3e-5).  fp16 ends at 65504: an activation beyond 6e4 poisons its item (NaN) AND flags the item's
// query (ChainArgs.bad_items); flagged queries are collected per chunk and re-run through the fp32 kernels at the end of
// the same call (p2s_api.hip: fallback) -- an arbitrary checkpoint cannot turn this mode into wrong values or an error
// (the reference's activations stay below 20 with the weights at hand: nothing is ever flagged there).
#include "p2s_common.h"
#include <cmath>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#ifndef P2S_BF16_MT
#define P2S_BF16_MT 64
#endif
#ifndef P2S_BF16_HOLD
#define P2S_BF16_HOLD 1      // split modes: pieces whose conv3 A fragments stay in registers (the rest: LDS per k-step)
#endif
// fp16 pair mode: pieces whose conv3 A fragments stay in registers / workgroups per CU the register budget is set for
// (53 KB of LDS allow 3).  Measured, ms per 4096 queries for both chain passes: HOLD 0 / WG 3 (127 VGPRs): 8.20;
// HOLD 1 / WG 3 (168 VGPRs + 20 B scratch): 8.09; HOLD 1 / WG 2: 8.50; HOLD 2 / WG 2: 8.31 -> everything from LDS, 3 per CU
#ifndef P2S_F16_HOLD
#define P2S_F16_HOLD 0
#endif
#ifndef P2S_F16_WG
#define P2S_F16_WG 3
#endif
constexpr int MT = P2S_BF16_MT;  // points per tile (64 or 128: larger tiles halve the L2 weight stream of conv3)
constexpr int NB = MT / 32;      // 32-row blocks per tile
constexpr int PPL = MT / 64;     // points per lane in the first layer
static_assert(MT == 64 || MT == 128, "tile of 64 or 128 points");
constexpr int HA = 64 + 8;      // halfs per row of the 64-channel buffer (16-byte rows, bank spread)
constexpr int HB = 128 + 8;     // halfs per row of the 128-channel buffer

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned short f2bf(float f) {       // round to nearest even (finite inputs)
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
// two values at once: one v_cvt_pk_bf16_f32 (hardware round-to-nearest-even); a in the low half
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

template <bool F16>
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    if (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pack_f16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}

__device__ __forceinline__ u32x4 bufld(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
}

// A fragment of rows row0 .. row0+31, k-block kb
__device__ __forceinline__ u32x4 lds_a(const unsigned short *buf, int H, int row0, int kb, int lane) {
    return *reinterpret_cast<const u32x4 *>(buf + (row0 + (lane & 31)) * H + 16 * kb + 8 * (lane >> 5));
}

__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// two fp32 values -> NS pieces each; out[p] = packed pair (a low half, b high half).
// bf16: piece p = bf16 of the residual.  fp16 pair: h0 = fp16(x), h1 = fp16((x - h0) * 2^11).
template <int NS, bool F16>
__device__ __forceinline__ void split_pair(float a, float b, unsigned (&out)[NS]) {
    if (F16) {
        out[0] = pack_f16(a, b);
        const f16x2 h = __builtin_bit_cast(f16x2, out[0]);
        if (NS > 1) out[NS > 1 ? 1 : 0] = pack_f16((a - (float)h[0]) * 2048.0f, (b - (float)h[1]) * 2048.0f);
        return;
    }
#pragma unroll
    for (int p = 0; p < NS; ++p) {
        out[p] = pack_bf16(a, b);
        if (p + 1 < NS) {
            a -= bf_hi(out[p]);
            b -= bf_lo(out[p]);
        }
    }
}

constexpr float F16_LIMIT = 6.0e4f;       // |activation| beyond this does not fit fp16 (max 65504)
constexpr float F16_SCALE = 1.0f / 2048.0f;

// Small layers run TRANSPOSED: D = W^T X^T, the weight fragment is the MFMA's first operand and the activation fragment the
// second (the register contents of an A and a B fragment are the same: 8 consecutive k of row / column l & 31), so a lane's
// 16 results are ONE point (l & 31) and four groups of 4 consecutive channels, (i & 3) + 8 (i >> 2) + 4 (l >> 5): in the
// [point][channel] tile that is one 8-byte store per group and piece (4 x ds_write_b64 instead of 16 x ds_write_b16 per
// piece; 272- / 144-byte rows: the 64 lanes of a store cover every bank twice).
// acc (+ bias, ReLU) -> pieces into buf[p][pt0 + point][ch0 + channel].  fp16 pair: the value is acc[0] + acc[1] * 2^-11
template <int NS, bool F16>
__device__ __forceinline__ void store_tile(const f32x16 (&acc)[F16 ? 2 : 1], unsigned short *buf, int H, int pstride, int pt0, int ch0,
                                           const float *__restrict__ bias, int lane, bool &range_bad) {
    unsigned short *dst = buf + (pt0 + (lane & 31)) * H + ch0 + 4 * (lane >> 5);
    const float *bsrc = bias + ch0 + 4 * (lane >> 5);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4 *>(bsrc + 8 * g);
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            v[t] = acc[0][4 * g + t];
            if (F16) v[t] = fmaf(acc[F16 ? 1 : 0][4 * g + t], F16_SCALE, v[t]);
            v[t] = fmaxf(v[t] + b[t], 0.0f);
            if (F16) range_bad = range_bad || !(v[t] <= F16_LIMIT);       // (NaN too: an operand beyond the range made inf * 0)
        }
        unsigned lo[NS], hi[NS];
        split_pair<NS, F16>(v[0], v[1], lo);
        split_pair<NS, F16>(v[2], v[3], hi);
#pragma unroll
        for (int p = 0; p < NS; ++p) {
            const u32x2 w = {lo[p], hi[p]};
            *reinterpret_cast<u32x2 *>(dst + p * pstride + 8 * g) = w;
        }
    }
}

// SUM: sym_op='sum' -- the pool over the points is a masked sum instead of the max (a compile-time variant: as a run-time
// branch inside the unrolled column-tile loop it cost the max kernels 40 VGPRs and pushed the fp16 pair kernel into scratch)
// Register budgets (workgroups per CU the compiler allocates for): fp16 pair 3 (127 VGPRs; sum pool 153); plain bf16 4 (127;
// its sum pool 3 -- at 128 VGPRs it spilled 15); split bf16 2 (144 / 164 VGPRs, sum pool 173 / 187: their 53 / 80 KB of LDS
// allow 3 / 2 workgroups anyway, and the three-piece mode is the one inside the accuracy contract).  No scratch in any of them.
template <int NS, bool F16, bool SUM = false>
__global__ __launch_bounds__(256, F16 ? P2S_F16_WG : ((MT == 64 && NS == 1) ? (SUM ? 3 : 4) : 2)) void p2s_chain_bf16_kernel(ChainArgs args) {
    constexpr int NA = F16 ? 2 : 1;                       // accumulators per tile (fp16 pair: second one scaled by 2^-11)
    extern __shared__ __attribute__((aligned(16))) unsigned short lds_bf16[];
    constexpr int SA = MT * HA, SB = MT * HB;             // halfs per piece
    unsigned short *bufA = lds_bf16;                      // [NS][MT][HA]
    unsigned short *bufB = lds_bf16 + NS * SA;            // [NS][MT][HB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int item = blockIdx.x, bsel = 0;
    if (item >= args.br[0].n_items) {
        item -= args.br[0].n_items;
        bsel = 1;
    }
    const ChainBranch &br = args.br[bsel];
    const int P = br.P, P1 = br.P1;
    const bool short_chain = br.short_chain != 0;
    const float *__restrict__ w0a = br.w0a;
    const float *__restrict__ b0a = br.b0a;
    // bf16 fragment arrays (the ChainBranch pointer fields are reused for them)
    // buffer descriptors (wave-uniform SGPRs) + scalar offsets + 16 * lane: no 64-bit VALU address math, no address VGPRs
    // piece q of a weight array sits args.piece_stride halfs (per-item W1': args.w1_piece_stride) behind piece 0
    const unsigned short *w1p = reinterpret_cast<const unsigned short *>(br.w1) + (long long)item * br.w1_item_stride;
    const long long ps = args.piece_stride, ps1 = br.w1_item_stride ? args.w1_piece_stride : args.piece_stride;
    // ONE descriptor per layer spanning all pieces; piece q is selected through the scalar offset (q * stride bytes):
    // 4 descriptors instead of 4 * NS (12 descriptors = 48 SGPRs spilled into VGPR lanes and were re-read in the loops)
    const int pb = (int)(ps * 2), pb1 = (int)(ps1 * 2);          // piece strides in bytes
    const __amdgpu_buffer_rsrc_t rs0b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short *>(reinterpret_cast<const unsigned short *>(br.w0b)), 0, (NS - 1) * pb + 4096 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(w1p), 0,
                                                                          (NS - 1) * pb1 + 4096 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short *>(reinterpret_cast<const unsigned short *>(br.w2)), 0, (NS - 1) * pb + 8192 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs3 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short *>(reinterpret_cast<const unsigned short *>(br.w3)), 0, (NS - 1) * pb + 128 * 1024 * 2, 0x00020000);
    const int lane16 = lane * 16;

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
    constexpr bool psum = SUM;                            // sym_op='sum': masked sum over the points instead of the max
#pragma unroll
    for (int i = 0; i < 8; ++i) rmax[i] = psum ? 0.0f : -INFINITY;
    bool bad = false;       // non-finite input poisons the item (torch propagates NaN through conv / ReLU / max)
    bool range_bad = false; // fp16 pair mode: an activation beyond the half range

    const int ntiles = (P + MT - 1) / MT;
    for (int tile = 0; tile < ntiles; ++tile) {
        // ---- this lane's point(s) (all 4 waves load the same points; past the end: the last point again) ----
#pragma unroll
        for (int pp = 0; pp < PPL; ++pp) {
            float x0, x1, x2;
            {
                int p = tile * MT + pp * 64 + lane;
                if (p >= P) p = P - 1;
                if (p < P1) {
                    const float *src = br.ptsA + ((long long)item * P1 + p) * 3;
                    x0 = src[0]; x1 = src[1]; x2 = src[2];
                } else {
                    const float *src = br.ptsB + ((long long)item * (P - P1) + (p - P1)) * 3;
                    x0 = src[0] - cx; x1 = src[1] - cy; x2 = src[2] - cz;
                }
            }
            if (has_rot) {
                const float y0 = R[0] * x0 + R[1] * x1 + R[2] * x2;
                const float y1 = R[3] * x0 + R[4] * x1 + R[5] * x2;
                const float y2 = R[6] * x0 + R[7] * x1 + R[8] * x2;
                x0 = y0; x1 = y1; x2 = y2;
            }
            bad = bad || !(fabsf(x0) <= 3.0e38f) || !(fabsf(x1) <= 3.0e38f) || !(fabsf(x2) <= 3.0e38f);
            // ---- first layer (K = 3, fp32 VALU): wave w -> channels [16w, 16w+16) of bufA ----
            unsigned short *dst = bufA + (pp * 64 + lane) * HA + 16 * wave;
#pragma unroll
            for (int c = 0; c < 16; c += 2) {
                float sv[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int o = 16 * wave + c + u;     // wave-uniform -> scalar loads
                    float v = b0a[o];
                    v = fmaf(w0a[o], x0, v);
                    v = fmaf(w0a[64 + o], x1, v);
                    v = fmaf(w0a[128 + o], x2, v);
                    sv[u] = fmaxf(v, 0.0f);
                }
                if (F16) range_bad = range_bad || !(sv[0] <= F16_LIMIT) || !(sv[1] <= F16_LIMIT);
                unsigned u[NS];
                split_pair<NS, F16>(sv[0], sv[1], u);
#pragma unroll
                for (int p = 0; p < NS; ++p) *reinterpret_cast<unsigned *>(dst + p * SA + c) = u[p];
            }
        }
        __syncthreads();          // bufA ready; every wave is past its conv3 reads of bufB (previous tile)

        if (!short_chain) {
            // ---- conv0b: bufA -> bufB[:, 0:64]; wave = (row block, column tile) ----
            {
                const int rt = wave >> 1, nt = wave & 1;          // row blocks rt, rt + 2, ...; column tile nt
                f32x16 acc[NB / 2][NA];
#pragma unroll
                for (int r = 0; r < NB / 2; ++r)
#pragma unroll
                    for (int t = 0; t < NA; ++t) acc[r][t] = f32x16{};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    u32x4 b[NS];
#pragma unroll
                    for (int q = 0; q < NS; ++q) b[q] = bufld(rs0b, lane16, q * pb + (nt * 4 + kb) * 1024);
#pragma unroll
                    for (int r = 0; r < NB / 2; ++r)
#pragma unroll
                        for (int p = NS - 1; p >= 0; --p) {              // small terms first
                            const u32x4 a = lds_a(bufA + p * SA, HA, 32 * (rt + 2 * r), kb, lane);
#pragma unroll
                            for (int q = NS - 1 - p; q >= 0; --q)
                                acc[r][F16 ? p + q : 0] = mfma_bf16<F16>(b[q], a, acc[r][F16 ? p + q : 0]);   // transposed: channels x points
                        }
                }
#pragma unroll
                for (int r = 0; r < NB / 2; ++r) store_tile<NS, F16>(acc[r], bufB, HB, SB, 32 * (rt + 2 * r), 32 * nt, br.b0b, lane, range_bad);
            }
            __syncthreads();
            // ---- conv1 (STN pass: shared weights; main pass: this item's W1' = W1 . trans2): bufB -> bufA ----
            {
                const int rt = wave >> 1, nt = wave & 1;
                f32x16 acc[NB / 2][NA];
#pragma unroll
                for (int r = 0; r < NB / 2; ++r)
#pragma unroll
                    for (int t = 0; t < NA; ++t) acc[r][t] = f32x16{};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    u32x4 b[NS];
#pragma unroll
                    for (int q = 0; q < NS; ++q) b[q] = bufld(rs1, lane16, q * pb1 + (nt * 4 + kb) * 1024);
#pragma unroll
                    for (int r = 0; r < NB / 2; ++r)
#pragma unroll
                        for (int p = NS - 1; p >= 0; --p) {
                            const u32x4 a = lds_a(bufB + p * SB, HB, 32 * (rt + 2 * r), kb, lane);
#pragma unroll
                            for (int q = NS - 1 - p; q >= 0; --q)
                                acc[r][F16 ? p + q : 0] = mfma_bf16<F16>(b[q], a, acc[r][F16 ? p + q : 0]);   // transposed: channels x points
                        }
                }
#pragma unroll
                for (int r = 0; r < NB / 2; ++r) store_tile<NS, F16>(acc[r], bufA, HA, SA, 32 * (rt + 2 * r), 32 * nt, br.b1, lane, range_bad);
            }
            __syncthreads();
        }
        // ---- conv2 (64 -> 128): bufA -> bufB; wave = column tile, both row blocks ----
        {
            f32x16 acc[NB][NA];
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
                for (int t = 0; t < NA; ++t) acc[r][t] = f32x16{};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                u32x4 b[NS];
#pragma unroll
                for (int q = 0; q < NS; ++q) b[q] = bufld(rs2, lane16, q * pb + (wave * 4 + kb) * 1024);
#pragma unroll
                for (int r = 0; r < NB; ++r)
#pragma unroll
                    for (int p = NS - 1; p >= 0; --p) {
                        const u32x4 a = lds_a(bufA + p * SA, HA, 32 * r, kb, lane);
#pragma unroll
                        for (int q = NS - 1 - p; q >= 0; --q)
                            acc[r][F16 ? p + q : 0] = mfma_bf16<F16>(b[q], a, acc[r][F16 ? p + q : 0]);   // transposed: channels x points
                    }
            }
#pragma unroll
            for (int r = 0; r < NB; ++r) store_tile<NS, F16>(acc[r], bufB, HB, SB, 32 * r, 32 * wave, br.b2, lane, range_bad);
        }
        __syncthreads();
        // ---- conv3 (128 -> 1024) + max over the 64 points: wave w owns column tiles [8w, 8w+8) ----
        // The A fragments of the first HOLD pieces stay in registers for all 8 column tiles (64 VGPRs per piece); the
        // other pieces are re-read from LDS in every k-step (conflict-free 16-byte reads, 2 x (NS - HOLD) per 2 NS (NS+1)/2
        // MFMAs).  Holding all three pieces of the split mode needs 192 VGPRs for A alone and spilled (r02: 48 B/lane,
        // 4.9x the algorithmic write traffic).
        {
            constexpr int HOLDW = F16 ? P2S_F16_HOLD : P2S_BF16_HOLD;
            constexpr int HOLD = (NS == 1) ? 1 : ((HOLDW < NS) ? HOLDW : NS);
            u32x4 af[HOLD > 0 ? HOLD : 1][NB][8];
#pragma unroll
            for (int p = 0; p < HOLD; ++p)
#pragma unroll
                for (int r = 0; r < NB; ++r)
#pragma unroll
                    for (int kb = 0; kb < 8; ++kb) af[p][r][kb] = lds_a(bufB + p * SB, HB, 32 * r, kb, lane);
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
                const int soff = (wave * 8 + ct) * 8 * 1024;        // bytes: 8 k-blocks of 64 lanes x 16 B per column tile
                f32x16 acc[NB][NA];
#pragma unroll
                for (int r = 0; r < NB; ++r)
#pragma unroll
                    for (int t = 0; t < NA; ++t) acc[r][t] = f32x16{};
                u32x4 b[NS];
                u32x4 a[NS][NB];
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) {
#pragma unroll
                    for (int q = 0; q < NS; ++q)
                        b[q] = bufld(rs3, lane16, q * pb + soff + kb * 1024);
#pragma unroll
                    for (int p = 0; p < NS; ++p)
#pragma unroll
                        for (int r = 0; r < NB; ++r)
                            a[p][r] = (p < HOLD) ? af[p < HOLD ? p : 0][r][kb] : lds_a(bufB + p * SB, HB, 32 * r, kb, lane);
#pragma unroll
                    for (int r = 0; r < NB; ++r)
#pragma unroll
                        for (int p = NS - 1; p >= 0; --p)
#pragma unroll
                            for (int q = NS - 1 - p; q >= 0; --q)
                                acc[r][F16 ? p + q : 0] = mfma_bf16<F16>(a[p][r], b[q], acc[r][F16 ? p + q : 0]);
                }
                float m;
                if (psum) {
                    const int nvalid = P - tile * MT;
                    m = 0.0f;
#pragma unroll
                    for (int r = 0; r < NB; ++r)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int row = 32 * r + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                            const float v = F16 ? acc[r][0][i] + acc[r][F16 ? 1 : 0][i] * F16_SCALE : acc[r][0][i];
                            m += (row < nvalid) ? v : 0.0f;
                        }
                    m += __shfl_xor(m, 32);
                    rmax[ct] += m;
                } else {
                    m = -INFINITY;
#pragma unroll
                    for (int r = 0; r < NB; ++r)
#pragma unroll
                        for (int i = 0; i < 16; i += 2) {
                            f32x2 v = {acc[r][0][i], acc[r][0][i + 1]};
                            if (F16) {                    // one v_pk_fma_f32 per pair, one v_max3_f32 per pair
                                const f32x2 lo = {acc[r][F16 ? 1 : 0][i], acc[r][F16 ? 1 : 0][i + 1]};
                                const f32x2 sc = {F16_SCALE, F16_SCALE};
                                v = __builtin_elementwise_fma(lo, sc, v);
                            }
                            m = fmaxf(fmaxf(m, v[0]), v[1]);
                        }
                    m = fmaxf(m, __shfl_xor(m, 32));
                    rmax[ct] = fmaxf(rmax[ct], m);
                }
                // keep the weight loads of the next column tile below this point: hoisting all 64 of them spills
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // (the next tile's first layer writes bufA, last read before the barrier above)
    }
    // ---- pooled output: bias (and ReLU for the STN trunks) commute with the max ----
    if (F16) {
        if (__ballot(range_bad) != 0ull) {
            bad = true;                                   // poison this wave's columns of the item ...
            if (lane == 0 && args.bad_items) args.bad_items[item] = 1;           // ... and queue the query for the fp32 kernels
        }
    }
    if (lane < 32) {
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            const int c = (wave * 8 + ct) * 32 + lane;
            float v = rmax[ct] + (psum ? (float)P * br.b3[c] : br.b3[c]);          // sum: the bias once per point
            if (br.relu_out) v = fmaxf(v, 0.0f);
            if (bad) v = __int_as_float(0x7fc00000);
            br.out[(long long)item * 1024 + c] = v;
        }
    }
}

// fp32 packed B fragments ([N/32][K/8][2][32][4]: k = 8 kg + 4 kk + t, n = 32 nt + j) -> bf16 fragments
// ([N/32][K/16][64 lanes][8]: k = 16 kb + 8 (lane >> 5) + t, n = 32 nt + (lane & 31)); one thread per output element
__global__ void p2s_pack_bf16_kernel(const float *__restrict__ src, unsigned short *__restrict__ dst, int K, int N,
                                     long long src_stride, long long dst_stride, int n_items, int piece, int f16,
                                     int *__restrict__ range_flag) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (long long)K * N;
    if (e >= per * n_items) return;
    const int item = (int)(e / per);
    int r = (int)(e % per);
    const int t = r & 7;
    r >>= 3;
    const int lane = r & 63;
    r >>= 6;
    const int KB = K / 16;
    const int kb = r % KB, nt = r / KB;
    const int k = 16 * kb + 8 * (lane >> 5) + t, j = lane & 31;
    const int kg = k >> 3, kk = (k >> 2) & 1, ts = k & 3;
    const long long si = ((((long long)nt * (K / 8) + kg) * 2 + kk) * 32 + j) * 4 + ts;
    float x = src[(long long)item * src_stride + si];
    if (f16) {                                        // fp16 pair: h0 = fp16(x); h1 = fp16((x - h0) * 2^11)
        if (range_flag && !(fabsf(x) <= F16_LIMIT)) atomicOr(range_flag, 1);      // a WEIGHT beyond the half range (or non-finite)
        _Float16 h0 = (_Float16)x;
        if (piece == 1) h0 = (_Float16)((x - (float)h0) * 2048.0f);
        dst[(long long)item * dst_stride + (e % per)] = __builtin_bit_cast(unsigned short, h0);
        return;
    }
    unsigned short h = f2bf(x);
    for (int q = 0; q < piece; ++q) {                 // piece q = bf16 of the residual after the pieces before it
        x -= __uint_as_float((unsigned)h << 16);
        h = f2bf(x);
    }
    dst[(long long)item * dst_stride + (e % per)] = h;
}

}  // namespace

template <int NS, bool F16, bool SUM>
static void launch_bf16(const ChainArgs &args, int n, size_t lds, hipStream_t stream) {
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void *)p2s_chain_bf16_kernel<NS, F16, SUM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((p2s_chain_bf16_kernel<NS, F16, SUM>), dim3(n), dim3(256), lds, stream, args);
}

int p2s_launch_chain_bf16(const ChainArgs &args, hipStream_t stream) {
    const int n = args.br[0].n_items + args.br[1].n_items;
    if (n <= 0) return P2S_OK;
    const int ns = args.ns < 1 ? 1 : args.ns;
    const size_t lds = (size_t)ns * MT * (HA + HB) * 2;
    // both branches of a launch pool alike (pass 2 of a sym_op='sum' model: sum; every other launch: max)
    const bool sum = args.br[0].pool_sum || (args.br[1].n_items > 0 && args.br[1].pool_sum);
    if (args.f16) {
        if (ns != 2) {
            p2s_set_error("p2s_launch_chain_bf16: the fp16 pair mode has 2 pieces, not %d", ns);
            return P2S_EINVAL;
        }
        if (sum) launch_bf16<2, true, true>(args, n, lds, stream);
        else launch_bf16<2, true, false>(args, n, lds, stream);
    } else if (ns == 1) {
        if (sum) launch_bf16<1, false, true>(args, n, lds, stream);
        else launch_bf16<1, false, false>(args, n, lds, stream);
    } else if (ns == 2) {
        if (sum) launch_bf16<2, false, true>(args, n, lds, stream);
        else launch_bf16<2, false, false>(args, n, lds, stream);
    } else if (ns == 3) {
        if (sum) launch_bf16<3, false, true>(args, n, lds, stream);
        else launch_bf16<3, false, false>(args, n, lds, stream);
    } else {
        p2s_set_error("p2s_launch_chain_bf16: %d pieces unsupported", ns);
        return P2S_EINVAL;
    }
    P2S_LAUNCH_CHECK("p2s_chain_bf16_kernel");
    return P2S_OK;
}

int p2s_launch_pack_bf16(const float *src, unsigned short *dst, int K, int N, long long src_stride, long long dst_stride,
                         int n_items, int piece, int f16, hipStream_t stream, int *range_flag) {
    const long long total = (long long)K * N * n_items;
    if (total <= 0) return P2S_OK;
    hipLaunchKernelGGL(p2s_pack_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, dst, K, N,
                       src_stride, dst_stride, n_items, piece, f16, range_flag);
    P2S_LAUNCH_CHECK("p2s_pack_bf16_kernel");
    return P2S_OK;
}
