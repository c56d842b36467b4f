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
#include <cstdlib>

namespace {

constexpr int MT = 64;    // points per tile
constexpr int SA = 68;    // LDS row stride (floats) of a 64-channel activation tile
constexpr int SB = 132;   // LDS row stride (floats) of a 128-channel activation tile

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 ldg4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ f32x4 lds4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ f32x4 bufld4(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    f32x4 r;
    r[0] = __uint_as_float(v[0]); r[1] = __uint_as_float(v[1]); r[2] = __uint_as_float(v[2]); r[3] = __uint_as_float(v[3]);
    return r;
}

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

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.0f;
    return z;
}

// B operand of one K=64 layer, one 32-column tile: 8 k-groups = 8 buffer loads (32 VGPRs).  Weights do not
// depend on the activations, so they are fetched BEFORE the barrier that guards the layer's input.
struct B8 {
    f32x4 v[8];
};
__device__ __forceinline__ void load_b8(B8 &b, __amdgpu_buffer_rsrc_t rsrc, int lane16, int soff) {
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) b.v[kg] = bufld4(rsrc, lane16, soff + kg * 1024);
}

// one K=64 layer: RT row tiles (rows row0 + 32 r) x one 32-column tile; A from LDS one k-group ahead, each
// ds_read in its own MFMA shadow; the first k-group accumulates into the literal 0 (no v_mov init)
template <int RT>
__device__ __forceinline__ void layer_k64(const float *src, int ss, int row0, const B8 &b, int lane,
                                          f32x16 (&acc)[RT]) {
    const float *ap = src + (row0 + (lane & 31)) * ss + 4 * (lane >> 5);
    f32x4 a[RT], na[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) a[r] = lds4(ap + 32 * r * ss);
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
        if (kg < 7) {
#pragma unroll
            for (int r = 0; r < RT; ++r) na[r] = lds4(ap + 32 * r * ss + 8 * (kg + 1));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < RT; ++r)
                acc[r] = (kg == 0 && t == 0) ? mfma32(a[r][0], b.v[0][0], zero16()) : mfma32(a[r][t], b.v[kg][t], acc[r]);
        if (kg < 7) {
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
#pragma unroll
            for (int r = 0; r < RT; ++r) a[r] = na[r];
        }
    }
}


// max over the 32 accumulator values of one output column (two row tiles): 1 v_max + 15 v_max3.
// Inline asm because fmaxf() on MFMA results makes hipcc prepend an sNaN-quieting v_max(x, x) to every
// operand (IEEE mode), which doubled the epilogue; (LLVM also folds med3(a, b, +inf) back to that form).
// hipcc does not pad hazards for asm operands, so the first statement carries the XDL-write -> VALU-read
// wait states itself (16-pass MFMA: 19 states; s_nop 15 + s_nop 4 = 21).  asm volatile keeps the order.
__device__ __forceinline__ float tile_colmax(const f32x16 &a, const f32x16 &b) {
    float m;
    asm volatile("s_nop 15\n\ts_nop 4\n\tv_max_f32 %0, %1, %2" : "=v"(m) : "v"(a[0]), "v"(b[0]));
#pragma unroll
    for (int i = 1; i < 16; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(a[i]), "v"(b[i]));
    return m;
}

// max of lane l and lane l^32 in both lanes, without an LDS round trip: v_permlane32_swap exchanges the
// upper half of its first operand with the lower half of the second
__device__ __forceinline__ float half_max(float m) {
    const unsigned u = __float_as_uint(m);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    float x = __uint_as_float(r[0]), y = __uint_as_float(r[1]), o;
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(x), "v"(y));
    return o;
}

// ---- 16-row tail (the last point tile of an item when at most 48 of its 64 rows are points) -------------------------
// Row tile 1 of such a tile holds <= 16 points (P = 300: 12, P = 1000: 8).  v_mfma_f32_16x16x1_4b_f32 multiplies FOUR
// independent 16x16 outer products per instruction, block = lane / 16 -- with the B register of the 32x32x2 layout
// unchanged (lane = 32 kk + col) the blocks are (cols 0-15, kk 0), (cols 16-31, kk 0), (cols 0-15, kk 1), (cols 16-31,
// kk 1): 16 rows x 32 columns x 2 k per instruction at half the cycles of the 32x32x2 form (8 passes), same weight
// fragments, same LDS tile.  The two kk partial sums of a column are added before the pool (one rounding more than
// the 32x32x2 chain on these <= 16 rows).  Saves a quarter of conv3 on one tile in 5 (P = 300) / 16 (P = 1000).
__device__ __forceinline__ f32x16 mfma16b(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, c, 0, 0, 0);
}
template <bool TAIL>
__device__ __forceinline__ f32x16 mfma_r1(float a, float b, f32x16 c) {
    if constexpr (TAIL) return mfma16b(a, b, c);
    else return mfma32(a, b, c);
}
// max over the 16 accumulator values of one output column of ONE row tile (see tile_colmax for the asm)
__device__ __forceinline__ float tile_colmax1(const f32x16 &a) {
    float m;
    asm volatile("s_nop 15\n\ts_nop 4\n\tv_max_f32 %0, %1, %2" : "=v"(m) : "v"(a[0]), "v"(a[1]));
#pragma unroll
    for (int i = 2; i < 16; i += 2) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(a[i]), "v"(a[i + 1]));
    return m;
}
// column max of the 4-block accumulator: D layout block b = regs 4b..4b+3, row = 4 (lane / 16) + reg, col = lane % 16.
// cols 0-15: blocks 0 + 2, cols 16-31: blocks 1 + 3.  Returns, in lane l, the max over the 8 rows that lane l and
// lane l ^ 16 hold of column (l & 31) -- the lane's own column in the 32x32 layout; the l ^ 32 exchange that follows
// in half_max() completes the 16 rows.  v_permlane16_swap(x, y) trades the odd 16-lane rows of x for the even ones of y.
__device__ __forceinline__ float tail_colmax(const f32x16 &c) {
    float sa[4], sb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sa[i] = c[i] + c[8 + i];
        sb[i] = c[4 + i] + c[12 + i];
    }
    float ma, mb;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(ma) : "v"(sa[0]), "v"(sa[1]), "v"(sa[2]));
    asm volatile("v_max_f32 %0, %0, %1" : "+v"(ma) : "v"(sa[3]));
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(mb) : "v"(sb[0]), "v"(sb[1]), "v"(sb[2]));
    asm volatile("v_max_f32 %0, %0, %1" : "+v"(mb) : "v"(sb[3]));
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(ma), __float_as_uint(mb), false, false);
    float x = __uint_as_float(r[0]), y = __uint_as_float(r[1]), o;
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(x), "v"(y));
    return o;
}

// sym_op='sum' (reference source/points_to_surf_model.py:213-214): sum over the VALID rows of the two row tiles of one
// output column.  Rows past the item's last point replicate that point (harmless for a max) and are masked here.
// Plain C++: the compiler pads the XDL-write -> VALU-read hazard itself and adds need no sNaN quieting.
__device__ __forceinline__ float tile_colsum(const f32x16 &a, const f32x16 &b, int nvalid, int lane) {
    float s = 0.0f;
    if (nvalid >= 64) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) s += b[i];
    } else {
        const int r0 = 4 * (lane >> 5);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = r0 + (i & 3) + 8 * (i >> 2);
            s += (r < nvalid) ? a[i] : 0.0f;
            s += (r + 32 < nvalid) ? b[i] : 0.0f;
        }
    }
    return s;
}
__device__ __forceinline__ float half_sum(float m) {
    const unsigned u = __float_as_uint(m);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <bool SUM>
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

    // eight named scalars, not an array: the pair index pr is a run-time value, an indexed array would live in scratch
    const float pool0 = SUM ? 0.0f : -INFINITY;
    float rm0 = pool0, rm1 = pool0, rm2 = pool0, rm3 = pool0, rm4 = pool0, rm5 = pool0, rm6 = pool0, rm7 = pool0;
    // torch's conv/ReLU/MaxPool propagate NaN (a non-finite coordinate poisons every channel of the
    // item); v_max_f32 does not.  Track non-finite inputs and poison the pooled output instead.
    bool bad = false;

    // buffer descriptors of the layer weights (wave-uniform): SGPR base + scalar offset + 16 * lane
    const int lane16 = lane * 16;
    const __amdgpu_buffer_rsrc_t rs0b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(br.w0b), 0, 4096 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w1), 0, 4096 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(br.w2), 0, 8192 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t w3rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w3), 0, 128 * 1024 * 4, 0x00020000);
    // this wave's 8 conv3 column tiles start at byte offset wave * 8 * 16 KB of the packed weights
    const int w3soff = wave * (8 * 16 * 1024);

    // coordinates of this lane's point of a tile (all 4 waves load the same 64 points; L1/L2 hits);
    // the next tile's point is fetched under conv3 of the current one
    auto load_point = [&](int tile, float &x0, float &x1, float &x2) {
        int p = tile * MT + lane;
        if (p >= P) p = P - 1;
        if (p < P1) {
            const float *src = br.ptsA + ((long long)item * P1 + p) * 3;
            x0 = src[0]; x1 = src[1]; x2 = src[2];
        } else {
            const float *src = br.ptsB + ((long long)item * (P - P1) + (p - P1)) * 3;
            x0 = src[0] - cx; x1 = src[1] - cy; x2 = src[2] - cz;
        }
    };

#ifdef P2S_DEV_ABLATE
    const int ablate = args.ablate;   // timing-only variants (wrong results): tools/ablate.sh builds with -DP2S_DEV_ABLATE
#else
    constexpr int ablate = 0;
#endif
    const int ntiles = (P + MT - 1) / MT;
    float nx0, nx1, nx2;
    load_point(0, nx0, nx1, nx2);
    for (int tile = 0; tile < ntiles; ++tile) {
      f32x4 bA0, bA1, aA0, aA1, bB0, bB1, aB0, aB1;   // conv3 operand register sets
      if (ablate != 1) {
        float x0 = nx0, x1 = nx1, x2 = nx2;
        B8 wb;
        if (!short_chain) load_b8(wb, rs0b, lane16, (wave & 1) * 8192);      // conv0b weights, in flight across the barrier
        else load_b8(wb, rs2, lane16, wave * 8192);
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
                f32x16 acc[1];
                layer_k64<1>(bufA, SA, 32 * rt, wb, lane, acc);
                load_b8(wb, rs1, lane16, nt * 8192);                           // conv1 weights
                store_tile_bias_relu(acc[0], bufB, SB, 32 * rt, 32 * nt, br.b0b, lane);
            }
            __syncthreads();
            // ---- conv1 (STN: shared weights; main: per-item W1' = W1 . trans2): bufB -> bufA ---------
            {
                const int rt = wave >> 1, nt = wave & 1;
                f32x16 acc[1];
                layer_k64<1>(bufB, SB, 32 * rt, wb, lane, acc);
                load_b8(wb, rs2, lane16, wave * 8192);                         // conv2 weights
                store_tile_bias_relu(acc[0], bufA, SA, 32 * rt, 32 * nt, br.b1, lane);
            }
            __syncthreads();
        }
        // ---- conv2: bufA[64x64] -> bufB[64x128]; wave w = column tile w, both row tiles --------------
        {
            f32x16 acc[2];
            layer_k64<2>(bufA, SA, 0, wb, lane, acc);
            // first conv3 weight fragments, in flight across the barrier
            bA0 = bufld4(w3rsrc, lane16, w3soff);
            bA1 = bufld4(w3rsrc, lane16, w3soff + 16 * 1024);
            store_tile_bias_relu(acc[0], bufB, SB, 0, 32 * wave, br.b2, lane);
            store_tile_bias_relu(acc[1], bufB, SB, 32, 32 * wave, br.b2, lane);
        }
        __syncthreads();
      } else {
        bA0 = bufld4(w3rsrc, lane16, w3soff);
        bA1 = bufld4(w3rsrc, lane16, w3soff + 16 * 1024);
      }
      if (ablate == 2) continue;
        // ---- conv3 (K = 128, N = 1024) + running max over points -----------------------------------
        // wave w owns channels [256w, 256w+256) = 8 column tiles, processed as 4 pairs with a
        // 2 (row tiles) x 2 (column tiles) register block: 64 accumulator registers.
        // One continuous software pipeline over the 4 x 16 k-groups: two operand register sets (A/B)
        // ping-pong, the operands of k-group g+1 are fetched while the 16 MFMAs of k-group g issue --
        // across pair boundaries too -- and every memory instruction sits in its own MFMA shadow.
        // P2S_TAIL (see mfma16b): the item's last tile with <= 48 points -- row tile 1 runs as 16 rows on the 4-block MFMA.
        // sum pool: padded rows are masked per row of the 32x32 layout (tile_colsum) -- keeps the full tile
        if (!SUM && tile == ntiles - 1 && P - tile * MT <= 48) {
#define P2S_TAIL 1
#include "p2s_chain_conv3.inl"
#undef P2S_TAIL
        } else {
#define P2S_TAIL 0
#include "p2s_chain_conv3.inl"
#undef P2S_TAIL
        }
        // next tile's first layer writes bufA, whose last readers (conv2) are behind the barrier above
    }

    // ---- pooled affine epilogue: out = [relu](max + bias) ------------------------------------------
    const bool any_bad = __ballot(bad) != 0ull;
    if (lane < 32) {
        float *out = br.out + (long long)item * 1024 + 256 * wave + lane;
        const float *b3 = br.b3 + 256 * wave + lane;
        const float rmax[8] = {rm0, rm1, rm2, rm3, rm4, rm5, rm6, rm7};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            float v = rmax[t] + (SUM ? (float)P * b3[32 * t] : b3[32 * t]);     // sum: the bias once per point
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
    unsigned short *outh = args.outh[e];
    if (outh) {
        // 16-bit fragment order: element (k, n) sits at [n / 32][k / 16][lane' = 32 ((k / 8) & 1) + n % 32][k % 8]; this
        // lane holds k = 32 jt + 8 g + 4 kk + t, t = 0..3: four consecutive halfs (8 bytes) per k-group g and piece
        outh += (long long)item * 4096;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int kg = 4 * jt + g;
            unsigned short *dst = outh + ((ot * 4 + (kg >> 1)) * 64 + (kg & 1) * 32 + (lane & 31)) * 8 + 4 * kk;
            float x[4] = {acc[4 * g + 0], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            if (args.f16 && args.bad_items &&
                !(fabsf(x[0]) <= 6.0e4f && fabsf(x[1]) <= 6.0e4f && fabsf(x[2]) <= 6.0e4f && fabsf(x[3]) <= 6.0e4f))
                args.bad_items[item] = 1;          // this item's folded weights leave the half range: its query re-runs in fp32
            for (int q = 0; q < args.ns; ++q) {
                unsigned short h[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (args.f16) {                    // fp16 pair: h0 = fp16(x); h1 = fp16((x - h0) * 2^11)
                        const _Float16 h0 = (_Float16)x[t];
                        h[t] = __builtin_bit_cast(unsigned short, h0);
                        x[t] = (x[t] - (float)h0) * 2048.0f;
                    } else {                           // bf16 pieces: round to nearest even of the residual
                        unsigned u = __float_as_uint(x[t]);
                        u += 0x7fffu + ((u >> 16) & 1u);
                        h[t] = (unsigned short)(u >> 16);
                        x[t] -= __uint_as_float((unsigned)h[t] << 16);
                    }
                }
                uint2 pk;
                pk.x = (unsigned)h[0] | ((unsigned)h[1] << 16);
                pk.y = (unsigned)h[2] | ((unsigned)h[3] << 16);
                *reinterpret_cast<uint2 *>(dst + (long long)q * args.h_piece_stride) = pk;
            }
        }
        return;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 v = {acc[4 * g + 0], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        *reinterpret_cast<f32x4 *>(out + ((ot * 8 + 4 * jt + g) * 64 + lane) * 4) = v;
    }
}

}  // namespace

int p2s_launch_chain(const ChainArgs &args_in, hipStream_t stream) {
    const int n = args_in.br[0].n_items + args_in.br[1].n_items;
    if (n <= 0) return P2S_OK;
    ChainArgs args = args_in;
    constexpr int padlds = 0;
#ifdef P2S_DEV_ABLATE
    static const int ablate = getenv("P2S_CHAIN_ABLATE") ? atoi(getenv("P2S_CHAIN_ABLATE")) : 0;
    args.ablate = ablate;
#endif
    // both branches of a launch pool alike (pass 2 of a sym_op='sum' model: sum; every other launch: max)
    if (args.br[0].pool_sum || (args.br[1].n_items > 0 && args.br[1].pool_sum))
        hipLaunchKernelGGL(p2s_chain_kernel<true>, dim3(n), dim3(256), padlds, stream, args);
    else
        hipLaunchKernelGGL(p2s_chain_kernel<false>, dim3(n), dim3(256), padlds, stream, args);
    P2S_LAUNCH_CHECK("p2s_chain_kernel");
    return P2S_OK;
}

int p2s_launch_fold(const FoldArgs &args, hipStream_t stream) {
    if (args.n_items <= 0) return P2S_OK;
    hipLaunchKernelGGL(p2s_fold_kernel, dim3(args.n_items, 2), dim3(256), 0, stream, args);
    P2S_LAUNCH_CHECK("p2s_fold_kernel");
    return P2S_OK;
}
