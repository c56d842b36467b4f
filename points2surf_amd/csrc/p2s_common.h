// Internal shared declarations of libp2s_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include "../../include/p2s_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void p2s_set_error(const char *fmt, ...);
// Process-wide grow-only scratch buffer per device (volume / iso-surface stages: ~0.5 ms of hipMalloc + hipFree per
// call otherwise).  A caller holds the lock for its whole call -- host threads serialise per device -- and synchronises
// its stream before it returns, so the buffer is idle whenever the lock is free.  p2s_release_scratch() frees it.
constexpr int P2S_MAX_DEVICES = 16;
class P2sScratchLock {
public:
    explicit P2sScratchLock(int device);
    ~P2sScratchLock();
    P2sScratchLock(const P2sScratchLock &) = delete;
    P2sScratchLock &operator=(const P2sScratchLock &) = delete;
    void *get(size_t bytes);          // nullptr on failure; valid until the next get() on this device
private:
    int dev_;
};
void p2s_cloud_pool_release(int device);     // p2s_cloud.hip: cached cloud arenas of the device

#define P2S_HIP_CHECK(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            p2s_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,    \
                          __LINE__);                                                          \
            return P2S_EHIP;                                                                  \
        }                                                                                     \
    } while (0)

#define P2S_LAUNCH_CHECK(name)                                                                \
    do {                                                                                      \
        hipError_t _e = hipGetLastError();                                                    \
        if (_e != hipSuccess) {                                                               \
            p2s_set_error("launch of %s failed: %s", name, hipGetErrorString(_e));            \
            return P2S_EHIP;                                                                  \
        }                                                                                     \
    } while (0)

// ---------------------------------------------------------------------------------------------
// Fused point-wise MLP chain + symmetric max-pool (p2s_chain.hip)
// ---------------------------------------------------------------------------------------------
struct ChainBranch {
    const float *ptsA;       // [n_items][P1][3]  first P1 points of every item (no centre)
    const float *ptsB;       // [n_items][P-P1][3] remaining points (centre subtracted), may be null
    const float *center;     // [n_items][3] or null
    const float *rot;        // [n_items][9] row-major 3x3 applied to every point, or null
    const float *w0a, *b0a;  // first (K=3) layer: [3][64], [64]
    const float *w0b, *b0b;  // packed 64x64 (full chain only)
    const float *w1, *b1;    // packed 64x64 (full chain only); per item if w1_item_stride != 0
    long long w1_item_stride;
    const float *w2, *b2;    // packed 64x128
    const float *w3, *b3;    // packed 128x1024
    float *out;              // [n_items][1024]
    int P, P1;
    int n_items;
    int relu_out;            // ReLU after the pooled affine (STN trunks) or not (PointNetfeat.conv3)
    int short_chain;         // 1: 3->64->128->1024 (QSTN trunk), 0: 3->64->64->64->128->1024
    int pool_sum;            // 1: the pool over the points is a SUM (sym_op='sum', main trunk only), out = sum + P * b3; 0: max
};
struct ChainArgs {
    ChainBranch br[2];       // br[0] items come first in the grid
    int ablate;              // only read when built with -DP2S_DEV_ABLATE (timing variants: 1 = conv3 only, 2 = all but conv3)
    // bf16 kernels: number of bf16 pieces per operand (1 = plain bf16, 2 / 3 = split precision) and the distance in
    // halfs between the pieces of a weight array (shared weights / per-item W1')
    int ns;
    long long piece_stride, w1_piece_stride;
    int f16;                 // 1: the pieces are an fp16 pair (h0, h1 * 2^11) with two accumulators (ns = 2)
    int *bad_items;          // fp16 pair mode: [items per branch] flag of every item (= query of the chunk) an activation of
                             // which left the half range -- the query is re-run through the fp32 kernels (p2s_api.hip: fallback)
};
int p2s_launch_chain(const ChainArgs &args, hipStream_t stream);
// bf16 variant (p2s_chain_bf16.hip): w0b / w1 / w2 / w3 point to bf16 fragment arrays, w1_item_stride counts halfs
int p2s_launch_chain_bf16(const ChainArgs &args, hipStream_t stream);
// fp32 packed B fragments -> bf16 fragments, n_items matrices of K x N
// range_flag (fp16 pair, may be null): raised when a weight does not fit the half range
int p2s_launch_pack_bf16(const float *src, unsigned short *dst, int K, int N, long long src_stride, long long dst_stride,
                         int n_items, int piece, int f16, hipStream_t stream, int *range_flag = nullptr);

// W1' = (BN-folded conv1) . trans2, written in packed B-fragment order.  grid.y = encoder
struct FoldArgs {
    const float *T[2];       // [n_items][64*64]   (I + fc3 output), row-major T[i][j]
    const float *m1t[2];     // packed conv1 weights
    float *out[2];           // [n_items][4096] packed (fp32 fragments; unused when outh is set)
    int n_items;
    // 16-bit encoder modes: W1' goes straight out as the chain kernel's 16-bit B fragments ([N/32][K/16][64 lanes][8]),
    // split into its pieces here (what p2s_pack_bf16_kernel did in 2 * ns extra launches per chunk): piece q of encoder e
    // item i at outh[e] + q * h_piece_stride + i * 4096 halfs.  ns pieces; f16: fp16 pair (h0, (x - h0) * 2^11)
    unsigned short *outh[2];
    long long h_piece_stride;
    int ns, f16;
    int *bad_items;          // fp16 pair: [n_items] flag of every item whose W1' does not fit the half range (-> fp32 fallback)
};
int p2s_launch_fold(const FoldArgs &args, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// Dense FC layers, M = batch of queries (p2s_gemm.hip)
// C[z][M][N] = act(A[z][M][K] . W[z] + bias[z]);  W packed [N/32][K/8][64][4]
// ---------------------------------------------------------------------------------------------
struct GemmArgs {
    const float *A; long long lda; long long a_z;
    const float *A2;              // optional: the activation is max(A, A2) element-wise (NaN-propagating) -- two partial max-pools
    long long a2_z;               // stride of A2 per z (may differ from a_z: with a_z = -a2_z both z see the same pair)
    int a2_add;                   // 1: the activation is A + A2 (two partial SUM-pools, sym_op='sum') instead of max(A, A2)
    // fp16 pair variant (p2s_gemm_f16_kernel; the encoder-side head layers of cfg.encoder_bf16 = 4): Wh[z] != NULL = the
    // weights as 16-bit B fragments [N/32][K/16][64 lanes][8], piece 1 wh_piece halfs behind piece 0; the activation is
    // split into its fp16 pair when it is staged.  bad_rows [M]: flag of every row with an activation beyond the half range
    const unsigned short *Wh[2];
    long long wh_piece;
    int *bad_rows;
    const float *W[2];
    const float *bias[2];
    float *C; long long ldc; long long c_z;
    int M, N, K, Z;
    int relu;
};
int p2s_launch_gemm(const GemmArgs &args, hipStream_t stream);

// fc4 (128 -> 2) + tanh^2 * r * sign + NaN->1
int p2s_launch_decoder_tail(const float *h3, const float *w4, const float *b4, const float *radius,
                            float *logits_out, float *sdf_out, int B, int K, int output_dim, hipStream_t stream);
// QSTN tail: fc3 (256 -> 4) + identity quaternion (in bias) -> rotation matrix [B][9]
int p2s_launch_qstn_tail(const float *h2, const float *w3, const float *b3, float *rot_out, int B, int K,
                         hipStream_t stream);
