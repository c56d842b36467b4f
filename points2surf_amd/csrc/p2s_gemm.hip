// Dense per-query layers with M = batch of queries, fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Replaces the reference's nn.Linear + BatchNorm1d + ReLU stacks:
//   STN head   fc1,bn4,relu / fc2,bn5,relu / fc3 (+identity)   source/points_to_surf_model.py:62-68
//   decoder    fc1_{local,global},bn1_*,relu / fc2,bn2,relu / fc3,bn3,relu / fc4   :335-350
// BatchNorm (eval) is folded into W/bias on the host; W is packed in B-fragment order so that
// each wave streams its own 32-column panel from L2 with coalesced 16-byte loads (no LDS for W);
// the activation tile [64 rows x 128 k] is staged in LDS once per k-chunk and shared by 4 waves.
#include "p2s_common.h"
#include <cstdlib>

namespace {

constexpr int GK = 128;    // k chunk staged in LDS
constexpr int GS = 132;    // LDS row stride

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// grid: (ceil(M / (32 RT)), N/128, Z); block 256: wave w -> column tile (blockIdx.y*4 + w), RT row tiles of 32.
// RT = 2 halves the weight stream per output; RT = 1 doubles the number of workgroups for the layers whose grid would
// not fill the chip (a 4096-query chunk gives the 512- and 256-column layers 512 / 256 workgroups of 64 rows for 1024
// workgroup slots: M = 4096 is all the parallelism there is).
template <int RT>
__global__ __launch_bounds__(256) void p2s_gemm_kernel(GemmArgs g) {
    constexpr int GM = 32 * RT;
    __shared__ __attribute__((aligned(16))) float As[GM * GS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int z = blockIdx.z;
    const int m0 = blockIdx.x * GM;
    const int nt = blockIdx.y * 4 + wave;
    const int KG = g.K / 8;
    const float *__restrict__ A = g.A + (long long)z * g.a_z;
    const float *__restrict__ A2 = g.A2 ? g.A2 + (long long)z * g.a2_z : nullptr;
    const float *__restrict__ Wp = g.W[z] + (long long)nt * KG * 256 + lane * 4;
    const float *__restrict__ bias = g.bias[z];
    float *__restrict__ C = g.C + (long long)z * g.c_z;

    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

    const float *a0p = As + (lane & 31) * GS + 4 * (lane >> 5);
    const float *a1p = a0p + 32 * GS;

    // the activation chunk [GM rows][128 k] goes global -> registers -> LDS; r04: the NEXT chunk's loads are issued before
    // the MFMAs of the current one and land in registers while they run (the kernel used to wait out the load latency
    // behind every barrier: 55 % of the fp32 MFMA peak).  8 rows per pass, 32 lanes x 16 B per row (coalesced).
    f32x4 stage[GM / 8];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int i = 0; i < GM / 8; ++i) {
            const int r = (tid >> 5) + 8 * i;
            int m = m0 + r;
            if (m >= g.M) m = g.M - 1;
            f32x4 v = *reinterpret_cast<const f32x4 *>(A + (long long)m * g.lda + kc + 4 * (tid & 31));
            if (A2) {                 // max-pool over two point sets = max of their pools; NaN wins like torch's max
                const f32x4 u = *reinterpret_cast<const f32x4 *>(A2 + (long long)m * g.lda + kc + 4 * (tid & 31));
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = g.a2_add ? v[t] + u[t] : ((u[t] > v[t] || u[t] != u[t]) ? u[t] : v[t]);
            }
            stage[i] = v;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < GM / 8; ++i) *reinterpret_cast<f32x4 *>(As + ((tid >> 5) + 8 * i) * GS + 4 * (tid & 31)) = stage[i];
    };
    fetch(0);
    for (int kc = 0; kc < g.K; kc += GK) {
        __syncthreads();              // every wave is done reading the previous chunk
        commit();
        __syncthreads();
        if (kc + GK < g.K) fetch(kc + GK);
        const float *wk = Wp + (long long)(kc / 8) * 256;
        f32x4 b = *reinterpret_cast<const f32x4 *>(wk);
#pragma unroll
        for (int kg = 0; kg < GK / 8; ++kg) {
            f32x4 nb = b;
            if (kg < GK / 8 - 1) nb = *reinterpret_cast<const f32x4 *>(wk + (kg + 1) * 256);
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(a0p + 8 * kg);
            f32x4 a1 = a0;
            if (RT == 2) a1 = *reinterpret_cast<const f32x4 *>(a1p + 8 * kg);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc0 = mfma32(a0[t], b[t], acc0);
                if (RT == 2) acc1 = mfma32(a1[t], b[t], acc1);
            }
            b = nb;
        }
    }
    const int col = nt * 32 + (lane & 31);
    const float bv = bias[col];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int r = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        float v0 = acc0[reg] + bv, v1 = acc1[reg] + bv;
        // NaN-propagating ReLU (torch.relu(NaN) = NaN; fmaxf would swallow it)
        if (g.relu) { v0 = (v0 < 0.f) ? 0.f : v0; v1 = (v1 < 0.f) ? 0.f : v1; }
        if (m0 + r < g.M) C[(long long)(m0 + r) * g.ldc + col] = v0;
        if (RT == 2 && m0 + 32 + r < g.M) C[(long long)(m0 + 32 + r) * g.ldc + col] = v1;
    }
}

// ---- fp16 pair variant (r04): the STN / QSTN head layers of the fp16-pair encoder mode -------------------------------
// Same decomposition as above; every operand as h0 + h1 * 2^-11 (h0 = fp16(x), h1 = fp16((x - h0) * 2^11)), a product =
// h0 h0' into one fp32 accumulator and (h1 h0' + h0 h1') into a second one that enters with 2^-11 in the epilogue: three
// v_mfma_f32_32x32x16_f16 per 16 k instead of eight v_mfma_f32_32x32x2_f32 (22 mantissa bits per operand; the dropped
// h1 h1' term is below fp32's own rounding) -- the arithmetic of p2s_chain_bf16_kernel<2, true>.  The weights are split
// once at model creation (p2s_pack_bf16_kernel), the activations when they are staged in LDS.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int HS = GK + 8;       // halfs per LDS row (272 B: 16-byte rows, bank spread)

__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pack_h2(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}

template <int RT>
__global__ __launch_bounds__(256) void p2s_gemm_f16_kernel(GemmArgs g) {
    constexpr int GM = 32 * RT;
    __shared__ __attribute__((aligned(16))) unsigned short As[2][GM * HS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int z = blockIdx.z;
    const int m0 = blockIdx.x * GM;
    const int nt = blockIdx.y * 4 + wave;
    const int KB = g.K / 16;
    const float *__restrict__ A = g.A + (long long)z * g.a_z;
    const float *__restrict__ A2 = g.A2 ? g.A2 + (long long)z * g.a2_z : nullptr;
    const unsigned short *__restrict__ Wp = g.Wh[z] + ((long long)nt * KB * 64 + lane) * 8;
    const float *__restrict__ bias = g.bias[z];
    float *__restrict__ C = g.C + (long long)z * g.c_z;

    f32x16 acc[RT][2];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][t][i] = 0.f;
    const int aoff = (lane & 31) * HS + 8 * (lane >> 5);

    // A[m0:m0+GM][kc:kc+128]: global -> registers (the next chunk's loads are in flight during the MFMAs) -> LDS as its fp16 pair
    f32x4 stage[GM / 8];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int i = 0; i < GM / 8; ++i) {
            const int r = (tid >> 5) + 8 * i;
            int m = m0 + r;
            if (m >= g.M) m = g.M - 1;
            f32x4 v = *reinterpret_cast<const f32x4 *>(A + (long long)m * g.lda + kc + 4 * (tid & 31));
            if (A2) {
                const f32x4 u = *reinterpret_cast<const f32x4 *>(A2 + (long long)m * g.lda + kc + 4 * (tid & 31));
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = g.a2_add ? v[t] + u[t] : ((u[t] > v[t] || u[t] != u[t]) ? u[t] : v[t]);
            }
            stage[i] = v;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < GM / 8; ++i) {
            const int r = (tid >> 5) + 8 * i;
            const f32x4 v = stage[i];
            // a row with an activation beyond the half range: its query goes through the fp32 kernels again (fallback)
            if (g.bad_rows && !(fabsf(v[0]) <= 6.0e4f && fabsf(v[1]) <= 6.0e4f && fabsf(v[2]) <= 6.0e4f && fabsf(v[3]) <= 6.0e4f))
                g.bad_rows[min(m0 + r, g.M - 1)] = 1;
            uint2 h0, h1;
            h0.x = pack_h2(v[0], v[1]);
            h0.y = pack_h2(v[2], v[3]);
            const f16x2 a = __builtin_bit_cast(f16x2, h0.x), b = __builtin_bit_cast(f16x2, h0.y);
            h1.x = pack_h2((v[0] - (float)a[0]) * 2048.0f, (v[1] - (float)a[1]) * 2048.0f);
            h1.y = pack_h2((v[2] - (float)b[0]) * 2048.0f, (v[3] - (float)b[1]) * 2048.0f);
            *reinterpret_cast<uint2 *>(&As[0][r * HS + 4 * (tid & 31)]) = h0;
            *reinterpret_cast<uint2 *>(&As[1][r * HS + 4 * (tid & 31)]) = h1;
        }
    };
    fetch(0);
    for (int kc = 0; kc < g.K; kc += GK) {
        __syncthreads();
        commit();
        __syncthreads();
        if (kc + GK < g.K) fetch(kc + GK);
        const unsigned short *wk = Wp + (long long)(kc / 16) * 512;
        u32x4 b0 = *reinterpret_cast<const u32x4 *>(wk), b1 = *reinterpret_cast<const u32x4 *>(wk + g.wh_piece);
#pragma unroll
        for (int kb = 0; kb < GK / 16; ++kb) {
            u32x4 n0 = b0, n1 = b1;
            if (kb < GK / 16 - 1) {
                n0 = *reinterpret_cast<const u32x4 *>(wk + (kb + 1) * 512);
                n1 = *reinterpret_cast<const u32x4 *>(wk + (kb + 1) * 512 + g.wh_piece);
            }
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const u32x4 a0 = *reinterpret_cast<const u32x4 *>(&As[0][aoff + 32 * r * HS + 16 * kb]);
                const u32x4 a1 = *reinterpret_cast<const u32x4 *>(&As[1][aoff + 32 * r * HS + 16 * kb]);
                acc[r][1] = mfma16(a1, b0, acc[r][1]);          // small terms first
                acc[r][1] = mfma16(a0, b1, acc[r][1]);
                acc[r][0] = mfma16(a0, b0, acc[r][0]);
            }
            b0 = n0;
            b1 = n1;
        }
    }
    const int col = nt * 32 + (lane & 31);
    const float bv = bias[col];
#pragma unroll
    for (int r2 = 0; r2 < RT; ++r2)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int r = 32 * r2 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            float v = (acc[r2][0][reg] + acc[r2][1][reg] * (1.0f / 2048.0f)) + bv;
            if (g.relu) v = (v < 0.f) ? 0.f : v;                // NaN-propagating ReLU
            if (m0 + r < g.M) C[(long long)(m0 + r) * g.ldc + col] = v;
        }
}

// fc4 (K -> 2, no BN) + post-processing.  One thread per query.
//   reference: source/points_to_surf_model.py:350, source/sdf_nn.py:11-21,
//              source/points_to_surf_eval.py:184-196,263-273,205-207
__global__ __launch_bounds__(256) void p2s_decoder_tail_kernel(const float *__restrict__ h3,
                                                               const float *__restrict__ w4,
                                                               const float *__restrict__ b4,
                                                               const float *__restrict__ radius,
                                                               float *__restrict__ logits_out,
                                                               float *__restrict__ sdf_out, int B, int K, int od) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= B) return;
    const float *h = h3 + (long long)q * K;
    if (od == 1) {
        // outputs = ['imp_surf']: one logit, distance = tanh(x)^2 * sign(x) (sdf_nn.py:6-8), * patch radius
        // (points_to_surf_eval.py:176-183); torch.sign(0) = 0, NaN stays NaN and becomes 1.0 (:205-207)
        float l = b4[0];
        for (int k = 0; k < K; k += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(h + k);
#pragma unroll
            for (int t = 0; t < 4; ++t) l = fmaf(v[t], w4[k + t], l);
        }
        if (logits_out) logits_out[q] = l;
        if (sdf_out) {
            const float th = tanhf(l);
            const float sg = (l > 0.0f) ? 1.0f : ((l < 0.0f) ? -1.0f : (l == 0.0f ? 0.0f : l));
            float sdf = ((th * th) * sg) * radius[q];
            if (sdf != sdf) sdf = 1.0f;
            sdf_out[q] = sdf;
        }
        return;
    }
    float l0 = b4[0], l1 = b4[1];
    for (int k = 0; k < K; k += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(h + k);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            l0 = fmaf(v[t], w4[2 * (k + t) + 0], l0);
            l1 = fmaf(v[t], w4[2 * (k + t) + 1], l1);
        }
    }
    if (logits_out) {
        logits_out[2 * q + 0] = l0;
        logits_out[2 * q + 1] = l1;
    }
    if (sdf_out) {
        const float th = tanhf(l0);
        const float mag = (th * th) * radius[q];
        float sdf = (l1 >= 0.0f) ? mag : -mag;
        if (sdf != sdf) sdf = 1.0f;
        sdf_out[q] = sdf;
    }
}

// QSTN tail: fc3 (K -> 4) with the identity quaternion folded into the bias, then
// batch_quat_to_rotmat (reference source/base/utils.py:13-46, same index pattern).
__global__ __launch_bounds__(256) void p2s_qstn_tail_kernel(const float *__restrict__ h2,
                                                            const float *__restrict__ w3,
                                                            const float *__restrict__ b3,
                                                            float *__restrict__ rot_out, int B, int K) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= B) return;
    const float *h = h2 + (long long)q * K;
    float qa = b3[0], qb = b3[1], qc = b3[2], qd = b3[3];
    for (int k = 0; k < K; ++k) {
        const float v = h[k];
        qa = fmaf(v, w3[4 * k + 0], qa);
        qb = fmaf(v, w3[4 * k + 1], qb);
        qc = fmaf(v, w3[4 * k + 2], qc);
        qd = fmaf(v, w3[4 * k + 3], qd);
    }
    const float qq[4] = {qa, qb, qc, qd};
    const float s = 2.0f / (__fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(qa, qa), __fmul_rn(qb, qb)), __fmul_rn(qc, qc)),
                                      __fmul_rn(qd, qd)));
#define H(i, j) __fmul_rn(qq[i], qq[j])
    float *R = rot_out + (long long)q * 9;
    R[0] = 1.0f - __fmul_rn(__fadd_rn(H(2, 2), H(3, 3)), s);
    R[1] = __fmul_rn(__fsub_rn(H(1, 2), H(3, 0)), s);
    R[2] = __fmul_rn(__fadd_rn(H(1, 3), H(2, 0)), s);
    R[3] = __fmul_rn(__fadd_rn(H(1, 2), H(3, 0)), s);
    R[4] = 1.0f - __fmul_rn(__fadd_rn(H(1, 1), H(3, 3)), s);
    R[5] = __fmul_rn(__fsub_rn(H(2, 3), H(1, 0)), s);
    R[6] = __fmul_rn(__fsub_rn(H(1, 3), H(2, 0)), s);
    R[7] = __fmul_rn(__fadd_rn(H(2, 3), H(1, 0)), s);
    R[8] = 1.0f - __fmul_rn(__fadd_rn(H(1, 1), H(2, 2)), s);
#undef H
}

}  // namespace

int p2s_launch_gemm(const GemmArgs &g, hipStream_t stream) {
    if (g.M <= 0) return P2S_OK;
    if (g.N % 128 != 0 || g.K % GK != 0 || g.Z < 1 || g.Z > 2) {
        p2s_set_error("gemm: unsupported shape M=%d N=%d K=%d Z=%d", g.M, g.N, g.K, g.Z);
        return P2S_EINVAL;
    }
    // 64-row workgroups when they fill the 1024 workgroup slots of the chip at least twice, else 32-row ones
    const long long wg64 = (long long)((g.M + 63) / 64) * (g.N / 128) * g.Z;
    const bool rt2 = wg64 >= 2048;
    if (g.Wh[0]) {                   // fp16 pair operands
        if (rt2) hipLaunchKernelGGL(p2s_gemm_f16_kernel<2>, dim3((g.M + 63) / 64, g.N / 128, g.Z), dim3(256), 0, stream, g);
        else hipLaunchKernelGGL(p2s_gemm_f16_kernel<1>, dim3((g.M + 31) / 32, g.N / 128, g.Z), dim3(256), 0, stream, g);
        P2S_LAUNCH_CHECK("p2s_gemm_f16_kernel");
        return P2S_OK;
    }
    if (rt2) {
        dim3 grid((g.M + 63) / 64, g.N / 128, g.Z);
        hipLaunchKernelGGL(p2s_gemm_kernel<2>, grid, dim3(256), 0, stream, g);
    } else {
        dim3 grid((g.M + 31) / 32, g.N / 128, g.Z);
        hipLaunchKernelGGL(p2s_gemm_kernel<1>, grid, dim3(256), 0, stream, g);
    }
    P2S_LAUNCH_CHECK("p2s_gemm_kernel");
    return P2S_OK;
}

int p2s_launch_decoder_tail(const float *h3, const float *w4, const float *b4, const float *radius,
                            float *logits_out, float *sdf_out, int B, int K, int output_dim, hipStream_t stream) {
    if (B <= 0) return P2S_OK;
    hipLaunchKernelGGL(p2s_decoder_tail_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, h3, w4, b4, radius,
                       logits_out, sdf_out, B, K, output_dim);
    P2S_LAUNCH_CHECK("p2s_decoder_tail_kernel");
    return P2S_OK;
}

int p2s_launch_qstn_tail(const float *h2, const float *w3, const float *b3, float *rot_out, int B, int K,
                         hipStream_t stream) {
    if (B <= 0) return P2S_OK;
    hipLaunchKernelGGL(p2s_qstn_tail_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, h2, w3, b3, rot_out, B, K);
    P2S_LAUNCH_CHECK("p2s_qstn_tail_kernel");
    return P2S_OK;
}
