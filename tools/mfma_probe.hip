// Micro-probe for the conv3 inner loop of p2s_chain_kernel (development aid, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// Variants isolate what throttles a v_mfma_f32_32x32x2_f32 stream: LDS operand reads, global (L2-resident)
// operand loads, register blocking, occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// MODE bit0: LDS A reads, bit1: global B loads (2 x dwordx4 per 16 MFMA), bit2: only ONE global load per 16 MFMA
// RT = row tiles per wave (2 or 4): accumulators = RT x 2
template <int MODE, int RT>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ w, float *__restrict__ out, int iters,
                                             int slices) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 128 * 132; i += 256) lds[i] = (float)(i & 7) * 0.01f;
    __syncthreads();
    const float *ap[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) ap[r] = lds + ((lane & 31) + 32 * r) * 132 + 4 * (lane >> 5);
    f32x16 acc[RT][2];
#pragma unroll
    for (int r = 0; r < RT; ++r) { acc[r][0] = zero16(); acc[r][1] = zero16(); }
    f32x4 a[RT], b0, b1;
#pragma unroll
    for (int r = 0; r < RT; ++r) a[r] = *(const f32x4 *)ap[r];
    b0 = a[0]; b1 = a[RT - 1];
    for (int it = 0; it < iters; ++it) {
        // one "pair": 16 k-groups
        const float *wb0 = w + (size_t)((wave * 8 + (it % 4) * 2 + ((it / 4) % slices) * 32) * 16) * 256 + lane * 4;
        const float *wb1 = wb0 + 16 * 256;
#pragma unroll 1
        for (int kg = 0; kg < 16; ++kg) {
            f32x4 na[RT], nb0 = b0, nb1 = b1;
            const int kn = (kg + 1) & 15;
            if (MODE & 2) {
                nb0 = *(const f32x4 *)(wb0 + kn * 256);
                if (!(MODE & 4)) nb1 = *(const f32x4 *)(wb1 + kn * 256);
            }
#pragma unroll
            for (int r = 0; r < RT; ++r) na[r] = (MODE & 1) ? *(const f32x4 *)(ap[r] + 8 * kn) : a[r];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    acc[r][0] = mfma32(a[r][t], b0[t], acc[r][0]);
                    acc[r][1] = mfma32(a[r][t], b1[t], acc[r][1]);
                }
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, RT, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8 * RT, 0);
#pragma unroll
            for (int r = 0; r < RT; ++r) a[r] = na[r];
            b0 = nb0; b1 = nb1;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[r][0][i] + acc[r][1][i];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int RT>
void run(const char *name, const float *w, float *out, int wg_per_cu, int lds_bytes) {
    const int iters = 64;
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)probe<MODE, RT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<MODE, RT>), dim3(grid), dim3(256), lds_bytes, 0, w, out, iters, 4);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * 4 * iters * 16 * 8 * RT;   // per wave: iters*16 groups * 8*RT
    const double tf = mfma * 4096 / (ms * 1e-3) / 1e12;
    printf("%-34s wg/cu=%d RT=%d  %.3f ms  %.1f TF  (%.1f%% of 157.3)\n", name, wg_per_cu, RT, ms, tf, tf / 157.3 * 100);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) printf("  error: %s\n", hipGetErrorString(e));
}

int main() {
    float *w, *out;
    const size_t wn = (size_t)4 * 1024 * 128;   // 4 slices of a 128x1024 weight matrix (2 MB)
    hipMalloc(&w, wn * 4);
    hipMalloc(&out, 256 * 8 * 256 * 4);
    std::vector<float> h(wn, 0.001f);
    hipMemcpy(w, h.data(), wn * 4, hipMemcpyHostToDevice);
    const int L1 = 128 * 132 * 4;          // 67.5 KB -> 2 WG/CU max; pad to control occupancy
    for (int occ = 1; occ <= 2; ++occ) {
        const int lds = (occ == 1) ? 120 * 1024 : L1;
        run<0, 2>("mfma only", w, out, occ, lds);
        run<1, 2>("+ LDS A reads", w, out, occ, lds);
        run<2, 2>("+ global B loads (2/16)", w, out, occ, lds);
        run<3, 2>("+ LDS + global (kernel's mix)", w, out, occ, lds);
        run<7, 2>("+ LDS + ONE global load /16", w, out, occ, lds);
        run<3, 4>("4x2 block: LDS + global (2/32)", w, out, occ, lds);
        run<0, 4>("4x2 block: mfma only", w, out, occ, lds);
    }
    return 0;
}
