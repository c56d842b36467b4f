// Latency probe for the primitives of the serial sub-sample kernel (one workgroup of 256 lanes on an otherwise idle
// GPU): dependent LDS reads, LDS atomics with return (random / same address), workgroup barriers, cold global loads.
//   hipcc --offload-arch=gfx950 -O3 -o tools/lds_probe.bin tools/lds_probe.hip && tools/lds_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void probe(const int *perm, const unsigned *big, long long *out) {
    __shared__ int a[16384];
    __shared__ int cnt;
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384; i += 256) a[i] = perm[i];
    if (tid == 0) cnt = 0;
    __syncthreads();
    long long t0, t1;
    int v = tid;
    // 1. dependent LDS reads (random)
    t0 = clock64();
    for (int k = 0; k < 64; ++k) v = a[v & 16383];
    t1 = clock64();
    if (tid == 0) out[0] = (t1 - t0) / 64;
    // 2. dependent LDS atomicOr with return (random)
    t0 = clock64();
    for (int k = 0; k < 64; ++k) v = atomicOr(&a[(v * 2654435761u >> 18) & 16383], 0) + k;
    t1 = clock64();
    if (tid == 0) out[1] = (t1 - t0) / 64;
    // 3. same-address LDS atomicAdd from 8 lanes per wave
    t0 = clock64();
    for (int k = 0; k < 64; ++k)
        if ((tid & 7) == 0) v += atomicAdd(&cnt, 1);
    t1 = clock64();
    if (tid == 0) out[2] = (t1 - t0) / 64;
    // 4. barrier (LDS-only flavour)
    t0 = clock64();
    for (int k = 0; k < 64; ++k) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    t1 = clock64();
    if (tid == 0) out[3] = (t1 - t0) / 64;
    // 5. __syncthreads
    t0 = clock64();
    for (int k = 0; k < 64; ++k) __syncthreads();
    t1 = clock64();
    if (tid == 0) out[4] = (t1 - t0) / 64;
    // 6. dependent cold global loads (pointer chase over 1 GiB, one lane)
    unsigned idx = 12345u + tid * 7919u;
    t0 = clock64();
    for (int k = 0; k < 32; ++k) idx = big[(idx * 2654435761u) >> 4];
    t1 = clock64();
    if (tid == 0) out[5] = (t1 - t0) / 32;
    // 7. 16 independent cold global loads per lane (all 256 lanes) -> time until all are back
    unsigned acc = 0;
    t0 = clock64();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += big[((idx + k * 40503u + tid) * 2654435761u) >> 4];
    t1 = clock64();
    if (tid == 0) out[6] = (t1 - t0);
    // 8. LDS write + barrier + read round (communication step)
    t0 = clock64();
    for (int k = 0; k < 64; ++k) {
        a[(tid + k) & 16383] = v;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        v += a[(tid * 5 + k) & 16383];
    }
    t1 = clock64();
    if (tid == 0) out[7] = (t1 - t0) / 64;
    out[8 + tid] = v + acc + idx;
}

int main() {
    std::vector<int> perm(16384);
    unsigned s = 1;
    for (auto &p : perm) { s = s * 1664525u + 1013904223u; p = (int)(s >> 8); }
    int *dperm; unsigned *big; long long *out;
    hipMalloc(&dperm, perm.size() * 4);
    hipMemcpy(dperm, perm.data(), perm.size() * 4, hipMemcpyHostToDevice);
    const size_t nbig = (size_t)1 << 28;   // 2^28 words = 1 GiB
    hipMalloc(&big, nbig * 4);
    hipMemset(big, 0x5a, nbig * 4);
    hipMalloc(&out, (8 + 256) * 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, dperm, big, out);
        hipDeviceSynchronize();
    }
    long long h[8];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("shader clocks: lds read %lld | lds atomic rtn %lld | same-addr atomic (8/wave) %lld | lds barrier %lld | "
           "__syncthreads %lld | cold global load %lld | 16 loads x 256 lanes %lld | write+barrier+read %lld\n",
           h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    return 0;
}
