// a6 (included by p2s_cloud.hip inside its anonymous namespace): the serial MT19937 kernels (randint for small requests, the
// legacy shuffle of clouds smaller than the sub-sample), gather, patch from given ids.
// ---------------------------------------------------------------------------------------------
// a6: numpy legacy RandomState.randint(0, N, size) on the device: MT19937 + masked rejection +
// ordered compaction.  One workgroup walks the stream block by block (the recurrence is serial
// across 624-word blocks; inside a block it has three internally parallel phases).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mt_mix(uint32_t u, uint32_t v) {
    const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// Out-of-place twist of one 624-word block by ONE wave.  With lane j-mapping j = 64*it + lane the three
// dependent phases of the recurrence chain through the lane's own registers
//   new[j] -> new[227+j] -> new[454+j]
// so every LDS read is from the old block (independent, issued back to back): no dependent LDS round trip.
__device__ __forceinline__ void mt_twist_wave(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int lane) {
    uint32_t v1[4], v2[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int j = 64 * it + lane;
        v1[it] = 0;
        if (j < 227) {
            v1[it] = src[j + 397] ^ mt_mix(src[j], src[j + 1]);
            dst[j] = v1[it];
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int j = 64 * it + lane;
        v2[it] = 0;
        if (j < 227) {
            v2[it] = v1[it] ^ mt_mix(src[227 + j], src[228 + j]);
            dst[227 + j] = v2[it];
        }
    }
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int j = 64 * it + lane;
        if (j < 169) dst[454 + j] = v2[it] ^ mt_mix(src[454 + j], src[455 + j]);
    }
    // new[623] = new[396] ^ mix(old[623], new[0]);  new[396] = v2 of j = 169 (it 2, lane 41), new[0] = v1 of j = 0
    const uint32_t n396 = __builtin_amdgcn_readlane(v2[2], 41);
    const uint32_t n0 = __builtin_amdgcn_readlane(v1[0], 0);
    if (lane == 0) dst[623] = n396 ^ mt_mix(src[623], n0);
}

// Two-wave pipeline: wave 0 twists block b+1 (out of place) while wave 1 tempers / mask-rejects /
// compacts block b into the output.  One workgroup barrier per 624-word block.  The waves run at raised
// priority: the kernel shares its CU with MFMA-saturated encoder waves and is pure latency.
__global__ __launch_bounds__(128) void p2s_mt_randint_kernel(uint32_t *__restrict__ state, uint32_t rng,
                                                             uint32_t mask, long long target,
                                                             int32_t *__restrict__ out) {
    __shared__ uint32_t st[2][624];
    __shared__ uint32_t stage[640];
    __shared__ int s_done[2];
    __shared__ int s_pos;
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 624; i += 128) st[0][i] = state[i];
    int pos = (int)state[624];
    if (tid == 0) {
        s_done[0] = 0;
        s_done[1] = 0;
        s_pos = 624;
    }
    __syncthreads();
    int cur = 0, start = pos;
    if (pos >= 624) {                       // numpy: "needs twist before the first draw"
        if (wave == 0) mt_twist_wave(st[0], st[1], lane);
        __syncthreads();
        cur = 1;
        start = 0;
    }
    long long produced = 0;                 // meaningful in the consumer wave only
    for (int iter = 0;; ++iter) {
        if (wave == 0) {
            mt_twist_wave(st[cur], st[cur ^ 1], lane);
        } else {
            // lane l owns words 10l .. 10l+9 (contiguous -> ordered compaction by an exclusive lane scan).
            // Branch-free: all 10 LDS reads are issued back to back; accepted words are compacted through
            // an LDS staging buffer and leave as coalesced 256-byte stores.
            const uint32_t *src = st[cur];
            uint32_t w[10];
            unsigned okmask = 0;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int idx = 10 * lane + j;
                w[j] = src[idx < 624 ? idx : 623];
            }
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int idx = 10 * lane + j;
                w[j] = mt_temper(w[j]) & mask;
                const unsigned ok = (idx < 624) & (idx >= start) & (w[j] <= rng);
                okmask |= ok << j;
            }
            const int c = __popc(okmask);
            // exclusive prefix of c (<= 10, 4 bits) over the lanes without any LDS traffic: one ballot +
            // mbcnt per bit plane (a shuffle scan would be six dependent ds_bpermute round trips)
            int excl = 0, total = 0;
#pragma unroll
            for (int bit = 0; bit < 4; ++bit) {
                const unsigned long long m = __ballot((c >> bit) & 1);
                excl += (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)) << bit;
                total += __popcll(m) << bit;
            }
            const long long need = target - produced;          // > 0
            int r = excl;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                if (okmask & (1u << j)) stage[r] = w[j];
                r += (okmask >> j) & 1u;
            }
            const int lim = (total < need) ? total : (int)need;
            {
                uint32_t sv[10];
#pragma unroll
                for (int it = 0; it < 10; ++it) sv[it] = stage[64 * it + lane];      // batched LDS reads
#pragma unroll
                for (int it = 0; it < 10; ++it)
                    if (out && 64 * it + lane < lim) out[produced + 64 * it + lane] = (int32_t)sv[it];
            }
            if (total >= need) {
                // the stream resumes after the word holding the need-th accepted value
                if (excl < need && need <= excl + c) {
                    int left = (int)need - excl;
                    int pos_end = 0;
#pragma unroll
                    for (int j = 0; j < 10; ++j) {
                        if ((okmask >> j) & 1u) {
                            if (--left == 0) pos_end = 10 * lane + j + 1;
                        }
                    }
                    s_pos = pos_end;
                }
                if (lane == 0) s_done[iter & 1] = 1;
                produced = target;
            } else {
                produced += total;
            }
        }
        // LDS-only synchronisation: __syncthreads() would add s_waitcnt vmcnt(0) and stall every block on
        // the completion of its (fire-and-forget) id stores
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s_done[iter & 1]) break;        // parity-indexed: the consumer may already be one block ahead
        cur ^= 1;
        start = 0;
    }
    // the block in which the target was reached stays the current block (the twister's look-ahead went to
    // the other buffer)
    for (int i = tid; i < 624; i += 128) state[i] = st[cur][i];
    if (tid == 0) state[624] = (uint32_t)s_pos;
}

__global__ void p2s_gather_kernel(const float *__restrict__ pts, const int32_t *__restrict__ ids, long long n,
                                  int n_points, float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int id = ids[i];
    if (id < 0) {            // zero padding of the N < sub_sample_size branch (reference source/base/utils.py:225-226)
        out[3 * i + 0] = out[3 * i + 1] = out[3 * i + 2] = 0.0f;
        return;
    }
    id = min(id, n_points - 1);
    out[3 * i + 0] = pts[3 * id + 0];
    out[3 * i + 1] = pts[3 * id + 1];
    out[3 * i + 2] = pts[3 * id + 2];
}

// ---------------------------------------------------------------------------------------------
// a6, clouds with FEWER points than the sub-sample size (reference source/base/utils.py:221-226):
//     pts_shuffled = pts_ms[:, :3]; rng.shuffle(pts_shuffled); pad with zeros
// The view is shuffled IN PLACE: shape.pts itself is permuted by every query, under the kd-tree (which holds its own
// float64 copy), so the patch of a later query gathers pts[knn ids] from the permuted array (Appendix A of
// SURVEY.md).  Reproduced literally: `perm` (device, persistent per cloud) maps the row of shape.pts to the original
// point; per query the state before the shuffle goes to perm_before (patch gather), the state after it is the
// sub-sample (+ -1 padding).  numpy legacy shuffle of a 2-D array: for i = n-1 .. 1: j = rk_interval(i) (masked
// rejection on 32-bit words); swap rows i, j.  One wave: lane 0 walks, all lanes twist.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void p2s_shuffle_pad_kernel(uint32_t *__restrict__ state, int *__restrict__ perm, int n,
                                                             long long nq, int n_sel, int *__restrict__ perm_before,
                                                             int *__restrict__ ids_out) {
    __shared__ uint32_t st[2][624];
    __shared__ int sp[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 624; i += 64) st[0][i] = state[i];
    for (int i = lane; i < n; i += 64) sp[i] = perm[i];
    int pos = (int)state[624];
    int cur = 0;
    __syncthreads();
    for (long long q = 0; q < nq; ++q) {
        if (perm_before)
            for (int i = lane; i < n; i += 64) perm_before[q * n + i] = sp[i];
        int i = n - 1;
        while (i >= 1) {                                  // uniform: i and pos are broadcast from lane 0
            if (pos >= 624) {
                mt_twist_wave(st[cur], st[cur ^ 1], lane);
                cur ^= 1;
                pos = 0;
                __syncthreads();
            }
            if (lane == 0) {
                // consume words of the current block until it is exhausted or the shuffle is done
                while (i >= 1 && pos < 624) {
                    uint32_t mask = (uint32_t)i;
                    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
                    const uint32_t v = mt_temper(st[cur][pos++]) & mask;
                    if (v <= (uint32_t)i) {
                        if ((int)v != i) {
                            const int t = sp[v];
                            sp[v] = sp[i];
                            sp[i] = t;
                        }
                        --i;
                    }
                }
            }
            i = __shfl(i, 0);
            pos = __shfl(pos, 0);
            __syncthreads();
        }
        if (ids_out) {
            for (int k = lane; k < n_sel; k += 64) ids_out[q * n_sel + k] = k < n ? sp[k] : -1;
        }
    }
    for (int i = lane; i < 624; i += 64) state[i] = st[cur][i];
    for (int i = lane; i < n; i += 64) perm[i] = sp[i];
    if (lane == 0) state[624] = (uint32_t)pos;
}

// patch from explicit kNN ids, rows looked up through the current permutation of shape.pts (NULL = identity):
// r = max ||q - p||_2 and (p - q) / r exactly as p2s_knn_kernel computes them
__global__ __launch_bounds__(64) void p2s_patch_from_ids_kernel(const float *__restrict__ pts, const int *__restrict__ ids,
                                                                const int *__restrict__ perm_before, int n,
                                                                const float *__restrict__ queries, long long nq, int k,
                                                                float *__restrict__ patch_out, float *__restrict__ radius_out) {
    const int lane = threadIdx.x;
    for (long long qi = blockIdx.x; qi < nq; qi += gridDim.x) {
        const float qxf = queries[3 * qi + 0], qyf = queries[3 * qi + 1], qzf = queries[3 * qi + 2];
        float smax = 0.0f;
        for (int j = lane; j < k; j += 64) {
            int id = ids[qi * k + j];
            if (perm_before) id = perm_before[qi * n + id];
            const float dx = qxf - pts[3 * id + 0], dy = qyf - pts[3 * id + 1], dz = qzf - pts[3 * id + 2];
            smax = fmaxf(smax, (dx * dx + dy * dy) + dz * dz);
        }
        for (int d = 32; d > 0; d >>= 1) smax = fmaxf(smax, __shfl_xor(smax, d));
        const float rad = sqrtf(smax);
        if (radius_out && lane == 0) radius_out[qi] = rad;
        if (patch_out) {
            for (int j = lane; j < k; j += 64) {
                int id = ids[qi * k + j];
                if (perm_before) id = perm_before[qi * n + id];
                float *dst = patch_out + (qi * k + j) * 3;
                dst[0] = (pts[3 * id + 0] - qxf) / rad;
                dst[1] = (pts[3 * id + 1] - qyf) / rad;
                dst[2] = (pts[3 * id + 2] - qzf) / rad;
            }
        }
    }
}
