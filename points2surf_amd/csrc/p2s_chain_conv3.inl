// conv3 (K = 128, N = 1024) + running pool of ONE point tile -- textually included twice by p2s_chain.hip inside
// p2s_chain_kernel, with P2S_TAIL = 0 (all 64 rows on v_mfma_f32_32x32x2_f32) and P2S_TAIL = 1 (the item's last tile
// when it holds <= 48 points: row tile 1 as 16 rows on the 4-block MFMA, see mfma16b).  An include rather than a
// lambda / function template: by-reference captures of the pooled registers sent them through scratch.
{
            constexpr bool TAIL = P2S_TAIL != 0;
            const float *a0p = bufB + (lane & 31) * SB + 4 * (lane >> 5);
            const float *a1p = bufB + (32 + (lane & (TAIL ? 15 : 31))) * SB + 4 * (lane >> 5);
#define P2S_MFMA16(A0, A1, B0, B1)                                   \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                  \
        c00 = mfma32(A0[t], B0[t], c00);                             \
        c01 = mfma32(A0[t], B1[t], c01);                             \
        c10 = mfma_r1<TAIL>(A1[t], B0[t], c10);                      \
        c11 = mfma_r1<TAIL>(A1[t], B1[t], c11);                      \
    }
// first k-group of a pair: accumulate into literal zero (inline constant C operand, no v_mov init)
#define P2S_MFMA16_FIRST(A0, A1, B0, B1)                             \
    c00 = mfma32(A0[0], B0[0], zero16());                            \
    c01 = mfma32(A0[0], B1[0], zero16());                            \
    c10 = mfma_r1<TAIL>(A1[0], B0[0], zero16());                     \
    c11 = mfma_r1<TAIL>(A1[0], B1[0], zero16());                     \
    _Pragma("unroll") for (int t = 1; t < 4; ++t) {                  \
        c00 = mfma32(A0[t], B0[t], c00);                             \
        c01 = mfma32(A0[t], B1[t], c01);                             \
        c10 = mfma_r1<TAIL>(A1[t], B0[t], c10);                      \
        c11 = mfma_r1<TAIL>(A1[t], B1[t], c11);                      \
    }
// fetch the operands of k-group KG of pair PR into a register set.  B comes through a buffer descriptor
// (SGPR base + scalar offset + one 32-bit lane offset): no 64-bit VALU address arithmetic in the loop.
#define P2S_FETCH(BS0, BS1, AS0, AS1, PR, KG)                                                          \
    BS0 = bufld4(w3rsrc, lane16, w3soff + (((2 * (PR)) * 16 + (KG)) * 1024));                           \
    BS1 = bufld4(w3rsrc, lane16, w3soff + (((2 * (PR) + 1) * 16 + (KG)) * 1024));                       \
    AS0 = lds4(a0p + 8 * (KG));                                                                         \
    AS1 = lds4(a1p + 8 * (KG));
// issue order of one 16-MFMA block: ONE memory instruction per MFMA shadow.  A VMEM/DS instruction costs
// tens of issue cycles; clustered at the block boundary (or sunk to first use, the scheduler's default)
// their issue time exceeds the 64-cycle shadow of one MFMA and the matrix pipe bubbles (measured: 13 %).
#define P2S_SPREAD()                                                                              \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            f32x16 c00, c01, c10, c11;
            aA0 = lds4(a0p);
            aA1 = lds4(a1p);
            if (tile + 1 < ntiles) load_point(tile + 1, nx0, nx1, nx2);   // lands during conv3
#pragma unroll 1
            for (int pr = 0; pr < 4; ++pr) {
                // k-group 0 (set A, C = 0) while set B <- k-group 1
                P2S_FETCH(bB0, bB1, aB0, aB1, pr, 1)
                P2S_MFMA16_FIRST(aA0, aA1, bA0, bA1)
                P2S_SPREAD()
#pragma unroll 1
                for (int kg = 1; kg < 15; kg += 2) {
                    P2S_FETCH(bA0, bA1, aA0, aA1, pr, kg + 1)
                    P2S_MFMA16(aB0, aB1, bB0, bB1)
                    P2S_SPREAD()
                    P2S_FETCH(bB0, bB1, aB0, aB1, pr, kg + 2)
                    P2S_MFMA16(aA0, aA1, bA0, bA1)
                    P2S_SPREAD()
                }
                // k-group 15 (set B) while set A <- k-group 0 of the next pair (the last pair re-fetches its own)
                const int prn = (pr < 3) ? pr + 1 : 3;
                P2S_FETCH(bA0, bA1, aA0, aA1, prn, 0)
                P2S_MFMA16(aB0, aB1, bB0, bB1)
                P2S_SPREAD()
                float m0, m1;
                if constexpr (SUM) {
                    const int nvalid = P - tile * MT;       // rows of this tile that are points of the item
                    m0 = half_sum(tile_colsum(c00, c10, nvalid, lane));
                    m1 = half_sum(tile_colsum(c01, c11, nvalid, lane));
                } else if constexpr (TAIL) {
                    const float t0 = tail_colmax(c10), t1 = tail_colmax(c11);
                    float u0 = tile_colmax1(c00), u1 = tile_colmax1(c01);
                    asm volatile("v_max_f32 %0, %0, %1" : "+v"(u0) : "v"(t0));
                    asm volatile("v_max_f32 %0, %0, %1" : "+v"(u1) : "v"(t1));
                    m0 = half_max(u0);
                    m1 = half_max(u1);
                } else {
                    m0 = half_max(tile_colmax(c00, c10));
                    m1 = half_max(tile_colmax(c01, c11));
                }
#define P2S_POOL(dst, v) dst = SUM ? dst + (v) : fmaxf(dst, (v))
                if (pr == 0) { P2S_POOL(rm0, m0); P2S_POOL(rm1, m1); }
                else if (pr == 1) { P2S_POOL(rm2, m0); P2S_POOL(rm3, m1); }
                else if (pr == 2) { P2S_POOL(rm4, m0); P2S_POOL(rm5, m1); }
                else { P2S_POOL(rm6, m0); P2S_POOL(rm7, m1); }
#undef P2S_POOL
            }
#undef P2S_MFMA16
#undef P2S_MFMA16_FIRST
#undef P2S_FETCH
#undef P2S_SPREAD
}
