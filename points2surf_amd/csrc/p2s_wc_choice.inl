// Weighted sub-sample (included by p2s_wchoice.hip inside its anonymous namespace): the exact cdf look-up (wc_gap, wc_gt,
// wc_finish), the LDS layout of one query and the COMPLETE algorithm of ``choice(n, size, replace=False, p)`` for one query
// (wc_full_query) -- what the ids kernel runs for every query and the chain kernel for the rare undecided candidate.
// ---------------------------------------------------------------------------------------------------------------
// choice: one workgroup, queries in order
// ---------------------------------------------------------------------------------------------------------------
struct WcLoc {
    int bin;
    double s, sprev;          // S_bin, S_{bin-1}
};

// searchsorted(cdf', x, 'right') = smallest i with fl((S_i - C(i)) / Stot_cur) > x, where the m_found ids sid[]
// (ascending) carry no mass any more: V[k] = S' at sid[k], C[k] = found mass up to and including sid[k].
// Step 1 (LDS only): the gap between two found ids that holds the answer, and the guide bucket to fetch.
struct WcGap {
    int lo, hi, bucket;
    double Ck;
};
__device__ __forceinline__ WcGap wc_gap(int n, int K, double Stot, double Stot_cur, double x, int m_found, const int *sid,
                                        const double *sV, const double *sC) {
    WcGap g;
    g.lo = 0;
    g.hi = n;
    g.Ck = 0.0;
    if (m_found) {
        int a = 0, b = m_found;                       // largest k in [0, m] with k == 0 or V[k-1]/Stot_cur <= x
        while (a < b) {
            const int mid = (a + b + 1) >> 1;
            if (sV[mid - 1] / Stot_cur <= x) a = mid;
            else b = mid - 1;
        }
        if (a) {
            g.lo = sid[a - 1] + 1;
            g.Ck = sC[a - 1];
        }
        if (a < m_found) g.hi = sid[a];
    }
    const double t = x * Stot_cur + g.Ck;
    int b = (int)((t / Stot) * (double)K);
    g.bucket = b < 0 ? 0 : (b > K - 1 ? K - 1 : b);
    return g;
}
// fl(S / St) > x without the division in all but razor-thin cases: S/St >= x(1+2^-52) rounds to at least the double
// above x, S/St < x rounds to at most x; t = fl(x*St) is within 2^-53 of x*St, so 1e-15 of slack decides both.
__device__ __forceinline__ bool wc_gt(double S, double St, double x) {
    const double t = x * St;
    if (S > t * (1.0 + 1e-15)) return true;
    if (S < t * (1.0 - 1e-15)) return false;
    return (S / St) > x;
}
// Step 2: exact answer from a starting index near it (S values fetched here; walks are short and rare)
__device__ __forceinline__ WcLoc wc_finish(const double *__restrict__ Sq, int n, int i, int g_lo, int g_hi, double Ck,
                                           double Stot_cur, double x) {
    i = i < g_lo ? g_lo : (i > g_hi - 1 ? g_hi - 1 : i);
    // ONE round trip decides the answers i .. i + 4 (a guide bucket holds one or two cdf boundaries on average; the six
    // values sit in one or two 64-byte sectors); anything else walks, one dependent read per step
    double s_m1 = (i > 0) ? Sq[i - 1] : 0.0;
    double s_0 = Sq[i];
    double sp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) sp[k] = Sq[i + 1 + k < n ? i + 1 + k : n - 1];
#define WC_PRED(sv) wc_gt((sv)-Ck, Stot_cur, x)
    if (WC_PRED(s_0)) {
        while (i > g_lo && WC_PRED(s_m1)) {
            --i;
            s_0 = s_m1;
            s_m1 = (i > 0) ? Sq[i - 1] : 0.0;
        }
        return {i, s_0, s_m1};
    }
    double prev = s_0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i + 1 + k >= n) return {n - 1, prev, prev};       // unreachable for valid tables
        if (WC_PRED(sp[k])) return {i + 1 + k, sp[k], prev};
        prev = sp[k];
    }
    i += 4;
    for (;;) {
        ++i;
        if (i >= n) return {n - 1, prev, prev};       // unreachable for valid tables; keeps the loop finite
        const double sv = Sq[i];
        if (WC_PRED(sv)) return {i, sv, prev};
        prev = sv;
    }
#undef WC_PRED
}

// ---------------------------------------------------------------------------------------------------------------
// LDS layout shared by the kernels below
// ---------------------------------------------------------------------------------------------------------------
constexpr int WC_HASH = 2048;        // open-addressing table: bin -> first draw index of the round
struct WcLds {
    double *fS, *sV, *sC;            // S at the found ids (numpy order) / V, C of the found ids sorted by id
    float *fP;                       // probability of the found ids (float32 values, exact)
    int *fid, *sid;                  // found ids in numpy's order / ascending
    uint32_t *hash;                  // (bin << 10 | draw) packed, 0xffffffff = empty
    uint32_t *bitmap;                // one bit per cloud point: found so far
    uint16_t *wpre;                  // set bits before each bitmap word
};
__host__ __device__ inline size_t wc_lds_bytes(int n) {
    const size_t BW = (size_t)((n + 31) >> 5);
    return (size_t)WC_MAX_SEL * (3 * 8 + 4 + 2 * 4) + WC_HASH * 4 + BW * 4 + ((BW * 2 + 15) & ~(size_t)15);
}
__device__ __forceinline__ WcLds wc_carve(unsigned char *base, int n) {
    const int BW = (n + 31) >> 5;
    WcLds l;
    l.fS = (double *)base;
    l.sV = l.fS + WC_MAX_SEL;
    l.sC = l.sV + WC_MAX_SEL;
    l.fP = (float *)(l.sC + WC_MAX_SEL);
    l.fid = (int *)(l.fP + WC_MAX_SEL);
    l.sid = l.fid + WC_MAX_SEL;
    l.hash = (uint32_t *)(l.sid + WC_MAX_SEL);
    l.bitmap = l.hash + WC_HASH;
    l.wpre = (uint16_t *)(l.bitmap + BW);
    return l;
}

__device__ __forceinline__ double wc_double(uint32_t w0, uint32_t w1) {      // numpy legacy random_sample
    return ((double)(w0 >> 5) * 67108864.0 + (double)(w1 >> 6)) / 9007199254740992.0;
}

// ---------------------------------------------------------------------------------------------------------------
// One query, start to end, by one workgroup: the reference algorithm (all rounds).  The bitmap must be all zero on
// entry and is all zero again on return.  Returns the number of random words consumed (uniform), or -1 if the words
// ran out / no progress (error code stored by the caller).  With WRITE the ids go to ids_out[0..nsel).
// ---------------------------------------------------------------------------------------------------------------
struct WcQuery {
    const double *Sq;
    const WcRec *Rq;
    double Stot;
    const uint32_t *words;       // word o of this query's first draw at words[0]
    long long words_left;        // words available from there
    int n, K, nsel;
};

template <bool WRITE>
__device__ __forceinline__ long long wc_full_query(const WcQuery &qa, const WcLds &l, int *wsum, double *wsumd, int32_t *ids_out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int BW = (qa.n + 31) >> 5;
    const double Stot = qa.Stot;
    double Stot_cur = Stot;
    int n_uniq = 0, m_found = 0, rounds = 0;
    long long o = 0;
    while (n_uniq < qa.nsel) {
        const int m = qa.nsel - n_uniq;
        const int per = (m + 255) >> 8;
        if (o + 2LL * m > qa.words_left || ++rounds > 64) return -1;     // uniform
        for (int i = tid; i < WC_HASH; i += 256) l.hash[i] = 0xffffffffu;
        // locate the bins of rand(m): A the doubles (two words each), B gap + guide record, C finish
        int bins[4];
        double sb[4], sp[4];
        unsigned valid = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bins[j] = 0;
            sb[j] = sp[j] = 0.0;
        }
        if (m_found == 0) {
            // first round (nothing found yet, every lane has up to 4 draws): straight-line code, lanes past the last
            // draw repeat it, so that all loads of a phase are in flight together
            uint2 wpair[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = tid * per + j;
                const bool ok = (j < per) & (d < m);
                valid |= (unsigned)ok << j;
                wpair[j] = *(const uint2 *)(qa.words + o + 2LL * (ok ? d : m - 1));      // o is even: 8-byte aligned
            }
            double xs[4];
            int st[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xs[j] = wc_double(wpair[j].x, wpair[j].y);
                int bk = (int)(xs[j] * (double)qa.K);                 // the bucket of x itself
                bk = bk > qa.K - 1 ? qa.K - 1 : bk;
                st[j] = qa.Rq[bk];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if ((valid >> j) & 1u) {
                    const WcLoc L = wc_finish(qa.Sq, qa.n, st[j], 0, qa.n, 0.0, Stot, xs[j]);
                    bins[j] = L.bin;
                    sb[j] = L.s;
                    sp[j] = L.sprev;
                }
            }
        } else {
            // redraw rounds: few draws (usually one per lane, < 64 lanes)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = tid * per + j;
                if (j < per && d < m) {
                    valid |= 1u << j;
                    const uint2 wp = *(const uint2 *)(qa.words + o + 2LL * d);
                    const double x = wc_double(wp.x, wp.y);
                    const WcGap g = wc_gap(qa.n, qa.K, Stot, Stot_cur, x, m_found, l.sid, l.sV, l.sC);
                    const WcLoc L = wc_finish(qa.Sq, qa.n, qa.Rq[g.bucket], g.lo, g.hi, g.Ck, Stot_cur, x);
                    bins[j] = L.bin;
                    sb[j] = L.s;
                    sp[j] = L.sprev;
                }
            }
        }
        __syncthreads();                                  // hash cleared
        // a bin drawn more than once in this round keeps its FIRST draw (np.unique(return_index) + sort):
        // hash bin -> smallest draw index
        unsigned slot[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            slot[j] = 0;
            if ((valid >> j) & 1u) {
                const uint32_t bin = (uint32_t)bins[j];
                const uint32_t packed = (bin << 10) | (uint32_t)(tid * per + j);
                atomicOr(&l.bitmap[bin >> 5], 1u << (bin & 31));
                uint32_t h = (bin * 2654435761u) >> 21;
                for (;;) {
                    uint32_t cur = l.hash[h];
                    if (cur == 0xffffffffu) {
                        const uint32_t old = atomicCAS(&l.hash[h], 0xffffffffu, packed);
                        if (old == 0xffffffffu) break;
                        cur = old;
                    }
                    if ((cur >> 10) == bin) {
                        atomicMin(&l.hash[h], packed);
                        break;
                    }
                    h = (h + 1) & (WC_HASH - 1);
                }
                slot[j] = h;
            }
        }
        __syncthreads();
        unsigned keep = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (((valid >> j) & 1u) && (l.hash[slot[j]] & 1023u) == (uint32_t)(tid * per + j)) keep |= 1u << j;
        // ordered compaction of the kept draws behind the ones found so far
        const int cnt = __popc(keep);
        int excl = 0, wtot = 0;
#pragma unroll
        for (int bit = 0; bit < 3; ++bit) {
            const unsigned long long mk = __ballot((cnt >> bit) & 1);
            excl += (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u)) << bit;
            wtot += __popcll(mk) << bit;
        }
        if (lane == 0) wsum[wave] = wtot;
        __syncthreads();
        int base = n_uniq, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) base += wsum[w];
            total += wsum[w];
        }
        int r = base + excl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if ((keep >> j) & 1u) {
                l.fid[r] = bins[j];
                l.fS[r] = sb[j];
                l.fP[r] = (float)(sb[j] - sp[j]);           // exact: the float32 probability itself
                ++r;
            }
        }
        n_uniq += total;
        o += 2LL * m;
        __syncthreads();
        if (n_uniq < qa.nsel) {
            // found ids in ascending order through the bitmap: rank = set bits below
            const int wper = (BW + 255) >> 8, w0 = tid * wper;
            int local = 0;
            for (int i = 0; i < wper; ++i)
                if (w0 + i < BW) local += __popc(l.bitmap[w0 + i]);
            int v = local;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int u = __shfl_up(v, off);
                if (lane >= off) v += u;
            }
            if (lane == 63) wsum[wave] = v;
            __syncthreads();
            int run = v - local;
#pragma unroll
            for (int w = 0; w < 4; ++w)
                if (w < wave) run += wsum[w];
            for (int i = 0; i < wper; ++i) {
                if (w0 + i < BW) {
                    l.wpre[w0 + i] = (uint16_t)run;
                    run += __popc(l.bitmap[w0 + i]);
                }
            }
            __syncthreads();
            for (int e = tid; e < n_uniq; e += 256) {
                const int f = l.fid[e];
                const int rank = l.wpre[f >> 5] + __popc(l.bitmap[f >> 5] & ((1u << (f & 31)) - 1u));
                l.sid[rank] = f;
                l.sV[rank] = l.fS[e];
                l.sC[rank] = (double)l.fP[e];
            }
            __syncthreads();
            // C = inclusive scan of the found masses (exact), V = S - C
            const int e0 = 4 * tid;
            double c[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = (e0 + j < n_uniq) ? l.sC[e0 + j] : 0.0;
            const double l1 = c[0], l2 = l1 + c[1], l3 = l2 + c[2], l4 = l3 + c[3];
            double vv = l4;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const double u = __shfl_up(vv, off);
                if (lane >= off) vv += u;
            }
            if (lane == 63) wsumd[wave] = vv;
            __syncthreads();
            double bs = vv - l4;
#pragma unroll
            for (int w = 0; w < 4; ++w)
                if (w < wave) bs += wsumd[w];
            const double cs[4] = {bs + l1, bs + l2, bs + l3, bs + l4};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (e0 + j < n_uniq) {
                    l.sC[e0 + j] = cs[j];
                    l.sV[e0 + j] -= cs[j];
                }
            }
            __syncthreads();
            Stot_cur = Stot - l.sC[n_uniq - 1];
            m_found = n_uniq;
        }
    }
    for (int e = tid; e < qa.nsel; e += 256) {
        const int f = l.fid[e];
        if (WRITE) ids_out[e] = f;
        l.bitmap[f >> 5] = 0;
    }
    __syncthreads();
    return o;
}
