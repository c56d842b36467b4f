// Weighted sub-sample, stage 3 (included by p2s_wchoice.hip inside its anonymous namespace): the ids, all queries in parallel.
// ids: one workgroup per query, every query's word offset known -> the full algorithm in parallel
__global__ __launch_bounds__(256) void wc_ids_kernel(WcArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wc_lds[];
    __shared__ int wsum[16];
    __shared__ double wsumd[16];
    const int tid = threadIdx.x;
    if (a.meta[1] != 0) return;
    const WcLds l = wc_carve(wc_lds, a.n);
    const int BW = (a.n + 31) >> 5;
    for (int i = tid; i < BW; i += 256) l.bitmap[i] = 0;
    __syncthreads();
    const int q = blockIdx.x;
    const long long base = a.fixed ? 0 : a.base[q];
    WcQuery qa;
    qa.Sq = a.S + (size_t)q * a.n;
    qa.Rq = a.R + (size_t)q * a.K;
    qa.Stot = a.stot[q];
    qa.words = a.words + base;
    qa.words_left = a.cap_words - base;
    qa.n = a.n;
    qa.K = a.K;
    qa.nsel = a.nsel;
    const long long used = wc_full_query<true>(qa, l, wsum, wsumd, a.ids_out + (size_t)q * a.nsel);
    if (a.fixed) {
        // rng.seed(42) before every query: the generator ends where the LAST query of the call left it
        if (tid == 0 && used < 0) a.meta[1] = 3;
        if (tid == 0 && q == a.nq - 1 && used >= 0) a.meta[0] = used;
        return;
    }
    // cross-check against the offsets pass: both must agree on where the next query starts
    const long long next = (q + 1 < a.nq) ? a.base[q + 1] : a.meta[0];
    if (tid == 0 && (used < 0 || base + used != next)) a.meta[1] = 4;
}
