/*
 * mcmc_model_f32.hip.h -- the FAST arithmetic mode of the model time-stepper (SURVEY 8d,
 * "FP32 fast mode"): DiseaseSpread::eval (covid.h:325-496) with the renewal convolution
 * -- 99 % of the arithmetic -- in PACKED single precision, everything around it (proposal,
 * capacity test, the feedback n_{i+1}, likelihood sum, prior, accept) in double as before.
 *
 * It is NOT the reference's arithmetic (`using Float = double`, mcmc.h:43) and is never
 * what a parity test or bench.py's `value` runs; it exists because north_star asks for the
 * FP32 vector roofline next to the FP64 one.  Selected per context with
 * mcmc_hip_set_precision(ctx, MCMC_HIP_PRECISION_F32_MODEL); its bar is statistical
 * (tests/test_gpu_fast_mode.py: log-likelihood within 1e-5 relative of the FP64 kernel,
 * posterior summaries of a chain ensemble within Monte-Carlo error).
 *
 * Same gather formulation and the same three phases as eval_model (mcmc_kernels.hip.h), so
 * only what differs is described here:
 *  - the seven accumulator families of a day sit in four float2 registers
 *    (MI,HI) (INC,MILD) (SER,REC) (DEAD,-) and one lag step of one day is four
 *    v_pk_fma_f32 -- the history value broadcast to both halves by op_sel, the table row
 *    read from SGPRs (one s_load_dwordx16 per two lag steps);
 *  - the per-lane histories are float (half the re-read traffic), kept in the same buffers;
 *  - the capacity correction of a source day is two packed multiply-adds per pair, with the
 *    row (-SFS, -RFS, 1-DFS, 0) prepared by the host (in double, rounded once): dead += e and
 *    dead -= e*DFS become one operation;
 *  - no cross-lane row tricks: rows 0..T-1 of the sequential part come from the scalar cache
 *    as well.
 *
 * MFMA = true (the default fast mode, MCMC_HIP_PRECISION_F32_MODEL) moves the whole contraction over the source
 * days BEFORE a block onto the FP32 matrix cores: for a tile of 64 chains and the block's T = MCMC_TM = 9 days,
 *     C[chain][7 t + c] = sum_k hist[k][chain] * table7[(j0 + t - k) * 7 + c]
 * is a (64 x K) x (K x 63) product whose B operand is Toeplitz in (k, t).  v_mfma_f32_32x32x2_f32 takes
 * M = 32 chains (the even or the odd ones of the wave), N = 32 columns and K = 2 source days: lane l supplies
 * A = hist[k0 + l/32][2 (l%32) + tile] (both chain tiles' values in one coalesced 8-byte load) and
 * B = table7[(j0 - k0 - l/32) * 7 + 32 q + l%32] for column tile q -- consecutive floats of the LDS image, which
 * has 7 floats per lag and no pad column, so 9 days x 7 curves fill 63 of the 64 columns.  Four MFMAs do the
 * multiply-adds of two source days; the capacity corrections are the same product with the e-history as A and
 * the table (-SFS -RFS 1-DFS), 3 floats per lag, as B: 27 of 32 columns, two MFMAs per pair.  At the end of the
 * block the sums of a chain go from the MFMA layout to chain-per-lane through an LDS buffer of the wave's own,
 * and the sequential part runs as before.  Summation order differs from the packed form (pairs of absolute
 * source days, then the in-block sources); both are FP32 and held to the same statistical bar.
 */
#pragma once

typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float8_t __attribute__((ext_vector_type(8)));

/* float tables of the fast mode: RFpos [rows][8] = MI HI INC MILD SER REC DEAD 0, RFneg [rows][4] =
 * -SFS -RFS 1-DFS 0; rows >= nLagRows are zero */
enum { P_INF = 0, P_IM, P_SR, P_D, P_N };     /* (MI,HI) (INC,MILD) (SER,REC) (DEAD,0) */

typedef const __attribute__((address_space(4))) float8_t* const_rowf8_ptr;
typedef const __attribute__((address_space(4))) float4_t* const_rowf4_ptr;
__device__ __forceinline__ float8_t smem_rowf8(const float* RFpos, int L) {
    return *(const_rowf8_ptr)(RFpos + (long long)L * 8);
}
__device__ __forceinline__ float4_t smem_rowf4(const float* RFneg, int L) {
    return *(const_rowf4_ptr)(RFneg + (long long)L * 4);
}
__device__ __forceinline__ float ld_rowf(const float* base_u, int row, unsigned lane) {
    const char* b = (const char*)(base_u + (size_t)row * MCMC_WAVE);
    return *(const float*)(b + ((lane & (MCMC_WAVE - 1)) << 2));
}
__device__ __forceinline__ float2_t fma2(float x, float a, float b, float2_t c) {
    return __builtin_elementwise_fma(float2_t{x, x}, float2_t{a, b}, c);
}
/* the four packed multiply-adds of one (day, lag) pair */
template <int T>
__device__ __forceinline__ void pair_pos(float2_t (&acc)[P_N][T], int t, float nk, const float8_t& r) {
    acc[P_INF][t] = fma2(nk, r[0], r[1], acc[P_INF][t]);
    acc[P_IM][t] = fma2(nk, r[2], r[3], acc[P_IM][t]);
    acc[P_SR][t] = fma2(nk, r[4], r[5], acc[P_SR][t]);
    acc[P_D][t] = fma2(nk, r[6], r[7], acc[P_D][t]);
}
/* the capacity correction e_k of the pair's source day (covid.h:418,426-428) */
template <int T>
__device__ __forceinline__ void pair_neg(float2_t (&acc)[P_N][T], int t, float ek, const float4_t& rn) {
    acc[P_SR][t] = fma2(ek, rn[0], rn[1], acc[P_SR][t]);
    acc[P_D][t] = fma2(ek, rn[2], rn[3], acc[P_D][t]);
}

/* Table rows of the lag loops.  Scalar loads return out of order, so every wait for one is a
 * wait for ALL outstanding ones (lgkmcnt(0)): however many steps ahead a row is requested, the
 * lead that counts is the time since the LAST request.  A packed lag step is only ~170 cycles
 * of arithmetic -- less than the scalar cache's latency -- so the rows travel in GROUPS of two
 * lag steps: one s_load_dwordx16 (two 8-float rows, consecutive in memory because lags descend)
 * is requested at the start of a group for the NEXT group and first read two steps later.  The
 * group in use is carried across chunks (`cur`); the rows the last chunk of a block requests
 * below lag T are in bounds and unused. */
typedef float float16u_t __attribute__((ext_vector_type(16))) __attribute__((aligned(32)));
typedef float float8u_t __attribute__((ext_vector_type(8))) __attribute__((aligned(16)));
typedef const __attribute__((address_space(4))) float16u_t* const_rowf16_ptr;
typedef const __attribute__((address_space(4))) float8u_t* const_rowf8u_ptr;
/* rows L and L+1 of the 8-float table: [0..7] = row L, [8..15] = row L+1 */
__device__ __forceinline__ float16u_t smem_rows2(const float* RFpos, int L) {
    return *(const_rowf16_ptr)(RFpos + (long long)L * 8);
}
/* rows L and L+1 of the 4-float correction table */
__device__ __forceinline__ float8u_t smem_rows2n(const float* RFneg, int L) {
    return *(const_rowf8u_ptr)(RFneg + (long long)L * 4);
}
__device__ __forceinline__ float8_t hi8(const float16u_t& v) { return float8_t{v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]}; }
__device__ __forceinline__ float8_t lo8(const float16u_t& v) { return float8_t{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]}; }
__device__ __forceinline__ float4_t hi4(const float8u_t& v) { return float4_t{v[4], v[5], v[6], v[7]}; }
__device__ __forceinline__ float4_t lo4(const float8u_t& v) { return float4_t{v[0], v[1], v[2], v[3]}; }

/* one chunk of T lags (Lc, Lc-1, ..., Lc-T+1) on the T days of the block; history windows as in
 * lag_chunk.  cur = rows (Lc-1, Lc) on entry, rows (Lc-T-1, Lc-T) on exit. */
template <int T>
__device__ __forceinline__ void lag_chunk_f(float2_t (&acc)[P_N][T], float (&W)[T], const float (&N)[T], float16u_t& cur,
                                            const float* RFpos, int Lc, const float* hn_next_u, unsigned lane) {
    static_assert(T % 2 == 0, "rows travel in pairs");
    const float* rowBase = RFpos + (long long)(Lc - T - 1) * 8;            /* one address per chunk */
#pragma unroll
    for (int g = 0; g < T / 2; ++g) {
        const float16u_t nxt = smem_rows2(rowBase, T - 2 - 2 * g);         /* rows Lc-2g-3, Lc-2g-2 */
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int u = 2 * g + h;
            const float8_t r = h == 0 ? hi8(cur) : lo8(cur);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int q = t + u;
                const float nk = (q < T) ? W[q < T ? q : 0] : N[q < T ? 0 : q - T];
                pair_pos<T>(acc, t, nk, r);
            }
            W[u] = ld_rowf(hn_next_u, u, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
        cur = nxt;
    }
}

template <int T>
__device__ __forceinline__ void lag_chunk_fe(float2_t (&acc)[P_N][T], float (&W)[T], const float (&N)[T], float (&EW)[T],
                                             const float (&EN)[T], float16u_t& cur, float8u_t& curn, const float* RFpos,
                                             const float* RFneg, int Lc, const float* hn_next_u, const float* he_next_u,
                                             unsigned lane) {
    const float* rowBase = RFpos + (long long)(Lc - T - 1) * 8;
    const float* rowBaseN = RFneg + (long long)(Lc - T - 1) * 4;
#pragma unroll
    for (int g = 0; g < T / 2; ++g) {
        const float16u_t nxt = smem_rows2(rowBase, T - 2 - 2 * g);
        const float8u_t nxtn = smem_rows2n(rowBaseN, T - 2 - 2 * g);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int u = 2 * g + h;
            const float8_t r = h == 0 ? hi8(cur) : lo8(cur);
            const float4_t rn = h == 0 ? hi4(curn) : lo4(curn);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int q = t + u;
                const float nk = (q < T) ? W[q < T ? q : 0] : N[q < T ? 0 : q - T];
                const float ek = (q < T) ? EW[q < T ? q : 0] : EN[q < T ? 0 : q - T];
                pair_pos<T>(acc, t, nk, r);
                pair_neg<T>(acc, t, ek, rn);
            }
            W[u] = ld_rowf(hn_next_u, u, lane);
            EW[u] = ld_rowf(he_next_u, u, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
        cur = nxt;
        curn = nxtn;
    }
}

/* the lags T-1 .. 1 whose source days lie before the block */
template <int T, bool WITH_E>
__device__ __forceinline__ void lag_tail_f(float2_t (&acc)[P_N][T], const float (&W)[T], const float* RFpos,
                                           const float* RFneg, const float* he_k0_u, unsigned lane) {
    float ek[WITH_E ? T - 1 : 1];
    if constexpr (WITH_E) {
#pragma unroll
        for (int q = 0; q < T - 1; ++q) ek[q] = he_k0_u[q * MCMC_WAVE + lane];
    }
#pragma unroll
    for (int v = 1; v < T; ++v) {
        const int L = T - v;
        const float8_t r = smem_rowf8(RFpos, L);
        float4_t rn;
        if constexpr (WITH_E) rn = smem_rowf4(RFneg, L);
#pragma unroll
        for (int t = 0; t < L; ++t) {
            pair_pos<T>(acc, t, W[t + v - 1], r);
            if constexpr (WITH_E) pair_neg<T>(acc, t, ek[t + v - 1], rn);
        }
    }
}


/* ------------------------------------------------------------ phase A on the matrix cores ---- */
typedef float float16v_t __attribute__((ext_vector_type(16)));

/* Streams nPairs pairs of source days through 2 x NT accumulator tiles.  hp: the lane's
 * A pointer of pair 0 (row k0 + lane/32, chain lane%32 of the float history; +32 = the other chain half; a
 * pair on = +128 floats).  li: the lane's LDS index of pair 0's B value.  NT = 2 (response table, 8 floats
 * per lag: tile 0 = days j0..j0+3, tile 1 at li + 32 = days j0+4..j0+7; a pair on = -16) or NT = 1 (correction
 * table, 4 floats per lag: one tile = 8 days x 4; a pair on = -8).  RING pairs of A values are in flight; the B
 * values of the NEXT pair are requested ahead of the current pair's MFMAs.  The ring is refilled
 * unconditionally (conditional refills cost the exact wait counts): reads ahead of the last pair stay inside
 * the history buffer (rows < j0 + 2 RING + 2) and the table's pad rows and are never used. */
/* The A values of a pair: the lane's row (k0 + lane/32) of the float history, chains 2n and 2n+1 (n = lane%32)
 * in one 8-byte load -- chain tile 0 holds the even chains of the wave, tile 1 the odd ones. */
__device__ __forceinline__ void ld_pair(float (&a)[2], const float* p) {
    const float2_t t = *(const float2_t*)p;
    a[0] = t.x;
    a[1] = t.y;
}
template <int RING>
__device__ __forceinline__ void mfma_ring_load(float (&ra)[RING][2], const float* hp) {
#pragma unroll
    for (int u = 0; u < RING; ++u) ld_pair(ra[u], hp + u * 128);
}
/* `ra`: the first RING pairs (mfma_ring_load, possibly issued long before, or a copy kept from an earlier block) */
template <int RING, int NT, int BSTEP>
__device__ __forceinline__ void mfma_stream(float16v_t (&C)[2][NT], float (&ra)[RING][2], const float* hp, const float* tab,
                                            int li, int nPairs) {
    float bv[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) bv[q] = tab[li + 32 * q];
    const float* hq = hp + RING * 128;
    int rem = nPairs;
    /* whole rings: the loop has ONE entry and ONE back edge and 2 RING loads are in flight at both, so the
     * wait ahead of slot u's MFMAs is exactly "all but the 2 RING - 2 younger loads" (an early exit inside the
     * unrolled body makes the wait-count pass assume the worst path and drain the ring to a quarter) */
#pragma unroll 1
    for (; rem >= RING; rem -= RING) {
        const float* tb = tab + (li - RING * BSTEP);         /* B of the pair after this ring: >= the pad rows */
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            float nb[NT];
#pragma unroll
            for (int q = 0; q < NT; ++q) nb[q] = tb[(RING - 1 - u) * BSTEP + 32 * q];
            __builtin_amdgcn_sched_barrier(0);
            const float a0 = ra[u][0], a1 = ra[u][1];
#pragma unroll
            for (int q = 0; q < NT; ++q) C[0][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[q], C[0][q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NT; ++q) C[1][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[q], C[1][q], 0, 0, 0);
            ld_pair(ra[u], hq + u * 128);
#pragma unroll
            for (int q = 0; q < NT; ++q) bv[q] = nb[q];
            __builtin_amdgcn_sched_barrier(0);
        }
        hq += RING * 128;
        li -= RING * BSTEP;
    }
    /* the last rem < RING pairs are in the ring already: no refills */
#pragma unroll
    for (int u = 0; u < RING - 1; ++u) {
        if (u < rem) {                                    /* no `break`: the loop stays unrolled, the ring in registers */
            li -= BSTEP;
            float nb[NT];
#pragma unroll
            for (int q = 0; q < NT; ++q) nb[q] = tab[li + 32 * q];
            __builtin_amdgcn_sched_barrier(0);
            const float a0 = ra[u][0], a1 = ra[u][1];
#pragma unroll
            for (int q = 0; q < NT; ++q) C[0][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[q], C[0][q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NT; ++q) C[1][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[q], C[1][q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NT; ++q) bv[q] = nb[q];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

/* All sources before the block (days kLo .. j0-1; ws, kLo and j0 are multiples of T) for the block's T days,
 * the capacity corrections of the source days [eLo, eHi] among them, and the hand-over to chain-per-lane.
 * Pairs of source days are (2p, 2p+1) in ABSOLUTE days.  With T = 9 a block may start on an odd day: the pair
 * that straddles the start is applied with the A value of its second day forced to zero (that day belongs to
 * the block), and a pair below kLo's parity reaches one day further back, whose contribution is an exact zero
 * (zeroed history below the first day, or lags >= nLagRows: rows of zeros).  The corrections touch three
 * curves only: their B tile is T days x (-SFS, -RFS, 1-DFS) in one 32-column tile, two MFMAs per pair. */
#define F32_RING 16           /* pairs of history values in flight in the main stream */
#define F32_RING_E 8          /* ... in the corrections' stream */
template <int NT>
__device__ __forceinline__ void mfma_once(float16v_t (&C)[2][NT], float a0, float a1, const float* tab, int li) {
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        const float bq = tab[li + 32 * q];
        C[0][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq, C[0][q], 0, 0, 0);
        C[1][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq, C[1][q], 0, 0, 0);
    }
}
template <int T>
__device__ __forceinline__ void mfma_sources(float2_t (&acc)[P_N][T], const LdsView& lds, float* tr, const float* hn_u,
                                             const float* he_u, int ws, int j0, int kLo, int eLo, int eHi,
                                             float (&ring0)[F32_RING][2], bool& haveRing0, unsigned lane, PhaseClock& pc) {
    static_assert(T * F32_NC <= 64 && T * F32_NE <= 32 && T * 4 <= F32_TR_JUNK, "block does not fit the MFMA tiles");
    constexpr int HOFF = 2 * T;
    const int half = (int)(lane >> 5), n = (int)(lane & 31);
    float16v_t C[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) C[a][b][i] = 0.0f;
    const int kE0 = kLo & ~1;                           /* first day of the first pair */
    const int nFull = (j0 - kE0) >> 1;                  /* pairs entirely before the block */
    const bool straddle = ((j0 - kE0) & 1) != 0;        /* pair (j0-1, j0): its second day is the block's first */
    /* corrections: the pairs touching [max(eLo, kLo), eHi]; whole ones first, then the straddling one */
    const int pLo = max(eLo, kLo) >> 1, pHi = min(eHi >> 1, (j0 >> 1) - 1);
    const bool withE = eHi >= eLo && pHi >= pLo;
    const bool tailE = straddle && eHi >= eLo && j0 - 1 <= eHi && j0 - 1 >= max(eLo, kLo);
    const int kE = withE ? 2 * pLo : kE0;
    const float* hpe = he_u + (kE + half + HOFF) * MCMC_WAVE + 2 * n;
    float re[F32_RING_E][2];
    if (withE) mfma_ring_load<F32_RING_E>(re, hpe);     /* requested before the main stream: the round trip hides there */
    /* ... and so are the values of the straddling pair's first day (only the lanes of the lower half use them) */
    float as[2] = {0.0f, 0.0f}, aes[2] = {0.0f, 0.0f};
    if (straddle) ld_pair(as, hn_u + (j0 - 1 + HOFF) * MCMC_WAVE + 2 * n);
    if (tailE) ld_pair(aes, he_u + (j0 - 1 + HOFF) * MCMC_WAVE + 2 * n);
    {
        /* Without lag skipping every block of a pass starts its stream with the SAME 32 history rows: once they
         * all lie before the block a copy stays in registers and the later blocks start without a memory round
         * trip. */
        const float* hp = hn_u + (kE0 + half + HOFF) * MCMC_WAVE + 2 * n;
        const int li = (j0 - kE0 - half + F32_TAB_PAD) * F32_NC + n;
        if (nFull > 0) {
            float ra[F32_RING][2];
            if (haveRing0 && kLo == ws) {
#pragma unroll
                for (int u = 0; u < F32_RING; ++u) { ra[u][0] = ring0[u][0]; ra[u][1] = ring0[u][1]; }
            } else {
                mfma_ring_load<F32_RING>(ra, hp);
                if (kLo == ws && nFull >= F32_RING) {
#pragma unroll
                    for (int u = 0; u < F32_RING; ++u) { ring0[u][0] = ra[u][0]; ring0[u][1] = ra[u][1]; }
                    haveRing0 = true;
                }
            }
            mfma_stream<F32_RING, 2, 2 * F32_NC>(C, ra, hp, lds.tabF, li, nFull);
        }
        if (straddle) mfma_once<2>(C, half ? 0.0f : as[0], half ? 0.0f : as[1], lds.tabF, (1 - half + F32_TAB_PAD) * F32_NC + n);
    }
    pc.stamp(PH_CHUNKS);
    float16v_t Ce[2][1];
    if (withE || tailE) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) Ce[a][0][i] = 0.0f;
        if (withE) mfma_stream<F32_RING_E, 1, 2 * F32_NE>(Ce, re, hpe, lds.tabN, (j0 - kE - half + F32_TAB_PAD) * F32_NE + n, pHi - pLo + 1);
        if (tailE) mfma_once<1>(Ce, half ? 0.0f : aes[0], half ? 0.0f : aes[1], lds.tabN, (1 - half + F32_TAB_PAD) * F32_NE + n);
    }
    pc.stamp(PH_CHUNKS_E);
    /* MFMA layout -> chain per lane through the wave's LDS buffer, one column tile at a time (the buffer is a
     * quarter of the wave's LDS share then: two workgroups fit a CU): element v of tile (mt, nt) is chain
     * 2 (8 (v/4) + 4 half + v%4) + mt, column 32 nt + n = (day, curve) 7 day + curve.  Tile 0 holds days
     * 0..F32_DSPLIT (of the last one the curves 0..3), tile 1 days F32_DSPLIT (curves 4..6) .. T-1; a chain's row
     * holds 8 floats per day and comes back as 16-byte reads (rows of 44 floats: lanes 8 apart share a bank group,
     * which a 16-byte read of 64 lanes needs eight passes for anyway). */
    constexpr int DS = F32_DSPLIT;
    static_assert(32 % F32_NC == 4 && (DS + 1) * 8 <= F32_TR_JUNK && (T - DS) * 8 <= F32_TR_JUNK && T > DS,
                  "the tile boundary has to fall between a day's two 16-byte halves");
    const int col1 = 32 + n;
    const int off0 = (n / F32_NC) * 8 + n % F32_NC;
    /* the 64th column goes to the row's unused tail instead of being predicated off */
    const int off1 = col1 < T * F32_NC ? (col1 / F32_NC - DS) * 8 + col1 % F32_NC : F32_TR_JUNK;
    const float4_t* mine = (const float4_t*)(tr + lane * F32_TR_ROW);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int v = 0; v < 16; ++v)
            tr[(2 * (8 * (v / 4) + 4 * half + (v % 4)) + mt) * F32_TR_ROW + off0] = C[mt][0][v];
    __builtin_amdgcn_s_waitcnt(0xc07f);                 /* lgkmcnt(0): the wave's own LDS writes have landed */
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t <= DS; ++t) {
        const float4_t lo = mine[2 * t];
        acc[P_INF][t] = float2_t{lo[0], lo[1]};
        acc[P_IM][t] = float2_t{lo[2], lo[3]};
        if (t < DS) {
            const float4_t hi = mine[2 * t + 1];
            acc[P_SR][t] = float2_t{hi[0], hi[1]};
            acc[P_D][t] = float2_t{hi[2], 0.0f};
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                 /* read back before the buffer is written again */
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int v = 0; v < 16; ++v)
            tr[(2 * (8 * (v / 4) + 4 * half + (v % 4)) + mt) * F32_TR_ROW + off1] = C[mt][1][v];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = DS; t < T; ++t) {
        const float4_t hi = mine[2 * (t - DS) + 1];
        acc[P_SR][t] = float2_t{hi[0], hi[1]};
        acc[P_D][t] = float2_t{hi[2], 0.0f};
        if (t > DS) {
            const float4_t lo = mine[2 * (t - DS)];
            acc[P_INF][t] = float2_t{lo[0], lo[1]};
            acc[P_IM][t] = float2_t{lo[2], lo[3]};
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (withE || tailE) {
        /* the corrections' tile through the same buffer: column n = (day, curve) 3 day + curve, 4 floats per day */
        const int offE = n < T * F32_NE ? (n / F32_NE) * 4 + n % F32_NE : F32_TR_JUNK;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int v = 0; v < 16; ++v)
                tr[(2 * (8 * (v / 4) + 4 * half + (v % 4)) + mt) * F32_TR_ROW + offE] = Ce[mt][0][v];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float4_t e4 = mine[t];
            acc[P_SR][t] += float2_t{e4[0], e4[1]};
            acc[P_D][t].x += e4[2];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
    pc.stamp(PH_TAIL);
}

/* -DMCMC_F32_CHECK (diagnosis builds, e.g. of another MCMC_TF): every row range the evaluation is about to
 * touch is compared with the size of its buffer first; the first violation is recorded in g_f32_check
 * (kind, low, high, bound) and the evaluation returns NaN instead of making the access. */
__device__ int g_f32_check[4];
#ifdef MCMC_F32_CHECK
#define F32_ROWS(kind, lo, hi, bound)                                                                    \
    do {                                                                                                 \
        if ((lo) < 0 || (hi) >= (bound)) {                                                               \
            if (lane == 0 && atomicCAS(&g_f32_check[0], 0, (kind)) == 0) {                               \
                g_f32_check[1] = (lo); g_f32_check[2] = (hi); g_f32_check[3] = (bound);                  \
            }                                                                                            \
            return __builtin_nan("");                                                                    \
        }                                                                                                \
    } while (0)
#else
#define F32_ROWS(kind, lo, hi, bound) do { } while (0)
#endif
enum { F32K_HIST_ZERO = 1, F32K_HIST_WINDOW, F32K_DAY_CONST, F32K_HIST_SKIP, F32K_TABLE_FIRST, F32K_TABLE_LAST, F32K_HIST_REFILL, F32K_HIST_DAYS };

/*
 * The fast-mode model evaluation for the chain of this lane; returns loglike (double).
 * B.hn_u / B.he_u are used as float rows (the buffers are sized for doubles), B.d1s_u stays
 * double.  M.RF / M.RFn are the float tables.
 */
template <int T, bool MFMA = false>
__device__ __forceinline__ double eval_model_f32(const DevModel& M, const LdsView& lds, const WaveBases& B,
                                                 const LaneScalars& lp, unsigned lane, WorkCount& work,
                                                 PhaseClock& pc) {
    const int md = M.md;
    const int horizon = M.simDays;
    const float* RFpos = M.RF;
    const float* RFneg = M.RFn;
    float* hn_u = (float*)B.hn_u;
    float* he_u = (float*)B.he_u;
    const double* xdisc_u = B.x_u + M.nB * MCMC_WAVE;
    constexpr int HOFF = 2 * T;
    const double invPop = 1.0 / M.pop;
    [[maybe_unused]] const int histF = 2 * (M.size + 4 * MCMC_T + MCMC_HOFF(MCMC_T));   /* float rows of hn / he (set_data's histRows doubles) */
    [[maybe_unused]] const int rowsF = M.size + 3 * T;                                 /* rows of RF / RFn */

    double loglike = 0;
    int start = (int)((double)md - lp.delay);                       /* covid.h:331 */
    const double frac = (double)md - lp.delay - (double)start;      /* covid.h:332 */
    start = min(max(start, 0), horizon);                            /* memory safety, as in eval_model */
    count_work(M, start, work);

#pragma unroll 1
    for (int shift = 1; shift >= 0; --shift) {                      /* covid.h:335 */
        const double w = (double)(1 - shift) * (1.0 - frac) + (double)shift * frac;
        const int s = min(start + shift, horizon);
        int ws = wave_min_i32(s);
        ws = min(ws, md);
        /* On the matrix cores a chain's sums must not depend on who shares its wave: an MFMA adds the two
         * products of a source-day pair in one step (not as two chained multiply-adds), so pairs and T-day
         * blocks are aligned to ABSOLUTE days, not to the wave's first day.  The up to T-1 days before ws are
         * "not started" days for every lane (zeros, handled below like any day i < s). */
        if constexpr (MFMA) ws -= ws % T;
        int keLo = 0x7fffffff, keHi = -1;

        float ncur = (s == ws) ? 1.0f : 0.0f;                       /* patient zero, covid.h:339 */
        F32_ROWS(F32K_HIST_ZERO, ws - 2 * T + HOFF, ws + HOFF, histF);
        F32_ROWS(F32K_HIST_WINDOW, ws - T + 1 + HOFF, ws - T + 1 + HOFF + 2 * T - 1, histF);
#pragma unroll 1
        for (int k = ws - 2 * T; k < ws; ++k) {
            hn_u[(k + HOFF) * MCMC_WAVE + lane] = 0.0f;
            he_u[(k + HOFF) * MCMC_WAVE + lane] = 0.0f;
        }
        hn_u[(ws + HOFF) * MCMC_WAVE + lane] = ncur;

        const int a0 = ws - T + 1;
        const float* hbase = hn_u + (a0 + HOFF) * MCMC_WAVE;
        const float* ebase = he_u + (a0 + HOFF) * MCMC_WAVE;
        float WA[T], WB[T];
        if constexpr (!MFMA) {
#pragma unroll
            for (int q = 0; q < T; ++q) {
                WA[q] = hbase[q * MCMC_WAVE + lane];
                WB[q] = hbase[(q + T) * MCMC_WAVE + lane];
            }
        }
        pc.stamp(PH_SETUP);
        double* ds_u = B.d1s_u + (shift == 1 ? 0 : M.nData * MCMC_WAVE);
        [[maybe_unused]] float ring0[MFMA ? 16 : 1][2];       /* mfma_sources: the stream's first history rows */
        [[maybe_unused]] bool haveRing0 = false;

#pragma unroll 1
        for (int j0 = ws; j0 < horizon; j0 += T) {
            float2_t acc[P_N][T];
#pragma unroll
            for (int c = 0; c < P_N; ++c)
#pragma unroll
                for (int t = 0; t < T; ++t) acc[c][t] = float2_t{0.0f, 0.0f};

            const int b = (j0 - ws) / T;
            /* what phase B needs from memory: requested while phase A still has work to cover the latency */
            double beh[T], pcfv[T], capv[T];
            auto request_block_inputs = [&]() {
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const int bi = min(max(j0 + t - md, 0), M.nB - 1);
                    beh[t] = B.x_u[bi * MCMC_WAVE + lane];
                    const int sg = __builtin_amdgcn_readfirstlane(lds.seg[j0 + t]);
                    const double dv = xdisc_u[max(sg, 0) * MCMC_WAVE + lane];
                    pcfv[t] = sg >= 0 ? dv : 1.0;                        /* covid.h:432-444 */
                    capv[t] = lds.capacity[j0 + t];
                }
            };
            if constexpr (MFMA) {
                /* ------------- phase A on the matrix cores: every source day before the block */
                int kLo = ws;                                        /* sources whose lags are all >= nLagRows: skipped */
                if (j0 - M.nLagRows + 1 - ws > 0) kLo += (j0 - M.nLagRows + 1 - ws) / T * T;
                F32_ROWS(F32K_DAY_CONST, j0, j0 + T - 1, M.simPad);
                F32_ROWS(F32K_HIST_DAYS, j0 + HOFF, j0 + T + HOFF, histF);
                F32_ROWS(F32K_HIST_REFILL, kLo - 1 + HOFF, j0 + 2 * 16 + 1 + HOFF, histF);   /* the A ring reads 16 pairs ahead */
                F32_ROWS(F32K_TABLE_FIRST, 0, j0 - kLo + 1 + T + 1, lds_rows_f32(M) + F32_TAB_SLACK);
                request_block_inputs();
                if (j0 > kLo) {
                    const int waveInBlock = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / MCMC_WAVE));
                    mfma_sources<T>(acc, lds, lds.tr + waveInBlock * F32_TR_FLOATS, hn_u, he_u, ws, j0, kLo, keLo, keHi, ring0, haveRing0, lane, pc);
                }
            } else {
                /* ------------- phase A */
                const bool anyE = keHi >= keLo;
                int Lc = j0 - ws + T - 1;
                int c0 = 0;
                F32_ROWS(F32K_DAY_CONST, j0, j0 + T - 1, M.simPad);
                F32_ROWS(F32K_HIST_DAYS, j0 + HOFF, j0 + T + HOFF, histF);              /* phase B: rows i and i+1 */
                F32_ROWS(F32K_HIST_REFILL, ws - T + 1 + HOFF, ws - T + 1 + HOFF + (b + 2) * T - 1, histF);   /* refills of the last chunk */
                if (Lc - T + 1 >= M.nLagRows) {                          /* rows of zeros: skipped (covid.h:378) */
                    c0 = (Lc - T + 1 - M.nLagRows) / T + 1;
                    if (c0 > b) c0 = b;
                    Lc -= c0 * T;
                    F32_ROWS(F32K_HIST_SKIP, ws - T + 1 + HOFF + c0 * T, ws - T + 1 + HOFF + c0 * T + 2 * T - 1, histF);
#pragma unroll
                    for (int q = 0; q < T; ++q) {
                        WA[q] = hbase[(c0 * T + q) * MCMC_WAVE + lane];
                        WB[q] = hbase[(c0 * T + q + T) * MCMC_WAVE + lane];
                    }
                }
                F32_ROWS(F32K_TABLE_FIRST, Lc - 1, Lc, rowsF);
                if (b > c0) F32_ROWS(F32K_TABLE_LAST, Lc - (b - 1 - c0) * T - T - 1, Lc, rowsF);   /* lowest row the last chunk requests */
                float16u_t cur = smem_rows2(RFpos, Lc - 1);
                const float* hnx = hbase + (c0 + 2) * T * MCMC_WAVE;
                int cLo = b, cHi = b - 1;
                if (anyE) {
                    const int num = keLo - a0 - 2 * T + 2;
                    cLo = num <= 0 ? 0 : (num + T - 1) / T;
                    cHi = (keHi - a0) / T;
                    cLo = max(cLo, c0);
                    cHi = min(cHi, b - 1);
                    if (cHi < cLo) { cLo = b; cHi = b - 1; }
                }
                const float* ek0 = ebase + cLo * T * MCMC_WAVE;
#pragma unroll 1
                for (int seg = 0; seg < 3; ++seg) {
                    int nch = (seg == 0) ? cLo - c0 : (seg == 1) ? cHi - cLo + 1 : b - 1 - cHi;
                    pc.count(seg == 1 ? PH_NCHUNK_E : PH_NCHUNK, nch);
                    unsigned laneL = lane;
                    asm volatile("" : "+v"(laneL));                      /* see eval_model: nothing pending at the loop header */
                    if (seg != 1) {
#pragma unroll 1
                        for (; nch >= 2; nch -= 2) {
                            lag_chunk_f<T>(acc, WA, WB, cur, RFpos, Lc, hnx, laneL);
                            lag_chunk_f<T>(acc, WB, WA, cur, RFpos, Lc - T, hnx + T * MCMC_WAVE, laneL);
                            hnx += 2 * T * MCMC_WAVE;
                            Lc -= 2 * T;
                        }
                        if (nch > 0) {
                            lag_chunk_f<T>(acc, WA, WB, cur, RFpos, Lc, hnx, laneL);
#pragma unroll
                            for (int q = 0; q < T; ++q) { const float tmp = WA[q]; WA[q] = WB[q]; WB[q] = tmp; }
                            hnx += T * MCMC_WAVE;
                            Lc -= T;
                        }
                    } else if (nch > 0) {
                        float EA[T], EB[T];
#pragma unroll
                        for (int q = 0; q < T; ++q) {
                            EA[q] = ek0[q * MCMC_WAVE + laneL];
                            EB[q] = ek0[(q + T) * MCMC_WAVE + laneL];
                        }
                        const float* ekx = ek0 + 2 * T * MCMC_WAVE;
                        float8u_t curn = smem_rows2n(RFneg, Lc - 1);
                        if (nch >= 2) __builtin_amdgcn_s_waitcnt(0x0F70);    /* vmcnt(0) ahead of the loop header */
#pragma unroll 1
                        for (; nch >= 2; nch -= 2) {
                            lag_chunk_fe<T>(acc, WA, WB, EA, EB, cur, curn, RFpos, RFneg, Lc, hnx, ekx, laneL);
                            lag_chunk_fe<T>(acc, WB, WA, EB, EA, cur, curn, RFpos, RFneg, Lc - T, hnx + T * MCMC_WAVE, ekx + T * MCMC_WAVE, laneL);
                            hnx += 2 * T * MCMC_WAVE;
                            ekx += 2 * T * MCMC_WAVE;
                            Lc -= 2 * T;
                        }
                        if (nch > 0) {
                            lag_chunk_fe<T>(acc, WA, WB, EA, EB, cur, curn, RFpos, RFneg, Lc, hnx, ekx, laneL);
#pragma unroll
                            for (int q = 0; q < T; ++q) { const float tmp = WA[q]; WA[q] = WB[q]; WB[q] = tmp; }
                            hnx += T * MCMC_WAVE;
                            Lc -= T;
                        }
                    }
                    pc.stamp(seg == 1 ? PH_CHUNKS_E : PH_CHUNKS);
                }
                request_block_inputs();
                pc.stamp(PH_CHUNKS);
                if (anyE && j0 - T + 1 <= keHi && j0 - 1 >= keLo)
                    lag_tail_f<T, true>(acc, WA, RFpos, RFneg, ebase + b * T * MCMC_WAVE, lane);
                else
                    lag_tail_f<T, false>(acc, WA, RFpos, RFneg, nullptr, lane);
                pc.stamp(PH_TAIL);
                if (b > 0) {
#pragma unroll
                    for (int q = 0; q < T; ++q) {
                        WA[q] = hbase[q * MCMC_WAVE + lane];
                        WB[q] = hbase[(q + T) * MCMC_WAVE + lane];
                    }
                }

            }
            /* ------------- phase B: the T days of the block, in order.  Its table rows 0..T-1: through the scalar
             * cache in the packed form; in the MFMA form from the LDS image into VGPRs (wave-uniform address =
             * broadcast read) -- 64 SGPRs of rows on top of the stream's bookkeeping made hipcc spill and restore
             * ~20 SGPRs per day */
            [[maybe_unused]] float8_t rowB[MFMA ? T : 1];
            if constexpr (MFMA) {
#pragma unroll
                for (int q = 0; q < T; ++q) rowB[q] = *(const float8_t*)(lds.rowsB + q * 8);
            }
            auto row_pos = [&](int q) -> float8_t {
                if constexpr (MFMA) return rowB[q];
                else return smem_rowf8(RFpos, q);
            };
            auto row_neg = [&](int q) -> float4_t {
                if constexpr (MFMA) return *(const float4_t*)(lds.rowsNB + q * 4);
                else return smem_rowf4(RFneg, q);
            };
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int i = j0 + t;
#pragma unroll
                for (int tt = t; tt < T; ++tt) pair_pos<T>(acc, tt, ncur, row_pos(tt - t));   /* covid.h:403-410 */
                /* hospital capacity (covid.h:413-429), decided in double */
                const double overCapacity = (double)acc[P_SR][t].x * M.probICU - capv[t];
                const double extraDeaths = M.probLethal * overCapacity;
                const bool hasE = (i >= s) && (extraDeaths > 0);
                const float e = hasE ? (float)extraDeaths : 0.0f;
                he_u[(i + HOFF) * MCMC_WAVE + lane] = e;
                if (__builtin_amdgcn_readfirstlane(__any(hasE) ? 1 : 0)) {
                    keLo = min(keLo, i);
                    keHi = max(keHi, i);
#pragma unroll
                    for (int tt = t; tt < T; ++tt) pair_neg<T>(acc, tt, e, row_neg(tt - t));
                }
                /* tomorrow's new infections (covid.h:446-462), in double */
                const double pcf = pcfv[t];
                double smoothbehavior = 1.0;
                if (i >= md) smoothbehavior = pcf * beh[t];
                const double totalBehavior = pcf * smoothbehavior;
                const double mild = (double)acc[P_IM][t].y;
                const double nSus = M.pop - (double)acc[P_IM][t].x - mild - mild - (double)acc[P_SR][t].x -
                                    (double)acc[P_SR][t].y - (double)acc[P_D][t].x;
                /* * (1/pop) instead of / pop: one rounding apart, and the division sequence is the
                 * longest link of the day-to-day dependency chain */
                double nnext = totalBehavior * nSus * invPop *
                               (lp.betaMild * (double)acc[P_INF][t].x + lp.betaHigh * (double)acc[P_INF][t].y);
                if (i < s) nnext = (i + 1 == s) ? 1.0 : 0.0;
                const float nnf = (float)nnext;
                hn_u[(i + 1 + HOFF) * MCMC_WAVE + lane] = nnf;
                if (i >= md && i - md < M.nData) ds_u[(i - md) * MCMC_WAVE + lane] = w * (double)acc[P_D][t].x;
                ncur = nnf;
            }
            if constexpr (!MFMA) {
                if (b == 0) {
#pragma unroll
                    for (int q = 0; q < T; ++q) {
                        WA[q] = hbase[q * MCMC_WAVE + lane];
                        WB[q] = hbase[(q + T) * MCMC_WAVE + lane];
                    }
                }
            }
            pc.stamp(PH_DAYS);
        }

        /* log-likelihood of this pass (covid.h:488-492), in double */
        {
            const double* d1_u = B.d1s_u;
            const double* d0_u = B.d1s_u + M.nData * MCMC_WAVE;
            constexpr int LB = MCMC_LL_BATCH;
            /* Whole batches without a per-day branch: the eight terms of a batch are independent chains of
             * ~15 dependent double operations each (the division), which the scheduler can only interleave if
             * no branch separates them; the sum itself stays in day order.  (The divisions still run one after
             * the other -- v_div_scale / v_div_fmas hand their flag over in VCC -- but the loads, LDS reads and
             * the arithmetic around them overlap: 281 -> 249 cycles per data day with one wave per SIMD.) */
            int d0 = 0;
#pragma unroll 1
            for (; d0 + LB <= M.nData; d0 += LB) {
                double dv[LB], obs[LB], s2[LB], term[LB];
#pragma unroll
                for (int j = 0; j < LB; ++j) dv[j] = d1_u[(d0 + j) * MCMC_WAVE + lane];
                if (shift == 0) {
                    double d2[LB];
#pragma unroll
                    for (int j = 0; j < LB; ++j) d2[j] = d0_u[(d0 + j) * MCMC_WAVE + lane];
#pragma unroll
                    for (int j = 0; j < LB; ++j) dv[j] = dv[j] + d2[j];
                }
#pragma unroll
                for (int j = 0; j < LB; ++j) {
                    obs[j] = lds.deaths[d0 + j];
                    s2[j] = lds.sigma2[d0 + j];
                }
#pragma unroll
                for (int j = 0; j < LB; ++j) {
                    const double delta = dv[j] - obs[j];
                    /* MFMA form: lds.sigma2 holds 1 / sigma^2 (stage_lds_f32) -- one multiplication instead of the
                     * ~12 dependent operations of a double division, one rounding apart from the reference's
                     * quotient, far inside this mode's tolerance */
                    if constexpr (MFMA) term[j] = w * (-0.5 * delta * delta * s2[j]);
                    else term[j] = w * (-0.5 * delta * delta / s2[j]);
                }
#pragma unroll
                for (int j = 0; j < LB; ++j) loglike += term[j];
            }
            /* the last nData % LB days */
#pragma unroll 1
            for (; d0 < M.nData; ++d0) {
                double dv = d1_u[d0 * MCMC_WAVE + lane];
                if (shift == 0) dv = dv + d0_u[d0 * MCMC_WAVE + lane];
                const double delta = dv - lds.deaths[d0];
                if constexpr (MFMA) loglike += w * (-0.5 * delta * delta * lds.sigma2[d0]);
                else loglike += w * (-0.5 * delta * delta / lds.sigma2[d0]);
            }
        }
        pc.stamp(PH_LOGLIKE);
    }
    return loglike;
}
