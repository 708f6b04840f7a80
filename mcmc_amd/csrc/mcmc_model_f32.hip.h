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
 *    read from SGPRs (one s_load_dwordx8 per lag step);
 *  - the per-lane histories are float (half the re-read traffic), kept in the same buffers;
 *  - the capacity correction of a source day is two packed multiply-adds per pair, with the
 *    row (-SFS, -RFS, 1-DFS, 0) prepared by the host (in double, rounded once): dead += e and
 *    dead -= e*DFS become one operation;
 *  - no cross-lane row tricks: rows 0..T-1 of the sequential part come from the scalar cache
 *    as well.
 */
#pragma once

typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float8_t __attribute__((ext_vector_type(8)));

#define MCMC_RF_STRIDE 16      /* floats per row of the float table: 8 positive curves, 4 correction terms, 4 pad */
enum { P_INF = 0, P_IM, P_SR, P_D, P_N };     /* (MI,HI) (INC,MILD) (SER,REC) (DEAD,0) */

typedef const __attribute__((address_space(4))) float8_t* const_rowf8_ptr;
typedef const __attribute__((address_space(4))) float4_t* const_rowf4_ptr;
__device__ __forceinline__ float8_t smem_rowf8(const float* RF, int L) {
    return *(const_rowf8_ptr)(RF + (long long)L * MCMC_RF_STRIDE);
}
__device__ __forceinline__ float4_t smem_rowf4(const float* RF, int L) {
    return *(const_rowf4_ptr)(RF + (long long)L * MCMC_RF_STRIDE + 8);
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

/* one chunk of T lags on the T days of the block; windows and row pipeline as in lag_chunk */
template <int T>
__device__ __forceinline__ void lag_chunk_f(float2_t (&acc)[P_N][T], float (&W)[T], const float (&N)[T], float8_t& r,
                                            const float* RF, int Lc, const float* hn_next_u, unsigned lane) {
    const float* rowBase = RF + (long long)(Lc - T) * MCMC_RF_STRIDE;    /* rows Lc-T .. Lc-1 */
    float8_t r1 = smem_rowf8(rowBase, T - 1);
#pragma unroll
    for (int u = 0; u < T; ++u) {
        const float8_t rnext = r1;
        r1 = smem_rowf8(rowBase, T - 2 - u);                              /* two steps ahead */
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int q = t + u;
            const float nk = (q < T) ? W[q < T ? q : 0] : N[q < T ? 0 : q - T];
            pair_pos<T>(acc, t, nk, r);
        }
        W[u] = ld_rowf(hn_next_u, u, lane);
        r = rnext;
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int T>
__device__ __forceinline__ void lag_chunk_fe(float2_t (&acc)[P_N][T], float (&W)[T], const float (&N)[T], float (&EW)[T],
                                             const float (&EN)[T], float8_t& r, float4_t& rn, const float* RF, int Lc,
                                             const float* hn_next_u, const float* he_next_u, unsigned lane) {
    const float* rowBase = RF + (long long)(Lc - T) * MCMC_RF_STRIDE;
#pragma unroll
    for (int u = 0; u < T; ++u) {
        const float8_t rnext = smem_rowf8(rowBase, T - 1 - u);
        const float4_t rnnext = smem_rowf4(rowBase, T - 1 - u);
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
        r = rnext;
        rn = rnnext;
        __builtin_amdgcn_sched_barrier(0);
    }
}

/* the lags T-1 .. 1 whose source days lie before the block */
template <int T, bool WITH_E>
__device__ __forceinline__ void lag_tail_f(float2_t (&acc)[P_N][T], const float (&W)[T], const float* RF,
                                           const float* he_k0_u, unsigned lane) {
    float ek[WITH_E ? T - 1 : 1];
    if constexpr (WITH_E) {
#pragma unroll
        for (int q = 0; q < T - 1; ++q) ek[q] = he_k0_u[q * MCMC_WAVE + lane];
    }
#pragma unroll
    for (int v = 1; v < T; ++v) {
        const int L = T - v;
        const float8_t r = smem_rowf8(RF, L);
        float4_t rn;
        if constexpr (WITH_E) rn = smem_rowf4(RF, L);
#pragma unroll
        for (int t = 0; t < L; ++t) {
            pair_pos<T>(acc, t, W[t + v - 1], r);
            if constexpr (WITH_E) pair_neg<T>(acc, t, ek[t + v - 1], rn);
        }
    }
}

/*
 * The fast-mode model evaluation for the chain of this lane; returns loglike (double).
 * B.hn_u / B.he_u are used as float rows (the buffers are sized for doubles), B.d1s_u stays
 * double.  `RF` = M.RF, the float table [rows][16].
 */
template <int T>
__device__ __forceinline__ double eval_model_f32(const DevModel& M, const LdsView& lds, const WaveBases& B,
                                                 const LaneScalars& lp, unsigned lane, unsigned long long& pairs,
                                                 PhaseClock& pc) {
    const int md = M.md;
    const int horizon = M.simDays;
    const float* RF = M.RF;
    float* hn_u = (float*)B.hn_u;
    float* he_u = (float*)B.he_u;
    const double* xdisc_u = B.x_u + M.nB * MCMC_WAVE;
    constexpr int HOFF = 2 * T;

    double loglike = 0;
    int start = (int)((double)md - lp.delay);                       /* covid.h:331 */
    const double frac = (double)md - lp.delay - (double)start;      /* covid.h:332 */
    start = min(max(start, 0), horizon);                            /* memory safety, as in eval_model */
    {
        long long m1 = M.size - min(start + 1, M.size), m0 = M.size - min(start, M.size);
        pairs += (unsigned long long)(m1 * (m1 + 1) / 2 + m0 * (m0 + 1) / 2);
    }

#pragma unroll 1
    for (int shift = 1; shift >= 0; --shift) {                      /* covid.h:335 */
        const double w = (double)(1 - shift) * (1.0 - frac) + (double)shift * frac;
        const int s = min(start + shift, horizon);
        int ws = wave_min_i32(s);
        ws = min(ws, md);
        int keLo = 0x7fffffff, keHi = -1;

        float ncur = (s == ws) ? 1.0f : 0.0f;                       /* patient zero, covid.h:339 */
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
#pragma unroll
        for (int q = 0; q < T; ++q) {
            WA[q] = hbase[q * MCMC_WAVE + lane];
            WB[q] = hbase[(q + T) * MCMC_WAVE + lane];
        }
        pc.stamp(PH_SETUP);
        double* ds_u = B.d1s_u + (shift == 1 ? 0 : M.nData * MCMC_WAVE);

#pragma unroll 1
        for (int j0 = ws; j0 < horizon; j0 += T) {
            float2_t acc[P_N][T];
#pragma unroll
            for (int c = 0; c < P_N; ++c)
#pragma unroll
                for (int t = 0; t < T; ++t) acc[c][t] = float2_t{0.0f, 0.0f};

            /* ------------- phase A */
            const int b = (j0 - ws) / T;
            const bool anyE = keHi >= keLo;
            int Lc = j0 - ws + T - 1;
            int c0 = 0;
            if (Lc - T + 1 >= M.nLagRows) {                          /* rows of zeros: skipped (covid.h:378) */
                c0 = (Lc - T + 1 - M.nLagRows) / T + 1;
                if (c0 > b) c0 = b;
                Lc -= c0 * T;
#pragma unroll
                for (int q = 0; q < T; ++q) {
                    WA[q] = hbase[(c0 * T + q) * MCMC_WAVE + lane];
                    WB[q] = hbase[(c0 * T + q + T) * MCMC_WAVE + lane];
                }
            }
            float8_t r = smem_rowf8(RF, Lc);
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
                        lag_chunk_f<T>(acc, WA, WB, r, RF, Lc, hnx, laneL);
                        lag_chunk_f<T>(acc, WB, WA, r, RF, Lc - T, hnx + T * MCMC_WAVE, laneL);
                        hnx += 2 * T * MCMC_WAVE;
                        Lc -= 2 * T;
                    }
                    if (nch > 0) {
                        lag_chunk_f<T>(acc, WA, WB, r, RF, Lc, hnx, laneL);
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
                    float4_t rn = smem_rowf4(RF, Lc);
                    if (nch >= 2) __builtin_amdgcn_s_waitcnt(0x0F70);    /* vmcnt(0) ahead of the loop header */
#pragma unroll 1
                    for (; nch >= 2; nch -= 2) {
                        lag_chunk_fe<T>(acc, WA, WB, EA, EB, r, rn, RF, Lc, hnx, ekx, laneL);
                        lag_chunk_fe<T>(acc, WB, WA, EB, EA, r, rn, RF, Lc - T, hnx + T * MCMC_WAVE, ekx + T * MCMC_WAVE, laneL);
                        hnx += 2 * T * MCMC_WAVE;
                        ekx += 2 * T * MCMC_WAVE;
                        Lc -= 2 * T;
                    }
                    if (nch > 0) {
                        lag_chunk_fe<T>(acc, WA, WB, EA, EB, r, rn, RF, Lc, hnx, ekx, laneL);
#pragma unroll
                        for (int q = 0; q < T; ++q) { const float tmp = WA[q]; WA[q] = WB[q]; WB[q] = tmp; }
                        hnx += T * MCMC_WAVE;
                        Lc -= T;
                    }
                }
                pc.stamp(seg == 1 ? PH_CHUNKS_E : PH_CHUNKS);
            }
            /* what phase B needs from memory, requested before the tail */
            double beh[T], pcfv[T], capv[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int bi = min(max(j0 + t - md, 0), M.nB - 1);
                beh[t] = B.x_u[bi * MCMC_WAVE + lane];
                const int sg = __builtin_amdgcn_readfirstlane(lds.seg[j0 + t]);
                const double dv = xdisc_u[max(sg, 0) * MCMC_WAVE + lane];
                pcfv[t] = sg >= 0 ? dv : 1.0;                        /* covid.h:432-444 */
                capv[t] = lds.capacity[j0 + t];
            }
            pc.stamp(PH_CHUNKS);
            if (anyE && j0 - T + 1 <= keHi && j0 - 1 >= keLo)
                lag_tail_f<T, true>(acc, WA, RF, ebase + b * T * MCMC_WAVE, lane);
            else
                lag_tail_f<T, false>(acc, WA, RF, nullptr, lane);
            pc.stamp(PH_TAIL);
            if (b > 0) {
#pragma unroll
                for (int q = 0; q < T; ++q) {
                    WA[q] = hbase[q * MCMC_WAVE + lane];
                    WB[q] = hbase[(q + T) * MCMC_WAVE + lane];
                }
            }

            /* ------------- phase B: the T days of the block, in order */
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int i = j0 + t;
#pragma unroll
                for (int tt = t; tt < T; ++tt) pair_pos<T>(acc, tt, ncur, smem_rowf8(RF, tt - t));   /* covid.h:403-410 */
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
                    for (int tt = t; tt < T; ++tt) pair_neg<T>(acc, tt, e, smem_rowf4(RF, tt - t));
                }
                /* tomorrow's new infections (covid.h:446-462), in double */
                const double pcf = pcfv[t];
                double smoothbehavior = 1.0;
                if (i >= md) smoothbehavior = pcf * beh[t];
                const double totalBehavior = pcf * smoothbehavior;
                const double mild = (double)acc[P_IM][t].y;
                const double nSus = M.pop - (double)acc[P_IM][t].x - mild - mild - (double)acc[P_SR][t].x -
                                    (double)acc[P_SR][t].y - (double)acc[P_D][t].x;
                double nnext = totalBehavior * nSus / M.pop *
                               (lp.betaMild * (double)acc[P_INF][t].x + lp.betaHigh * (double)acc[P_INF][t].y);
                if (i < s) nnext = (i + 1 == s) ? 1.0 : 0.0;
                const float nnf = (float)nnext;
                hn_u[(i + 1 + HOFF) * MCMC_WAVE + lane] = nnf;
                if (i >= md && i - md < M.nData) ds_u[(i - md) * MCMC_WAVE + lane] = w * (double)acc[P_D][t].x;
                ncur = nnf;
            }
            if (b == 0) {
#pragma unroll
                for (int q = 0; q < T; ++q) {
                    WA[q] = hbase[q * MCMC_WAVE + lane];
                    WB[q] = hbase[(q + T) * MCMC_WAVE + lane];
                }
            }
            pc.stamp(PH_DAYS);
        }

        /* log-likelihood of this pass (covid.h:488-492), in double */
        {
            const double* d1_u = B.d1s_u;
            const double* d0_u = B.d1s_u + M.nData * MCMC_WAVE;
            constexpr int LB = MCMC_LL_BATCH;
#pragma unroll 1
            for (int d0 = 0; d0 < M.nData; d0 += LB) {
                double dv[LB];
#pragma unroll
                for (int j = 0; j < LB; ++j) dv[j] = d1_u[min(d0 + j, M.nData - 1) * MCMC_WAVE + lane];
                if (shift == 0) {
                    double d2[LB];
#pragma unroll
                    for (int j = 0; j < LB; ++j) d2[j] = d0_u[min(d0 + j, M.nData - 1) * MCMC_WAVE + lane];
#pragma unroll
                    for (int j = 0; j < LB; ++j) dv[j] = dv[j] + d2[j];
                }
#pragma unroll
                for (int j = 0; j < LB; ++j) {
                    const int di = d0 + j;
                    if (di < M.nData) {
                        const double delta = dv[j] - lds.deaths[di];
                        loglike += w * (-0.5 * delta * delta / lds.sigma2[di]);
                    }
                }
            }
        }
        pc.stamp(PH_LOGLIKE);
    }
    return loglike;
}
