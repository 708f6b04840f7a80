/*
 * mcmc_kernels.hip.h -- CDNA4 (gfx950) device code of the Metropolis-Hastings hot path.
 *
 * One chain per lane; 64 chains per wavefront advance in lockstep.  All chain state
 * lives in HBM as structure-of-arrays [field][chain] so that every per-lane access of a
 * wave is one contiguous 512-byte line.  Citations are file:line under /root/reference.
 *
 * THE MODEL TIME-STEPPER (DiseaseSpread::eval, covid.h:325-496) -- how it is laid out
 * ------------------------------------------------------------------------------------
 * The reference SCATTERS: on day k it adds n_k * cumsum_c(d) into eight length-`size`
 * buffers for all d (project(), covid.h:376-386), so each chain drags 8*size doubles
 * of read-modify-write state.  Here every destination day is instead GATHERED:
 *
 *     buf_c[j] = sum over k <= j of  n_k * R_c[j-k]  (+ capacity corrections e_k)
 *
 * with the addends of one destination applied in ascending k, and within one k in the
 * reference's order (8 projections, flat add, 3 negative projections: covid.h:403-428).
 * That is exactly the sequence of floating-point additions buf_c[j] receives in the
 * reference, so the result is bit-identical, but a lane now only stores its infection
 * history n_k (and e_k): 2*size doubles.
 *
 * Days are processed in blocks of T.  A block keeps 7*T accumulators in VGPRs
 * (asymptomaticBuf is projected with traj.mild, covid.h:406, so it equals mildBuf and
 * needs none of its own).  Phase A walks the lags L = j-k from large to small (= k
 * ascending); at each lag ONE history value per lane is loaded (coalesced 8 B/lane)
 * and feeds 7*T multiply-adds whose second operand R_c[L] is wave-uniform and comes
 * in through the scalar cache as an SGPR operand -- no VGPR, no LDS traffic for the
 * table.  Phase B then runs the T days of the block in order (the feedback
 * n_{i+1} = f(buffers[i]) is inherently sequential, covid.h:461-462), pushing each new
 * n_i into the later days of the block from registers.
 *
 * Only what the log-likelihood can see is simulated in the sampling kernels: days
 * [0, maxDelay+nData).  The nPredictDays tail (covid.h:297) cannot influence `loglike`
 * (covid.h:488-492 reads dead[i+maxDelay], i < nData, and buffers at day j only depend
 * on days <= j); the derived curves over the full horizon are produced on request by
 * the DERIVE instantiation (the reference's compute_derived_late idea, mcmc.h:100-104).
 *
 * This file must be compiled with -ffp-contract=off: every `a*b+c` below is two
 * roundings unless it goes through madd<true>.
 */
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mcmc_detmath.h"

#define MCMC_MAXDISC 8
#define MCMC_WAVE 64

/* constant address space: loads through it are invariant, so a wave-uniform address
 * becomes an s_load (scalar cache -> SGPR) instead of a per-lane vector load. */
typedef const double __attribute__((address_space(4))) cdouble_t;
typedef const int __attribute__((address_space(4))) cint_t;
typedef double double8_t __attribute__((ext_vector_type(8)));
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef const double8_t __attribute__((address_space(4))) cdouble8_t;
typedef const double4_t __attribute__((address_space(4))) cdouble4_t;

/* accumulator families */
enum { A_MI = 0, A_HI, A_INC, A_MILD, A_SER, A_REC, A_DEAD, A_NPOS };

struct DevModel {
    int nData, md, size, nB, nDisc, simDays, nLagRows, hasPrior;
    long long nP;                 /* padded chain count = SoA stride                  */
    double pop, probICU, probLethal, smoothLs, smoothL, dNData;
    float discThr;                /* 2/nNotFixed, float division done on the host    */
    unsigned discFixedMask;
    const double* Rpos;           /* [nLagRows][8]: MI HI INC MILD SER REC DEAD 0    */
    const double* Rneg;           /* [nLagRows][4]: SFS RFS DFS 0                    */
    const double* capacity;       /* [simPad]                                        */
    const double* deaths;         /* [nData]                                         */
    const double* sigma2;         /* [nData]  sigma[i]*sigma[i] (covid.h:491)        */
    const int* seg;               /* [simPad] piecewise-constant segment of day i, -1 = none */
};

/* ------------------------------------------------------------------- pcg32 ---- */
/* wjakob/pcg32 (submodule, not vendored): the stream every chain draws, mcmc.h:999 */
struct Pcg32 {
    uint64_t state, inc;
    __device__ __forceinline__ uint32_t next_uint() {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
    }
    __device__ __forceinline__ float next_float() {
        return __uint_as_float((next_uint() >> 9) | 0x3f800000u) - 1.0f;
    }
    __device__ __forceinline__ double next_double() {
        return __longlong_as_double(((uint64_t)next_uint() << 20) | 0x3ff0000000000000ULL) - 1.0;
    }
};

__host__ __device__ inline void pcg32_seed(uint64_t& state, uint64_t& inc, uint64_t initstate,
                                           uint64_t initseq) {
    const uint64_t mult = 0x5851f42d4c957f2dULL;
    state = 0u;
    inc = (initseq << 1u) | 1u;
    state = state * mult + inc;
    state += initstate;
    state = state * mult + inc;
}

/* --------------------------------------------------------------- arithmetic ---- */
template <bool FUSE>
__device__ __forceinline__ double madd(double a, double b, double c) {
    if constexpr (FUSE) return __builtin_fma(a, b, c);
    else return c + a * b;
}

/* mcmc.h:49-53; std::round is half-away-from-zero */
__device__ __forceinline__ double bound_reflect(double val, double lower, double upper) {
    val = 0.5 * (val - lower) / (upper - lower);
    val = 2.0 * fabs(val - round(val));
    val = lower + val * (upper - lower);
    return val;
}

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

/* LDS image of the observed series and the per-day constants, staged once per block */
struct LdsView {
    const double* capacity;   /* [simPad] */
    const double* deaths;     /* [nData]  */
    const double* sigma2;     /* [nData]  */
    const int* seg;           /* [simPad] */
};

__device__ inline LdsView stage_lds(const DevModel& M, double* lds, int simPad) {
    double* cap = lds;
    double* dea = cap + simPad;
    double* sg2 = dea + M.nData;
    int* seg = (int*)(sg2 + M.nData);
    for (int i = threadIdx.x; i < simPad; i += blockDim.x) {
        cap[i] = M.capacity[i];
        seg[i] = M.seg[i];
    }
    for (int i = threadIdx.x; i < M.nData; i += blockDim.x) {
        dea[i] = M.deaths[i];
        sg2[i] = M.sigma2[i];
    }
    __syncthreads();
    LdsView v;
    v.capacity = cap; v.deaths = dea; v.sigma2 = sg2; v.seg = seg;
    return v;
}
__host__ __device__ inline size_t lds_bytes(int simPad, int nData) {
    return sizeof(double) * ((size_t)simPad + 2 * (size_t)nData) + sizeof(int) * (size_t)simPad + 8;
}

/*
 * Addressing convention.  Every per-chain array is SoA [row][nP].  A wave addresses it as
 * (uniform base + uniform row offset)[lane]: the row part stays in SGPRs and the lane part
 * is ONE shared 32-bit VGPR offset, so a load/store costs no address arithmetic on the
 * vector ALU and no address registers (global_load ... v_off, s[base:base+1]).
 * All `*_u` pointers below are wave-uniform and already advanced to the wave's first chain.
 */
struct WaveBases {
    const double* x_u;     /* parameter vector being evaluated, element k at x_u[k*nP + lane] */
    double* hn_u;          /* infection history,  day k at hn_u[(k+HOFF)*nP + lane]             */
    double* he_u;          /* extra-deaths history, same indexing                               */
    double* d1s_u;         /* shift-1 weighted deaths of data day i at d1s_u[i*nP + lane]       */
    double* curves_u;      /* DERIVE only: curve c, day i at curves_u[(c*size+i)*nP + lane]     */
};

/* the scalars of a parameter vector that the day loop needs all the time */
struct LaneScalars {
    double betaMild, betaHigh, delay;
};

/* ------------------------------------------------------- phase A building blocks ---- */
/* One chunk = T consecutive lags Lc, Lc-1, ..., Lc-T+1 applied to the T days of the block.
 * History window: element q in [0, 2T-1) is W[q] for q < T and N[q-T] otherwise, holding
 * n[kbase + q].  After step u the slot W[u] is dead and is refilled with the history value
 * 2T further on (it becomes the `N` of the chunk after next: ping-pong, no register moves).
 * WITH_E adds the capacity corrections of covid.h:416-429 from the e-history. */
template <int T, bool FUSE, bool WITH_E>
__device__ __forceinline__ void lag_chunk(double (&acc)[A_NPOS][T], double (&W)[T], double (&N)[T],
                                          cdouble8_t* Rpos, cdouble4_t* Rneg, int Lc,
                                          const double* hn_next_u, const double* he_k0_u,
                                          long long nP, unsigned lane) {
#pragma unroll
    for (int u = 0; u < T; ++u) {
        const double8_t r = Rpos[Lc - u];
        double4_t rn;
        if constexpr (WITH_E) rn = Rneg[Lc - u];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int q = t + u;
            const double nk = (q < T) ? W[q < T ? q : 0] : N[q < T ? 0 : q - T];
            acc[A_MI][t] = madd<FUSE>(nk, r[0], acc[A_MI][t]);
            acc[A_HI][t] = madd<FUSE>(nk, r[1], acc[A_HI][t]);
            acc[A_INC][t] = madd<FUSE>(nk, r[2], acc[A_INC][t]);
            acc[A_MILD][t] = madd<FUSE>(nk, r[3], acc[A_MILD][t]);
            acc[A_SER][t] = madd<FUSE>(nk, r[4], acc[A_SER][t]);
            acc[A_REC][t] = madd<FUSE>(nk, r[5], acc[A_REC][t]);
            acc[A_DEAD][t] = madd<FUSE>(nk, r[6], acc[A_DEAD][t]);
            if constexpr (WITH_E) {
                /* rare path (only source days with hospital overflow): no register window,
                 * the e-history is re-read (L1/L2-resident) */
                const double ek = (he_k0_u + (long long)q * nP)[lane];
                const double nek = -ek;
                acc[A_DEAD][t] = acc[A_DEAD][t] + ek;                       /* add(), covid.h:418 */
                acc[A_SER][t] = madd<FUSE>(nek, rn[0], acc[A_SER][t]);      /* covid.h:426 */
                acc[A_REC][t] = madd<FUSE>(nek, rn[1], acc[A_REC][t]);      /* covid.h:427 */
                acc[A_DEAD][t] = madd<FUSE>(nek, rn[2], acc[A_DEAD][t]);    /* covid.h:428 */
            }
        }
        /* slot u of W is dead from here on: prefetch for the chunk after next */
        W[u] = (hn_next_u + (long long)u * nP)[lane];
    }
}

/* the lags T-1 .. 1: only destinations t < L have their source day before the block */
template <int T, bool FUSE, bool WITH_E>
__device__ __forceinline__ void lag_tail(double (&acc)[A_NPOS][T], const double (&W)[T],
                                         cdouble8_t* Rpos, cdouble4_t* Rneg, const double* he_k0_u,
                                         long long nP, unsigned lane) {
#pragma unroll
    for (int v = 1; v < T; ++v) {
        const int L = T - v;
        const double8_t r = Rpos[L];
        double4_t rn;
        if constexpr (WITH_E) rn = Rneg[L];
#pragma unroll
        for (int t = 0; t < L; ++t) {
            const double nk = W[t + v - 1];
            acc[A_MI][t] = madd<FUSE>(nk, r[0], acc[A_MI][t]);
            acc[A_HI][t] = madd<FUSE>(nk, r[1], acc[A_HI][t]);
            acc[A_INC][t] = madd<FUSE>(nk, r[2], acc[A_INC][t]);
            acc[A_MILD][t] = madd<FUSE>(nk, r[3], acc[A_MILD][t]);
            acc[A_SER][t] = madd<FUSE>(nk, r[4], acc[A_SER][t]);
            acc[A_REC][t] = madd<FUSE>(nk, r[5], acc[A_REC][t]);
            acc[A_DEAD][t] = madd<FUSE>(nk, r[6], acc[A_DEAD][t]);
            if constexpr (WITH_E) {
                const double ek = (he_k0_u + (long long)(t + v - 1) * nP)[lane];
                const double nek = -ek;
                acc[A_DEAD][t] = acc[A_DEAD][t] + ek;
                acc[A_SER][t] = madd<FUSE>(nek, rn[0], acc[A_SER][t]);
                acc[A_REC][t] = madd<FUSE>(nek, rn[1], acc[A_REC][t]);
                acc[A_DEAD][t] = madd<FUSE>(nek, rn[2], acc[A_DEAD][t]);
            }
        }
    }
}

/* rows of the history arrays are offset so that index k = -HOFF is row 0 */
#define MCMC_HOFF(T) (2 * (T))

/*
 * DiseaseSpread::eval (covid.h:325-496) for the chain of this lane.  Returns loglike.
 * pairs += number of (day,lag) pairs the reference performs for this evaluation over the
 * full horizon `size` (the SURVEY 8d work unit).
 */
template <int T, bool FUSE, bool DERIVE>
__device__ double eval_model(const DevModel& M, const LdsView& lds, const WaveBases& B,
                             const LaneScalars& lp, unsigned lane, unsigned long long& pairs) {
    const long long nP = M.nP;
    const int md = M.md;
    const int horizon = DERIVE ? M.size : M.simDays;
    cdouble8_t* Rpos = (cdouble8_t*)M.Rpos;
    cdouble4_t* Rneg = (cdouble4_t*)M.Rneg;
    cdouble_t* Rp = (cdouble_t*)M.Rpos;      /* lag 0 .. T-1, used inside a block */
    cdouble_t* Rn = (cdouble_t*)M.Rneg;
    const double* xdisc_u = B.x_u + (long long)M.nB * nP;

    double loglike = 0;                                             /* covid.h:329 */
    int start = (int)((double)md - lp.delay);                       /* covid.h:331 */
    const double frac = (double)md - lp.delay - (double)start;      /* covid.h:332 */
    /* memory safety: every day index below derives from these clamped values */
    start = min(max(start, 0), horizon);
    {
        long long m1 = M.size - min(start + 1, M.size), m0 = M.size - min(start, M.size);
        pairs += (unsigned long long)(m1 * (m1 + 1) / 2 + m0 * (m0 + 1) / 2);
    }

#pragma unroll 1
    for (int shift = 1; shift >= 0; --shift) {                      /* covid.h:335 */
        const double w = (double)(1 - shift) * (1.0 - frac) + (double)shift * frac; /* covid.h:333 */
        const int s = min(start + shift, horizon);
        int ws = wave_min_i32(s);
        ws = min(ws, md);                  /* cover every data day even if no lane has started */
        /* wave-uniform day range that holds capacity corrections (empty: lo > hi) */
        int keLo = 0x7fffffff, keHi = -1;

        /* zero the 2T history rows below ws (read as k < ws by the first blocks) and plant
         * patient zero (covid.h:339): n_s = 1, n_k = 0 for k < s */
#pragma unroll 1
        for (int k = ws - 2 * T; k < ws; ++k) {
            (B.hn_u + (long long)(k + MCMC_HOFF(T)) * nP)[lane] = 0.0;
            (B.he_u + (long long)(k + MCMC_HOFF(T)) * nP)[lane] = 0.0;
        }
        double ncur = (s == ws) ? 1.0 : 0.0;
        (B.hn_u + (long long)(ws + MCMC_HOFF(T)) * nP)[lane] = ncur;

        if constexpr (DERIVE) {
            /* days before the wave's first start: all buffers are zero there */
#pragma unroll 1
            for (int i = 0; i < ws; ++i) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    double* o = B.curves_u + ((long long)c * M.size + i) * nP;
                    if (shift == 1) o[lane] = w * 0.0; else o[lane] += w * 0.0;
                }
                (B.curves_u + ((long long)8 * M.size + i) * nP)[lane] = 1.0; /* covid.h:349 */
            }
        }

#pragma unroll 1
        for (int j0 = ws; j0 < horizon; j0 += T) {
            double acc[A_NPOS][T];
#pragma unroll
            for (int c = 0; c < A_NPOS; ++c)
#pragma unroll
                for (int t = 0; t < T; ++t) acc[c][t] = 0.0;

            /* ------------- phase A: sources k < j0, lags descending = k ascending */
            const int b = (j0 - ws) / T;                     /* full chunks of T lags */
            {
                double WA[T], WB[T];
                const int a0 = ws - T + 1;                   /* k of window element 0 */
                const double* hbase = B.hn_u + (long long)(a0 + MCMC_HOFF(T)) * nP;
                const double* ebase = B.he_u + (long long)(a0 + MCMC_HOFF(T)) * nP;
                const bool anyE = keHi >= keLo;
#pragma unroll
                for (int q = 0; q < T; ++q) {
                    WA[q] = (hbase + (long long)q * nP)[lane];
                    WB[q] = (hbase + (long long)(q + T) * nP)[lane];
                }
                int cb = 0;
                int Lc = j0 - ws + T - 1;
                /* chunk c covers source days [a0+c*T, a0+c*T+2T-2] */
                auto chunk_has_e = [&](int c) {
                    const int klo = a0 + c * T, khi = klo + 2 * T - 2;
                    return anyE && klo <= keHi && khi >= keLo;
                };
                /* lags >= nLag (500, covid.h:378) carry R = 0 rows: chunks made only of them
                 * just keep the window sliding */
#pragma unroll 1
                while (cb + 2 <= b) {
                    const double* hnx = hbase + (long long)(cb + 2) * T * nP;
                    const double* ek0 = ebase + (long long)cb * T * nP;
                    if (Lc - T + 1 < M.nLagRows) {
                        if (chunk_has_e(cb)) lag_chunk<T, FUSE, true>(acc, WA, WB, Rpos, Rneg, Lc, hnx, ek0, nP, lane);
                        else lag_chunk<T, FUSE, false>(acc, WA, WB, Rpos, Rneg, Lc, hnx, ek0, nP, lane);
                    } else {
#pragma unroll
                        for (int u = 0; u < T; ++u) WA[u] = (hnx + (long long)u * nP)[lane];
                    }
                    hnx += (long long)T * nP;
                    ek0 += (long long)T * nP;
                    if (Lc - 2 * T + 1 < M.nLagRows) {
                        if (chunk_has_e(cb + 1)) lag_chunk<T, FUSE, true>(acc, WB, WA, Rpos, Rneg, Lc - T, hnx, ek0, nP, lane);
                        else lag_chunk<T, FUSE, false>(acc, WB, WA, Rpos, Rneg, Lc - T, hnx, ek0, nP, lane);
                    } else {
#pragma unroll
                        for (int u = 0; u < T; ++u) WB[u] = (hnx + (long long)u * nP)[lane];
                    }
                    cb += 2;
                    Lc -= 2 * T;
                }
                const double* ekb = ebase + (long long)b * T * nP;
                if (cb < b) {
                    const double* hnx = hbase + (long long)(cb + 2) * T * nP;
                    const double* ek0 = ebase + (long long)cb * T * nP;
                    if (chunk_has_e(cb)) lag_chunk<T, FUSE, true>(acc, WA, WB, Rpos, Rneg, Lc, hnx, ek0, nP, lane);
                    else lag_chunk<T, FUSE, false>(acc, WA, WB, Rpos, Rneg, Lc, hnx, ek0, nP, lane);
                    if (chunk_has_e(b)) lag_tail<T, FUSE, true>(acc, WB, Rpos, Rneg, ekb, nP, lane);
                    else lag_tail<T, FUSE, false>(acc, WB, Rpos, Rneg, ekb, nP, lane);
                } else {
                    if (chunk_has_e(b)) lag_tail<T, FUSE, true>(acc, WA, Rpos, Rneg, ekb, nP, lane);
                    else lag_tail<T, FUSE, false>(acc, WA, Rpos, Rneg, ekb, nP, lane);
                }
            }

            /* ------------- phase B: the T days of the block, in order */
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int i = j0 + t;
                /* today's new infections into today and the rest of the block (covid.h:403-410) */
#pragma unroll
                for (int tt = t; tt < T; ++tt) {
                    const int d = tt - t;
                    acc[A_MI][tt] = madd<FUSE>(ncur, Rp[d * 8 + 0], acc[A_MI][tt]);
                    acc[A_HI][tt] = madd<FUSE>(ncur, Rp[d * 8 + 1], acc[A_HI][tt]);
                    acc[A_INC][tt] = madd<FUSE>(ncur, Rp[d * 8 + 2], acc[A_INC][tt]);
                    acc[A_MILD][tt] = madd<FUSE>(ncur, Rp[d * 8 + 3], acc[A_MILD][tt]);
                    acc[A_SER][tt] = madd<FUSE>(ncur, Rp[d * 8 + 4], acc[A_SER][tt]);
                    acc[A_REC][tt] = madd<FUSE>(ncur, Rp[d * 8 + 5], acc[A_REC][tt]);
                    acc[A_DEAD][tt] = madd<FUSE>(ncur, Rp[d * 8 + 6], acc[A_DEAD][tt]);
                }
                /* hospital capacity (covid.h:413-429) */
                const double overCapacity = acc[A_SER][t] * M.probICU - lds.capacity[i];
                const double extraDeaths = M.probLethal * overCapacity;
                const bool hasE = (i >= s) && (extraDeaths > 0);
                const double e = hasE ? extraDeaths : 0.0;
                (B.he_u + (long long)(i + MCMC_HOFF(T)) * nP)[lane] = e;
                if (__builtin_amdgcn_readfirstlane(__any(hasE) ? 1 : 0)) {
                    keLo = min(keLo, i);
                    keHi = max(keHi, i);
                    const double ne = -e;
#pragma unroll
                    for (int tt = t; tt < T; ++tt) {
                        const int d = tt - t;
                        acc[A_DEAD][tt] = acc[A_DEAD][tt] + e;
                        acc[A_SER][tt] = madd<FUSE>(ne, Rn[d * 4 + 0], acc[A_SER][tt]);
                        acc[A_REC][tt] = madd<FUSE>(ne, Rn[d * 4 + 1], acc[A_REC][tt]);
                        acc[A_DEAD][tt] = madd<FUSE>(ne, Rn[d * 4 + 2], acc[A_DEAD][tt]);
                    }
                }
                /* piecewise-constant factor and behaviour (covid.h:432-454) */
                const int sg = lds.seg[i];                          /* wave-uniform */
                double pcf = 1.0;
                if (sg >= 0) pcf = (xdisc_u + (long long)sg * nP)[lane];
                double smoothbehavior = 1.0;
                if (i >= md) {
                    const int bi = min(i - md, M.nB - 1);
                    smoothbehavior = pcf * (B.x_u + (long long)bi * nP)[lane];
                }
                const double totalBehavior = pcf * smoothbehavior;
                /* tomorrow's new infections (covid.h:461-462); asymptomatic == mild buffer */
                const double nSus = M.pop - acc[A_INC][t] - acc[A_MILD][t] - acc[A_MILD][t] -
                                    acc[A_SER][t] - acc[A_REC][t] - acc[A_DEAD][t];
                double nnext = totalBehavior * nSus / M.pop *
                               (lp.betaMild * acc[A_MI][t] + lp.betaHigh * acc[A_HI][t]);
                if (i < s) nnext = (i + 1 == s) ? 1.0 : 0.0;       /* not started yet */
                (B.hn_u + (long long)(i + 1 + MCMC_HOFF(T)) * nP)[lane] = nnext;

                /* log-likelihood term of data day i-md (covid.h:467-492) */
                if (i >= md && i - md < M.nData) {
                    const int di = i - md;
                    double* dslot = B.d1s_u + (long long)di * nP;
                    double dead;
                    if (shift == 1) {
                        dead = w * acc[A_DEAD][t];                 /* multset, covid.h:475 */
                        dslot[lane] = dead;
                    } else {
                        dead = dslot[lane] + w * acc[A_DEAD][t];   /* multadd, covid.h:484 */
                    }
                    const double delta = dead - lds.deaths[di];
                    loglike += w * (-0.5 * delta * delta / lds.sigma2[di]);
                }
                if constexpr (DERIVE) {
                    if (i < M.size) {
                        const double vals[8] = {acc[A_MI][t], acc[A_HI][t], acc[A_INC][t], acc[A_MILD][t],
                                                acc[A_MILD][t], acc[A_SER][t], acc[A_REC][t], acc[A_DEAD][t]};
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            double* o = B.curves_u + ((long long)c * M.size + i) * nP;
                            if (shift == 1) o[lane] = w * vals[c]; else o[lane] += w * vals[c];
                        }
                        (B.curves_u + ((long long)8 * M.size + i) * nP)[lane] = (i >= s) ? totalBehavior : 1.0;
                    }
                }
                ncur = nnext;
            }
        }
    }
    return loglike;
}

/* SmoothnessPrior::eval (mcmc.h:197-214) on behavior = x[0..nB) */
__device__ inline double eval_prior(const DevModel& M, const double* x_u, unsigned lane) {
    const long long nP = M.nP;
    const int n = M.nB;
    const double dx = M.smoothL / (double)n;
    double loglike = 0;
    double fm = x_u[lane], f0 = (x_u + nP)[lane];
#pragma unroll 4
    for (int i = 1; i + 1 < n; ++i) {
        const double fp = (x_u + (long long)(i + 1) * nP)[lane];
        const double dd = fp + fm - 2 * f0;
        loglike -= dd * dd / dx / dx / dx;
        fm = f0;
        f0 = fp;
    }
    loglike *= M.smoothLs * M.smoothLs / M.smoothL;
    return loglike;
}

__device__ inline void load_lane_scalars(const DevModel& M, const double* x_u, unsigned lane, LaneScalars& lp) {
    const long long nP = M.nP;
    lp.betaMild = (x_u + (long long)(M.nB + M.nDisc) * nP)[lane];
    lp.betaHigh = (x_u + (long long)(M.nB + M.nDisc + 1) * nP)[lane];
    lp.delay = (x_u + (long long)(M.nB + M.nDisc + 2) * nP)[lane];
}

/*
 * DiseaseSpread::step up to the eval call (covid.h:499-622): reads the current vector xc,
 * writes the complete proposal vector xp, returns its scalars in lp and whether the
 * behaviour function changed (only then the smoothness prior has to be re-evaluated).
 */
__device__ inline bool propose(const DevModel& M, Pcg32& rnd, const double* xc_u, double* xp_u,
                               unsigned lane, double stepsize, LaneScalars& lp) {
    const long long nP = M.nP;
    load_lane_scalars(M, xc_u, lane, lp);
    double disc[MCMC_MAXDISC];
#pragma unroll
    for (int m = 0; m < MCMC_MAXDISC; ++m)
        disc[m] = (m < M.nDisc) ? (xc_u + (long long)(M.nB + m) * nP)[lane] : 1.0;
    const double sfmin = fmin(stepsize, 1.0);

    const bool big1 = rnd.next_float() < 0.6f;                      /* covid.h:503-507 */
    const bool big2 = rnd.next_float() < 0.6f;
    const bool big3 = rnd.next_float() < 0.6f;
    const bool big4 = rnd.next_float() < 0.6f;
    const bool big5 = rnd.next_float() < 0.6f;

    if (rnd.next_float() < 0.3f) {                                  /* covid.h:510-515 */
        lp.delay += (rnd.next_double() - 0.5) * 0.5 * 0.1 * sfmin;
        if (big1) lp.delay += (rnd.next_double() - 0.5) * sfmin;
        lp.delay = bound_reflect(lp.delay, 5.0, (double)M.md);
    }
    if (rnd.next_float() < 0.3f) {                                  /* covid.h:518-532 */
        if (!big2) lp.betaMild += (rnd.next_double() - 0.5) * 0.1 * sfmin;
        else lp.betaMild += (rnd.next_double() - 0.5) * sfmin;
        lp.betaMild = bound_reflect(lp.betaMild, 0.0, 100.0);
        if (!big3) lp.betaHigh += (rnd.next_double() - 0.5) * 0.1 * sfmin;
        else lp.betaHigh += (rnd.next_double() - 0.5) * sfmin;
        lp.betaHigh = bound_reflect(lp.betaHigh, 0.0, 100.0);
        if (lp.betaHigh < lp.betaMild) {
            const double t = lp.betaHigh; lp.betaHigh = lp.betaMild; lp.betaMild = t;
        }
    }
    if (rnd.next_float() < 0.9f) {                                  /* covid.h:557-598 */
        const bool reverse = rnd.next_double() < 0.5;
        const int nD = M.nDisc;
        /* one visit of entry i, all indices compile-time so disc[] stays in registers */
#define MCMC_VISIT(i)                                                                        \
        if ((i) < nD && !((M.discFixedMask >> (i)) & 1u)) {                                  \
            const float xx = rnd.next_float();                                               \
            if (xx < M.discThr) {                                                            \
                if (!big4) disc[i] += (rnd.next_double() - 0.5) * 0.1 * sfmin;               \
                else disc[i] += (rnd.next_double() - 0.5) * sfmin;                           \
                double lower = 0, upper = 1;                                                 \
                if ((i) >= 1) upper = disc[(i) >= 1 ? (i) - 1 : 0];                          \
                if ((i) < nD - 1) lower = disc[(i) + 1 < MCMC_MAXDISC ? (i) + 1 : 0];        \
                disc[i] = bound_reflect(disc[i], lower, upper);                              \
            }                                                                                \
        }
        if (!reverse) {
            MCMC_VISIT(0) MCMC_VISIT(1) MCMC_VISIT(2) MCMC_VISIT(3)
            MCMC_VISIT(4) MCMC_VISIT(5) MCMC_VISIT(6) MCMC_VISIT(7)
        } else {
            MCMC_VISIT(7) MCMC_VISIT(6) MCMC_VISIT(5) MCMC_VISIT(4)
            MCMC_VISIT(3) MCMC_VISIT(2) MCMC_VISIT(1) MCMC_VISIT(0)
        }
#undef MCMC_VISIT
    }
    bool doB = false;
    double omega = 0, A = 0;
    if (rnd.next_float() < 0.3f) {                                  /* covid.h:601-622 */
        doB = true;
        const double xx = rnd.next_double();
        omega = 2 * 3.1415 / M.dNData * 5 * xx * xx;
        A = sfmin * 0.1 * (rnd.next_double() - 0.5);
        if (big5) A += sfmin * (rnd.next_double() - 0.5);
    }
    /* write the proposal vector */
    if (__any(doB)) {
#pragma unroll 2
        for (int i = 0; i < M.nB; ++i) {
            double v = (xc_u + (long long)i * nP)[lane];
            if (doB) {
                v += A * (mcmc_cos(omega * (double)i) - 1);
                v = bound_reflect(v, 0.0, 2.0);
            }
            (xp_u + (long long)i * nP)[lane] = v;
        }
    } else {
#pragma unroll 4
        for (int i = 0; i < M.nB; ++i) (xp_u + (long long)i * nP)[lane] = (xc_u + (long long)i * nP)[lane];
    }
#pragma unroll
    for (int m = 0; m < MCMC_MAXDISC; ++m)
        if (m < M.nDisc) (xp_u + (long long)(M.nB + m) * nP)[lane] = disc[m];
    (xp_u + (long long)(M.nB + M.nDisc) * nP)[lane] = lp.betaMild;
    (xp_u + (long long)(M.nB + M.nDisc + 1) * nP)[lane] = lp.betaHigh;
    (xp_u + (long long)(M.nB + M.nDisc + 2) * nP)[lane] = lp.delay;
    return doB;
}

/* ------------------------------------------------------------ chain state ---- */
struct ChainState {
    double* params;               /* [dim][nP] current state of every chain             */
    double* prop;                 /* [dim][nP] proposal scratch                         */
    double* ll;                   /* [2][nP]: model, prior                              */
    double* stepsize;             /* [nP]                                               */
    uint64_t* rng;                /* [2][nP]: state, inc                                */
    int* nAccept;                 /* [nP]                                               */
    unsigned long long* pairs;    /* [nP]                                               */
    double* hn; double* he;       /* [histRows][nP]                                     */
    double* d1s;                  /* [nData][nP]                                        */
    double* mean;                 /* [dim][nP] or null                                  */
    double* recLL;                /* [nSavedCap][nP]                                    */
    double* recSamples;           /* [nSavedCap][dim][nP] or null                       */
    long long nChains;
    int dim;
};

enum { SF_RANDOM = 1, SF_RESET_ACC = 2, SF_ADJUST_END = 4, SF_RECORD = 8, SF_MEAN = 16 };

/* mcmc.h:1110-1112 */
__device__ __forceinline__ double rate2corr(double x) {
    return (1 + 0.726484 * x * x * x * x) / (0.82051 + 0.0427315 / (x + 0.0001));
}

#define MCMC_KERNEL_ATTRS __launch_bounds__(MCMC_WAVE) __attribute__((amdgpu_waves_per_eu(2, 2)))

/*
 * n_steps Metropolis-Hastings steps for every chain: Target::step (mcmc.h:750-783) with
 * State::step_random/given_subspace (mcmc.h:494-610), DiseaseSpread::step (covid.h:499-627),
 * SmoothnessPrior::eval (mcmc.h:197-214), State::accept (mcmc.h:653-661) and the per-step
 * bookkeeping of MetropolisChain::run (mcmc.h:1023-1127) driven by `flags`.
 */
template <int T, bool FUSE>
__global__ MCMC_KERNEL_ATTRS void mh_step_kernel(DevModel M, ChainState S, int n_steps,
                                                 const double* __restrict__ betas,
                                                 const int* __restrict__ flags, int slot0, int simPad) {
    extern __shared__ double lds_raw[];
    const LdsView lds = stage_lds(M, lds_raw, simPad);
    const unsigned lane = threadIdx.x;
    const long long wave0 = (long long)blockIdx.x * MCMC_WAVE;      /* first chain of this wave */
    const long long nP = M.nP;
    const long long bufStride = (long long)S.dim * nP;
    const double* xc_u = S.params + wave0;
    double* xcw_u = S.params + wave0;
    double* xp_u = S.prop + wave0;

    Pcg32 rnd;
    rnd.state = (S.rng + wave0)[lane];
    rnd.inc = (S.rng + nP + wave0)[lane];
    double llModel = (S.ll + wave0)[lane], llPrior = (S.ll + nP + wave0)[lane];
    double stepsize = (S.stepsize + wave0)[lane];
    int nAccept = (S.nAccept + wave0)[lane];
    unsigned long long pairs = (S.pairs + wave0)[lane];
    WaveBases B;
    B.x_u = xp_u; B.hn_u = S.hn + wave0; B.he_u = S.he + wave0; B.d1s_u = S.d1s + wave0;
    B.curves_u = nullptr;
    int slot = slot0;

#pragma unroll 1
    for (int g = 0; g < n_steps; ++g) {
        const int fl = flags[g];
        const double beta = betas[g];

        if (fl & SF_RESET_ACC) nAccept = 0;                         /* mcmc.h:1032,1090 */
        if (fl & SF_RECORD) {                                       /* mcmc.h:1037-1062 */
            double l = 0;
            l += llModel;
            if (M.hasPrior) l += llPrior;
            (S.recLL + (long long)slot * nP + wave0)[lane] = 0.0 + beta * l; /* log(weight()==1) = 0 */
            if (S.recSamples) {
                double* dst = S.recSamples + (long long)slot * bufStride + wave0;
                for (int k = 0; k < S.dim; ++k) (dst + (long long)k * nP)[lane] = (xc_u + (long long)k * nP)[lane];
            }
            ++slot;
        }
        if ((fl & SF_MEAN) && S.mean) {                             /* mcmc.h:1064-1065 */
            double* mu = S.mean + wave0;
            for (int k = 0; k < S.dim; ++k) (mu + (long long)k * nP)[lane] += (xc_u + (long long)k * nP)[lane];
        }

        if (fl & SF_RANDOM) {                                       /* mcmc.h:494-500 */
            const int nState = M.hasPrior ? 2 : 1;
            int head;
            do {
                const float f = rnd.next_float() * (float)nState;
                head = min((int)f, nState - 1);
            } while (head != 0);    /* SmoothnessPrior has no steppable coords: redraw */
        }

        LaneScalars lp;
        const bool behChanged = propose(M, rnd, xc_u, xp_u, lane, stepsize, lp);
        const double llm = eval_model<T, FUSE, false>(M, lds, B, lp, lane, pairs);
        double llp = 0.0;
        double delta = 0;                                           /* mcmc.h:604-606 */
        delta += -llModel + llm;
        if (M.hasPrior) {
            llp = llPrior;
            if (__any(behChanged)) {
                const double v = eval_prior(M, xp_u, lane);
                if (behChanged) llp = v;
            }
            delta += -llPrior + llp;
        }
        delta += 0.0;                                               /* log(wNew)-log(wOld), mcmc.h:765 */
        delta *= beta;                                              /* mcmc.h:766 */
        const double u = (double)rnd.next_float();
        const bool accept = u < 1.0 * mcmc_exp(delta);              /* mcmc.h:776 */
        if (__any(accept)) {                                        /* State::accept, mcmc.h:653-661 */
            for (int k = 0; k < S.dim; ++k) {
                const double v = (xp_u + (long long)k * nP)[lane];
                if (accept) (xcw_u + (long long)k * nP)[lane] = v;
            }
        }
        if (accept) {
            llModel = llm;
            llPrior = llp;
            ++nAccept;
        }
        if (fl & SF_ADJUST_END) {                                   /* mcmc.h:1107-1114 */
            const double acceptRate = (double)nAccept / 20;
            stepsize *= rate2corr(acceptRate);
        }
    }

    (S.rng + wave0)[lane] = rnd.state;
    (S.ll + wave0)[lane] = llModel;
    (S.ll + nP + wave0)[lane] = llPrior;
    (S.stepsize + wave0)[lane] = stepsize;
    (S.nAccept + wave0)[lane] = nAccept;
    (S.pairs + wave0)[lane] = pairs;
}

/* State::eval (mcmc.h:385-393) of parameter vectors laid out [dim][nP]: llOut[2][nP] */
template <int T, bool FUSE>
__global__ MCMC_KERNEL_ATTRS void eval_kernel(DevModel M, ChainState S, const double* x, double* llOut,
                                              int simPad) {
    extern __shared__ double lds_raw[];
    const LdsView lds = stage_lds(M, lds_raw, simPad);
    const unsigned lane = threadIdx.x;
    const long long wave0 = (long long)blockIdx.x * MCMC_WAVE;
    const long long nP = M.nP;
    WaveBases B;
    B.x_u = x + wave0; B.hn_u = S.hn + wave0; B.he_u = S.he + wave0; B.d1s_u = S.d1s + wave0;
    B.curves_u = nullptr;
    LaneScalars lp;
    load_lane_scalars(M, B.x_u, lane, lp);
    unsigned long long pairs = 0;
    const double llm = eval_model<T, FUSE, false>(M, lds, B, lp, lane, pairs);
    const double llp = M.hasPrior ? eval_prior(M, B.x_u, lane) : 0.0;
    (llOut + wave0)[lane] = llm;
    (llOut + nP + wave0)[lane] = llp;
}

/* derived output coords (covid.h:467-485) of parameter vectors [dim][nP]:
 * curves[(c*size+i)*nP + chain] */
template <int T, bool FUSE>
__global__ MCMC_KERNEL_ATTRS void derive_kernel(DevModel M, ChainState S, const double* x, double* curves,
                                                int simPad) {
    extern __shared__ double lds_raw[];
    const LdsView lds = stage_lds(M, lds_raw, simPad);
    const unsigned lane = threadIdx.x;
    const long long wave0 = (long long)blockIdx.x * MCMC_WAVE;
    WaveBases B;
    B.x_u = x + wave0; B.hn_u = S.hn + wave0; B.he_u = S.he + wave0; B.d1s_u = S.d1s + wave0;
    B.curves_u = curves + wave0;
    LaneScalars lp;
    load_lane_scalars(M, B.x_u, lane, lp);
    unsigned long long pairs = 0;
    (void)eval_model<T, FUSE, true>(M, lds, B, lp, lane, pairs);
}

/* [n][dim] chain-major (host layout) <-> [dim][nP] SoA */
__global__ void transpose_in_kernel(const double* __restrict__ src, double* __restrict__ dst,
                                    long long n, int dim, long long nP) {
    const long long chain = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (chain >= nP) return;
    const long long srcChain = chain < n ? chain : n - 1;           /* pad lanes copy the last chain */
    for (int k = 0; k < dim; ++k) dst[(long long)k * nP + chain] = src[srcChain * dim + k];
}
__global__ void transpose_out_kernel(const double* __restrict__ src, double* __restrict__ dst,
                                     long long n, int dim, long long nP) {
    const long long chain = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (chain >= n) return;
    for (int k = 0; k < dim; ++k) dst[chain * dim + k] = src[(long long)k * nP + chain];
}
__global__ void scale_kernel(double* __restrict__ a, long long count, double f) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) a[i] *= f;
}

/* cross-chain diagnostics: per-wave partial sums by wavefront shuffle reduction, then one
 * atomic add per wave and quantity.  out[0]=sum nAccept, out[2]=sum ll, out[3]=sum ll^2,
 * out[4+2k]=sum x_k, out[5+2k]=sum x_k^2 */
__global__ __launch_bounds__(MCMC_WAVE) void diagnostics_kernel(DevModel M, ChainState S, double* out) {
    const long long chain = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nP = M.nP;
    const bool live = chain < S.nChains;
    const double* xb = S.params + chain;
    const double acc = live ? (double)S.nAccept[chain] : 0.0;
    double l = 0;
    if (live) { l += S.ll[chain]; if (M.hasPrior) l += S.ll[nP + chain]; }
    const double sAcc = wave_sum_f64(acc), sL = wave_sum_f64(l), sL2 = wave_sum_f64(l * l);
    if (threadIdx.x == 0) {
        atomicAdd(&out[0], sAcc);
        atomicAdd(&out[2], sL);
        atomicAdd(&out[3], sL2);
    }
    for (int k = 0; k < S.dim; ++k) {
        const double x = live ? xb[(long long)k * nP] : 0.0;
        const double sx = wave_sum_f64(x), sx2 = wave_sum_f64(x * x);
        if (threadIdx.x == 0) {
            atomicAdd(&out[4 + 2 * k], sx);
            atomicAdd(&out[5 + 2 * k], sx2);
        }
    }
}
