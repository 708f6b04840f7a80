/*
 * mcmc_kernels.hip.h -- CDNA4 (gfx950) device code of the Metropolis-Hastings hot path.
 *
 * One chain per lane; 64 chains per wavefront advance in lockstep.  All chain state
 * lives in HBM as tile-major structure-of-arrays [tile of 64 chains][field][lane] so that
 * every per-lane access of a wave is one contiguous 512-byte line and everything a wave
 * touches of one array is one contiguous block.  Citations are file:line under
 * /root/reference.
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
 * ascending) in chunks of T lags: per lag ONE history value per lane is loaded
 * (coalesced 8 B/lane, requested two chunks = 16 lags ahead) and feeds 7*T multiply-adds
 * whose second operand R_c[L] is wave-uniform: one row of the response table, fetched by one
 * scalar load two lags ahead of use and read from SGPRs (MCMC_SMEM_TABLE; the workgroup's LDS
 * image of the table serves the sequential part, the e-term rows and the strict / derive
 * paths).  Source days that carry a
 * hospital-capacity correction form a wave-uniform day range; only the chunks reading
 * that range take the variant with the extra e-terms.  Phase B then runs the T days of
 * the block in order (the feedback n_{i+1} = f(buffers[i]) is inherently sequential,
 * covid.h:461-462), pushing each new n_i into the later days of the block from registers.
 * The per-day likelihood terms are summed in a streaming pass after the day loop.
 *
 * A wave runs alone on its SIMD (64k chains = 1024 waves = one per SIMD of the chip), so
 * nothing hides a stall for it: every load in the lag loop is software-pipelined, and the
 * code is shaped so that the compiler keeps the accumulators and both register windows in
 * place across the loop (no copies, no spills inside it; tools/asm_blocks.py audits that).
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
#define MCMC_WAVES_PER_BLOCK 4     /* one wave per SIMD of a CU share one LDS image */

typedef double double8_t __attribute__((ext_vector_type(8)));
typedef double double4_t __attribute__((ext_vector_type(4)));

/* accumulator families */
enum { A_MI = 0, A_HI, A_INC, A_MILD, A_SER, A_REC, A_DEAD, A_NPOS };

struct DevModel {
    int nData, md, size, nB, nDisc, simDays, nLagRows, hasPrior;
    int tabRows, simPad;
    double pop, probICU, probLethal, smoothLs, smoothL, dNData;
    float discThr;                /* 2/nNotFixed, float division done on the host    */
    unsigned discFixedMask;
    const double* Rpos;           /* [tabRows][8]: MI HI INC MILD SER REC DEAD 0; rows >= nLagRows zero */
    const double* Rneg;           /* [tabRows][4]: SFS RFS DFS 1; rows >= nLagRows zero */
    const double* capacity;       /* [simPad]                                        */
    const double* deaths;         /* [nData]                                         */
    const double* sigma2;         /* [nData]  sigma[i]*sigma[i] (covid.h:491)        */
    const int* seg;               /* [simPad] piecewise-constant segment of day i, -1 = none */
    const float* RF;              /* fast mode only (mcmc_model_f32.hip.h): [rowsF][8] floats, or null */
    const float* RFn;             /* fast mode only: [rowsF][4] floats                                  */
    /* multi-region model (extension, see "Multi-region" below): R region blocks per chain */
    int nRegions, dimRegion;      /* 1 and nB+nDisc+3 for the reference's single-region State            */
    const double* regDeaths;      /* [R][nData]                                                          */
    const double* regSigma2;      /* [R][nData]                                                          */
    const double* regPop;         /* [R] population, cap0, cap_incr_rate of every region                 */
    const double* regCap0;
    const double* regCapRate;
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

/* ------------------------------------------------ optional in-kernel phase stamps ---- */
/* Built with -DMCMC_STAMPS the kernels accumulate, per phase of an MH step, the shader
 * cycles (s_memtime) every wave spends there into g_phase_cycles; mcmc_hip_debug_phase_cycles
 * reads them.  Without the macro the hooks compile to nothing. */
enum { PH_PROPOSE = 0, PH_SETUP, PH_CHUNKS, PH_TAIL, PH_DAYS, PH_LOGLIKE, PH_PRIOR, PH_ACCEPT, PH_RECORD, PH_CHUNKS_E, PH_NCHUNK, PH_NCHUNK_E, PH_WAIT, PH_EDAYS, PH_EHULL, PH_COUNT };
__device__ unsigned long long g_phase_cycles[16];
#define MCMC_WAVE_LOG 8192            /* tiles the -DMCMC_STAMPS per-wave log holds */
__device__ unsigned long long g_wave_log[4 * MCMC_WAVE_LOG];   /* per tile: launch cycles, (XCC_ID << 32) | HW_ID, e-chunks, e-days << 32 | hull days */
#ifdef MCMC_STAMPS
struct PhaseClock {
    unsigned long long t[PH_COUNT];
    unsigned long long last;
    __device__ __forceinline__ void start() {
#pragma unroll
        for (int k = 0; k < PH_COUNT; ++k) t[k] = 0;
        last = __builtin_amdgcn_s_memtime();
    }
    __device__ __forceinline__ void stamp(int k) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        t[k] += now - last;
        last = now;
    }
    __device__ __forceinline__ void count(int k, int n) { t[k] += (unsigned long long)n; }
    __device__ __forceinline__ void flush(unsigned lane, int tile) {
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < PH_COUNT; ++k) atomicAdd(&g_phase_cycles[k], t[k]);
            /* slot 15: the slowest wave's cycles over its launch (load imbalance = this / mean) */
            unsigned long long tot = 0;
#pragma unroll
            for (int k = 0; k < PH_COUNT; ++k)
                if (k != PH_NCHUNK && k != PH_NCHUNK_E && k != PH_EDAYS && k != PH_EHULL) tot += t[k];
            atomicMax(&g_phase_cycles[15], tot);
            if (tile < MCMC_WAVE_LOG) {
                const unsigned long long hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
                g_wave_log[4 * tile] = tot;
                g_wave_log[4 * tile + 1] = (xcc << 32) | hw;
                g_wave_log[4 * tile + 2] = t[PH_NCHUNK_E];
                g_wave_log[4 * tile + 3] = (t[PH_EDAYS] << 32) | t[PH_EHULL];
            }
        }
    }
};
#else
struct PhaseClock {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void stamp(int) {}
    __device__ __forceinline__ void count(int, int) {}
    __device__ __forceinline__ void flush(unsigned, int) {}
};
#endif

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
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

/*
 * LDS image, staged ONCE per workgroup and shared by its four waves: the response table
 * (what the phase-A inner loop multiplies with) and the observed series / per-day
 * constants.  The table rows are wave-uniform reads: every lane of a wave reads the same
 * address, which the LDS serves as a broadcast.
 */
struct LdsView {
    const double* Rpos;       /* [ldsRows][8] */
    const double* Rneg;       /* [ldsRows][4] */
    const double* capacity;   /* [simPad] */
    const double* deaths;     /* [nData]  */
    const double* sigma2;     /* [nData]  */
    const int* seg;           /* [simPad] */
    /* fast mode on the matrix cores only (stage_lds_f32, mcmc_model_f32.hip.h); null otherwise */
    const float* tabF;        /* [F32_TAB_PAD + rowsL + F32_TAB_SLACK][7] float response table, pad rows of zeros in front */
    const float* tabN;        /* [same rows][3] capacity-correction table (-SFS -RFS 1-DFS) */
    const float* rowsB;       /* [MCMC_TM][8]: rows 0..MCMC_TM-1 of the response table for the sequential part */
    const float* rowsNB;      /* [MCMC_TM][4]: ... of the correction table */
    float* tr;                /* [waves of the workgroup][64][F32_TR_ROW] transposition buffers */
};

__host__ __device__ inline int lds_rows(const DevModel& M, bool derive, int T) {
    /* lags reach horizon+T-2; rows >= nLagRows are zero and a chunk made only of them is
     * skipped, so nothing above row nLagRows+T-2 is ever read */
    const int h = derive ? M.size : M.simDays;
    const int a = h + 2 * T, b = M.nLagRows + T;
    const int r = a < b ? a : b;
    return r < M.tabRows ? r : M.tabRows;
}
__host__ __device__ inline size_t lds_bytes(const DevModel& M, bool derive, int T) {
    return sizeof(double) * ((size_t)12 * lds_rows(M, derive, T) + (size_t)M.simPad + 2 * (size_t)M.nData) +
           sizeof(int) * (size_t)M.simPad + 16;
}

__device__ inline LdsView stage_lds(const DevModel& M, double* lds, bool derive, int T) {
    const int rows = lds_rows(M, derive, T);
    double* rp = lds;
    double* rn = rp + (size_t)8 * rows;
    double* cap = rn + (size_t)4 * rows;
    double* dea = cap + M.simPad;
    double* sg2 = dea + M.nData;
    int* seg = (int*)(sg2 + M.nData);
    for (int i = threadIdx.x; i < 8 * rows; i += blockDim.x) rp[i] = M.Rpos[i];
    for (int i = threadIdx.x; i < 4 * rows; i += blockDim.x) rn[i] = M.Rneg[i];
    for (int i = threadIdx.x; i < M.simPad; i += blockDim.x) {
        cap[i] = M.capacity[i];
        seg[i] = M.seg[i];
    }
    for (int i = threadIdx.x; i < M.nData; i += blockDim.x) {
        dea[i] = M.deaths[i];
        sg2[i] = M.sigma2[i];
    }
    __syncthreads();
    LdsView v;
    v.Rpos = rp; v.Rneg = rn; v.capacity = cap; v.deaths = dea; v.sigma2 = sg2; v.seg = seg;
    v.tabF = nullptr; v.tabN = nullptr; v.rowsB = nullptr; v.rowsNB = nullptr; v.tr = nullptr;
    return v;
}

/* LDS image of the fast mode on the matrix cores: no double tables, the per-day constants as above, the two
 * float tables the MFMA B operands are read from -- SEVEN floats per lag (no pad column: a block of MCMC_TM = 9
 * days x 7 curves fills 63 of the 64 MFMA columns) and three per lag for the corrections --, rows 0..MCMC_TM-1 of
 * both once more in 16-byte aligned rows for the sequential part, and one 64 x F32_TR_ROW float transposition
 * buffer per wave (one 32-column MFMA tile passes through it at a time).  Lags reach horizon + MCMC_TM (last day of the last block against the first source day, one
 * more for the pair alignment); sources all of whose lags are >= nLagRows are skipped in whole blocks, so no row
 * above nLagRows + 2 MCMC_TM + 2 is read either. */
#ifndef MCMC_TM
#define MCMC_TM 9             /* days per block of the MFMA form */
#endif
#define F32_NC 7              /* floats per lag of the MFMA response table */
#define F32_NE 3              /* ... of the correction table */
#define F32_TAB_PAD 2         /* the B operand of the pair after the last one is requested too: rows -1 and -2 */
#define F32_TAB_SLACK 2       /* the unused 64th column / lanes 27..31 of the correction tile read past the last row */
#define F32_DSPLIT (32 / F32_NC)   /* the day whose curves straddle the two 32-column MFMA tiles */
#define F32_TR_JUNK 40        /* floats per chain of the hand-over buffer that carry data: 5 days x 8, one column tile at a time */
#define F32_TR_ROW 44         /* ... and its row pitch (slot F32_TR_JUNK takes the unused columns' values) */
#define F32_TR_FLOATS (MCMC_WAVE * F32_TR_ROW)
static_assert(MCMC_TM * F32_NC <= 64 && MCMC_TM * F32_NE <= 32 && (F32_DSPLIT + 1) * 8 <= F32_TR_JUNK &&
              (MCMC_TM - F32_DSPLIT) * 8 <= F32_TR_JUNK && F32_TR_JUNK < F32_TR_ROW, "block does not fit the MFMA tiles");
__host__ __device__ inline int lds_rows_f32(const DevModel& M) {
    const int a = M.simDays + MCMC_TM + 3, b = M.nLagRows + 2 * MCMC_TM + 6;
    return a < b ? a : b;
}
struct F32Image { size_t nd, ni, ntab, nrowsB, total; };      /* element counts of the parts, bytes in total */
__host__ __device__ inline F32Image f32_image(const DevModel& M) {
    F32Image I;
    I.nd = ((size_t)M.simPad + 2 * (size_t)M.nData + 1) & ~(size_t)1;            /* doubles, 16-byte multiple */
    I.ni = ((size_t)M.simPad + 3) & ~(size_t)3;                                   /* ints */
    I.ntab = ((size_t)(F32_TAB_PAD + lds_rows_f32(M) + F32_TAB_SLACK) * (F32_NC + F32_NE) + 3) & ~(size_t)3;   /* floats */
    I.nrowsB = (size_t)MCMC_TM * 12;                                               /* floats */
    I.total = sizeof(double) * I.nd + sizeof(int) * I.ni +
              sizeof(float) * (I.ntab + I.nrowsB + (size_t)MCMC_WAVES_PER_BLOCK * F32_TR_FLOATS) + 16;
    return I;
}
__host__ __device__ inline size_t lds_bytes_f32(const DevModel& M) { return f32_image(M).total; }
__device__ inline LdsView stage_lds_f32(const DevModel& M, double* lds) {
    const F32Image I = f32_image(M);
    const int rowsAll = F32_TAB_PAD + lds_rows_f32(M) + F32_TAB_SLACK;
    double* cap = lds;
    double* dea = cap + M.simPad;
    double* sg2 = dea + M.nData;
    int* seg = (int*)(lds + I.nd);
    float* tf = (float*)(seg + I.ni);
    float* tn = tf + (size_t)rowsAll * F32_NC;
    float* rb = tf + I.ntab;
    float* rnb = rb + (size_t)MCMC_TM * 8;
    float* tr = rb + I.nrowsB;
    for (int i = threadIdx.x; i < M.simPad; i += blockDim.x) {
        cap[i] = M.capacity[i];
        seg[i] = M.seg[i];
    }
    for (int i = threadIdx.x; i < M.nData; i += blockDim.x) {
        dea[i] = M.deaths[i];
        sg2[i] = 1.0 / M.sigma2[i];     /* this image holds the RECIPROCAL: the fast mode multiplies (eval_model_f32) */
    }
    /* the global float tables have 8 / 4 floats per lag and zeros from nLagRows on (mcmc_hip_set_tables) */
    for (int i = threadIdx.x; i < rowsAll * F32_NC; i += blockDim.x) {
        const int r = i / F32_NC - F32_TAB_PAD, c = i % F32_NC;
        tf[i] = r >= 0 ? M.RF[r * 8 + c] : 0.0f;
    }
    for (int i = threadIdx.x; i < rowsAll * F32_NE; i += blockDim.x) {
        const int r = i / F32_NE - F32_TAB_PAD, c = i % F32_NE;
        tn[i] = r >= 0 ? M.RFn[r * 4 + c] : 0.0f;
    }
    for (int i = threadIdx.x; i < MCMC_TM * 8; i += blockDim.x) rb[i] = M.RF[i];
    for (int i = threadIdx.x; i < MCMC_TM * 4; i += blockDim.x) rnb[i] = M.RFn[i];
    __syncthreads();
    LdsView v;
    v.Rpos = nullptr; v.Rneg = nullptr; v.capacity = cap; v.deaths = dea; v.sigma2 = sg2; v.seg = seg;
    v.tabF = tf; v.tabN = tn; v.rowsB = rb; v.rowsNB = rnb; v.tr = tr;
    return v;
}
/* PREC: 0 = double (the reference's arithmetic), 1 = fast mode with packed FP32 multiply-adds, 2 = fast mode
 * with the convolution on the matrix cores */
template <int PREC>
__device__ inline LdsView stage_lds_for(const DevModel& M, double* lds, int T) {
    if constexpr (PREC == 2) return stage_lds_f32(M, lds);
    else return stage_lds(M, lds, false, T);
}

/*
 * Addressing convention.  Chains are grouped in TILES of 64 (one wave).  Every per-chain
 * array is tile-major: element (row, chain) of an array with `rows` rows lives at
 *     base[((chain/64) * rows + row) * 64 + chain%64]
 * so that (a) each access of a wave is one contiguous 512-byte line and (b) everything one
 * wave ever touches of an array is ONE contiguous block (a 405-day infection history is
 * 207 KB per wave instead of 405 lines 512 KB apart), which keeps it inside a handful of
 * pages for the TLBs.  All `*_u` pointers below are wave-uniform and already advanced to
 * the wave's tile; the lane part is the single 32-bit index `lane`.
 */
struct WaveBases {
    const double* x_u;     /* parameter vector being evaluated, element k at x_u[k*64 + lane]  */
    double* hn_u;          /* infection history,  day k at hn_u[(k+HOFF)*64 + lane]             */
    double* he_u;          /* extra-deaths history, same indexing                               */
    double* d1s_u;         /* weighted deaths of data day i: first pass at d1s_u[i*64 + lane],
                              second pass at d1s_u[(nData+i)*64 + lane]                         */
    double* curves_u;      /* DERIVE only: curve c, day i at curves_u[(c*size+i)*64 + lane]     */
};

/* the scalars of a parameter vector that the day loop needs all the time */
struct LaneScalars {
    double betaMild, betaHigh, delay;
};

__device__ __forceinline__ double8_t lds_row8(const double* Rpos, int L) {
    return *(const double8_t*)(Rpos + (size_t)L * 8);
}
#ifndef MCMC_SMEM_TABLE
#define MCMC_SMEM_TABLE 1
#endif
#ifndef MCMC_SMEM_LEAD2
#define MCMC_SMEM_LEAD2 1      /* the row of lag step u+2 is requested during step u (+2 % over one step of lead) */
#endif
#ifndef MCMC_SMEM_LEAD2_E
#define MCMC_SMEM_LEAD2_E 0
#endif
/* The lag loops take their table rows through the SCALAR cache: one s_load_dwordx16 per lag
 * step from the global copy of the table (rows are wave-uniform), and the 56 multiply-adds of
 * the step read the row straight from SGPRs.  Against four LDS broadcast reads into 16 VGPRs
 * per step this saves ~5 % of the loop's instructions and 32 VGPRs, measured +4 % on the
 * whole benchmark.  (Reads only; nothing is ever written through the scalar cache.)
 * MCMC_SMEM_TABLE=0 keeps the LDS rows.  Rows 0..T-1 of the sequential part, the e-term rows
 * and the strict / derive paths use the LDS image. */
#if MCMC_SMEM_TABLE
#define MCMC_TAB(lds, M) (M).Rpos
#else
#define MCMC_TAB(lds, M) (lds).Rpos
#endif
typedef const __attribute__((address_space(4))) double8_t* const_row_ptr;
__device__ __forceinline__ double8_t smem_row8(const double* RposGlobal, int L) {
    return *(const_row_ptr)(RposGlobal + (size_t)L * 8);
}
__device__ __forceinline__ double4_t lds_row4(const double* Rneg, int L) {
    return *(const double4_t*)(Rneg + (size_t)L * 4);
}

/* element (row, lane) of a wave's tile: uniform base + 32-bit lane offset, the form the
 * hardware addresses as SGPR base + VGPR offset + immediate (no 64-bit vector add) */
__device__ __forceinline__ double ld_row(const double* base_u, int row, unsigned lane) {
    const char* b = (const char*)(base_u + (size_t)row * MCMC_WAVE);
    return *(const double*)(b + ((lane & (MCMC_WAVE - 1)) << 3));
}

/* ------------------------------------------------------- phase A building blocks ---- */
/* One chunk = T consecutive lags Lc, Lc-1, ..., Lc-T+1 applied to the T days of the block.
 * History window: element q in [0, 2T-1) is W[q] for q < T and N[q-T] otherwise, holding
 * n[kbase + q].  After step u the slot W[u] is dead and is refilled with the history value
 * 2T further on (it becomes the `N` of the chunk after next: ping-pong, no register moves).
 * `r` carries the table row of the chunk's first lag in and that of the next chunk out, so
 * the read of a row (scalar load, or LDS with MCMC_SMEM_TABLE=0) is always issued one full
 * step (56 multiply-adds) ahead of use. */
template <int T, bool FUSE>
__device__ __forceinline__ void lag_chunk(double (&acc)[A_NPOS][T], double (&W)[T], const double (&N)[T],
                                          double8_t& r, const double* Rpos, int Lc,
                                          const double* hn_next_u, unsigned lane) {
#if MCMC_SMEM_TABLE
    const double* rowBase = Rpos + (long long)(Lc - T) * 8;      /* rows Lc-T .. Lc-1, one address per chunk */
#if MCMC_SMEM_LEAD2
    double8_t r1 = smem_row8(rowBase, T - 1);
#endif
#endif
#pragma unroll
    for (int u = 0; u < T; ++u) {
#if MCMC_SMEM_TABLE && MCMC_SMEM_LEAD2
        const double8_t rnext = r1;
        r1 = smem_row8(rowBase, T - 2 - u);                          /* row Lc-u-2: two steps ahead */
#elif MCMC_SMEM_TABLE
        const double8_t rnext = smem_row8(rowBase, T - 1 - u);
#else
        const double8_t rnext = lds_row8(Rpos, Lc - u - 1);
#endif
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
        }
        /* slot u of W is dead from here on: prefetch for the chunk after next */
        W[u] = ld_row(hn_next_u, u, lane);
        r = rnext;
#if MCMC_SMEM_TABLE
        __builtin_amdgcn_sched_barrier(0);      /* keep every step's prefetch in its step */
#endif
    }
}

/* The same chunk when some source day of it carries a hospital-capacity correction e_k
 * (covid.h:416-429): after the eight projections of day k come add(e_k) and the three
 * negative projections, in that order.  The e-history has a register window of its own
 * (EW/EN, refilled like W/N) and the row of the three *FromSerious curves is read one step
 * ahead like r. */
template <int T, bool FUSE>
__device__ __forceinline__ void lag_chunk_e(double (&acc)[A_NPOS][T], double (&W)[T], const double (&N)[T],
                                            double (&EW)[T], const double (&EN)[T], double8_t& r, double4_t& rn,
                                            const double* Rpos, const double* Rneg, int Lc,
                                            const double* hn_next_u, const double* he_next_u, unsigned lane) {
#if MCMC_SMEM_TABLE
    const double* rowBase = Rpos + (long long)(Lc - T) * 8;
#if MCMC_SMEM_LEAD2_E
    double8_t r1 = smem_row8(rowBase, T - 1);
#endif
#endif
#pragma unroll
    for (int u = 0; u < T; ++u) {
#if MCMC_SMEM_TABLE && MCMC_SMEM_LEAD2_E
        const double8_t rnext = r1;
        r1 = smem_row8(rowBase, T - 2 - u);
#elif MCMC_SMEM_TABLE
        const double8_t rnext = smem_row8(rowBase, T - 1 - u);
#else
        const double8_t rnext = lds_row8(Rpos, Lc - u - 1);
#endif
        const double4_t rnnext = lds_row4(Rneg, Lc - u - 1);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int q = t + u;
            const double nk = (q < T) ? W[q < T ? q : 0] : N[q < T ? 0 : q - T];
            const double ek = (q < T) ? EW[q < T ? q : 0] : EN[q < T ? 0 : q - T];
            acc[A_MI][t] = madd<FUSE>(nk, r[0], acc[A_MI][t]);
            acc[A_HI][t] = madd<FUSE>(nk, r[1], acc[A_HI][t]);
            acc[A_INC][t] = madd<FUSE>(nk, r[2], acc[A_INC][t]);
            acc[A_MILD][t] = madd<FUSE>(nk, r[3], acc[A_MILD][t]);
            acc[A_SER][t] = madd<FUSE>(nk, r[4], acc[A_SER][t]);
            acc[A_REC][t] = madd<FUSE>(nk, r[5], acc[A_REC][t]);
            acc[A_DEAD][t] = madd<FUSE>(nk, r[6], acc[A_DEAD][t]);
            const double nek = -ek;
            /* add(), covid.h:418 -- like project() it stops after traj.nDays days (covid.h:388):
             * rn[3] is 1 for the lags below nLagRows and 0 above; fma(ek, 1, acc) IS acc + ek
             * (the product is exact, one rounding), so this costs nothing and cuts the flat
             * add off exactly where the reference does */
            acc[A_DEAD][t] = __builtin_fma(ek, rn[3], acc[A_DEAD][t]);
            acc[A_SER][t] = madd<FUSE>(nek, rn[0], acc[A_SER][t]);      /* covid.h:426 */
            acc[A_REC][t] = madd<FUSE>(nek, rn[1], acc[A_REC][t]);      /* covid.h:427 */
            acc[A_DEAD][t] = madd<FUSE>(nek, rn[2], acc[A_DEAD][t]);    /* covid.h:428 */
        }
        W[u] = ld_row(hn_next_u, u, lane);
        EW[u] = ld_row(he_next_u, u, lane);
        r = rnext;
        rn = rnnext;
        /* keep every lag step's prefetch where it is: left alone, the scheduler sometimes
         * gathers the loads at the end of the unrolled body, right before their use */
#if MCMC_E_SCHEDBAR
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
}


/* ---------------------------------------------- rows 0..T-1 across the lanes (FUSE) ---- */
/* The sequential part of a block (phase B and the tail) multiplies with the table rows of
 * the lags 0..T-1 only.  With FMA contraction those 8 rows x 7 curves live ACROSS THE LANES
 * of four VGPR pairs instead of being re-read from LDS (or parked in AccVGPRs) for every
 * multiply-add: lane 8*h + c of each 16-lane row of pair g holds R[2g+h][c], and
 * v_fmac_f64_dpp with row_newbcast broadcasts that lane as the multiplicand -- one
 * instruction per multiply-add and the same fused operation as v_fmac_f64, bit for bit.
 * The leading s_nop covers the two wait states a DPP source needs after a VALU write
 * (the compiler may have just restored the pair from an AccVGPR; it cannot see into the
 * asm).  All 64 lanes are active wherever these are used. */
#ifndef MCMC_LL_BATCH
#define MCMC_LL_BATCH 16     /* rows per memory round trip in the likelihood pass, ...  */
#endif
#ifndef MCMC_PROP_BATCH
#define MCMC_PROP_BATCH 16   /* ... the proposal loop ...                              */
#endif
#ifndef MCMC_ACC_BATCH
#define MCMC_ACC_BATCH 32    /* ... and the accept copy (a lone wave has 512 registers) */
#endif
#ifndef MCMC_E_DRAIN
#define MCMC_E_DRAIN 1
#endif
#ifndef MCMC_LANE_LAUNDER
#define MCMC_LANE_LAUNDER 1
#endif
#ifndef MCMC_E_SCHEDBAR
#define MCMC_E_SCHEDBAR 1
#endif
#ifndef MCMC_PLAIN_DRAIN
#define MCMC_PLAIN_DRAIN 1
#endif
#ifndef MCMC_DPP_ROWS
#define MCMC_DPP_ROWS 1
#endif
/* which block sizes take their rows 0..T-1 across the lanes: even T (a register pair holds two rows) */
#define MCMC_ROWS_IN_LANES(T, FUSE) ((FUSE) && ((T) % 2 == 0) && (T) >= 8 && MCMC_DPP_ROWS)   /* the derive kernel (T = 4) keeps the LDS rows: a cross-check of the two forms */
template <int T>
struct RowRegs {
    double pos[(T + 1) / 2];      /* rows 0..T-1 of Rpos, two rows per pair                          */
    double neg[(T + 3) / 4];      /* rows 0..T-1 of Rneg, four rows per pair: lane 4*(d&3)+c          */
};
template <int T>
__device__ __forceinline__ RowRegs<T> load_row_regs(const LdsView& lds, unsigned lane) {
    RowRegs<T> R;
#pragma unroll
    for (int g = 0; g < (T + 1) / 2; ++g) R.pos[g] = lds.Rpos[(2 * g + ((lane >> 3) & 1)) * 8 + (lane & 7)];
#pragma unroll
    for (int h = 0; h < (T + 3) / 4; ++h) R.neg[h] = lds.Rneg[(4 * h + ((lane >> 2) & 3)) * 4 + (lane & 3)];
    return R;
}
#define MCMC_DPP_FMAC(D, LANE) "v_fmac_f64_dpp " D ", %[row], %[nk] row_newbcast:" #LANE " row_mask:0xf bank_mask:0xf\n\t"
/* acc[c][tt] += R[row][c] * nk for the seven curves; HALF = row & 1 */
template <int T, int HALF>
__device__ __forceinline__ void dpp_row7(double (&acc)[A_NPOS][T], int tt, double rowv, double nk) {
    if constexpr (HALF == 0)
        asm("s_nop 1\n\t" MCMC_DPP_FMAC("%0", 0) MCMC_DPP_FMAC("%1", 1) MCMC_DPP_FMAC("%2", 2) MCMC_DPP_FMAC("%3", 3)
            MCMC_DPP_FMAC("%4", 4) MCMC_DPP_FMAC("%5", 5) MCMC_DPP_FMAC("%6", 6)
            : "+v"(acc[0][tt]), "+v"(acc[1][tt]), "+v"(acc[2][tt]), "+v"(acc[3][tt]), "+v"(acc[4][tt]),
              "+v"(acc[5][tt]), "+v"(acc[6][tt])
            : [row] "v"(rowv), [nk] "v"(nk));
    else
        asm("s_nop 1\n\t" MCMC_DPP_FMAC("%0", 8) MCMC_DPP_FMAC("%1", 9) MCMC_DPP_FMAC("%2", 10) MCMC_DPP_FMAC("%3", 11)
            MCMC_DPP_FMAC("%4", 12) MCMC_DPP_FMAC("%5", 13) MCMC_DPP_FMAC("%6", 14)
            : "+v"(acc[0][tt]), "+v"(acc[1][tt]), "+v"(acc[2][tt]), "+v"(acc[3][tt]), "+v"(acc[4][tt]),
              "+v"(acc[5][tt]), "+v"(acc[6][tt])
            : [row] "v"(rowv), [nk] "v"(nk));
}
/* the two rows of a pair applied to the destination days tt (even row) and tt+1 (odd row) */
template <int T>
__device__ __forceinline__ void dpp_row14(double (&acc)[A_NPOS][T], int tt, double rowv, double nk) {
    asm("s_nop 1\n\t" MCMC_DPP_FMAC("%0", 0) MCMC_DPP_FMAC("%1", 1) MCMC_DPP_FMAC("%2", 2) MCMC_DPP_FMAC("%3", 3)
        MCMC_DPP_FMAC("%4", 4) MCMC_DPP_FMAC("%5", 5) MCMC_DPP_FMAC("%6", 6)
        MCMC_DPP_FMAC("%7", 8) MCMC_DPP_FMAC("%8", 9) MCMC_DPP_FMAC("%9", 10) MCMC_DPP_FMAC("%10", 11)
        MCMC_DPP_FMAC("%11", 12) MCMC_DPP_FMAC("%12", 13) MCMC_DPP_FMAC("%13", 14)
        : "+v"(acc[0][tt]), "+v"(acc[1][tt]), "+v"(acc[2][tt]), "+v"(acc[3][tt]), "+v"(acc[4][tt]),
          "+v"(acc[5][tt]), "+v"(acc[6][tt]),
          "+v"(acc[0][tt + 1]), "+v"(acc[1][tt + 1]), "+v"(acc[2][tt + 1]), "+v"(acc[3][tt + 1]), "+v"(acc[4][tt + 1]),
          "+v"(acc[5][tt + 1]), "+v"(acc[6][tt + 1])
        : [row] "v"(rowv), [nk] "v"(nk));
}
/* the capacity correction of one source day on destination day tt at lag d (covid.h:418,
 * 426-428): dead += e, then the three negative projections with nek = -e; SLOT = d & 3 */
template <int T, int SLOT>
__device__ __forceinline__ void dpp_neg3(double (&acc)[A_NPOS][T], int tt, double negv, double ek, double nek) {
    acc[A_DEAD][tt] = acc[A_DEAD][tt] + ek;
#define MCMC_DPP_NEG(L0, L1, L2)                                                                                   \
    asm("s_nop 1\n\t"                                                                                              \
        "v_fmac_f64_dpp %0, %[row], %[nk] row_newbcast:" #L0 " row_mask:0xf bank_mask:0xf\n\t"                      \
        "v_fmac_f64_dpp %1, %[row], %[nk] row_newbcast:" #L1 " row_mask:0xf bank_mask:0xf\n\t"                      \
        "v_fmac_f64_dpp %2, %[row], %[nk] row_newbcast:" #L2 " row_mask:0xf bank_mask:0xf\n\t"                      \
        : "+v"(acc[A_SER][tt]), "+v"(acc[A_REC][tt]), "+v"(acc[A_DEAD][tt])                                         \
        : [row] "v"(negv), [nk] "v"(nek))
    if constexpr (SLOT == 0) MCMC_DPP_NEG(0, 1, 2);
    else if constexpr (SLOT == 1) MCMC_DPP_NEG(4, 5, 6);
    else if constexpr (SLOT == 2) MCMC_DPP_NEG(8, 9, 10);
    else MCMC_DPP_NEG(12, 13, 14);
#undef MCMC_DPP_NEG
}
template <int T>
__device__ __forceinline__ void dpp_neg3_at(double (&acc)[A_NPOS][T], int tt, int d, const RowRegs<T>& RR, double ek, double nek) {
    const double negv = RR.neg[d >> 2];
    switch (d & 3) {
        case 0: dpp_neg3<T, 0>(acc, tt, negv, ek, nek); break;
        case 1: dpp_neg3<T, 1>(acc, tt, negv, ek, nek); break;
        case 2: dpp_neg3<T, 2>(acc, tt, negv, ek, nek); break;
        default: dpp_neg3<T, 3>(acc, tt, negv, ek, nek); break;
    }
}

/* the lags T-1 .. 1: only destinations t < L have their source day before the block */
template <int T, bool FUSE, bool WITH_E>
__device__ __forceinline__ void lag_tail(double (&acc)[A_NPOS][T], const double (&W)[T],
                                         const double* Rpos, const double* Rneg, const double* he_k0_u,
                                         unsigned lane, const RowRegs<T>& RR) {
    if constexpr (MCMC_ROWS_IN_LANES(T, FUSE)) {
        double ek[WITH_E ? T - 1 : 1];
        if constexpr (WITH_E) {
#pragma unroll
            for (int q = 0; q < T - 1; ++q) ek[q] = he_k0_u[q * MCMC_WAVE + lane];
        }
#pragma unroll
        for (int v = 1; v < T; ++v) {
            const int L = T - v;
#pragma unroll
            for (int t = 0; t < L; ++t) {
                const double nk = W[t + v - 1];
                if (L & 1) dpp_row7<T, 1>(acc, t, RR.pos[L >> 1], nk);
                else dpp_row7<T, 0>(acc, t, RR.pos[L >> 1], nk);
                if constexpr (WITH_E) dpp_neg3_at<T>(acc, t, L, RR, ek[t + v - 1], -ek[t + v - 1]);
            }
        }
        return;
    }
#pragma unroll
    for (int v = 1; v < T; ++v) {
        const int L = T - v;
        const double8_t r = lds_row8(Rpos, L);
        double4_t rn;
        if constexpr (WITH_E) rn = lds_row4(Rneg, L);
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
                const double ek = he_k0_u[(t + v - 1) * MCMC_WAVE + lane];
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

/* Work accounting of one evaluation, per chain.  `algo` = the (day, lag) pairs the reference
 * performs over its full horizon `size` (the SURVEY 8d work unit: 8 compartments, 16 flop per
 * pair); `exec` = the pairs this kernel's sampling path executes for the chain (days up to
 * maxDelay+nData only, 7 accumulator families = 14 flop per pair).  Both follow project()'s
 * d < nDays cut-off (covid.h:378): a source day reaches at most nLag later days. */
struct WorkCount {
    unsigned long long algo, exec;
};
__device__ __forceinline__ long long tri_capped(long long m, long long cap) {
    return m <= cap ? m * (m + 1) / 2 : cap * (cap + 1) / 2 + (m - cap) * cap;
}
__device__ __forceinline__ void count_work(const DevModel& M, int start, WorkCount& work) {
    const long long cap = M.nLagRows;
    const long long a1 = M.size - min(start + 1, M.size), a0 = M.size - min(start, M.size);
    const long long e1 = M.simDays - min(start + 1, M.simDays), e0 = M.simDays - min(start, M.simDays);
    work.algo += (unsigned long long)(tri_capped(a1, cap) + tri_capped(a0, cap));
    work.exec += (unsigned long long)(tri_capped(e1, cap) + tri_capped(e0, cap));
}

/*
 * DiseaseSpread::eval (covid.h:325-496) for the chain of this lane.  Returns loglike and adds
 * the evaluation's work to `work`.
 */
template <int T, bool FUSE, bool DERIVE>
__device__ __forceinline__ double eval_model(const DevModel& M, const LdsView& lds, const WaveBases& B,
                                             const LaneScalars& lp, unsigned lane, WorkCount& work,
                                             PhaseClock& pc) {
    const int md = M.md;
    const int horizon = DERIVE ? M.size : M.simDays;
    const double* xdisc_u = B.x_u + M.nB * MCMC_WAVE;

    double loglike = 0;                                             /* covid.h:329 */
    int start = (int)((double)md - lp.delay);                       /* covid.h:331 */
    const double frac = (double)md - lp.delay - (double)start;      /* covid.h:332 */
    /* memory safety: every day index below derives from these clamped values */
    start = min(max(start, 0), horizon);
    count_work(M, start, work);

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
        double ncur = (s == ws) ? 1.0 : 0.0;
#pragma unroll 1
        for (int k = ws - 2 * T; k < ws; ++k) {
            B.hn_u[(k + MCMC_HOFF(T)) * MCMC_WAVE + lane] = 0.0;
            B.he_u[(k + MCMC_HOFF(T)) * MCMC_WAVE + lane] = 0.0;
        }
        B.hn_u[(ws + MCMC_HOFF(T)) * MCMC_WAVE + lane] = ncur;

        if constexpr (DERIVE) {
            /* days before the wave's first start: all buffers are zero there */
#pragma unroll 1
            for (int i = 0; i < ws; ++i) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    double* o = B.curves_u + ((size_t)c * M.size + i) * MCMC_WAVE;
                    if (shift == 1) o[lane] = w * 0.0; else o[lane] += w * 0.0;
                }
                B.curves_u[((size_t)8 * M.size + i) * MCMC_WAVE + lane] = 1.0; /* covid.h:349 */
            }
        }

        /* the window of the first chunk is the same for every block of this shift: history
         * rows a0 .. a0+2T-1 with a0 = ws-T+1 */
        const int a0 = ws - T + 1;
        const double* hbase = B.hn_u + (a0 + MCMC_HOFF(T)) * MCMC_WAVE;
        const double* ebase = B.he_u + (a0 + MCMC_HOFF(T)) * MCMC_WAVE;
        double WA[T], WB[T];
#pragma unroll
        for (int q = 0; q < T; ++q) {
            WA[q] = hbase[q * MCMC_WAVE + lane];
            WB[q] = hbase[(q + T) * MCMC_WAVE + lane];
        }
        pc.stamp(PH_SETUP);
        double* ds_u = B.d1s_u + (shift == 1 ? 0 : M.nData * MCMC_WAVE);

#pragma unroll 1
        for (int j0 = ws; j0 < horizon; j0 += T) {
            double acc[A_NPOS][T];
#pragma unroll
            for (int c = 0; c < A_NPOS; ++c)
#pragma unroll
                for (int t = 0; t < T; ++t) acc[c][t] = 0.0;

            double beh[T];
            /* ------------- phase A: sources k < j0, lags descending = k ascending */
            const int b = (j0 - ws) / T;                     /* full chunks of T lags */
            const int bRun = b;
            bool anyE = keHi >= keLo;
            int Lc = j0 - ws + T - 1;                        /* first lag of chunk 0 */
            int c0 = 0;
            /* lags >= nLagRows are rows of zeros (the d < nDays cut-off of project(),
             * covid.h:378): chunks made only of them are skipped, the window starts later */
            if (Lc - T + 1 >= M.nLagRows) {
                c0 = (Lc - T + 1 - M.nLagRows) / T + 1;
                if (c0 > b) c0 = b;
                Lc -= c0 * T;
#pragma unroll
                for (int q = 0; q < T; ++q) {
                    WA[q] = hbase[(c0 * T + q) * MCMC_WAVE + lane];
                    WB[q] = hbase[(c0 * T + q + T) * MCMC_WAVE + lane];
                }
            }
#if MCMC_SMEM_TABLE
            double8_t r = smem_row8(M.Rpos, Lc);
#else
            double8_t r = lds_row8(lds.Rpos, Lc);
#endif
            const double* hnx = hbase + (c0 + 2) * T * MCMC_WAVE;   /* refill source of chunk c0 */
            /* Source days that carry a capacity correction form the wave-uniform range
             * [keLo, keHi]; chunk c reads source days [a0+c*T, a0+c*T+2T-2].  So the chunks
             * split into three runs: plain, with e-terms, plain.  Each run is a loop of its own
             * (two chunks per iteration, the window halves swapping roles). */
            const int c0r = min(c0, bRun);
            int cLo = bRun, cHi = bRun - 1;
            if (anyE) {
                const int num = keLo - a0 - 2 * T + 2;
                cLo = num <= 0 ? 0 : (num + T - 1) / T;
                cHi = (keHi - a0) / T;
                cLo = max(cLo, c0r);
                cHi = min(cHi, bRun - 1);
                if (cHi < cLo) { cLo = bRun; cHi = bRun - 1; }
            }
            const double* ek0 = ebase + cLo * T * MCMC_WAVE;
            /* The lane index the lag loops address with is re-defined ahead of each run, as an opaque
             * copy.  Otherwise the compiler may reload its (spilled) 64-bit lane offset just ahead of a
             * loop; a load still pending at the loop header makes the wait-count pass put a full vmcnt(0)
             * INSIDE the loop, which drains the window prefetch every iteration.
             * The three runs are three pieces of straight-line code, not one loop over a run index: as a
             * loop, the 56 accumulators and both windows went through a register-to-register copy at
             * every run's entry (~230 instructions, up to three times per block). */
            auto run_plain = [&](int nch, bool firstRun) {
                pc.count(PH_NCHUNK, nch);
                unsigned laneL = lane;
#if MCMC_LANE_LAUNDER
                asm volatile("" : "+v"(laneL));
#endif
                /* The block's first run is entered with the previous block's phase-B STORES still pending as far
                 * as the compiler can tell (they were issued after this block's window loads).  gfx9 counts loads
                 * and stores in one vmcnt and they may complete out of order with respect to each other, so with
                 * both kinds pending every wait for a window register becomes vmcnt(0) -- and, the loop header
                 * merging that state into every iteration, a full drain of the window prefetch at the top of
                 * EVERY iteration of this loop (found in round 3: tools/isa_guard.py only looked at the last of
                 * the two plain loops).  In real time those stores and loads finished during phase B: draining
                 * here, once per block, costs nothing and lets the loop wait with vmcnt(T + 1). */
#if MCMC_PLAIN_DRAIN
                if (firstRun && nch >= 2) __builtin_amdgcn_s_waitcnt(0x0F70);    /* vmcnt(0) */
#endif
#pragma unroll 1
                for (; nch >= 2; nch -= 2) {
                    lag_chunk<T, FUSE>(acc, WA, WB, r, MCMC_TAB(lds, M), Lc, hnx, laneL);
                    lag_chunk<T, FUSE>(acc, WB, WA, r, MCMC_TAB(lds, M), Lc - T, hnx + T * MCMC_WAVE, laneL);
                    hnx += 2 * T * MCMC_WAVE;
                    Lc -= 2 * T;
                }
                if (nch > 0) {
                    lag_chunk<T, FUSE>(acc, WA, WB, r, MCMC_TAB(lds, M), Lc, hnx, laneL);
#pragma unroll
                    for (int q = 0; q < T; ++q) { const double tmp = WA[q]; WA[q] = WB[q]; WB[q] = tmp; }
                    hnx += T * MCMC_WAVE;
                    Lc -= T;
                }
                pc.stamp(PH_CHUNKS);
            };
            run_plain(cLo - c0r, true);
            {
                int nch = cHi - cLo + 1;
                pc.count(PH_NCHUNK_E, nch);
                unsigned laneL = lane;
#if MCMC_LANE_LAUNDER
                asm volatile("" : "+v"(laneL));
#endif
                if (nch > 0) {
                    /* e-window of the run's first chunk (one exposed round trip per block),
                     * refilled two chunks ahead from then on */
                    double EA[T], EB[T];
#pragma unroll
                    for (int q = 0; q < T; ++q) {
                        EA[q] = ek0[q * MCMC_WAVE + laneL];
                        EB[q] = ek0[(q + T) * MCMC_WAVE + laneL];
                    }
                    const double* ekx = ek0 + 2 * T * MCMC_WAVE;
                    double4_t rn = lds_row4(lds.Rneg, Lc);
                    /* the e-window just requested is needed by the first chunk anyway; with it
                     * still in flight at the loop header the wait counts inside the loop come
                     * out far too tight for the steady state */
#if MCMC_E_DRAIN
                    if (nch >= 2) __builtin_amdgcn_s_waitcnt(0x0F70);    /* vmcnt(0) */
#endif
#pragma unroll 1
                    for (; nch >= 2; nch -= 2) {
                        lag_chunk_e<T, FUSE>(acc, WA, WB, EA, EB, r, rn, MCMC_TAB(lds, M), lds.Rneg, Lc, hnx, ekx, laneL);
                        lag_chunk_e<T, FUSE>(acc, WB, WA, EB, EA, r, rn, MCMC_TAB(lds, M), lds.Rneg, Lc - T,
                                             hnx + T * MCMC_WAVE, ekx + T * MCMC_WAVE, laneL);
                        hnx += 2 * T * MCMC_WAVE;
                        ekx += 2 * T * MCMC_WAVE;
                        Lc -= 2 * T;
                    }
                    if (nch > 0) {
                        lag_chunk_e<T, FUSE>(acc, WA, WB, EA, EB, r, rn, MCMC_TAB(lds, M), lds.Rneg, Lc, hnx, ekx, laneL);
#pragma unroll
                        for (int q = 0; q < T; ++q) { const double tmp = WA[q]; WA[q] = WB[q]; WB[q] = tmp; }
                        hnx += T * MCMC_WAVE;
                        Lc -= T;
                    }
                }
                pc.stamp(PH_CHUNKS_E);
            }
            run_plain(bRun - 1 - cHi, false);
            /* what phase B needs from memory is requested before the tail's 7*T*(T-1)/2
             * multiply-adds: behaviour values and piecewise factors of the block's days.
             * Rows 0..T-1 across the lanes, for the tail and phase B of this block (re-read per
             * block: six LDS reads, and the registers stay free during the lag loops) */
            RowRegs<T> RR;
            if constexpr (MCMC_ROWS_IN_LANES(T, FUSE)) RR = load_row_regs<T>(lds, lane);
            double pcfv[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int bi = min(max(j0 + t - md, 0), M.nB - 1);
                beh[t] = B.x_u[bi * MCMC_WAVE + lane];
                /* piecewise-constant factor of the day's breakpoint interval (covid.h:432-444).
                 * Loaded here and not inside phase B: a load there makes every day wait for
                 * ALL outstanding vector memory operations, the history stores included. */
                const int sg = __builtin_amdgcn_readfirstlane(lds.seg[j0 + t]);
                const double dv = xdisc_u[max(sg, 0) * MCMC_WAVE + lane];
                pcfv[t] = sg >= 0 ? dv : 1.0;
            }
            pc.stamp(PH_CHUNKS);
            {
                /* the tail reads source days [j0-T+1, j0-1] */
                if (anyE && j0 - T + 1 <= keHi && j0 - 1 >= keLo)
                    lag_tail<T, FUSE, true>(acc, WA, lds.Rpos, lds.Rneg, ebase + b * T * MCMC_WAVE, lane, RR);
                else
                    lag_tail<T, FUSE, false>(acc, WA, lds.Rpos, lds.Rneg, nullptr, lane, RR);
            }
            pc.stamp(PH_TAIL);
            /* the window registers are free now: request the first window of the NEXT block
             * before the sequential phase B, unless this block is about to write into it
             * (only the first block of a shift does: its days are ws .. ws+T-1 <= a0+2T-1) */
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
                /* today's new infections into today and the rest of the block (covid.h:403-410):
                 * with FMA contraction from the rows held across the lanes (RowRegs); otherwise
                 * the rows 0..T-1 are re-read from LDS every day (the opaque zero keeps the
                 * compiler from parking all of them in AccVGPRs, which costs two register
                 * moves per multiply-add) */
                if constexpr (MCMC_ROWS_IN_LANES(T, FUSE)) {
#pragma unroll
                    for (int d = 0; d < T - t; d += 2) {
                        if (d + 1 < T - t) dpp_row14<T>(acc, t + d, RR.pos[d >> 1], ncur);
                        else dpp_row7<T, 0>(acc, t + d, RR.pos[d >> 1], ncur);
                    }
                } else {
                    int rowOff = 0;
                    asm volatile("" : "+s"(rowOff));
                    const double* rpB = lds.Rpos + rowOff;
#pragma unroll
                    for (int tt = t; tt < T; ++tt) {
                        const double8_t r0 = lds_row8(rpB, tt - t);
                        acc[A_MI][tt] = madd<FUSE>(ncur, r0[0], acc[A_MI][tt]);
                        acc[A_HI][tt] = madd<FUSE>(ncur, r0[1], acc[A_HI][tt]);
                        acc[A_INC][tt] = madd<FUSE>(ncur, r0[2], acc[A_INC][tt]);
                        acc[A_MILD][tt] = madd<FUSE>(ncur, r0[3], acc[A_MILD][tt]);
                        acc[A_SER][tt] = madd<FUSE>(ncur, r0[4], acc[A_SER][tt]);
                        acc[A_REC][tt] = madd<FUSE>(ncur, r0[5], acc[A_REC][tt]);
                        acc[A_DEAD][tt] = madd<FUSE>(ncur, r0[6], acc[A_DEAD][tt]);
                    }
                }
                /* hospital capacity (covid.h:413-429) */
                const double overCapacity = acc[A_SER][t] * M.probICU - lds.capacity[i];
                const double extraDeaths = M.probLethal * overCapacity;
                const bool hasE = (i >= s) && (extraDeaths > 0);
                const double e = hasE ? extraDeaths : 0.0;
                B.he_u[(i + MCMC_HOFF(T)) * MCMC_WAVE + lane] = e;
                if (__builtin_amdgcn_readfirstlane(__any(hasE) ? 1 : 0)) {
                    keLo = min(keLo, i);
                    keHi = max(keHi, i);
                    pc.count(PH_EDAYS, 1);
                    const double ne = -e;
                    if constexpr (MCMC_ROWS_IN_LANES(T, FUSE)) {
#pragma unroll
                        for (int tt = t; tt < T; ++tt) dpp_neg3_at<T>(acc, tt, tt - t, RR, e, ne);
                    } else {
#pragma unroll
                        for (int tt = t; tt < T; ++tt) {
                            const double4_t rn = lds_row4(lds.Rneg, tt - t);
                            acc[A_DEAD][tt] = acc[A_DEAD][tt] + e;
                            acc[A_SER][tt] = madd<FUSE>(ne, rn[0], acc[A_SER][tt]);
                            acc[A_REC][tt] = madd<FUSE>(ne, rn[1], acc[A_REC][tt]);
                            acc[A_DEAD][tt] = madd<FUSE>(ne, rn[2], acc[A_DEAD][tt]);
                        }
                    }
                }
                /* piecewise-constant factor and behaviour (covid.h:432-454) */
                const double pcf = pcfv[t];
                double smoothbehavior = 1.0;
                if (i >= md) smoothbehavior = pcf * beh[t];
                const double totalBehavior = pcf * smoothbehavior;
                /* tomorrow's new infections (covid.h:461-462); asymptomatic == mild buffer */
                const double nSus = M.pop - acc[A_INC][t] - acc[A_MILD][t] - acc[A_MILD][t] -
                                    acc[A_SER][t] - acc[A_REC][t] - acc[A_DEAD][t];
                double nnext = totalBehavior * nSus / M.pop *
                               (lp.betaMild * acc[A_MI][t] + lp.betaHigh * acc[A_HI][t]);
                if (i < s) nnext = (i + 1 == s) ? 1.0 : 0.0;       /* not started yet */
                B.hn_u[(i + 1 + MCMC_HOFF(T)) * MCMC_WAVE + lane] = nnext;

                /* this pass's weighted deaths of data day i-md (covid.h:475,484); the
                 * likelihood terms are summed after the day loop */
                if (i >= md && i - md < M.nData)
                    ds_u[(i - md) * MCMC_WAVE + lane] = w * acc[A_DEAD][t];
                if constexpr (DERIVE) {
                    if (i < M.size) {
                        const double vals[8] = {acc[A_MI][t], acc[A_HI][t], acc[A_INC][t], acc[A_MILD][t],
                                                acc[A_MILD][t], acc[A_SER][t], acc[A_REC][t], acc[A_DEAD][t]};
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            double* o = B.curves_u + ((size_t)c * M.size + i) * MCMC_WAVE;
                            if (shift == 1) o[lane] = w * vals[c]; else o[lane] += w * vals[c];
                        }
                        B.curves_u[((size_t)8 * M.size + i) * MCMC_WAVE + lane] = (i >= s) ? totalBehavior : 1.0;
                    }
                }
                ncur = nnext;
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
        pc.count(PH_EHULL, keHi >= keLo ? keHi - keLo + 1 : 0);
        /* log-likelihood of this pass (covid.h:488-492), data days in order: the output
         * coord `dead` is the first pass's weighted deaths (multset, covid.h:475), plus this
         * pass's in the second (multadd, covid.h:484).  A streaming loop over two per-lane
         * rows and the LDS-resident observations. */
        {
            /* eight data days per trip: all loads of a trip are requested before the first
             * term is formed, so one memory round trip is paid per eight days, not per day */
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
                    term[j] = w * (-0.5 * delta * delta / s2[j]);
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
                loglike += w * (-0.5 * delta * delta / lds.sigma2[d0]);
            }
        }
        pc.stamp(PH_LOGLIKE);
    }
    return loglike;
}

/* days per register block of the fast mode.  8 on performance grounds: a -DMCMC_TF=16 build (fewer history
 * loads per multiply-add) is memory-safe -- a -DMCMC_F32_CHECK build of it ran the fast-mode tests on an
 * MI355X with every row range compared against its buffer, no violation, and the plain build ran them clean
 * (profiles/r02/f_fast_mode_tf16_checked_run.txt) -- but its unrolled 16-day sequential part spills 473 SGPRs
 * and 68 VGPRs to scratch and it is 14 % slower (23.2 against 27.1 M steps/s). */
#ifndef MCMC_TF
#define MCMC_TF 8
#endif
static_assert(MCMC_TF == 8 || MCMC_TF == 16, "fast-mode block sizes validated on hardware: 8 (product) and 16");
#include "mcmc_model_f32.hip.h"

/* t / dx / dx / dx of SmoothnessPrior::eval (mcmc.h:208).  When dx = L/n is a power of two
 * (the notebooks' L = n gives dx = 1) each division is the multiplication by the exact
 * reciprocal: the same real number rounded once, hence the same double -- without the three
 * ~12-instruction division sequences per element.  The branch is wave-uniform. */
__device__ __forceinline__ bool is_pow2(double x) {
    return x > 0 && (__double_as_longlong(x) & 0x000fffffffffffffLL) == 0 && x < 1e300 && x > 1e-300;
}
__device__ __forceinline__ double div3(double t, double dx, bool pow2, double inv) {
    if (pow2) return t * inv * inv * inv;
    return t / dx / dx / dx;
}

/* SmoothnessPrior::eval (mcmc.h:197-214) on behavior = x[0..nB) */
__device__ inline double eval_prior(const DevModel& M, const double* x_u, unsigned lane) {
    const int n = M.nB;
    const double dx = M.smoothL / (double)n;
    const bool dxPow2 = is_pow2(dx);
    const double invDx = 1.0 / dx;
    double loglike = 0;
    double fm = x_u[lane], f0 = x_u[MCMC_WAVE + lane];
#pragma unroll 4
    for (int i = 1; i + 1 < n; ++i) {
        const double fp = x_u[(i + 1) * MCMC_WAVE + lane];
        const double dd = fp + fm - 2 * f0;
        loglike -= div3(dd * dd, dx, dxPow2, invDx);
        fm = f0;
        f0 = fp;
    }
    loglike *= M.smoothLs * M.smoothLs / M.smoothL;
    return loglike;
}

__device__ inline void load_lane_scalars(const DevModel& M, const double* x_u, unsigned lane, LaneScalars& lp) {
    lp.betaMild = x_u[(M.nB + M.nDisc) * MCMC_WAVE + lane];
    lp.betaHigh = x_u[(M.nB + M.nDisc + 1) * MCMC_WAVE + lane];
    lp.delay = x_u[(M.nB + M.nDisc + 2) * MCMC_WAVE + lane];
}

/*
 * DiseaseSpread::step up to the eval call (covid.h:499-622): reads the current vector xc,
 * writes the complete proposal vector xp, returns its scalars in lp and whether the
 * behaviour function changed (only then the smoothness prior has to be re-evaluated).
 */
__device__ inline bool propose(const DevModel& M, Pcg32& rnd, const double* __restrict__ xc_u,
                               double* __restrict__ xp_u, unsigned lane, double stepsize, LaneScalars& lp,
                               double& priorOut, bool withBetas = true) {
    load_lane_scalars(M, xc_u, lane, lp);
    double disc[MCMC_MAXDISC];
#pragma unroll
    for (int m = 0; m < MCMC_MAXDISC; ++m)
        disc[m] = (m < M.nDisc) ? xc_u[(M.nB + m) * MCMC_WAVE + lane] : 1.0;
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
    /* withBetas is false only for a region of the multi-region model that does not own the shared
     * betas: it skips their branch, gate draw included */
    if (withBetas && rnd.next_float() < 0.3f) {                     /* covid.h:518-532 */
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
    /* write the proposal vector; where some lane's behaviour function changes, the
     * smoothness prior of the proposal (mcmc.h:197-214) is summed on the way, from the
     * values just written: same terms, same order as a pass of its own over them */
    priorOut = 0.0;
    if (__any(doB)) {
        const int n = M.nB;
        const double dx = M.smoothL / (double)n;
        const bool dxPow2 = is_pow2(dx);
        const double invDx = 1.0 / dx;
        double loglike = 0;
        double fm = 0, f0 = 0;
        constexpr int PB = MCMC_PROP_BATCH;   /* loads of a trip are requested together */
#pragma unroll 1
        for (int i0 = 0; i0 < n; i0 += PB) {
            double vv[PB];
#pragma unroll
            for (int j = 0; j < PB; ++j) vv[j] = xc_u[min(i0 + j, n - 1) * MCMC_WAVE + lane];
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const int i = i0 + j;
                if (i < n) {
                    /* straight-line: the lanes that do not move their behaviour function compute a value
                     * too (omega = A = 0) and keep the old one -- cheaper for the wave than a per-lane
                     * branch around ~60 double-precision operations */
                    const double moved = bound_reflect(vv[j] + A * (mcmc_cos(omega * (double)i) - 1), 0.0, 2.0);
                    const double v = doB ? moved : vv[j];
                    xp_u[i * MCMC_WAVE + lane] = v;
                    if (i >= 2) {
                        const double dd = v + fm - 2 * f0;
                        loglike -= div3(dd * dd, dx, dxPow2, invDx);
                    }
                    fm = f0;
                    f0 = v;
                }
            }
        }
        loglike *= M.smoothLs * M.smoothLs / M.smoothL;
        priorOut = loglike;
    } else {
#pragma unroll 8
        for (int i = 0; i < M.nB; ++i) xp_u[i * MCMC_WAVE + lane] = xc_u[i * MCMC_WAVE + lane];
    }
#pragma unroll
    for (int m = 0; m < MCMC_MAXDISC; ++m)
        if (m < M.nDisc) xp_u[(M.nB + m) * MCMC_WAVE + lane] = disc[m];
    xp_u[(M.nB + M.nDisc) * MCMC_WAVE + lane] = lp.betaMild;
    xp_u[(M.nB + M.nDisc + 1) * MCMC_WAVE + lane] = lp.betaHigh;
    xp_u[(M.nB + M.nDisc + 2) * MCMC_WAVE + lane] = lp.delay;
    return doB;
}

/* ------------------------------------------------------------ chain state ---- */
/* all arrays tile-major (see "Addressing convention"); nTiles = ceil(nChains/64) */
struct ChainState {
    double* params;               /* [nTiles][dim][64] current state of every chain      */
    double* prop;                 /* [nTiles][dim][64] proposal scratch                  */
    double* ll;                   /* [nTiles][2R][64]: (model, prior) of every region    */
    double* stepsize;             /* [nTiles][R][64]                                     */
    uint64_t* rng;                /* [nTiles][2][64]: state, inc                         */
    int* nAccept;                 /* [nTiles][64]                                        */
    unsigned long long* pairs;    /* [nTiles][2][64]: WorkCount algo, exec               */
    double* hn; double* he;       /* [nTiles][histRows][64]                              */
    double* d1s;                  /* [nTiles][2*nData][64]                               */
    double* mean;                 /* [nTiles][dim][64] or null                           */
    double* recLL;                /* [nSavedCap][nTiles][64]                             */
    double* recSamples;           /* [nSavedCap][nTiles][dim][64] or null                */
    const double* chainBeta;      /* [nTiles][64] inverse temperature of the chain's rung, or null */
    double* llProp;               /* [nTiles][R][64] multi-region only: proposed model log-likelihoods */
    long long nChains;
    int nTiles;
    int dim;
    int histRows;
};

enum { SF_RANDOM = 1, SF_RESET_ACC = 2, SF_ADJUST_END = 4, SF_RECORD = 8, SF_MEAN = 16 };

/* mcmc.h:1110-1112 */
__device__ __forceinline__ double rate2corr(double x) {
    return (1 + 0.726484 * x * x * x * x) / (0.82051 + 0.0427315 / (x + 0.0001));
}

#define MCMC_BLOCK (MCMC_WAVE * MCMC_WAVES_PER_BLOCK)
#ifndef MCMC_WPE_MIN
#define MCMC_WPE_MIN 1
#endif
#define MCMC_KERNEL_ATTRS __launch_bounds__(MCMC_BLOCK) __attribute__((amdgpu_waves_per_eu(MCMC_WPE_MIN, 2)))

__device__ __forceinline__ WaveBases wave_bases(const DevModel& M, const ChainState& S, int tile,
                                                const double* x_u) {
    WaveBases B;
    B.x_u = x_u;
    B.hn_u = S.hn + (size_t)tile * S.histRows * MCMC_WAVE;
    B.he_u = S.he + (size_t)tile * S.histRows * MCMC_WAVE;
    B.d1s_u = S.d1s + (size_t)tile * 2 * M.nData * MCMC_WAVE;
    B.curves_u = nullptr;
    return B;
}

/* ------------------------------------------------------------ multi-region ---- */
/* EXTENSION (BASELINE.json configs[4]; the reference has one region, covid.h:284-321): a chain's
 * State is R region blocks [DiseaseSpread_r (, SmoothnessPrior_r)] -- each the complete covid.h
 * model with observations, population and hospital capacity of its own -- sharing betaMild and
 * betaHigh, which region 0 owns.  Parameter vector: R blocks of the single-region layout; the beta
 * slots of every block mirror region 0's.  One MH step moves ONE region, the head, by SYSTEMATIC
 * scan: head = (number of random-subspace steps so far) mod R, a given-subspace step reuses it.
 * The head is the same for every chain, so a wavefront never diverges on it (a per-chain random
 * head would make every wave evaluate ~min(R, 64) regions per step for 64 useful evaluations).
 * Region 0's step includes the betas' branch; if a lane's betas changed, all other regions are its
 * dependents and are re-evaluated with them (mcmc.h:539-602); delta sums (-old + new) over the
 * evaluated likelihoods in State order DS_0, P_0, DS_1, P_1, ... (mcmc.h:604-606).  With R = 1
 * every operation is the single-region kernel's.  Oracle: orc_multi_step / orc_multi_run.
 *
 * Region data is wave-uniform but differs between the regions a wave walks through, so each wave
 * stages the current region's observations and capacity curve into an LDS slice of its own
 * (behind the workgroup's shared image; no barrier: only this wave reads it). */
__host__ __device__ inline size_t region_slice_doubles(const DevModel& M) {
    return (size_t)M.simPad + 2 * (size_t)M.nData;
}
__device__ __forceinline__ LdsView stage_region(const DevModel& M, const LdsView& shared, double* slice, int q,
                                                unsigned lane, double& popOut) {
    double* cap = slice;
    double* dea = cap + M.simPad;
    double* sg2 = dea + M.nData;
    const double cap0 = M.regCap0[q], rate = M.regCapRate[q];
    for (int i = lane; i < M.simPad; i += MCMC_WAVE) cap[i] = cap0 + fmax(rate * (double)(i - M.md), 0.0);   /* covid.h:314 */
    for (int i = lane; i < M.nData; i += MCMC_WAVE) {
        dea[i] = M.regDeaths[(size_t)q * M.nData + i];
        sg2[i] = M.regSigma2[(size_t)q * M.nData + i];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);        /* lgkmcnt(0): the slice is written before any lane reads it */
    __builtin_amdgcn_wave_barrier();
    popOut = M.regPop[q];
    LdsView v = shared;
    v.capacity = cap; v.deaths = dea; v.sigma2 = sg2;
    return v;
}

/*
 * n_steps Metropolis-Hastings steps for every chain: Target::step (mcmc.h:750-783) with
 * State::step_random/given_subspace (mcmc.h:494-610), DiseaseSpread::step (covid.h:499-627),
 * SmoothnessPrior::eval (mcmc.h:197-214), State::accept (mcmc.h:653-661) and the per-step
 * bookkeeping of MetropolisChain::run (mcmc.h:1023-1127) driven by `flags`.
 * MULTI: the multi-region State (see above); headCounter = random-subspace steps before this launch.
 */
template <int T, bool FUSE, int PREC = 0, bool MULTI = false>
__global__ MCMC_KERNEL_ATTRS void mh_step_kernel(DevModel M, ChainState S, int n_steps,
                                                 const double* __restrict__ betas,
                                                 const int* __restrict__ flags, int slot0, long long headCounter) {
    static_assert(!(MULTI && PREC != 0), "the multi-region model runs in double only");
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    const LdsView lds = stage_lds_for<PREC>(M, lds_raw, T);
    const unsigned lane = threadIdx.x & (MCMC_WAVE - 1);
    const int tile = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * MCMC_WAVES_PER_BLOCK + threadIdx.x / MCMC_WAVE));
    if (tile >= S.nTiles) return;
    const size_t tileVec = (size_t)S.dim * MCMC_WAVE;
    const double* xc_u = S.params + (size_t)tile * tileVec;
    double* xcw_u = S.params + (size_t)tile * tileVec;
    double* xp_u = S.prop + (size_t)tile * tileVec;
    uint64_t* rng_u = S.rng + (size_t)tile * 2 * MCMC_WAVE;
    const int R = MULTI ? M.nRegions : 1;
    const int dimR = MULTI ? M.dimRegion : S.dim;
    double* ll_u = S.ll + (size_t)tile * 2 * R * MCMC_WAVE;
    double* ss_u = S.stepsize + (size_t)tile * R * MCMC_WAVE;
    const size_t flat = (size_t)tile * MCMC_WAVE + lane;

    Pcg32 rnd;
    rnd.state = rng_u[lane];
    rnd.inc = rng_u[MCMC_WAVE + lane];
    double llModel = 0, llPrior = 0, stepsize = 1;
    if constexpr (!MULTI) {
        llModel = ll_u[lane];
        llPrior = ll_u[MCMC_WAVE + lane];
        stepsize = ss_u[lane];
    }
    int nAccept = S.nAccept[flat];
    unsigned long long* pairs_u = S.pairs + (size_t)tile * 2 * MCMC_WAVE;
    WorkCount work;
    work.algo = pairs_u[lane];
    work.exec = pairs_u[MCMC_WAVE + lane];
    const WaveBases B = wave_bases(M, S, tile, xp_u);
    int slot = slot0;
    PhaseClock pc;
    pc.start();
    /* tempering ladder (extension): the chain's rung multiplies the target's beta */
    const double rungBeta = S.chainBeta ? S.chainBeta[flat] : 1.0;
    /* multi-region: this wave's LDS slice, the head of the scan */
    double* slice = nullptr;
    double* llp_u = nullptr;
    long long scan = headCounter;
    int head = 0;
    if constexpr (MULTI) {
        const int waveInBlock = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / MCMC_WAVE));
        slice = lds_raw + (lds_bytes(M, false, T) + 7) / 8 + (size_t)waveInBlock * region_slice_doubles(M);
        llp_u = S.llProp + (size_t)tile * R * MCMC_WAVE;
        head = scan > 0 ? (int)((scan - 1) % R) : 0;
    }

#pragma unroll 1
    for (int g = 0; g < n_steps; ++g) {
        const int fl = flags[g];
        const double beta = betas[g] * rungBeta;

        if (fl & SF_RESET_ACC) nAccept = 0;                         /* mcmc.h:1032,1090 */
        if (fl & SF_RECORD) {                                       /* mcmc.h:1037-1062 */
            double l = 0;
            if constexpr (MULTI) {
                for (int q = 0; q < R; ++q) {                       /* State::loglikelihood, mcmc.h:612-618 */
                    l += ll_u[(2 * q) * MCMC_WAVE + lane];
                    if (M.hasPrior) l += ll_u[(2 * q + 1) * MCMC_WAVE + lane];
                }
            } else {
                l += llModel;
                if (M.hasPrior) l += llPrior;
            }
            S.recLL[((size_t)slot * S.nTiles + tile) * MCMC_WAVE + lane] = 0.0 + beta * l; /* log(weight()==1) = 0 */
            if (S.recSamples) {
                double* dst = S.recSamples + ((size_t)slot * S.nTiles + tile) * tileVec;
                for (int k = 0; k < S.dim; ++k) dst[k * MCMC_WAVE + lane] = xc_u[k * MCMC_WAVE + lane];
            }
            ++slot;
        }
        if ((fl & SF_MEAN) && S.mean) {                             /* mcmc.h:1064-1065 */
            double* mu = S.mean + (size_t)tile * tileVec;
            for (int k = 0; k < S.dim; ++k) mu[k * MCMC_WAVE + lane] += xc_u[k * MCMC_WAVE + lane];
        }

        if (fl & SF_RANDOM) {                                       /* mcmc.h:494-500 */
            const int nState = M.hasPrior ? 2 : 1;
            int hd;
            do {
                const float f = rnd.next_float() * (float)nState;
                hd = min((int)f, nState - 1);
            } while (hd != 0);    /* SmoothnessPrior has no steppable coords: redraw */
            if constexpr (MULTI) { head = (int)(scan % R); ++scan; }
        }

        pc.stamp(PH_RECORD);
        LaneScalars lp;
        double llpNew;
        bool accept;
        if constexpr (!MULTI) {
            const bool behChanged = propose(M, rnd, xc_u, xp_u, lane, stepsize, lp, llpNew);
            pc.stamp(PH_PROPOSE);
            double llm;
            if constexpr (PREC != 0) llm = eval_model_f32<(PREC == 2 ? MCMC_TM : MCMC_TF), PREC == 2>(M, lds, B, lp, lane, work, pc);   /* fast mode */
            else llm = eval_model<T, FUSE, false>(M, lds, B, lp, lane, work, pc);
            double llp = 0.0;
            double delta = 0;                                           /* mcmc.h:604-606 */
            delta += -llModel + llm;
            if (M.hasPrior) {
                llp = behChanged ? llpNew : llPrior;     /* unchanged behaviour: the prior keeps its value */
                delta += -llPrior + llp;
            }
            pc.stamp(PH_PRIOR);
            delta += 0.0;                                               /* log(wNew)-log(wOld), mcmc.h:765 */
            delta *= beta;                                              /* mcmc.h:766 */
            const double u = (double)rnd.next_float();
            accept = u < 1.0 * mcmc_exp(delta);                         /* mcmc.h:776 */
            if (__any(accept)) {                                        /* State::accept, mcmc.h:653-661 */
                constexpr int AB = MCMC_ACC_BATCH;
#pragma unroll 1
                for (int k0 = 0; k0 < S.dim; k0 += AB) {
                    double vv[AB];
#pragma unroll
                    for (int j = 0; j < AB; ++j) vv[j] = xp_u[min(k0 + j, S.dim - 1) * MCMC_WAVE + lane];
#pragma unroll
                    for (int j = 0; j < AB; ++j)
                        if (accept && k0 + j < S.dim) xcw_u[(k0 + j) * MCMC_WAVE + lane] = vv[j];
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
        } else {
            const double* xcH_u = xc_u + (size_t)head * dimR * MCMC_WAVE;
            double* xpH_u = xp_u + (size_t)head * dimR * MCMC_WAVE;
            const int iBeta = M.nB + M.nDisc;
            const double bmOld = xcH_u[iBeta * MCMC_WAVE + lane], bhOld = xcH_u[(iBeta + 1) * MCMC_WAVE + lane];
            stepsize = ss_u[head * MCMC_WAVE + lane];
            const bool behChanged = propose(M, rnd, xcH_u, xpH_u, lane, stepsize, lp, llpNew, head == 0);
            pc.stamp(PH_PROPOSE);
            const bool needAll = (lp.betaMild != bmOld) || (lp.betaHigh != bhOld);
            const bool waveAll = __builtin_amdgcn_readfirstlane(__any(needAll) ? 1 : 0) != 0;
            double llp = 0.0;
            double delta = 0;                                           /* mcmc.h:604-606, State order */
#pragma unroll 1
            for (int q = 0; q < R; ++q) {
                if (q != head && !waveAll) continue;
                DevModel Mq = M;
                const LdsView ldsq = stage_region(M, lds, slice, q, lane, Mq.pop);
                WaveBases Bq = B;
                LaneScalars lpq = lp;
                if (q == head) {
                    Bq.x_u = xpH_u;
                } else {                                 /* a dependent: its own coordinates, the head's new betas */
                    Bq.x_u = xc_u + (size_t)q * dimR * MCMC_WAVE;
                    lpq.delay = Bq.x_u[(iBeta + 2) * MCMC_WAVE + lane];
                }
                const double llm = eval_model<T, FUSE, false>(Mq, ldsq, Bq, lpq, lane, work, pc);
                llp_u[q * MCMC_WAVE + lane] = llm;
                if (q == head || needAll) delta += -ll_u[(2 * q) * MCMC_WAVE + lane] + llm;
                if (q == head && M.hasPrior) {
                    const double llpOld = ll_u[(2 * q + 1) * MCMC_WAVE + lane];
                    llp = behChanged ? llpNew : llpOld;
                    delta += -llpOld + llp;
                }
            }
            pc.stamp(PH_PRIOR);
            delta += 0.0;
            delta *= beta;
            const double u = (double)rnd.next_float();
            accept = u < 1.0 * mcmc_exp(delta);
            if (__any(accept)) {
                double* xcH_w = xcw_u + (size_t)head * dimR * MCMC_WAVE;
                constexpr int AB = 16;
#pragma unroll 1
                for (int k0 = 0; k0 < dimR; k0 += AB) {
                    double vv[AB];
#pragma unroll
                    for (int j = 0; j < AB; ++j) vv[j] = xpH_u[min(k0 + j, dimR - 1) * MCMC_WAVE + lane];
#pragma unroll
                    for (int j = 0; j < AB; ++j)
                        if (accept && k0 + j < dimR) xcH_w[(k0 + j) * MCMC_WAVE + lane] = vv[j];
                }
                for (int q = 0; q < R; ++q) {
                    if (q != head && !waveAll) continue;
                    const double llm = llp_u[q * MCMC_WAVE + lane];
                    if (accept && (q == head || needAll)) {
                        ll_u[(2 * q) * MCMC_WAVE + lane] = llm;
                        xcw_u[((size_t)q * dimR + iBeta) * MCMC_WAVE + lane] = lp.betaMild;      /* the shared betas */
                        xcw_u[((size_t)q * dimR + iBeta + 1) * MCMC_WAVE + lane] = lp.betaHigh;
                    }
                }
                if (accept && M.hasPrior) ll_u[(2 * head + 1) * MCMC_WAVE + lane] = llp;
            }
            if (accept) ++nAccept;
            if (fl & SF_ADJUST_END) {                                   /* the head's own step size, mcmc.h:1107-1114 */
                const double acceptRate = (double)nAccept / 20;
                stepsize *= rate2corr(acceptRate);
            }
            ss_u[head * MCMC_WAVE + lane] = stepsize;
        }
        pc.stamp(PH_ACCEPT);
    }
    pc.flush(lane, tile);

    rng_u[lane] = rnd.state;
    if constexpr (!MULTI) {
        ll_u[lane] = llModel;
        ll_u[MCMC_WAVE + lane] = llPrior;
        ss_u[lane] = stepsize;
    }
    S.nAccept[flat] = nAccept;
    pairs_u[lane] = work.algo;
    pairs_u[MCMC_WAVE + lane] = work.exec;
}

/* State::eval (mcmc.h:385-393) of parameter vectors laid out [nTiles][dim][64]:
 * llOut[nTiles][2R][64] (R = 1 unless MULTI: every region's model and prior log-likelihood) */
template <int T, bool FUSE, int PREC = 0, bool MULTI = false>
__global__ MCMC_KERNEL_ATTRS void eval_kernel(DevModel M, ChainState S, const double* x, double* llOut) {
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    const LdsView lds = stage_lds_for<PREC>(M, lds_raw, T);
    const unsigned lane = threadIdx.x & (MCMC_WAVE - 1);
    const int tile = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * MCMC_WAVES_PER_BLOCK + threadIdx.x / MCMC_WAVE));
    if (tile >= S.nTiles) return;
    WaveBases B = wave_bases(M, S, tile, x + (size_t)tile * S.dim * MCMC_WAVE);
    WorkCount work = {0, 0};
    PhaseClock pc;
    pc.start();
    if constexpr (MULTI) {
        const int R = M.nRegions, dimR = M.dimRegion;
        const int waveInBlock = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / MCMC_WAVE));
        double* slice = lds_raw + (lds_bytes(M, false, T) + 7) / 8 + (size_t)waveInBlock * region_slice_doubles(M);
        const double* x0_u = B.x_u;
#pragma unroll 1
        for (int q = 0; q < R; ++q) {
            DevModel Mq = M;
            const LdsView ldsq = stage_region(M, lds, slice, q, lane, Mq.pop);
            B.x_u = x0_u + (size_t)q * dimR * MCMC_WAVE;
            LaneScalars lp;
            load_lane_scalars(M, B.x_u, lane, lp);
            const double llm = eval_model<T, FUSE, false>(Mq, ldsq, B, lp, lane, work, pc);
            const double llp = M.hasPrior ? eval_prior(M, B.x_u, lane) : 0.0;
            llOut[((size_t)tile * 2 * R + 2 * q) * MCMC_WAVE + lane] = llm;
            llOut[((size_t)tile * 2 * R + 2 * q + 1) * MCMC_WAVE + lane] = llp;
        }
    } else {
        LaneScalars lp;
        load_lane_scalars(M, B.x_u, lane, lp);
        double llm;
        if constexpr (PREC != 0) llm = eval_model_f32<(PREC == 2 ? MCMC_TM : MCMC_TF), PREC == 2>(M, lds, B, lp, lane, work, pc);
        else llm = eval_model<T, FUSE, false>(M, lds, B, lp, lane, work, pc);
        const double llp = M.hasPrior ? eval_prior(M, B.x_u, lane) : 0.0;
        llOut[((size_t)tile * 2) * MCMC_WAVE + lane] = llm;
        llOut[((size_t)tile * 2 + 1) * MCMC_WAVE + lane] = llp;
    }
}

/* multi-region: the beta slots of every block mirror region 0's (the shared betas) */
__global__ void mirror_betas_kernel(double* __restrict__ params, long long nP, int R, int dimR, int iBeta) {
    const long long chain = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (chain >= nP) return;
    double* x = params + (chain / MCMC_WAVE) * (long long)R * dimR * MCMC_WAVE + chain % MCMC_WAVE;
    const double bm = x[(long long)iBeta * MCMC_WAVE], bh = x[(long long)(iBeta + 1) * MCMC_WAVE];
    for (int q = 1; q < R; ++q) {
        x[((long long)q * dimR + iBeta) * MCMC_WAVE] = bm;
        x[((long long)q * dimR + iBeta + 1) * MCMC_WAVE] = bh;
    }
}

/* derived output coords (covid.h:467-485) of parameter vectors [nTiles][dim][64]:
 * curves[nTiles][9*size][64] */
template <int T, bool FUSE>
__global__ MCMC_KERNEL_ATTRS void derive_kernel(DevModel M, ChainState S, const double* x, double* curves) {
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    const LdsView lds = stage_lds(M, lds_raw, true, T);
    const unsigned lane = threadIdx.x & (MCMC_WAVE - 1);
    const int tile = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * MCMC_WAVES_PER_BLOCK + threadIdx.x / MCMC_WAVE));
    if (tile >= S.nTiles) return;
    WaveBases B = wave_bases(M, S, tile, x + (size_t)tile * S.dim * MCMC_WAVE);
    B.curves_u = curves + (size_t)tile * 9 * M.size * MCMC_WAVE;
    LaneScalars lp;
    load_lane_scalars(M, B.x_u, lane, lp);
    WorkCount work = {0, 0};
    PhaseClock pc;
    pc.start();
    (void)eval_model<T, FUSE, true>(M, lds, B, lp, lane, work, pc);
}

/* [n][rows] chain-major (host layout) <-> [nTiles][rows][64] tiles; pad lanes of the last
 * tile copy the last chain */
template <typename Tp>
__global__ void tile_in_kernel(const Tp* __restrict__ src, Tp* __restrict__ dst, long long n, int rows,
                               long long nP) {
    const long long chain = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (chain >= nP) return;
    const long long srcChain = chain < n ? chain : n - 1;
    Tp* d = dst + (chain / MCMC_WAVE) * rows * MCMC_WAVE + chain % MCMC_WAVE;
    for (int k = 0; k < rows; ++k) d[(long long)k * MCMC_WAVE] = src[srcChain * rows + k];
}
template <typename Tp>
__global__ void tile_out_kernel(const Tp* __restrict__ src, Tp* __restrict__ dst, long long n, int rows) {
    const long long chain = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (chain >= n) return;
    const Tp* s = src + (chain / MCMC_WAVE) * rows * MCMC_WAVE + chain % MCMC_WAVE;
    for (int k = 0; k < rows; ++k) dst[chain * rows + k] = s[(long long)k * MCMC_WAVE];
}
/* the recorded points of ONE chain, coordinates [coordFirst, coordFirst+coordCount), out of the
 * tile-major record [slot][tile][dim][64] as a dense [nSaved][coordCount] block */
__global__ void chain_samples_kernel(const double* __restrict__ rec, double* __restrict__ out, int nSaved, int nTiles,
                                     int dim, long long tile, int lane, int coordFirst, int coordCount) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nSaved * coordCount) return;
    const long long s = i / coordCount;
    const int k = (int)(i % coordCount);
    out[i] = rec[((s * nTiles + tile) * dim + coordFirst + k) * MCMC_WAVE + lane];
}

__global__ void scale_kernel(double* __restrict__ a, long long count, double f) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) a[i] *= f;
}

/* cross-chain diagnostics: per-wave partial sums by wavefront shuffle reduction, then one
 * atomic add per wave and quantity.  out[0]=sum nAccept, out[2]=sum ll, out[3]=sum ll^2,
 * out[4+2k]=sum x_k, out[5+2k]=sum x_k^2 */
__global__ __launch_bounds__(MCMC_WAVE) void diagnostics_kernel(DevModel M, ChainState S, double* out) {
    const int tile = blockIdx.x;
    const unsigned lane = threadIdx.x;
    const long long chain = (long long)tile * MCMC_WAVE + lane;
    const bool live = chain < S.nChains;
    const double* xb = S.params + (size_t)tile * S.dim * MCMC_WAVE;
    const int R = M.nRegions > 1 ? M.nRegions : 1;
    const double* ll_u = S.ll + (size_t)tile * 2 * R * MCMC_WAVE;
    const double acc = live ? (double)S.nAccept[chain] : 0.0;
    double l = 0;
    if (live)
        for (int q = 0; q < R; ++q) { l += ll_u[(2 * q) * MCMC_WAVE + lane]; if (M.hasPrior) l += ll_u[(2 * q + 1) * MCMC_WAVE + lane]; }
    const double sAcc = wave_sum_f64(acc), sL = wave_sum_f64(l), sL2 = wave_sum_f64(l * l);
    if (threadIdx.x == 0) {
        atomicAdd(&out[0], sAcc);
        atomicAdd(&out[2], sL);
        atomicAdd(&out[3], sL2);
    }
    for (int k = 0; k < S.dim; ++k) {
        const double x = live ? xb[k * MCMC_WAVE + lane] : 0.0;
        const double sx = wave_sum_f64(x), sx2 = wave_sum_f64(x * x);
        if (threadIdx.x == 0) {
            atomicAdd(&out[4 + 2 * k], sx);
            atomicAdd(&out[5 + 2 * k], sx2);
        }
    }
}

/* ------------------------------------------------------------- tempering ladder ---- */
/* Replica exchange (EXTENSION, BASELINE configs[3]; the reference has no such step).  The
 * chains [L*K, (L+1)*K) form ladder L; chainAt[L][r] is the chain sitting on rung r, whose
 * inverse temperature is ladderBeta[r].  One round proposes the exchange of the rungs
 * (r, r+1) for r = parity, parity+2, ...: with l = State::loglikelihood (mcmc.h:612-618)
 *     accept  iff  double(u) < exp((beta_r - beta_{r+1}) * (l_{r+1} - l_r)),
 * u = nextFloat() of the ladder's own stream pcg32(swap_seed, ladder id), one draw per pair
 * whether accepted or not.  Accepting swaps the TEMPERATURES of the two chains (16 bytes),
 * not their states.  One thread per ladder. */
__global__ __launch_bounds__(256) void ladder_swap_kernel(ChainState S, int hasPrior, int K, long long nLadders,
                                                          const double* __restrict__ ladderBeta,
                                                          int* __restrict__ chainAt, int* __restrict__ rungOf,
                                                          uint64_t* __restrict__ swapRng, double* __restrict__ chainBeta,
                                                          int parity, unsigned long long* __restrict__ nAccepted) {
    const long long L = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (L >= nLadders) return;
    Pcg32 rnd;
    rnd.state = swapRng[2 * L];
    rnd.inc = swapRng[2 * L + 1];
    for (int r = parity; r + 1 < K; r += 2) {
        const int a = chainAt[L * K + r], b = chainAt[L * K + r + 1];
        const size_t ia = ((size_t)(a / MCMC_WAVE) * 2) * MCMC_WAVE + (size_t)(a % MCMC_WAVE);
        const size_t ib = ((size_t)(b / MCMC_WAVE) * 2) * MCMC_WAVE + (size_t)(b % MCMC_WAVE);
        double la = 0, lb = 0;
        la += S.ll[ia];
        lb += S.ll[ib];
        if (hasPrior) { la += S.ll[ia + MCMC_WAVE]; lb += S.ll[ib + MCMC_WAVE]; }
        const double delta = (ladderBeta[r] - ladderBeta[r + 1]) * (lb - la);
        const double u = (double)rnd.next_float();
        if (u < mcmc_exp(delta)) {
            chainAt[L * K + r] = b;
            chainAt[L * K + r + 1] = a;
            rungOf[a] = r + 1;
            rungOf[b] = r;
            chainBeta[a] = ladderBeta[r + 1];
            chainBeta[b] = ladderBeta[r];
            atomicAdd(&nAccepted[r], 1ULL);
        }
    }
    swapRng[2 * L] = rnd.state;
}

/* ------------------------------------------------------------------ toy targets ---- */
/* MyLike1 (test.h:14-67) and FourGaussians (test.h:69-136) under MetropolisChain::run
 * (mcmc.h:1005-1138) with a one-likelihood State: one chain per lane, the whole run in one
 * launch.  Chain-flat layouts: x [dim][n], recLL [slot][n], recS [slot][dim][n].
 * TOY_ROSENBROCK (EXTENSION, BASELINE configs[0] "Gaussian/Rosenbrock"; the reference has no such
 * target) follows FourGaussians' pattern -- one 2-D coordinate "pos", a random-walk step of
 * stepsizeCorrectionFac * (nextFloat() - 0.5) per component (test.h:105), no bounds:
 *     loglike = -((1 - x)^2 + difficulty * (y - x^2)^2) / 20
 * (difficulty = 100: the Rosenbrock density of Goodman & Weare 2010; x ~ N(1, 10),
 * y | x ~ N(x^2, 10 / difficulty)). */
enum { TOY_MYLIKE1 = 0, TOY_FOURGAUSS = 1, TOY_ROSENBROCK = 2 };

template <int KIND>
__device__ __forceinline__ double toy_loglike(const double (&x)[3], double difficulty) {
    if constexpr (KIND == TOY_MYLIKE1) {
        const double dp[2][3] = {{2, 2, 0}, {-2, -2, 0}};
        double loglike = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) loglike -= (dp[i][j] - x[j]) * (dp[i][j] - x[j]) * 0.5;
        return loglike;
    } else if constexpr (KIND == TOY_ROSENBROCK) {
        const double u = 1 - x[0], v = x[1] - x[0] * x[0];
        return -(u * u + difficulty * (v * v)) / 20;
    } else {
        const double a = x[0], y = x[1];
        double loglike;
        loglike = mcmc_exp(-difficulty * ((a + 0.5) * (a + 0.5) / (2 * 0.5 * 0.5) + (y + 1) * (y + 1) / (2 * 0.2 * 0.2)));
        loglike += mcmc_exp(-difficulty * ((a - 1) * (a - 1) / (2 * 0.2 * 0.2) + (y - 0.5) * (y - 0.5) / (2 * 0.5 * 0.5)));
        loglike += mcmc_exp(-difficulty * ((a + y) * (a + y) / (2 * 1) + (a - y - 2) * (a - y - 2) / (2 * 0.2 * 0.2)));
        loglike += mcmc_exp(-difficulty * ((a + y) * (a + y) / (2 * 1) + (a - y + 2) * (a - y + 2) / (2 * 0.1)));
        return mcmc_log(loglike);
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void toy_eval_kernel(const double* __restrict__ x, double* __restrict__ ll,
                                                       long long n, double difficulty) {
    constexpr int DIM = KIND == TOY_MYLIKE1 ? 3 : 2;
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    double v[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < DIM; ++k) v[k] = x[(long long)k * n + c];
    ll[c] = toy_loglike<KIND>(v, difficulty);
}

template <int KIND>
__global__ __launch_bounds__(256) void toy_run_kernel(double* __restrict__ x, double* __restrict__ stepsizes,
                                                      const uint64_t* __restrict__ ids, long long n,
                                                      double difficulty, const double* __restrict__ betas,
                                                      int nSamples, int nBurnin, int nAdjust, int thinning,
                                                      double* __restrict__ recLL, double* __restrict__ recS,
                                                      double* __restrict__ meanOut, int* __restrict__ nAcceptOut,
                                                      uint64_t* __restrict__ rngIO, int meanContinue) {
    constexpr int DIM = KIND == TOY_MYLIKE1 ? 3 : 2;
    constexpr int NOUT = KIND == TOY_MYLIKE1 ? 4 : 2;                /* + the coordinate "max", test.h:37 */
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    Pcg32 rnd;
    if (rngIO) {                                                     /* a chain's rnd lives across run() calls, mcmc.h:994 */
        rnd.state = rngIO[2 * c];
        rnd.inc = rngIO[2 * c + 1];
    } else {
        pcg32_seed(rnd.state, rnd.inc, 0, ids[c]);                   /* rnd(0, chainid), mcmc.h:999 */
    }
    double cur[3] = {0, 0, 0}, prop[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < DIM; ++k) cur[k] = x[(long long)k * n + c];
    double ll = toy_loglike<KIND>(cur, difficulty);
    double sf = stepsizes[c];
    int nAccept = 0, slot = 0;
    /* mean = deep copy of the evaluated initial state (mcmc.h:1002) */
    double mean[4] = {cur[0], cur[1], cur[2], fmax(fmax(cur[0], cur[1]), cur[2])};
    if (meanOut && meanContinue) {                                   /* a later run() adds to the mean it has */
#pragma unroll
        for (int k = 0; k < NOUT; ++k) mean[k == DIM && NOUT > DIM ? 3 : k] = meanOut[(long long)k * n + c];
    }
    for (int i = 0; i < nSamples + nAdjust; ++i) {                   /* mcmc.h:1023 */
        const double beta = betas[i];
        if (i == nAdjust) nAccept = 0;                               /* mcmc.h:1032 */
        const double curMax = fmax(fmax(cur[0], cur[1]), cur[2]);    /* test.h:37 */
        if (i >= nBurnin && (i % thinning) == 0) {                   /* mcmc.h:1037-1062 */
            double l = 0;
            l += ll;
            if (recLL) recLL[(long long)slot * n + c] = 0.0 + beta * l;
            if (recS) {
#pragma unroll
                for (int k = 0; k < DIM; ++k) recS[((long long)slot * NOUT + k) * n + c] = cur[k];
                if constexpr (NOUT > DIM) recS[((long long)slot * NOUT + DIM) * n + c] = curMax;
            }
            ++slot;
        }
        if (i > nBurnin && meanOut) {                                /* mcmc.h:1064-1065 */
#pragma unroll
            for (int k = 0; k < DIM; ++k) mean[k] += cur[k];
            mean[3] += curMax;
        }
        int reps = 1;
        if (i < nAdjust) { nAccept = 0; reps = 20; }                 /* mcmc.h:1090-1105 */
        for (int j = 0; j < reps; ++j) {
            if (j == 0) (void)rnd.next_float();                      /* step_random_subspace, mcmc.h:498 */
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
                if constexpr (KIND == TOY_MYLIKE1) {
                    prop[k] = cur[k] + 10 * sf * ((double)rnd.next_float() - 0.5);            /* test.h:50 */
                } else if constexpr (KIND == TOY_ROSENBROCK) {
                    prop[k] = cur[k] + sf * ((double)rnd.next_float() - 0.5);                 /* unbounded */
                } else {
                    prop[k] = cur[k] + sf * ((double)rnd.next_float() - 0.5);                 /* test.h:105 */
                    prop[k] = bound_reflect(prop[k], -5.0, 5.0);
                }
            }
            const double llp = toy_loglike<KIND>(prop, difficulty);
            double delta = 0;
            delta += -ll + llp;
            delta += 0.0;
            delta *= beta;
            if ((double)rnd.next_float() < 1.0 * mcmc_exp(delta)) {  /* mcmc.h:776 */
#pragma unroll
                for (int k = 0; k < DIM; ++k) cur[k] = prop[k];
                ll = llp;
                ++nAccept;
            }
        }
        if (i < nAdjust) sf *= rate2corr((double)nAccept / 20);      /* mcmc.h:1107-1114 */
    }
#pragma unroll
    for (int k = 0; k < DIM; ++k) x[(long long)k * n + c] = cur[k];
    stepsizes[c] = sf;
    nAcceptOut[c] = nAccept;
    if (rngIO) rngIO[2 * c] = rnd.state;
    if (meanOut) {                                                   /* mcmc.h:1133-1134 */
        const double f = 1.0 / (double)(nAdjust + nSamples - nBurnin);   /* 1/0 = inf there too */
#pragma unroll
        for (int k = 0; k < DIM; ++k) meanOut[(long long)k * n + c] = mean[k] * f;
        if constexpr (NOUT > DIM) meanOut[(long long)DIM * n + c] = mean[3] * f;
    }
}
