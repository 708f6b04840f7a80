/*
 * mcmc_wave_kernel.hip.h -- the SMALL-ENSEMBLE mapping of the Metropolis-Hastings step: one chain per
 * WAVEFRONT instead of one chain per lane (EXTENSION beside north_star's mapping; same arithmetic, same bits).
 *
 * Why: with one chain per lane BASELINE configs[1] (1 024 chains x 200 days) is 16 wavefronts on a chip of
 * 1 024 SIMDs.  Inside one chain every accumulator buf_c[j] receives its addends n_k * R_c[j-k] in ascending
 * source day k (that order is the reference's, covid.h:376-386, and fixes the bits), but the days of a block and
 * the two integer-shift passes of an evaluation (covid.h:335) are independent of each other.  So LANE l OWNS ONE
 * DAY of a 32-day block of ONE pass -- pass l >> 5, day l & 31 -- and keeps that day's seven sums in registers;
 * both passes run in lockstep, block for block and day for day.
 *
 *   phase A  (sources before the block)   at lag L lane l needs R_c[L] * n[its day - L]: the table value is
 *            wave-uniform, the history value the lane's own.  The lags go in groups of 16: lane n of every 16-lane
 *            row reads the table (curve-major in LDS) at lag (top - n), and v_fmac_f64_dpp row_newbcast:j
 *            multiplies lane j's table value with the lane's own history value of lag (top - j).  Descending lags
 *            are ascending source days.  Groups whose source days carry a hospital-capacity correction
 *            (covid.h:413-429) do that day's three operations after its seven projections, in the reference's
 *            order: n_k * R_c[L], then + e_k (deaths), then - e_k * FromSerious[L].
 *   phase B  (the block's 32 days in order; the feedback n_{i+1} = f(buffers[i]) is sequential, covid.h:461-462)
 *            today's n_i reaches the later days' sums with one v_fmac_f64_dpp row_newbcast per compartment (the
 *            multiplier is the lane's row of a small lag table: lag (its day - today), a row of zeros once its
 *            day is over), and every lane runs the day's scalar arithmetic on its own seven sums -- the lane
 *            whose day it is produces the number the next push broadcasts, a lane whose day has passed
 *            recomputes what it computed then.  Nothing is selected or stored inside the loop; after the block
 *            all 64 lanes store their day's results at once.  ~42 instructions per day for both passes together.
 *
 * (The first version ran phase A on the FP64 matrix cores -- v_mfma_f64_16x16x4_f64 is bit for bit the chain
 * fma(a3,b3,fma(a2,b2,fma(a1,b1,fma(a0,b0,C)))), tools/ubench/mfma_f64.hip -- at 7.0 M steps/s on configs[1];
 * gfx950 issues that instruction every 149 cycles, the vector form here runs 18.6 M.  DESIGN.md 4b has the history.)
 *
 * Everything a chain touches during a launch -- parameter vector, proposal, both histories, weighted deaths --
 * lives in the wave's slice of LDS; HBM is read at the start of a launch and written at its end (and by
 * recording).  The proposal's behaviour nodes, the prior's and the likelihood's terms are computed one per
 * lane and summed in the reference's order.  Both arithmetic modes (template FUSE = cfg.contract; without
 * contraction a term is a DPP move, a multiply and an add), only FP64, only the single-region State.  The host
 * picks the mapping by ensemble size (mcmc_hip_set_mapping); tests/test_gpu_wave_mapping.py holds it bit-exact to
 * the oracle AND to the chain-per-lane kernel.
 */
#pragma once

#define MCMC_WV_T 16          /* lanes of a DPP row: days per HALF block, lags per group of phase A */
#define MCMC_WV_B 32          /* days per block: lane l carries day l & 31 of the block for pass l >> 5 */
#define MCMC_WV_HLO 32        /* zero history entries kept below day 0 (phase A reads down to s - 32) */
#define MCMC_WV_HHI 32        /* ... and above the simulated horizon (phase A reads up to j0 + 30, as zeros; j0 <= horizon) */
#define MCMC_WV_MAX_WAVES 8   /* chains (waves) per workgroup at most: two per SIMD */
#define MCMC_WV_NTAB 10       /* doubles per lag of the table in LDS: MI HI INC MILD SER REC DEAD (one row), -SFS -RFS -DFS (curve-major) */

typedef double double2_t __attribute__((ext_vector_type(2)));
#define MCMC_WV_KROW 12       /* doubles per row of the phase-B lag table: the 8 of Rpos, the 4 of Rneg */
#define MCMC_WV_KROWS (2 * MCMC_WV_B - 1)      /* lags -(B-1) .. B-1 */

__host__ __device__ inline size_t wave_slice_doubles(const DevModel& M, int dim) {
    const size_t histLen = (size_t)M.simDays + MCMC_WV_HLO + MCMC_WV_HHI;
    return 4 * histLen + 2 * (size_t)M.nData + 2 * (size_t)dim;
}
/* rows (lags) of the table in LDS: phase A reads lags up to (days simulated so far) + 32; every lag from
 * nLagRows on is a zero (the d < nDays cut-off of project(), covid.h:378) and reads the one zero row kept for them */
__host__ __device__ inline int wave_tab_rows(const DevModel& M) {
    const int a = M.simDays + MCMC_WV_B + 1, b = M.nLagRows + 1;
    return a < b ? a : b;
}
/* doubles of the part of LDS the waves of a workgroup share (stage_wave_image): the table; capacity,
 * deaths, sigma^2; the segment index (ints); then, 16-byte aligned, the phase-B lag table */
__host__ __device__ inline size_t wave_image_doubles(const DevModel& M) {
    const size_t d = (size_t)MCMC_WV_NTAB * wave_tab_rows(M) + (size_t)M.simPad + 2 * (size_t)M.nData + ((size_t)M.simPad + 1) / 2;
    return (d + 1) / 2 * 2;
}
__host__ __device__ inline size_t wave_shared_doubles(const DevModel& M) {
    return wave_image_doubles(M) + (size_t)MCMC_WV_KROWS * MCMC_WV_KROW;
}
/* LDS of a workgroup of `waves` chains: the shared part + one slice per wave */
__host__ __device__ inline size_t wave_lds_bytes(const DevModel& M, int dim, int waves) {
    return (wave_shared_doubles(M) + (size_t)waves * wave_slice_doubles(M, dim)) * sizeof(double);
}

/* the wave's slice */
struct WaveSlice {
    double* hn[2];     /* [histLen] infections per day of either pass, day k at hn[p][k + MCMC_WV_HLO] */
    double* he[2];     /* [histLen] extra deaths per day                            */
    double* d1;        /* [nData] weighted deaths of the first pass (shift 1)       */
    double* d0;        /* [nData] ... of the second                                 */
    double* terms;     /* [max(nData, nB)] per-day / per-node terms awaiting their ordered sum: the room of he[0], which is
                          dead while the prior's terms (before the evaluation) and the likelihood's (after its days) exist */
    double* xc;        /* [dim] current parameter vector                            */
    double* xp;        /* [dim] proposal                                            */
    int histLen;
};

__device__ __forceinline__ void wave_sync() {
    /* LDS operations of one wave execute in order; this only keeps the compiler from moving accesses across */
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double readlane_f64(double v, int srcLane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srcLane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srcLane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double uniform_f64(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

/* The row exchange of gfx950, as measured by tools/ubench/permlane_swap.hip (rows = groups of 16 lanes):
 *   v_permlane16_swap a, b:  a' = (a.row0, b.row0, a.row2, b.row2)   b' = (a.row1, b.row1, a.row3, b.row3)   */
__device__ __forceinline__ void swap16_f64(double& a, double& b) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}
/* rows 0, 1 <- row 0 and rows 2, 3 <- row 2 of v (EVEN), or rows 0, 1 <- row 1 and rows 2, 3 <- row 3 (odd) */
template <bool EVEN>
__device__ __forceinline__ double row_pair_bcast(double v) {
    double a = v, b = v;
    swap16_f64(a, b);
    return EVEN ? a : b;
}

/* The workgroup's shared image.  (1) The response table: phase A gives lane n of every 16-lane row the table at lag
 * (top - n) and broadcasts it with DPP.  The seven curves of a lag are one row of SEVEN doubles, rt[lag][7] -- one
 * address per group, the curves at constant offsets; rows 56 bytes apart put the 16 lanes of a row on 16 different
 * bank pairs (14 n mod 64 is distinct for n = 0..15).  The three FromSerious curves, used only by the groups with
 * corrections, follow curve-major, rn[c][lag], and negated: fma(e, -g, s) is fma(-e, g, s), bit for bit.  (2) The per-day constants and the
 * observed series, as stage_lds.  (3) The phase-B lag table: row (lag + B - 1) holds, for the lags -(B-1) .. B-1
 * between a destination day of a block and the block's current day, what a destination multiplies today's numbers
 * with -- [0..7] the row of Rpos (today's new infections; zeros for lag < 0: the destination's day is over) and
 * [8..11] the row of Rneg (today's capacity correction; zeros for lag <= 0: today's own sums take theirs from
 * registers, covid.h:413-429).  Zeros make the multiply-add leave a finished sum as it is, which phase B relies on. */
struct WaveImage {
    LdsView lds;              /* capacity, deaths, sigma2, seg (no row-major tables) */
    const double* rt;         /* [rows][7]: MI HI INC MILD SER REC DEAD */
    const double* rn;         /* [3][rows]: -SFS -RFS -DFS */
    const double* lagTab;     /* [MCMC_WV_KROWS][MCMC_WV_KROW] */
    int rows;
};
__device__ inline WaveImage stage_wave_image(const DevModel& M, double* lds_raw) {
    const int rows = wave_tab_rows(M);
    double* rt = lds_raw;
    double* cap = rt + (size_t)MCMC_WV_NTAB * rows;             /* behind rt[rows][7] and rn[3][rows] */
    double* dea = cap + M.simPad;
    double* sg2 = dea + M.nData;
    int* seg = (int*)(sg2 + M.nData);
    double* kt = lds_raw + wave_image_doubles(M);
    double* rn = rt + (size_t)7 * rows;
    for (int i = threadIdx.x; i < 7 * rows; i += blockDim.x) {
        const int L = i / 7, c = i % 7;
        rt[i] = L < M.nLagRows ? M.Rpos[(size_t)L * 8 + c] : 0.0;
    }
    for (int i = threadIdx.x; i < 3 * rows; i += blockDim.x) {
        const int c = i / rows, L = i % rows;
        rn[i] = L < M.nLagRows ? -M.Rneg[(size_t)L * 4 + c] : 0.0;
    }
    for (int i = threadIdx.x; i < M.simPad; i += blockDim.x) {
        cap[i] = M.capacity[i];
        seg[i] = M.seg[i];
    }
    for (int i = threadIdx.x; i < M.nData; i += blockDim.x) {
        dea[i] = M.deaths[i];
        sg2[i] = M.sigma2[i];
    }
    for (int i = threadIdx.x; i < MCMC_WV_KROWS * MCMC_WV_KROW; i += blockDim.x) {
        const int lag = i / MCMC_WV_KROW - (MCMC_WV_B - 1), c = i % MCMC_WV_KROW;
        double v = 0.0;
        if (lag < M.nLagRows) {
            if (c < 8 && lag >= 0) v = M.Rpos[(size_t)lag * 8 + c];
            if (c >= 8 && lag >= 1) v = M.Rneg[(size_t)lag * 4 + (c - 8)];
        }
        kt[i] = v;
    }
    __syncthreads();
    WaveImage I;
    I.lds.Rpos = nullptr; I.lds.Rneg = nullptr; I.lds.capacity = cap; I.lds.deaths = dea; I.lds.sigma2 = sg2; I.lds.seg = seg;
    I.lds.tabF = nullptr; I.lds.tabN = nullptr; I.lds.rowsB = nullptr; I.lds.rowsNB = nullptr; I.lds.tr = nullptr;
    I.rt = rt; I.rn = rn; I.lagTab = kt; I.rows = rows;
    return I;
}

/* Phase A, 16 lags of all 32 days x 2 passes: S[c] += (table of lane j of the lane's own row) * h[j], j = 0..15 in
 * order, for the seven compartments.  Lane n of every row holds the table at lag (top - n), h[j] is the lane's
 * own source day of lag (top - j): descending lags are ascending source days, the reference's order. */
#define MCMC_WV_F(D, T, H, L) "v_fmac_f64_dpp " D ", " T ", " H " row_newbcast:" #L " row_mask:0xf bank_mask:0xf\n\t"
#define MCMC_WV_LAG7(H, L)                                                                                         \
    MCMC_WV_F("%0", "%[t0]", H, L) MCMC_WV_F("%1", "%[t1]", H, L) MCMC_WV_F("%2", "%[t2]", H, L)                   \
    MCMC_WV_F("%3", "%[t3]", H, L) MCMC_WV_F("%4", "%[t4]", H, L) MCMC_WV_F("%5", "%[t5]", H, L)                   \
    MCMC_WV_F("%6", "%[t6]", H, L)
#define MCMC_WV_S7 "+v"(S[0]), "+v"(S[1]), "+v"(S[2]), "+v"(S[3]), "+v"(S[4]), "+v"(S[5]), "+v"(S[6])
#define MCMC_WV_T7 [t0] "v"(t[0]), [t1] "v"(t[1]), [t2] "v"(t[2]), [t3] "v"(t[3]), [t4] "v"(t[4]), [t5] "v"(t[5]), [t6] "v"(t[6])
__device__ __forceinline__ void wave_lags16(double (&S)[7], const double (&t)[7], const double* h) {
    asm("s_nop 1\n\t" MCMC_WV_LAG7("%[h0]", 0) MCMC_WV_LAG7("%[h1]", 1) MCMC_WV_LAG7("%[h2]", 2) MCMC_WV_LAG7("%[h3]", 3)
        MCMC_WV_LAG7("%[h4]", 4) MCMC_WV_LAG7("%[h5]", 5) MCMC_WV_LAG7("%[h6]", 6) MCMC_WV_LAG7("%[h7]", 7)
        MCMC_WV_LAG7("%[h8]", 8) MCMC_WV_LAG7("%[h9]", 9) MCMC_WV_LAG7("%[h10]", 10) MCMC_WV_LAG7("%[h11]", 11)
        MCMC_WV_LAG7("%[h12]", 12) MCMC_WV_LAG7("%[h13]", 13) MCMC_WV_LAG7("%[h14]", 14) MCMC_WV_LAG7("%[h15]", 15)
        : MCMC_WV_S7
        : MCMC_WV_T7, [h0] "v"(h[0]), [h1] "v"(h[1]), [h2] "v"(h[2]), [h3] "v"(h[3]), [h4] "v"(h[4]), [h5] "v"(h[5]),
          [h6] "v"(h[6]), [h7] "v"(h[7]), [h8] "v"(h[8]), [h9] "v"(h[9]), [h10] "v"(h[10]), [h11] "v"(h[11]),
          [h12] "v"(h[12]), [h13] "v"(h[13]), [h14] "v"(h[14]), [h15] "v"(h[15]));
}
/* the same for four lags whose source days may carry capacity corrections: after the seven projections of a source
 * day come + e (deaths; m = 1 within the table's lags, the flat add() of covid.h:387-391) and the three negative
 * projections of e (q = the negated FromSerious curves), in the reference's order (covid.h:413-429) */
#define MCMC_WV_LAG_E(H, E, L)                                                                                     \
    MCMC_WV_LAG7(H, L) MCMC_WV_F("%6", "%[m]", E, L) MCMC_WV_F("%4", "%[q0]", E, L) MCMC_WV_F("%5", "%[q1]", E, L) \
    MCMC_WV_F("%6", "%[q2]", E, L)
#define MCMC_WV_LAGS4_E(L0, L1, L2, L3)                                                                            \
    asm("s_nop 1\n\t" MCMC_WV_LAG_E("%[h0]", "%[e0]", L0) MCMC_WV_LAG_E("%[h1]", "%[e1]", L1)                       \
        MCMC_WV_LAG_E("%[h2]", "%[e2]", L2) MCMC_WV_LAG_E("%[h3]", "%[e3]", L3)                                     \
        : MCMC_WV_S7                                                                                               \
        : MCMC_WV_T7, [m] "v"(m), [q0] "v"(q[0]), [q1] "v"(q[1]), [q2] "v"(q[2]), [h0] "v"(h[0]), [h1] "v"(h[1]),    \
          [h2] "v"(h[2]), [h3] "v"(h[3]), [e0] "v"(e[0]), [e1] "v"(e[1]), [e2] "v"(e[2]), [e3] "v"(e[3]))
template <int J0>
__device__ __forceinline__ void wave_lags4_e(double (&S)[7], const double (&t)[7], const double (&q)[3], double m,
                                             const double* h, const double* e) {
    if constexpr (J0 == 0) MCMC_WV_LAGS4_E(0, 1, 2, 3);
    else if constexpr (J0 == 4) MCMC_WV_LAGS4_E(4, 5, 6, 7);
    else if constexpr (J0 == 8) MCMC_WV_LAGS4_E(8, 9, 10, 11);
    else MCMC_WV_LAGS4_E(12, 13, 14, 15);
}

/* cfg.contract = 0: every term is a multiply and an add (covid.h:384 as the C++ abstract machine has it).  The
 * DPP form exists for the fused instruction only (v_mul_f64 / v_add_f64 take no DPP source on gfx950), so the
 * broadcast is a v_mov_b64_dpp of its own: three instructions per term instead of one. */
template <int L>
__device__ __forceinline__ void bcast7(double (&b)[7], const double (&t)[7]) {
#define MCMC_WV_MOV(D, S) "v_mov_b64_dpp " D ", " S " row_newbcast:%[lane] row_mask:0xf bank_mask:0xf\n\t"
    asm("s_nop 1\n\t" MCMC_WV_MOV("%0", "%7") MCMC_WV_MOV("%1", "%8") MCMC_WV_MOV("%2", "%9") MCMC_WV_MOV("%3", "%10")
        MCMC_WV_MOV("%4", "%11") MCMC_WV_MOV("%5", "%12") MCMC_WV_MOV("%6", "%13")
        : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]), "=&v"(b[4]), "=&v"(b[5]), "=&v"(b[6])
        : "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6]), [lane] "n"(L));
}
template <int L>
__device__ __forceinline__ double bcast1(double v) {
    double o;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%[lane] row_mask:0xf bank_mask:0xf" : "=&v"(o) : "v"(v), [lane] "n"(L));
    return o;
}
#undef MCMC_WV_MOV
template <int L>
__device__ __forceinline__ void strict_lag(double (&S)[7], const double (&t)[7], double h) {
    double b[7];
    bcast7<L>(b, t);
#pragma unroll
    for (int c = 0; c < 7; ++c) S[c] = S[c] + h * b[c];
}
template <int L>
__device__ __forceinline__ void strict_lag_e(double (&S)[7], const double (&t)[7], const double (&q)[3], double m, double h, double e) {
    strict_lag<L>(S, t, h);
    const double bm = bcast1<L>(m), b0 = bcast1<L>(q[0]), b1 = bcast1<L>(q[1]), b2 = bcast1<L>(q[2]);
    S[6] = S[6] + e * bm;
    S[4] = S[4] + e * b0;
    S[5] = S[5] + e * b1;
    S[6] = S[6] + e * b2;
}
__device__ __forceinline__ void strict_lags16(double (&S)[7], const double (&t)[7], const double* h) {
    strict_lag<0>(S, t, h[0]); strict_lag<1>(S, t, h[1]); strict_lag<2>(S, t, h[2]); strict_lag<3>(S, t, h[3]);
    strict_lag<4>(S, t, h[4]); strict_lag<5>(S, t, h[5]); strict_lag<6>(S, t, h[6]); strict_lag<7>(S, t, h[7]);
    strict_lag<8>(S, t, h[8]); strict_lag<9>(S, t, h[9]); strict_lag<10>(S, t, h[10]); strict_lag<11>(S, t, h[11]);
    strict_lag<12>(S, t, h[12]); strict_lag<13>(S, t, h[13]); strict_lag<14>(S, t, h[14]); strict_lag<15>(S, t, h[15]);
}
__device__ __forceinline__ void strict_lags16_e(double (&S)[7], const double (&t)[7], const double (&q)[3], double m,
                                                const double* h, const double* e) {
    strict_lag_e<0>(S, t, q, m, h[0], e[0]); strict_lag_e<1>(S, t, q, m, h[1], e[1]);
    strict_lag_e<2>(S, t, q, m, h[2], e[2]); strict_lag_e<3>(S, t, q, m, h[3], e[3]);
    strict_lag_e<4>(S, t, q, m, h[4], e[4]); strict_lag_e<5>(S, t, q, m, h[5], e[5]);
    strict_lag_e<6>(S, t, q, m, h[6], e[6]); strict_lag_e<7>(S, t, q, m, h[7], e[7]);
    strict_lag_e<8>(S, t, q, m, h[8], e[8]); strict_lag_e<9>(S, t, q, m, h[9], e[9]);
    strict_lag_e<10>(S, t, q, m, h[10], e[10]); strict_lag_e<11>(S, t, q, m, h[11], e[11]);
    strict_lag_e<12>(S, t, q, m, h[12], e[12]); strict_lag_e<13>(S, t, q, m, h[13], e[13]);
    strict_lag_e<14>(S, t, q, m, h[14], e[14]); strict_lag_e<15>(S, t, q, m, h[15], e[15]);
}
/* today's number, from lane srcLane of the lane's own row, times the lane's lag-table row, strict form */
__device__ __forceinline__ void strict_push7(double (&S)[7], int srcLane, double nb, const double (&k)[7]) {
    double b;
    switch (srcLane) {                     /* a constant wherever this is called: the day loop is unrolled */
        case 0: b = bcast1<0>(nb); break;   case 1: b = bcast1<1>(nb); break;   case 2: b = bcast1<2>(nb); break;
        case 3: b = bcast1<3>(nb); break;   case 4: b = bcast1<4>(nb); break;   case 5: b = bcast1<5>(nb); break;
        case 6: b = bcast1<6>(nb); break;   case 7: b = bcast1<7>(nb); break;   case 8: b = bcast1<8>(nb); break;
        case 9: b = bcast1<9>(nb); break;   case 10: b = bcast1<10>(nb); break; case 11: b = bcast1<11>(nb); break;
        case 12: b = bcast1<12>(nb); break; case 13: b = bcast1<13>(nb); break; case 14: b = bcast1<14>(nb); break;
        default: b = bcast1<15>(nb); break;
    }
#pragma unroll
    for (int c = 0; c < 7; ++c) S[c] = S[c] + b * k[c];
}

/* S[c] += (nb of lane L of the lane's own row) * k[c] for the seven compartments: v_fmac_f64 with a DPP source is the
 * same fused operation as v_fmac_f64.  The s_nop covers the two wait states a DPP source needs after a VALU
 * write, which the compiler cannot see into the asm to provide.  All 64 lanes are active wherever this is used. */
#define MCMC_WV_FMAC(D, K, L) "v_fmac_f64_dpp " D ", %[nb], " K " row_newbcast:" #L " row_mask:0xf bank_mask:0xf\n\t"
#define MCMC_WV_PUSH_CASE(L)                                                                                            \
    case L:                                                                                                             \
        asm("s_nop 1\n\t" MCMC_WV_FMAC("%0", "%[k0]", L) MCMC_WV_FMAC("%1", "%[k1]", L) MCMC_WV_FMAC("%2", "%[k2]", L)  \
            MCMC_WV_FMAC("%3", "%[k3]", L) MCMC_WV_FMAC("%4", "%[k4]", L) MCMC_WV_FMAC("%5", "%[k5]", L)                \
            MCMC_WV_FMAC("%6", "%[k6]", L)                                                                              \
            : "+v"(S[0]), "+v"(S[1]), "+v"(S[2]), "+v"(S[3]), "+v"(S[4]), "+v"(S[5]), "+v"(S[6])                        \
            : [nb] "v"(nb), [k0] "v"(k0), [k1] "v"(k1), [k2] "v"(k2), [k3] "v"(k3), [k4] "v"(k4), [k5] "v"(k5),          \
              [k6] "v"(k6));                                                                                            \
        break;
__device__ __forceinline__ void wave_push7(double (&S)[7], int srcLane, double nb, double k0, double k1, double k2,
                                           double k3, double k4, double k5, double k6) {
    switch (srcLane) {                     /* a constant wherever this is called: the day loop is unrolled */
        MCMC_WV_PUSH_CASE(0) MCMC_WV_PUSH_CASE(1) MCMC_WV_PUSH_CASE(2) MCMC_WV_PUSH_CASE(3)
        MCMC_WV_PUSH_CASE(4) MCMC_WV_PUSH_CASE(5) MCMC_WV_PUSH_CASE(6) MCMC_WV_PUSH_CASE(7)
        MCMC_WV_PUSH_CASE(8) MCMC_WV_PUSH_CASE(9) MCMC_WV_PUSH_CASE(10) MCMC_WV_PUSH_CASE(11)
        MCMC_WV_PUSH_CASE(12) MCMC_WV_PUSH_CASE(13) MCMC_WV_PUSH_CASE(14)
        default: MCMC_WV_PUSH_CASE(15)
    }
}
#undef MCMC_WV_PUSH_CASE

/* sum = (((sum + sgn t[0]) + sgn t[1]) + ...) + sgn t[count-1], sgn = +1 or -1: the reference's running sums (covid.h:488-492,
 * mcmc.h:197-214) are ordered, so this is one dependent chain -- of v_fmac_f64_dpp, which IS the add (x * 1 is exact),
 * with the addend of step i broadcast from lane i & 15 of each row: every row reads the same 16 terms from LDS, so
 * every lane runs the same chain and ends with the same sum.  Eight groups of 16 are requested before their 128 adds,
 * which keeps LDS latency off the chain.  t[count .. count + 15] must be zeros (a ragged last group adds them). */
__device__ __forceinline__ void ordered_add16(double& sum, double t, double sgn) {
#define MCMC_WV_ADD(L) "v_fmac_f64_dpp %0, %[t], %[s] row_newbcast:" #L " row_mask:0xf bank_mask:0xf\n\t"
    asm("s_nop 1\n\t" MCMC_WV_ADD(0) MCMC_WV_ADD(1) MCMC_WV_ADD(2) MCMC_WV_ADD(3) MCMC_WV_ADD(4) MCMC_WV_ADD(5) MCMC_WV_ADD(6)
        MCMC_WV_ADD(7) MCMC_WV_ADD(8) MCMC_WV_ADD(9) MCMC_WV_ADD(10) MCMC_WV_ADD(11) MCMC_WV_ADD(12) MCMC_WV_ADD(13)
        MCMC_WV_ADD(14) MCMC_WV_ADD(15)
        : "+v"(sum) : [t] "v"(t), [s] "v"(sgn));
#undef MCMC_WV_ADD
}
__device__ __forceinline__ double ordered_sum(double sum, const double* t, int count, unsigned lane, double sgn) {
    const double* tp = t + (lane & 15);
#pragma unroll 1
    for (int base = 0; base < count; base += 128) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tp[base + 16 * j];           /* beyond count: inside the wave's slice, not added */
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (base + 16 * j < count) ordered_add16(sum, v[j], sgn);
    }
    return sum;
}

/* x / pop (covid.h:461).  hipcc's double division is: scale both operands if an exponent is extreme (v_div_scale),
 * y = two Newton steps on v_rcp of the denominator, q = x y, r = fma(-pop, q, x), fma(r, y, q), un-scale (v_div_fmas),
 * special cases (v_div_fixup).  pop never changes, so y is formed once per launch (wave_rcp_pop, the same
 * instructions), and where no scaling happens -- frexp exponent of x in [ELO, EHI], pop of ordinary size -- the
 * remaining four instructions ARE the division, bit for bit.  eval_model_wave checks every numerator it used and
 * has the whole evaluation redone with the compiler's division if one fell outside (never, on a sane epidemic). */
#define MCMC_WV_DIV_ELO (-900)
#define MCMC_WV_DIV_EHI 700
__device__ __forceinline__ double wave_rcp_pop(double pop) {
    double y = __builtin_amdgcn_rcp(pop);
    double e = __builtin_fma(-pop, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-pop, y, 1.0);
    y = __builtin_fma(y, e, y);
    return y;
}
__device__ __forceinline__ bool wave_fast_div_ok(double pop) {
    const int ex = __builtin_amdgcn_frexp_exp(pop);
    return ex > -100 && ex < 100;
}
template <bool FASTDIV>
__device__ __forceinline__ double wave_div_pop(double x, double pop, double rcpPop) {
    if constexpr (FASTDIV) {
        const double q = x * rcpPop;
        const double r = __builtin_fma(-pop, q, x);
        return __builtin_amdgcn_div_fixup(__builtin_fma(r, rcpPop, q), pop, x);
    } else {
        return x / pop;
    }
}

/*
 * DiseaseSpread::eval (covid.h:325-496) of the parameter vector W.xp for the chain of this WAVE.  Every lane
 * returns the same loglike.  Same operations on every accumulator, in the same order, as eval_model<T, true>.
 *
 * The reference's two integer-shift passes (covid.h:335: first day start + 1, then start) are independent until
 * their likelihood terms are summed, so they run here IN LOCKSTEP, block for block and day for day: pass index
 * p = 0 is shift 1 (the first in the reference's order), p = 1 is shift 0.  Two independent matrix-instruction
 * chains keep the matrix pipe at its throughput instead of one chain's latency (116 against 62 ns per
 * instruction, tools/ubench/mfma_f64.hip), and the two sequential day chains fill each other's latency gaps.
 * The day loop is straight-line for that reason (its one branch, a capacity correction today, is wave-uniform).
 */
template <bool FASTDIV, bool FUSE>
__device__ __forceinline__ double eval_model_wave(const DevModel& M, const WaveImage& I, const WaveSlice& W,
                                                  const LaneScalars& lp, unsigned lane, WorkCount& work, PhaseClock& pc,
                                                  double rcpPop, bool& redo) {
    const LdsView& lds = I.lds;
    const double* const lagTab = I.lagTab;
    const int md = M.md;
    const int horizon = M.simDays;
    const int n16 = (int)(lane & 15);                          /* the lane's place in its DPP row */
    const int rowsM1 = I.rows - 1;
    const double* xdisc = W.xp + M.nB;

    int start = (int)((double)md - lp.delay);                       /* covid.h:331 */
    const double frac = (double)md - lp.delay - (double)start;      /* covid.h:332 */
    start = min(max(start, 0), horizon);
    start = __builtin_amdgcn_readfirstlane(start);
    count_work(M, start, work);

    const bool hiHalf = lane >= 32;                              /* phase B: the pass whose days the lane carries ... */
    const int dayInBlk = (int)(lane & 31);                       /* ... and which day of a block */
    int sP[2];
    double wP[2];
    double nb = 1.0;                                              /* what today's push broadcasts; patient zero on day s, covid.h:339 */
    int keLo[2], keHi[2];                                           /* source days with a capacity correction */
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int shift = 1 - p;
        wP[p] = (double)(1 - shift) * (1.0 - frac) + (double)shift * frac;      /* covid.h:333 */
        sP[p] = min(start + shift, horizon);
        keLo[p] = 0x7fffffff; keHi[p] = -1;
    }
    double* const hnDst = (hiHalf ? W.hn[1] : W.hn[0]) + MCMC_WV_HLO;
    double* const heDst = (hiHalf ? W.he[1] : W.he[0]) + MCMC_WV_HLO;
    double* const dsDst = hiHalf ? W.d0 : W.d1;
    const double wBoth = hiHalf ? wP[1] : wP[0];
    const double* const kLane = lagTab + dayInBlk * MCMC_WV_KROW;  /* the lane's row of lag (its day - day t of the block) is kLane + (B-1-t) rows */
    const double ng0S = M.Rneg[0], ng0R = M.Rneg[1], ng0D = M.Rneg[2];            /* FromSerious at lag 0: today's own sums */
    for (int k = (int)lane; k < W.histLen; k += MCMC_WAVE) {
        W.hn[0][k] = 0.0; W.he[0][k] = 0.0; W.hn[1][k] = 0.0; W.he[1][k] = 0.0;
    }
    wave_sync();
    pc.stamp(PH_SETUP);

    unsigned long long redoMask = 0;
#pragma unroll 1
    for (int blk = 0; sP[1] + blk * MCMC_WV_B < horizon; ++blk) {
        const int j0P[2] = {sP[0] + blk * MCMC_WV_B, sP[1] + blk * MCMC_WV_B};
        const int iB = (hiHalf ? j0P[1] : j0P[0]) + dayInBlk;                   /* the lane's day */
        /* ---------- phase A: sources k < j0, by descending lag (= ascending source day, the reference's order,
         * covid.h:376-386).  Lane l owns the seven sums of ITS day; at lag L its source day is iB - L.  Entries >= j0
         * of the history are still zero (phase B stores them later), entries below the first day are zero pads,
         * so neither the in-block sources nor the days before patient zero need a mask.  The lags 1 .. 32 blk + 32
         * (the top one only meets zero pads) go in groups of 16: lane n of every row reads the table at lag
         * (top - n), DPP broadcasts it.  Groups that hold a source day with a capacity correction in EITHER pass
         * take the form with the correction's operations (a correction of zero changes nothing). */
        double Sx[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (blk > 0) {
            int eLo = 0x7fffffff, eHi = -1;                          /* the lags with corrections, both passes */
#pragma unroll
            for (int p = 0; p < 2; ++p)
                if (keHi[p] >= keLo[p]) {
                    eLo = min(eLo, j0P[p] - keHi[p]);
                    eHi = max(eHi, j0P[p] + (MCMC_WV_B - 1) - keLo[p]);
                }
            const double* const hN = hnDst + iB;                     /* source day of lag L: hN[-L] */
            const double* const hE = heDst + iB;
            struct LagOps { double t[7]; double h[MCMC_WV_T]; };
            auto load_ops = [&](int top, LagOps& o) {
                const double* tp = I.rt + min(top - n16, rowsM1) * 7;
#pragma unroll
                for (int c = 0; c < 7; ++c) o.t[c] = tp[c];
                const double* hp = hN - top;
#pragma unroll
                for (int j = 0; j < MCMC_WV_T; ++j) o.h[j] = hp[j];
            };
            auto run_plain = [&](const LagOps& o) {
                pc.count(PH_NCHUNK, 1);
                if constexpr (FUSE) {
                    wave_lags16(Sx, o.t, o.h);
                } else {
                    strict_lags16(Sx, o.t, o.h);
                }
            };
            auto run_e = [&](int top, const LagOps& o) {
                pc.count(PH_NCHUNK_E, 1);
                const int lagN = top - n16;
                const double* tp = I.rn + min(lagN, rowsM1);
                const double q[3] = {tp[0], tp[I.rows], tp[2 * I.rows]};
                const double m = lagN < M.nLagRows ? 1.0 : 0.0;
                double e[MCMC_WV_T];
                const double* ep = hE - top;
#pragma unroll
                for (int j = 0; j < MCMC_WV_T; ++j) e[j] = ep[j];
                if constexpr (FUSE) {
                    wave_lags4_e<0>(Sx, o.t, q, m, o.h, e);
                    wave_lags4_e<4>(Sx, o.t, q, m, o.h + 4, e + 4);
                    wave_lags4_e<8>(Sx, o.t, q, m, o.h + 8, e + 8);
                    wave_lags4_e<12>(Sx, o.t, q, m, o.h + 12, e + 12);
                } else {
                    strict_lags16_e(Sx, o.t, q, m, o.h, e);
                }
            };
            /* The groups (tops 32 blk + 32, ..., 32, 16) form three runs -- plain, with corrections, plain -- each a
             * tight loop of its own.  A plain run takes two groups per trip, each one's operands requested while the
             * other's multiply-adds run; a run of odd length starts with a single group. */
            int top = MCMC_WV_B * blk + MCMC_WV_B;
            int eTop = 0, eBot = 1;                                  /* tops of the groups with corrections: eTop >= top' >= eBot */
            if (eHi >= eLo) {
                eTop = min(top, (eHi + MCMC_WV_T - 1) / MCMC_WV_T * MCMC_WV_T);         /* the group that holds lag eHi */
                eBot = max((max(eLo, 1) + MCMC_WV_T - 1) / MCMC_WV_T * MCMC_WV_T, MCMC_WV_T);
            }
            LagOps oa, ob;
            auto plain_run = [&](int stop) {                         /* the groups with stop <= top' <= top */
                if (top < stop) return;
                load_ops(top, oa);
                if (((top - stop) / MCMC_WV_T) % 2 == 0) {           /* odd number of groups */
                    run_plain(oa);
                    top -= MCMC_WV_T;
                    if (top < stop) return;
                    load_ops(top, oa);
                }
#pragma unroll 1
                for (; top >= stop; top -= 2 * MCMC_WV_T) {
                    load_ops(top - MCMC_WV_T, ob);
                    __builtin_amdgcn_sched_barrier(0);               /* each request stays in front of the other group's multiply-adds */
                    run_plain(oa);
                    load_ops(max(top - 2 * MCMC_WV_T, MCMC_WV_T), oa);   /* the last trip's is not used */
                    __builtin_amdgcn_sched_barrier(0);
                    run_plain(ob);
                }
            };
            plain_run(eTop >= eBot ? eTop + MCMC_WV_T : MCMC_WV_T);
            if (eTop >= eBot) {
#pragma unroll 1
                for (; top >= eBot; top -= MCMC_WV_T) {
                    load_ops(top, oa);
                    run_e(top, oa);
                }
                plain_run(MCMC_WV_T);
            }
        }
        pc.stamp(PH_CHUNKS);
        /* ---------- phase B: the block's days in order, both passes day for day.  Lane l's sums only change while
         * its day is not over (the lag table holds zeros for the days that are), and whatever it computes from
         * them on the days after its own it computes again, the same: so every lane runs the day arithmetic on
         * every day, the lane whose day it is feeds the next push, and the results are stored once, from all
         * lanes, after the block. */
        const bool live = iB < horizon;                                          /* days beyond compute garbage nobody reads */
        /* the block's first day enters the history only now: phase A takes a source day k >= j0 for zero, which
         * keeps the in-block sources (phase B's) out of it without a mask.  nb of lane 31 / 63 is that day's n */
        if (dayInBlk == MCMC_WV_B - 1 && iB - (MCMC_WV_B - 1) < horizon) hnDst[iB - (MCMC_WV_B - 1)] = nb;
        /* what the day needs besides the sums: capacity and pcf * smoothbehavior (covid.h:432-454) */
        double capL, tbL;
        {
            const int iC = min(iB, M.simPad - 1);
            capL = live ? lds.capacity[iC] : __builtin_inf();      /* a day beyond the horizon never has a correction */
            const int sg = lds.seg[iC];
            const double pcf = sg >= 0 ? xdisc[max(sg, 0)] : 1.0;
            double smoothbehavior = 1.0;
            if (iC >= md) smoothbehavior = pcf * W.xp[min(max(iC - md, 0), M.nB - 1)];
            tbL = pcf * smoothbehavior;
        }
        double nn = 0.0, eL = 0.0, deadL = 0.0, numL = 0.0;
        /* the lane's table row of a day is requested a day ahead: LDS latency stays off the sequential chain */
        struct LagRow { double2_t k01, k23, k45; double k6; };
        auto lag_row = [&](int t) {
            const double* kr = kLane + (MCMC_WV_B - 1 - t) * MCMC_WV_KROW;
            LagRow r;
            r.k01 = *(const double2_t*)(kr); r.k23 = *(const double2_t*)(kr + 2); r.k45 = *(const double2_t*)(kr + 4);
            r.k6 = kr[6];
            return r;
        };
        LagRow kc = lag_row(0);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half == 1 && !(j0P[1] + MCMC_WV_T < horizon)) break;            /* wave-uniform: nothing of the second half is simulated */
#pragma unroll
            for (int u = 0; u < MCMC_WV_T; ++u) {
                const int t = half * MCMC_WV_T + u;
                const double* kr = kLane + (MCMC_WV_B - 1 - t) * MCMC_WV_KROW;
                const LagRow kn = lag_row(min(t + 1, MCMC_WV_B - 1));
                /* today's new infections into today and the rest of the block (covid.h:403-410) */
                if constexpr (FUSE) {
                    wave_push7(Sx, (u + MCMC_WV_T - 1) & (MCMC_WV_T - 1), nb, kc.k01[0], kc.k01[1], kc.k23[0], kc.k23[1],
                               kc.k45[0], kc.k45[1], kc.k6);
                } else {
                    const double k7[7] = {kc.k01[0], kc.k01[1], kc.k23[0], kc.k23[1], kc.k45[0], kc.k45[1], kc.k6};
                    strict_push7(Sx, (u + MCMC_WV_T - 1) & (MCMC_WV_T - 1), nb, k7);
                }
                kc = kn;
                const double vMI = Sx[0], vHI = Sx[1], vINC = Sx[2], vMILD = Sx[3];
                double vSER = Sx[4], vREC = Sx[5], vDEAD = Sx[6];
                /* hospital capacity (covid.h:413-429).  Straight-line: without a correction e = +0, and adding
                 * +0 or the product -0 * x changes no sum (none of them is ever -0) */
                const double overCapacity = vSER * M.probICU - capL;
                const double extraDeaths = M.probLethal * overCapacity;
                const unsigned long long gt = __builtin_amdgcn_fcmp(extraDeaths, 0.0, 2 /* ordered > */);
                eL = extraDeaths > 0 ? extraDeaths : 0.0;
                const double ne = -eL;
                vDEAD = vDEAD + eL;                                                 /* today's own sums */
                vSER = madd<FUSE>(ne, ng0S, vSER);
                vREC = madd<FUSE>(ne, ng0R, vREC);
                vDEAD = madd<FUSE>(ne, ng0D, vDEAD);
                deadL = vDEAD;
                /* tomorrow's new infections (covid.h:461-462); asymptomatic == mild buffer */
                const double nSus = M.pop - vINC - vMILD - vMILD - vSER - vREC - vDEAD;
                numL = tbL * nSus;
                const double perCapita = wave_div_pop<FASTDIV>(numL, M.pop, rcpPop), force = lp.betaMild * vMI + lp.betaHigh * vHI;
                nn = perCapita * force;
                double nn2 = 0.0;                                                   /* the same product in a second register: see below */
                if (half == 0)      /* (hipcc pads an asm's outputs like any VALU write: the two wait states before v_permlane16_swap are its) */
                    asm("v_mul_f64 %0, %1, %2" : "=v"(nn2) : "v"(perCapita), "v"(force));
                /* The later days' sums: + e (deaths), then the projections.  The test is "a correction in ANY lane" (one
                 * branch on the compare's mask): a lane whose day has passed keeps its correction in the mask, so from
                 * a block's first correction on the branch is taken for nothing on some days -- today's e is zero
                 * then and changes nothing (measured against a test of today's two lanes: 4 % faster on configs[1]).
                 * The branch stays behind the day's arithmetic, where the old values of the sums it changes are dead
                 * (hipcc otherwise sinks the arithmetic below it and copies the three sums on the common path). */
                asm volatile("" : "+v"(nn), "+v"(nn2), "+v"(deadL), "+v"(eL));
                if (gt) {
                    const double e0 = readlane_f64(eL, t), e1 = readlane_f64(eL, 32 + t);
                    const double eb = hiHalf ? e1 : e0, neb = -eb;
                    const double2_t n01 = *(const double2_t*)(kr + 8), n23 = *(const double2_t*)(kr + 10);
                    Sx[6] = madd<FUSE>(eb, n23[1], Sx[6]);                          /* the flat add: 1 within the table's lags */
                    Sx[4] = madd<FUSE>(neb, n01[0], Sx[4]);
                    Sx[5] = madd<FUSE>(neb, n01[1], Sx[5]);
                    Sx[6] = madd<FUSE>(neb, n23[0], Sx[6]);
                }
                /* tomorrow's push reads lane u of every row: the second half's lanes get the first half's number (their
                 * own nn is not theirs yet).  v_permlane16_swap exchanges between two registers, hence the second product */
                if (half == 0) {
                    swap16_f64(nn, nn2);
                }
                nb = nn;
            }
        }
        nb = row_pair_bcast<false>(nn);                                             /* day 31's, for the next block's day 0 */
        if (dayInBlk < MCMC_WV_B - 1 && iB + 1 < horizon) hnDst[iB + 1] = nn;     /* day 31's: by the next block, see above */
        if (live) {
            heDst[iB] = eL;
            if (iB >= md && iB - md < M.nData) dsDst[iB - md] = wBoth * deadL;      /* covid.h:475,484 */
        }
        {
            const unsigned long long hm = __builtin_amdgcn_ballot_w64(eL > 0);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const unsigned m32 = (unsigned)(hm >> (32 * p));
                if (m32) {
                    keLo[p] = min(keLo[p], j0P[p] + (int)__builtin_ctz(m32));
                    keHi[p] = max(keHi[p], j0P[p] + 31 - (int)__builtin_clz(m32));
                }
            }
        }
        if constexpr (FASTDIV) {
            /* did every division of this block stay where the short form is the long one?  (each lane's last numerator
             * is its own day's; frexp gives 0 for zero, infinity and NaN, which v_div_fixup settles in both forms) */
            const unsigned ex = (unsigned)(__builtin_amdgcn_frexp_exp(numL) - MCMC_WV_DIV_ELO);
            redoMask |= __builtin_amdgcn_ballot_w64(live && ex > (unsigned)(MCMC_WV_DIV_EHI - MCMC_WV_DIV_ELO));
        }
        pc.stamp(PH_DAYS);
        wave_sync();
    }
    /* log-likelihood (covid.h:488-492): the first pass's terms, then the second's, one term per lane, each pass summed
     * in day order onto the same sum */
    double loglike = 0;                                             /* covid.h:329 */
#pragma unroll 1
    for (int p = 0; p < 2; ++p) {
        const double w = uniform_f64(wP[p]);
        for (int i = (int)lane; i < M.nData; i += MCMC_WAVE) {
            double dv = W.d1[i];
            if (p == 1) dv = dv + W.d0[i];
            const double delta = dv - lds.deaths[i];
            W.terms[i] = w * (-0.5 * delta * delta / lds.sigma2[i]);
        }
        if (lane < 16) W.terms[M.nData + lane] = 0.0;
        wave_sync();
        loglike = ordered_sum(loglike, W.terms, M.nData, lane, 1.0);
        wave_sync();
    }
    pc.stamp(PH_LOGLIKE);
    redo = redoMask != 0;
    return loglike;
}

/*
 * DiseaseSpread::step up to the eval call (covid.h:499-622) for the chain of this wave: the draws and the scalar
 * moves run uniformly in every lane, the behaviour nodes one per lane.  Mirrors propose() line by line.
 */
__device__ inline bool propose_wave(const DevModel& M, Pcg32& rnd, const WaveSlice& W, unsigned lane, double stepsize,
                                    LaneScalars& lp, double& priorOut) {
    const double* xc = W.xc;
    double* xp = W.xp;
    lp.betaMild = xc[M.nB + M.nDisc];
    lp.betaHigh = xc[M.nB + M.nDisc + 1];
    lp.delay = xc[M.nB + M.nDisc + 2];
    double disc[MCMC_MAXDISC];
#pragma unroll
    for (int k = 0; k < MCMC_MAXDISC; ++k) disc[k] = (k < M.nDisc) ? xc[M.nB + k] : 1.0;
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
    doB = __builtin_amdgcn_readfirstlane(doB ? 1 : 0) != 0;
    priorOut = 0.0;
    const int nB = M.nB;
    if (doB) {
        for (int i = (int)lane; i < nB; i += MCMC_WAVE)
            xp[i] = bound_reflect(xc[i] + A * (mcmc_cos(omega * (double)i) - 1), 0.0, 2.0);
        wave_sync();
        /* SmoothnessPrior::eval of the proposal (mcmc.h:197-214): the terms one per lane, their sum in node order */
        const double dx = M.smoothL / (double)nB;
        const bool dxPow2 = is_pow2(dx);
        const double invDx = 1.0 / dx;
        for (int i = 2 + (int)lane; i < nB; i += MCMC_WAVE) {
            const double v = xp[i], f0 = xp[i - 1], fm = xp[i - 2];
            const double dd = v + fm - 2 * f0;
            W.terms[i] = div3(dd * dd, dx, dxPow2, invDx);
        }
        if (lane < 16) W.terms[nB + lane] = 0.0;
        wave_sync();
        double loglike = ordered_sum(0.0, W.terms + 2, nB - 2, lane, -1.0);
        loglike *= M.smoothLs * M.smoothLs / M.smoothL;
        priorOut = loglike;
        wave_sync();
    } else {
        for (int i = (int)lane; i < nB; i += MCMC_WAVE) xp[i] = xc[i];
    }
    if (lane < (unsigned)M.nDisc) {
        double v = disc[0];
#pragma unroll
        for (int k = 1; k < MCMC_MAXDISC; ++k) v = (lane == (unsigned)k) ? disc[k] : v;
        xp[nB + lane] = v;
    }
    if (lane == 0) {
        xp[nB + M.nDisc] = lp.betaMild;
        xp[nB + M.nDisc + 1] = lp.betaHigh;
        xp[nB + M.nDisc + 2] = lp.delay;
    }
    wave_sync();
    return doB;
}

/*
 * n_steps Metropolis-Hastings steps for every chain, one chain per wavefront: the same schedule, draws,
 * arithmetic and results as mh_step_kernel<T, true, 0, false> (see there for the reference citations).
 * Grid: ceil(nChains / 4) workgroups of 4 waves sharing one LDS image of the tables and the series.
 */
/* (the strict form needs more than the 256 registers a second wave per SIMD leaves: four chains per workgroup at most) */
template <bool FUSE>
__global__ __launch_bounds__(MCMC_WAVE * (FUSE ? MCMC_WV_MAX_WAVES : MCMC_WAVES_PER_BLOCK)) __attribute__((amdgpu_waves_per_eu(1, 2))) void mh_step_wave_kernel(DevModel M, ChainState S, int n_steps,
                                                      const double* __restrict__ betas,
                                                      const int* __restrict__ flags, int slot0) {
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    const WaveImage I = stage_wave_image(M, lds_raw);
    const unsigned lane = threadIdx.x & (MCMC_WAVE - 1);
    const int waveInBlock = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / MCMC_WAVE));
    const int wavesPerBlock = (int)(blockDim.x / MCMC_WAVE);       /* 4, or fewer where the series leaves LDS for fewer slices */
    const long long chain = (long long)blockIdx.x * wavesPerBlock + waveInBlock;
    if (chain >= S.nChains) return;
    const int tile = (int)(chain / MCMC_WAVE);
    const int lit = (int)(chain % MCMC_WAVE);                       /* the chain's lane in the tile-major arrays */
    const int dim = S.dim;

    WaveSlice W;
    W.histLen = M.simDays + MCMC_WV_HLO + MCMC_WV_HHI;
    {
        double* p = lds_raw + wave_shared_doubles(M) + (size_t)waveInBlock * wave_slice_doubles(M, dim);
        W.hn[0] = p; p += W.histLen;
        W.hn[1] = p; p += W.histLen;
        W.he[0] = p; p += W.histLen;
        W.he[1] = p; p += W.histLen;
        W.d1 = p; p += M.nData;
        W.d0 = p; p += M.nData;
        W.terms = W.he[0];                 /* histLen = simDays + 36 > nData, nB */
        W.xc = p; p += dim;
        W.xp = p;
    }
    double* xg = S.params + (size_t)tile * dim * MCMC_WAVE + lit;   /* element k at xg[k * 64] */
    for (int k = (int)lane; k < dim; k += MCMC_WAVE) W.xc[k] = xg[(size_t)k * MCMC_WAVE];
    for (int k = (int)lane; k < M.nData; k += MCMC_WAVE) { W.d1[k] = 0.0; W.d0[k] = 0.0; }
    uint64_t* rng_g = S.rng + (size_t)tile * 2 * MCMC_WAVE + lit;
    double* ll_g = S.ll + (size_t)tile * 2 * MCMC_WAVE + lit;
    double* ss_g = S.stepsize + (size_t)tile * MCMC_WAVE + lit;
    const size_t flat = (size_t)tile * MCMC_WAVE + lit;
    unsigned long long* pairs_g = S.pairs + (size_t)tile * 2 * MCMC_WAVE + lit;

    Pcg32 rnd;
    rnd.state = rng_g[0];
    rnd.inc = rng_g[MCMC_WAVE];
    double llModel = ll_g[0], llPrior = ll_g[MCMC_WAVE], stepsize = ss_g[0];
    int nAccept = S.nAccept[flat];
    WorkCount work;
    work.algo = pairs_g[0];
    work.exec = pairs_g[MCMC_WAVE];
    const double rungBeta = S.chainBeta ? S.chainBeta[flat] : 1.0;
    int slot = slot0;
    const bool fastDiv = wave_fast_div_ok(M.pop);
    const double rcpPop = wave_rcp_pop(M.pop);
    PhaseClock pc;
    pc.start();

#pragma unroll 1
    for (int g = 0; g < n_steps; ++g) {
        const int fl = flags[g];
        const double beta = betas[g] * rungBeta;
        if (fl & SF_RESET_ACC) nAccept = 0;                         /* mcmc.h:1032,1090 */
        if (fl & SF_RECORD) {                                       /* mcmc.h:1037-1062 */
            double l = 0;
            l += llModel;
            if (M.hasPrior) l += llPrior;
            if (lane == 0) S.recLL[((size_t)slot * S.nTiles + tile) * MCMC_WAVE + lit] = 0.0 + beta * l;
            if (S.recSamples) {
                double* dst = S.recSamples + ((size_t)slot * S.nTiles + tile) * (size_t)dim * MCMC_WAVE + lit;
                for (int k = (int)lane; k < dim; k += MCMC_WAVE) dst[(size_t)k * MCMC_WAVE] = W.xc[k];
            }
            ++slot;
        }
        if ((fl & SF_MEAN) && S.mean) {                             /* mcmc.h:1064-1065 */
            double* mu = S.mean + (size_t)tile * dim * MCMC_WAVE + lit;
            for (int k = (int)lane; k < dim; k += MCMC_WAVE) mu[(size_t)k * MCMC_WAVE] += W.xc[k];
        }
        if (fl & SF_RANDOM) {                                       /* mcmc.h:494-500 */
            const int nState = M.hasPrior ? 2 : 1;
            int hd;
            do {
                const float f = rnd.next_float() * (float)nState;
                hd = min((int)f, nState - 1);
            } while (hd != 0);
        }
        LaneScalars lp;
        double llpNew;
        pc.stamp(PH_RECORD);
        const bool behChanged = propose_wave(M, rnd, W, lane, stepsize, lp, llpNew);
        pc.stamp(PH_PROPOSE);
        /* the short division first; the compiler's if a numerator left the range where the two are the same */
        double llm;
        {
            const WorkCount work0 = work;
            bool redo = !fastDiv;
            if (fastDiv) llm = eval_model_wave<true, FUSE>(M, I, W, lp, lane, work, pc, rcpPop, redo);
            if (redo) {
                work = work0;
                llm = eval_model_wave<false, FUSE>(M, I, W, lp, lane, work, pc, rcpPop, redo);
            }
        }
        double llp = 0.0;
        double delta = 0;                                           /* mcmc.h:604-606 */
        delta += -llModel + llm;
        if (M.hasPrior) {
            llp = behChanged ? llpNew : llPrior;
            delta += -llPrior + llp;
        }
        delta += 0.0;                                               /* mcmc.h:765 */
        delta *= beta;                                              /* mcmc.h:766 */
        const double u = (double)rnd.next_float();
        const bool accept = __builtin_amdgcn_readfirstlane((u < 1.0 * mcmc_exp(delta)) ? 1 : 0) != 0;   /* mcmc.h:776 */
        if (accept) {                                               /* State::accept, mcmc.h:653-661 */
            for (int k = (int)lane; k < dim; k += MCMC_WAVE) W.xc[k] = W.xp[k];
            llModel = llm;
            llPrior = llp;
            ++nAccept;
        }
        wave_sync();
        if (fl & SF_ADJUST_END) {                                   /* mcmc.h:1107-1114 */
            const double acceptRate = (double)nAccept / 20;
            stepsize *= rate2corr(acceptRate);
        }
        pc.stamp(PH_ACCEPT);
    }
    pc.flush(lane, (int)chain);

    for (int k = (int)lane; k < dim; k += MCMC_WAVE) xg[(size_t)k * MCMC_WAVE] = W.xc[k];
    if (lane == 0) {
        rng_g[0] = rnd.state;
        ll_g[0] = llModel;
        ll_g[MCMC_WAVE] = llPrior;
        ss_g[0] = stepsize;
        S.nAccept[flat] = nAccept;
        pairs_g[0] = work.algo;
        pairs_g[MCMC_WAVE] = work.exec;
    }
}
