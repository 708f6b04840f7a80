/*
 * mcmc_wave_kernel.hip.h -- the SMALL-ENSEMBLE mapping of the Metropolis-Hastings step: one chain per
 * WAVEFRONT instead of one chain per lane (EXTENSION beside north_star's mapping; same arithmetic, same bits).
 *
 * Why: with one chain per lane BASELINE configs[1] (1 024 chains x 200 days) is 16 wavefronts on a chip of
 * 1 024 SIMDs.  The model's only parallelism inside one chain is (destination day) x (compartment): every
 * accumulator buf_c[j] receives its addends n_k * R_c[j-k] in ascending source day k (that order is the
 * reference's, covid.h:376-386, and fixes the bits), but the 7 compartments and the days of a block are
 * independent of each other.  This kernel gives a chain a whole wavefront and spreads a block of 16 days x 7
 * compartments over it:
 *
 *   phase A  (sources before the block)   C[c][t] = sum over lags L of R_c[L] * n_{j0+t-L}  is a matrix product
 *            with A[m = c][k = L] = table, B[k = L][n = t] = history (Toeplitz), and v_mfma_f64_16x16x4_f64 is,
 *            bit for bit, the chain fma(a3,b3,fma(a2,b2,fma(a1,b1,fma(a0,b0,C)))) (tools/ubench/mfma_f64.hip:
 *            0 mismatches) -- with the K slots holding DESCENDING lags that is ascending source days, the order
 *            the parity bar needs.  One MFMA = 4 lags x 16 days x 7 compartments; both operands come from LDS
 *            (the workgroup's table image, the wave's own history).  Lags whose source day carries a hospital-
 *            capacity correction (covid.h:413-429) go one lag per MFMA, the K slots holding that source day's
 *            three operations in the reference's order: n_k * R_c[L], then + e_k (deaths), then - e_k * FromSerious[L].
 *   phase B  (the block's 16 days in order; the feedback n_{i+1} = f(buffers[i]) is sequential, covid.h:461-462)
 *            today's n_i goes into today's and the later days' accumulators from registers (one multiply-add per
 *            lane), the seven sums of today are fetched with v_readlane and the day's scalar arithmetic runs
 *            uniformly in every lane.
 *
 * Everything a chain touches during a launch -- parameter vector, proposal, both histories, weighted deaths --
 * lives in the wave's slice of LDS; HBM is read at the start of a launch and written at its end (and by
 * recording).  The proposal's behaviour nodes, the prior's and the likelihood's terms are computed one per
 * lane and summed in the reference's order.  Only FP64 with FMA contraction (cfg.contract = 1: an MFMA is
 * fused), only the single-region State.  The host picks the mapping by ensemble size (mcmc_hip_set_mapping);
 * tests/test_gpu_wave_mapping.py holds it bit-exact to the oracle AND to the chain-per-lane kernel.
 */
#pragma once

#define MCMC_WV_T 16          /* the 16 columns of the MFMA tile: days per HALF block */
#define MCMC_WV_B 32          /* days per block: rows 0..6 of the tile hold the block's days 0..15, rows 8..14 its days 16..31 */
#define MCMC_WV_HLO 16        /* zero history entries kept below day 0 (phase A reads down to s - 15) */
#define MCMC_WV_HHI 20        /* ... and above the simulated horizon (phase A reads up to j0 + 18, as zeros; j0 <= horizon) */

__host__ __device__ inline size_t wave_slice_doubles(const DevModel& M, int dim) {
    const size_t histLen = (size_t)M.simDays + MCMC_WV_HLO + MCMC_WV_HHI;
    return 4 * histLen + 2 * (size_t)M.nData + 2 * (size_t)dim + MCMC_WAVE;
}
/* LDS of a workgroup of `waves` chains: the shared image of tables and series + one slice per wave */
__host__ __device__ inline size_t wave_lds_bytes(const DevModel& M, int dim, int waves) {
    return ((lds_bytes(M, false, MCMC_WV_T) + 7) / 8 + (size_t)waves * wave_slice_doubles(M, dim)) * sizeof(double);
}

/* the wave's slice */
struct WaveSlice {
    double* hn[2];     /* [histLen] infections per day of either pass, day k at hn[p][k + MCMC_WV_HLO] */
    double* he[2];     /* [histLen] extra deaths per day                            */
    double* dummy;     /* [64] one slot per lane for the stores of the branch-free day loop that must not happen */
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
__device__ __forceinline__ double both_f64(double lo, double hi, bool hiHalf) { return hiHalf ? hi : lo; }
__device__ __forceinline__ double uniform_f64(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

/* What phase B multiplies with never changes: the lane that holds (compartment row q / 4 + q, day n) of a block
 * needs, on the block's day t, the table rows of lag n - t -- or nothing if its day is already over.  Loaded once
 * per launch; a zero stands for "day over" (the multiply-add then leaves the finished sum, which nobody reads
 * again, unchanged). */
struct WaveRows {
    double p0[MCMC_WV_T];     /* R_q[n - t]      (MI, HI, INC, MILD) */
    double p1[MCMC_WV_T];     /* R_{4+q}[n - t]  (SER, REC, DEAD, 0) */
    double ng[MCMC_WV_T];     /* the FromSerious curve of SER / REC / DEAD at lag n - t, else 0 */
    double f0[MCMC_WV_T];     /* the same three at lag 16 + n - t: what a day t < 16 pushes into the block's second half */
    double f1[MCMC_WV_T];
    double fg[MCMC_WV_T];
    double ng0[3];            /* the three FromSerious curves at lag 0 (today's own sums), the same in every lane */
};
__device__ __forceinline__ void load_wave_rows(const LdsView& lds, unsigned lane, WaveRows& R) {
    const int q = (int)(lane >> 4), n = (int)(lane & 15);
#pragma unroll
    for (int t = 0; t < MCMC_WV_T; ++t) {
        const int d = n - t;
        const int dd = max(d, 0);
        const double a = lds.Rpos[dd * 8 + q], b = lds.Rpos[dd * 8 + 4 + q], g = lds.Rneg[dd * 4 + min(q, 2)];
        R.p0[t] = d >= 0 ? a : 0.0;
        R.p1[t] = d >= 0 ? b : 0.0;
        R.ng[t] = (d >= 0 && q < 3) ? g : 0.0;
        const int df = MCMC_WV_T + n - t;                               /* 1 .. 31 */
        R.f0[t] = lds.Rpos[df * 8 + q];
        R.f1[t] = lds.Rpos[df * 8 + 4 + q];
        R.fg[t] = q < 3 ? lds.Rneg[df * 4 + q] : 0.0;
    }
    R.ng0[0] = lds.Rneg[0]; R.ng0[1] = lds.Rneg[1]; R.ng0[2] = lds.Rneg[2];
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
 * The day loop is branch-free for that reason (stores go to a per-lane dummy slot when they must not happen).
 */
__device__ __forceinline__ double eval_model_wave(const DevModel& M, const LdsView& lds, const WaveSlice& W,
                                                  const LaneScalars& lp, unsigned lane, WorkCount& work, int ldsRows,
                                                  const WaveRows& BR, PhaseClock& pc) {
    const int md = M.md;
    const int horizon = M.simDays;
    const int q = (int)(lane >> 4), n = (int)(lane & 15);     /* MFMA result: rows q, q+4, q+8, q+12 of column n */
    const int kk = q, m = n;                                   /* MFMA operands: A[m][kk], B[kk][n]               */
    const int rowsM1 = ldsRows - 1;
    const double* xdisc = W.xp + M.nB;
    double* const dummy = W.dummy + lane;                       /* where a store that must not happen goes */

    int start = (int)((double)md - lp.delay);                       /* covid.h:331 */
    const double frac = (double)md - lp.delay - (double)start;      /* covid.h:332 */
    start = min(max(start, 0), horizon);
    start = __builtin_amdgcn_readfirstlane(start);
    count_work(M, start, work);

    const bool hiHalf = lane >= 32;                              /* phase B: which pass's day arithmetic the lane carries */
    int sP[2];
    double wP[2], ncurS[2];                                       /* n of the current day of either pass, the same in every lane */
    double ncurB = 1.0;                                           /* ... and as the lanes of the day chain hold it */
    double* hnP[2] = {W.hn[0], W.hn[1]};
    double* heP[2] = {W.he[0], W.he[1]};
    int keLo[2], keHi[2];                                           /* source days with a capacity correction */
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int shift = 1 - p;
        wP[p] = (double)(1 - shift) * (1.0 - frac) + (double)shift * frac;      /* covid.h:333 */
        sP[p] = min(start + shift, horizon);
        ncurS[p] = 1.0;                                             /* patient zero on day s, covid.h:339 */
        keLo[p] = 0x7fffffff; keHi[p] = -1;
    }
    double* const hnDst = (hiHalf ? W.hn[1] : W.hn[0]) + MCMC_WV_HLO;
    double* const heDst = (hiHalf ? W.he[1] : W.he[0]) + MCMC_WV_HLO;
    double* const dsDst = hiHalf ? W.d0 : W.d1;
    const double wBoth = hiHalf ? wP[1] : wP[0];
    for (int k = (int)lane; k < W.histLen; k += MCMC_WAVE) {
        W.hn[0][k] = 0.0; W.he[0][k] = 0.0; W.hn[1][k] = 0.0; W.he[1][k] = 0.0;
    }
    wave_sync();
    pc.stamp(PH_SETUP);

#pragma unroll 1
    for (int blk = 0; sP[1] + blk * MCMC_WV_B < horizon; ++blk) {
        const int j0P[2] = {sP[0] + blk * MCMC_WV_B, sP[1] + blk * MCMC_WV_B};
        /* ---------- phase A: sources k < j0 (entries >= j0 of the history are still zero).  Tile rows m < 8: the
         * block's days 0..15 at lag L; rows m >= 8: its days 16..31, i.e. the same source day at lag L + 16.  The
         * table operand is the same for both passes, the history operand is each pass's own.  Lag groups whose
         * source days carry corrections in EITHER pass form one range of lags (a correction of zero changes
         * nothing), so the groups split into three runs -- plain, with corrections, plain -- each a tight loop
         * of its own with both tiles in fixed registers. */
        double4_t c0 = {0.0, 0.0, 0.0, 0.0}, c1 = {0.0, 0.0, 0.0, 0.0};
        if (blk > 0) {
            const int Lhi = blk * MCMC_WV_B + (MCMC_WV_T - 1);
            const int Loff = (m >> 3) * MCMC_WV_T, col = m & 7;
            const double* hnB0 = hnP[0] + MCMC_WV_HLO + j0P[0] + n;   /* source day of lag L: j0 + n - L */
            const double* hnB1 = hnP[1] + MCMC_WV_HLO + j0P[1] + n;
            int eLo = 0x7fffffff, eHi = -1;                          /* the lags with corrections, both passes */
#pragma unroll
            for (int p = 0; p < 2; ++p)
                if (keHi[p] >= keLo[p]) {
                    eLo = min(eLo, j0P[p] - keHi[p]);
                    eHi = max(eHi, j0P[p] + (MCMC_WV_T - 1) - keLo[p]);
                }
            const bool anyE = eHi >= eLo && eHi >= 1;
            int L0 = Lhi;
            auto plain_run = [&](int stop) {                         /* the groups with L0 >= stop */
                if (L0 < stop) return;
                /* operands of the NEXT group are requested before the current group's matrix instructions are issued */
                double aN = lds.Rpos[min(max(L0 - kk, 0) + Loff, rowsM1) * 8 + col];
                double bN0 = hnB0[-(L0 - kk)], bN1 = hnB1[-(L0 - kk)];
#pragma unroll 1
                for (; L0 >= stop; L0 -= 4) {
                    const double a = aN, b0 = bN0, b1 = bN1;
                    const int Ln = L0 - 4 - kk;                      /* below lag 1 the history operand is zero anyway */
                    aN = lds.Rpos[min(max(Ln, 0) + Loff, rowsM1) * 8 + col];
                    bN0 = hnB0[-max(Ln, -3)];
                    bN1 = hnB1[-max(Ln, -3)];
                    pc.count(PH_NCHUNK, 2);
                    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, c1, 0, 0, 0);
                }
            };
            plain_run(anyE ? max(eHi + 4, 1) : 1);                   /* a plain group has L0 - 3 > eHi */
            if (anyE) {
                /* one lag per matrix instruction, the K slots holding the source day's three operations in the
                 * reference's order: n_k R_c[L]; + e_k (deaths: the flat add() of covid.h:387-391, cut after
                 * traj.nDays lags like project()); - e_k x the three FromSerious curves; nothing.  What a lane
                 * feeds depends on its slot only: */
                const double* eTab = kk == 0 ? lds.Rpos : lds.Rneg;
                const int eStride = kk == 0 ? 8 : 4;
                const int eOff = kk == 0 ? col : kk == 1 ? 3 : min(max(col - A_SER, 0), 2);
                const bool eValid = kk == 0 || (kk == 1 && col == A_DEAD) || (kk == 2 && col >= A_SER && col <= A_DEAD);
                const double* eH0 = (kk == 0 ? hnP[0] : heP[0]) + MCMC_WV_HLO + j0P[0] + n;
                const double* eH1 = (kk == 0 ? hnP[1] : heP[1]) + MCMC_WV_HLO + j0P[1] + n;
                const int Ltop = L0;
                const int stop = max(eLo, 1);
                while (L0 >= stop) L0 -= 4;                           /* the groups of this run */
                const int Lbot = max(L0 + 1, 1);
                pc.count(PH_NCHUNK_E, 2 * max(Ltop - Lbot + 1, 0));
#pragma unroll 1
                for (int L = Ltop; L >= Lbot; --L) {
                    const int Lr = min(L + Loff, rowsM1);
                    const double t = eTab[Lr * eStride + eOff];
                    const double x0 = eH0[-L], x1 = eH1[-L];
                    const double ae = eValid ? t : 0.0;
                    const double be0 = kk == 3 ? 0.0 : (kk == 2 ? -x0 : x0), be1 = kk == 3 ? 0.0 : (kk == 2 ? -x1 : x1);
                    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ae, be0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ae, be1, c1, 0, 0, 0);
                }
                plain_run(1);
            }
        }
        pc.stamp(PH_CHUNKS);
        /* ---------- phase B: the block's days in order, both passes day for day.  cc[p][0] = (MI, HI, INC, MILD)[q] and
         * cc[p][1] = (SER, REC, DEAD, -)[q] of day n; cc[p][2], cc[p][3] the same of day 16 + n.  The day's scalar
         * arithmetic is the same for both passes on different numbers: lanes 0..31 carry pass 0's, lanes 32..63
         * pass 1's, so ONE sequential chain of operations serves both. */
        double cc[2][4] = {{c0[0], c0[1], c0[2], c0[3]}, {c1[0], c1[1], c1[2], c1[3]}};
        double keLoB = hiHalf ? (double)keLo[1] : (double)keLo[0], keHiB = hiHalf ? (double)keHi[1] : (double)keHi[0];
        const int j0B = hiHalf ? j0P[1] : j0P[0];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half == 1 && !(j0P[1] + MCMC_WV_T < horizon)) break;        /* wave-uniform: nothing of the second half is simulated */
            /* what the days need besides the sums, the lane's own pass: capacity and pcf * smoothbehavior (covid.h:432-454) */
            double capB[MCMC_WV_T], tbB[MCMC_WV_T];
#pragma unroll
            for (int u = 0; u < MCMC_WV_T; ++u) {
                const int iB = min(j0B + half * MCMC_WV_T + u, M.simPad - 1);
                capB[u] = lds.capacity[iB];
                const int sg = lds.seg[iB];
                const double pcf = sg >= 0 ? xdisc[max(sg, 0)] : 1.0;
                double smoothbehavior = 1.0;
                if (iB >= md) smoothbehavior = pcf * W.xp[min(max(iB - md, 0), M.nB - 1)];
                tbB[u] = pcf * smoothbehavior;
            }
#pragma unroll
            for (int u = 0; u < MCMC_WV_T; ++u) {
                const int iB = j0B + half * MCMC_WV_T + u;                          /* the lane's pass's day */
                const bool writer = (lane & 31) == 0 && iB < horizon;               /* days beyond compute garbage nobody reads */
                *(writer ? hnDst + iB : dummy) = ncurB;
                /* today's new infections into today and the rest of the block (covid.h:403-410) */
                double va[2], vb[2];                                                /* the half that holds today */
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    if (half == 0) {
                        cc[p][0] = __builtin_fma(ncurS[p], BR.p0[u], cc[p][0]);
                        cc[p][1] = __builtin_fma(ncurS[p], BR.p1[u], cc[p][1]);
                        cc[p][2] = __builtin_fma(ncurS[p], BR.f0[u], cc[p][2]);
                        cc[p][3] = __builtin_fma(ncurS[p], BR.f1[u], cc[p][3]);
                        va[p] = cc[p][0]; vb[p] = cc[p][1];
                    } else {
                        cc[p][2] = __builtin_fma(ncurS[p], BR.p0[u], cc[p][2]);
                        cc[p][3] = __builtin_fma(ncurS[p], BR.p1[u], cc[p][3]);
                        va[p] = cc[p][2]; vb[p] = cc[p][3];
                    }
                }
                /* both lanes' values are fetched, then selected: a select AROUND the (convergent) lane reads becomes branches */
#define MCMC_BOTH(v, srcLane) both_f64(readlane_f64((v)[0], srcLane), readlane_f64((v)[1], srcLane), hiHalf)
                const double vMI = MCMC_BOTH(va, u), vHI = MCMC_BOTH(va, 16 + u), vINC = MCMC_BOTH(va, 32 + u), vMILD = MCMC_BOTH(va, 48 + u);
                double vSER = MCMC_BOTH(vb, u), vREC = MCMC_BOTH(vb, 16 + u), vDEAD = MCMC_BOTH(vb, 32 + u);
#undef MCMC_BOTH
                /* hospital capacity (covid.h:413-429).  Straight-line: without a correction e = +0, and adding
                 * +0 or the product -0 * x changes no sum (none of them is ever -0) */
                const double overCapacity = vSER * M.probICU - capB[u];
                const double extraDeaths = M.probLethal * overCapacity;
                const bool hasE = extraDeaths > 0;
                const double e = hasE ? extraDeaths : 0.0, ne = -e;
                *(writer ? heDst + iB : dummy) = e;
                keLoB = (hasE && iB < horizon) ? fmin(keLoB, (double)iB) : keLoB;
                keHiB = (hasE && iB < horizon) ? fmax(keHiB, (double)iB) : keHiB;
#pragma unroll
                for (int p = 0; p < 2; ++p) {                                       /* the later days' sums: + e (deaths), then the projections */
                    const double eS = readlane_f64(e, 32 * p);
                    const double eDead = (q == 2) ? eS : 0.0;
                    if (half == 0) {
                        cc[p][1] = cc[p][1] + eDead;
                        cc[p][1] = __builtin_fma(-eS, BR.ng[u], cc[p][1]);
                        cc[p][3] = cc[p][3] + eDead;
                        cc[p][3] = __builtin_fma(-eS, BR.fg[u], cc[p][3]);
                    } else {
                        cc[p][3] = cc[p][3] + eDead;
                        cc[p][3] = __builtin_fma(-eS, BR.ng[u], cc[p][3]);
                    }
                }
                vDEAD = vDEAD + e;                                                  /* today's own sums, the same operations */
                vSER = __builtin_fma(ne, BR.ng0[0], vSER);
                vREC = __builtin_fma(ne, BR.ng0[1], vREC);
                vDEAD = __builtin_fma(ne, BR.ng0[2], vDEAD);
                /* tomorrow's new infections (covid.h:461-462); asymptomatic == mild buffer */
                const double nSus = M.pop - vINC - vMILD - vMILD - vSER - vREC - vDEAD;
                const double nnext = tbB[u] * nSus / M.pop * (lp.betaMild * vMI + lp.betaHigh * vHI);
                const bool stD = writer && iB >= md && iB - md < M.nData;
                *(stD ? dsDst + (iB - md) : dummy) = wBoth * vDEAD;                 /* covid.h:475,484 */
                ncurB = nnext;
                ncurS[0] = readlane_f64(nnext, 0);
                ncurS[1] = readlane_f64(nnext, 32);
            }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            keLo[p] = (int)readlane_f64(keLoB, 32 * p);
            keHi[p] = (int)readlane_f64(keHiB, 32 * p);
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
        wave_sync();
#pragma unroll 8
        for (int i = 0; i < M.nData; ++i) loglike += W.terms[i];
        wave_sync();
    }
    pc.stamp(PH_LOGLIKE);
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
        wave_sync();
        double loglike = 0;
#pragma unroll 8
        for (int i = 2; i < nB; ++i) loglike -= W.terms[i];
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
__global__ MCMC_KERNEL_ATTRS void mh_step_wave_kernel(DevModel M, ChainState S, int n_steps,
                                                      const double* __restrict__ betas,
                                                      const int* __restrict__ flags, int slot0) {
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    const LdsView lds = stage_lds(M, lds_raw, false, MCMC_WV_T);
    const int ldsRows = lds_rows(M, false, MCMC_WV_T);
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
        double* p = lds_raw + (lds_bytes(M, false, MCMC_WV_T) + 7) / 8 + (size_t)waveInBlock * wave_slice_doubles(M, dim);
        W.hn[0] = p; p += W.histLen;
        W.hn[1] = p; p += W.histLen;
        W.he[0] = p; p += W.histLen;
        W.he[1] = p; p += W.histLen;
        W.dummy = p; p += MCMC_WAVE;
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
    WaveRows BR;
    load_wave_rows(lds, lane, BR);
    wave_sync();
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
        const double llm = eval_model_wave(M, lds, W, lp, lane, work, ldsRows, BR, pc);
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
