/*
 * mcmc_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See
 * mcmc_oracle.h for scope and pinning status.  Every function cites the reference
 * lines (file:line under /root/reference) it restates.  Compile with
 *   gcc -O2 -ffp-contract=off -mfma     (contraction off: each C operation below is
 * one IEEE operation; the only fused operations are the explicit fma() calls that the
 * `contract` flag enables).
 *
 * The formulation here is deliberately the reference's own: project() SCATTERS each
 * day's new infections forward into eight full-length buffers (covid.h:376-386).  The
 * HIP kernels compute the same sums as a GATHER over a per-lane infection history; the
 * two agree bit for bit because every destination element receives the same addends in
 * the same order.
 */
#include "mcmc_oracle.h"
#include "../include/mcmc_detmath.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ pcg32 ---- */
/* wjakob/pcg32 pcg32.h (submodule not vendored): PCG32_DEFAULT_STATE/STREAM/MULT,
 * seed(), nextUInt(), nextFloat(), nextDouble().  Used at mcmc.h:999 (rnd(0,chainid)),
 * covid.h:102 (default ctor). */
#define ORC_PCG32_DEFAULT_STATE  0x853c49e6748fea9bULL
#define ORC_PCG32_DEFAULT_STREAM 0xda3e39cb94b95bdbULL
#define ORC_PCG32_MULT           0x5851f42d4c957f2dULL

void orc_pcg32_default(orc_pcg32* r) {
    r->state = ORC_PCG32_DEFAULT_STATE;
    r->inc = ORC_PCG32_DEFAULT_STREAM;
}
uint32_t orc_pcg32_next_uint(orc_pcg32* r) {
    uint64_t old = r->state;
    r->state = old * ORC_PCG32_MULT + r->inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}
void orc_pcg32_seed(orc_pcg32* r, uint64_t initstate, uint64_t initseq) {
    r->state = 0u;
    r->inc = (initseq << 1u) | 1u;
    orc_pcg32_next_uint(r);
    r->state += initstate;
    orc_pcg32_next_uint(r);
}
float orc_pcg32_next_float(orc_pcg32* r) {
    union { uint32_t u; float f; } x;
    x.u = (orc_pcg32_next_uint(r) >> 9) | 0x3f800000u;
    return x.f - 1.0f;
}
double orc_pcg32_next_double(orc_pcg32* r) {
    union { uint64_t u; double d; } x;
    x.u = ((uint64_t)orc_pcg32_next_uint(r) << 20) | 0x3ff0000000000000ULL;
    return x.d - 1.0;
}

/* ---------------------------------------------------------------- helpers ---- */
/* mcmc.h:49-53; std::round = half away from zero = C round() */
double orc_bound(double val, double lower, double upper) {
    val = 0.5 * (val - lower) / (upper - lower);
    val = 2.0 * fabs(val - round(val));
    val = lower + val * (upper - lower);
    return val;
}

/* enoki::erfinv (enoki/special.h, submodule not vendored) restated as the published
 * single-precision polynomial of M. Giles, "Approximating the erfinv function",
 * evaluated in double.  Parity unpinned; only feeds the response-table precompute. */
double orc_erfinv(double x) {
    double w = -log((1.0 - x) * (1.0 + x));
    double p;
    if (w < 5.0) {
        w = w - 2.5;
        p = 2.81022636e-08;
        p = 3.43273939e-07 + p * w;
        p = -3.5233877e-06 + p * w;
        p = -4.39150654e-06 + p * w;
        p = 0.00021858087 + p * w;
        p = -0.00125372503 + p * w;
        p = -0.00417768164 + p * w;
        p = 0.246640727 + p * w;
        p = 1.50140941 + p * w;
    } else {
        w = sqrt(w) - 3.0;
        p = -0.000200214257;
        p = 0.000100950558 + p * w;
        p = 0.00134934322 + p * w;
        p = -0.00367342844 + p * w;
        p = 0.00573950773 + p * w;
        p = -0.0076224613 + p * w;
        p = 0.00943887047 + p * w;
        p = 1.00167406 + p * w;
        p = 2.83297682 + p * w;
    }
    return p * x;
}
/* mcmc.h:45-47 */
double orc_to_normal(double u) { return M_SQRT2 * orc_erfinv(2.0 * u - 1.0); }

/* covid.h:75-90 */
void orc_disease_params_default(orc_disease_params* p) {
    p->timeIncub = 4; p->timeIncubSigma = 2; p->probAsymp = 0.1; p->probSerious = 0.08;
    p->probLethal = 0.004; p->probLethalDailyWhenSeriousUntreated = 0.2;
    p->probICUIfSerious = 0.3; p->timeMildDuration = 10; p->timeMildDurationSigma = 5;
    p->timeMildToSerious = 8; p->timeMildToSeriousSigma = 3; p->timeSeriousToRec = 10;
    p->timeSeriousToRecSigma = 5; p->timeSeriousToDeath = 9; p->timeSeriousToDeathSigma = 5;
}

/* ---------------------------------------------------- AvgDiseaseTrajectory ---- */
enum { C_INCUB = 0, C_ASYMP, C_MILD, C_INFMILD, C_INFHIGH, C_SERIOUS, C_DEAD, C_RECOV,
       C_SFS, C_RFS, C_DFS };

/* covid.h:262-271.  NB: takes the generator BY VALUE, so the draw is a peek: the
 * caller's generator does not advance (quirk reproduced). */
static double sample_time(orc_pcg32 rnd, double mean, double sig) {
    double s = sqrt(log(sig * sig / (mean * mean) + 1));
    double expmu = mean * exp(-0.5 * s * s);
    double sample;
    do {
        sample = orc_pcg32_next_double(&rnd);
    } while (sample <= 0 || sample >= 1);
    double n = orc_to_normal(sample) * s;
    return expmu * exp(n);
}
/* covid.h:273-280 (the out-of-range message is dropped, the clamp kept) */
static int t2i(double t) {
    double v = ORC_NGRID * (t / ORC_NDAYS);
    if (!(v < (double)ORC_NGRID)) return ORC_NGRID - 1;   /* size_t(v) > nGrid-1 */
    return (int)(size_t)v;
}

/* covid.h:100-203.  The three *FromSerious arrays are not value-initialised by the
 * reference's ctor init-list (indeterminate there); zero here. */
void orc_traj_generate(const orc_disease_params* p, uint64_t n_traj, double* grid) {
    double* a[ORC_NCURVES];
    for (int c = 0; c < ORC_NCURVES; ++c) {
        a[c] = grid + (size_t)c * ORC_NGRID;
        memset(a[c], 0, sizeof(double) * ORC_NGRID);
    }
    orc_pcg32 rnd;
    orc_pcg32_default(&rnd);                                        /* covid.h:102 */
    for (uint64_t k = 0; k < n_traj; ++k) {
        a[C_INCUB][0] += 1;
        double t = sample_time(rnd, p->timeIncub, p->timeIncubSigma);
        a[C_INCUB][t2i(t)] -= 1;
        if (orc_pcg32_next_double(&rnd) < p->probAsymp) {           /* covid.h:113 */
            a[C_ASYMP][t2i(t)] += 1;
            a[C_INFHIGH][t2i(t)] += 1;
            double t2 = sample_time(rnd, p->timeMildDuration, p->timeMildDurationSigma);
            a[C_ASYMP][t2i(t + t2)] -= 1;
            a[C_RECOV][t2i(t + t2)] += 1;
            a[C_INFHIGH][t2i(t + t2)] -= 1;
        } else {
            a[C_MILD][t2i(t)] += 1;
            a[C_INFMILD][t2i(t)] += 1;
            if (t2i(t) - 1 > 0) {
                a[C_INFHIGH][t2i(t) - 1] += 1;
                a[C_INFHIGH][t2i(t)] -= 1;
            }
            double x = orc_pcg32_next_double(&rnd);                 /* covid.h:131 */
            if (x < 1 - (p->probSerious + p->probLethal) / (1 - p->probAsymp)) {
                double t2 = sample_time(rnd, p->timeMildDuration, p->timeMildDurationSigma);
                a[C_MILD][t2i(t + t2)] -= 1;
                a[C_INFMILD][t2i(t + t2)] -= 1;
                a[C_RECOV][t2i(t + t2)] += 1;
            } else {
                /* mean passed as sigma: covid.h:142 */
                double t2 = sample_time(rnd, p->timeMildToSerious, p->timeMildToSerious);
                a[C_INFMILD][t2i(t + t2)] -= 1;
                a[C_SERIOUS][t2i(t + t2)] += 1;
                a[C_MILD][t2i(t + t2)] -= 1;
                double t3;
                if (x < 1 - p->probLethal / (1 - p->probAsymp)) {
                    t3 = sample_time(rnd, p->timeSeriousToRec, p->timeSeriousToRecSigma);
                    a[C_RECOV][t2i(t + t2 + t3)] += 1;
                } else {
                    t3 = sample_time(rnd, p->timeSeriousToDeath, p->timeSeriousToDeathSigma);
                    a[C_DEAD][t2i(t + t2 + t3)] += 1;
                }
                a[C_SERIOUS][t2i(t + t2 + t3)] -= 1;
            }
        }
        double t3;
        double x = orc_pcg32_next_double(&rnd);                     /* covid.h:165 */
        a[C_SFS][0] += 1;
        if (x < p->probSerious / (p->probLethal + p->probSerious)) {
            t3 = sample_time(rnd, p->timeSeriousToRec, p->timeSeriousToRecSigma);
            a[C_RFS][t2i(t3)] += 1;
        } else {
            t3 = sample_time(rnd, p->timeSeriousToDeath, p->timeSeriousToDeathSigma);
            a[C_DFS][t2i(t3)] += 1;
        }
        a[C_SFS][t2i(t3)] -= 1;
    }
    /* covid.h:178-202; nTrajectories is a size_t there, so a[i]/nTrajectories divides
     * by the exactly converted double. */
    double nt = (double)n_traj;
    for (int c = 0; c < ORC_NCURVES; ++c) a[c][0] /= nt;
    for (size_t t = 1; t < ORC_NGRID; ++t)
        for (int c = 0; c < ORC_NCURVES; ++c) a[c][t] = a[c][t - 1] + a[c][t] / nt;
}

/* covid.h:221-235 */
double orc_traj_get_day(const double* ar, size_t day) {
    if (day == 0) return ar[0];
    if (day == ORC_NDAYS) return 0;
    size_t i1 = (size_t)(ORC_NGRID * ((double)(day - 1) / ORC_NDAYS));
    size_t i2 = (size_t)(ORC_NGRID * ((double)day / ORC_NDAYS));
    if (i2 > ORC_NGRID - 1) return 0;
    return ar[i2] - ar[i1];
}

/* the cumsum that project() (covid.h:377-384) forms is independent of k and n */
void orc_traj_response(const double* grid, int n_lag, double* R) {
    for (int c = 0; c < ORC_NCURVES; ++c) {
        double cumsum = 0;
        for (int d = 0; d < n_lag; ++d) {
            cumsum += orc_traj_get_day(grid + (size_t)c * ORC_NGRID, (size_t)d);
            R[(size_t)c * n_lag + d] = cumsum;
        }
    }
}

/* ------------------------------------------------------ DiseaseSpread::eval ---- */
/* covid.h:376-386: dest[d+k] += n*cumsum for d < min(nDays, size-k) */
static void project(const orc_covid_model* m, int size, int k, double* dest, int curve, double n) {
    if (m->faithful && m->grid) {
        const double* src = m->grid + (size_t)curve * ORC_NGRID;
        double cumsum = 0;
        for (int d = 0; d < ORC_NDAYS && d + k < size; ++d) {
            cumsum += orc_traj_get_day(src, (size_t)d);
            if (m->contract) dest[d + k] = fma(n, cumsum, dest[d + k]);
            else             dest[d + k] += n * cumsum;
        }
        return;
    }
    const double* Rc = m->R + (size_t)curve * m->n_lag;
    if (m->contract) {
        for (int d = 0; d < ORC_NDAYS && d + k < size; ++d) dest[d + k] = fma(n, Rc[d], dest[d + k]);
    } else {
        for (int d = 0; d < ORC_NDAYS && d + k < size; ++d) dest[d + k] += n * Rc[d];
    }
}
/* covid.h:387-391 */
static void add_flat(int size, int k, double* dest, double n) {
    for (int d = 0; d < ORC_NDAYS && d + k < size; ++d) dest[d + k] += n;
}

double orc_covid_eval(const orc_covid_model* m, const double* x, double* curves, int sim_days) {
    const int size = orc_size(m);                                   /* covid.h:327 */
    const int nB = orc_nb(m), nDisc = m->n_disc, md = m->max_delay;
    const double* behavior = x;
    const double* disc = x + nB;
    const double betaMild = x[nB + nDisc], betaHigh = x[nB + nDisc + 1], delay = x[nB + nDisc + 2];
    const int horizon = (sim_days > 0 && sim_days < size) ? sim_days : size;

    /* 8 buffers (covid.h:653-660) + the 9 output coords (covid.h:302-311) */
    double* buf = (double*)calloc((size_t)8 * size, sizeof(double));
    double* out = (double*)calloc((size_t)9 * size, sizeof(double));
    double *bMI = buf, *bHI = buf + size, *bInc = buf + 2 * size, *bAsy = buf + 3 * size,
           *bMild = buf + 4 * size, *bSer = buf + 5 * size, *bRec = buf + 6 * size,
           *bDead = buf + 7 * size;
    double* oDead = out + 7 * size;
    double* oTB = out + 8 * size;

    double loglike = 0;                                             /* covid.h:329 */
    int start = (int)(md - delay);                                  /* covid.h:331 */
    double frac = md - delay - start;                               /* covid.h:332 */

    for (int shift = 1; shift >= 0; --shift) {                      /* covid.h:335 */
        const double w = (1 - shift) * (1 - frac) + shift * frac;   /* covid.h:333 */
        double newlyInfected = 1;                                   /* covid.h:339 */
        for (int i = 0; i < size; ++i) oTB[i] = 1;                  /* covid.h:349 */
        memset(buf, 0, sizeof(double) * 8 * size);                  /* covid.h:351-358 */

        for (int i = start + shift; i < horizon; ++i) {             /* covid.h:373 */
            project(m, size, i, bMI, C_INFMILD, newlyInfected);     /* covid.h:403-410 */
            project(m, size, i, bHI, C_INFHIGH, newlyInfected);
            project(m, size, i, bInc, C_INCUB, newlyInfected);
            project(m, size, i, bAsy, C_MILD, newlyInfected);       /* traj.mild: covid.h:406 */
            project(m, size, i, bMild, C_MILD, newlyInfected);
            project(m, size, i, bSer, C_SERIOUS, newlyInfected);
            project(m, size, i, bRec, C_RECOV, newlyInfected);
            project(m, size, i, bDead, C_DEAD, newlyInfected);

            double capacity = m->cap0 + fmax(m->cap_incr_rate * (i - md), 0.0); /* covid.h:314 */
            double overCapacity = bSer[i] * m->prob_icu_if_serious - capacity;  /* covid.h:413 */
            double extraDeaths = m->prob_lethal_daily_untreated * overCapacity; /* covid.h:414 */
            if (extraDeaths > 0) {                                  /* covid.h:416-429 */
                add_flat(size, i, bDead, extraDeaths);
                project(m, size, i, bSer, C_SFS, -extraDeaths);
                project(m, size, i, bRec, C_RFS, -extraDeaths);
                project(m, size, i, bDead, C_DFS, -extraDeaths);
            }

            int found = -1;                                         /* covid.h:432-444 */
            for (int mm = 0; mm + 1 < nDisc; ++mm) {
                if ((m->disc_days[mm] <= i - md) && (i - md < m->disc_days[mm + 1])) { found = mm; break; }
            }
            double pcf = 1;
            if ((found == -1) && (i - md >= m->disc_days[nDisc - 1])) found = nDisc - 1;
            if (found != -1) pcf = disc[found];

            double smoothbehavior = 1;                              /* covid.h:446-452 */
            if (i >= md) {
                if (i - md < nB) smoothbehavior = pcf * behavior[i - md];
                else             smoothbehavior = pcf * behavior[nB - 1];
            }
            oTB[i] = pcf * smoothbehavior;                          /* covid.h:454 */

            double nSusceptible = m->pop_size - bInc[i] - bAsy[i] - bMild[i] - bSer[i] - bRec[i] - bDead[i];
            newlyInfected = oTB[i] * nSusceptible / m->pop_size *
                            (betaMild * bMI[i] + betaHigh * bHI[i]); /* covid.h:461-462 */
        }

        /* covid.h:467-485: blend into the output coords */
        for (int c = 0; c < 8; ++c) {
            double* o = out + (size_t)c * size;
            const double* b = buf + (size_t)c * size;
            if (shift == 1) for (int i = 0; i < size; ++i) o[i] = w * b[i];
            else            for (int i = 0; i < size; ++i) o[i] += w * b[i];
        }
        /* covid.h:488-492: uses the BLENDED-so-far output coords "dead" */
        for (int i = 0; i < m->n_data; ++i) {
            double delta = oDead[i + md] - m->deaths[i];
            loglike += w * (-0.5 * delta * delta / (m->sigma[i] * m->sigma[i]));
        }
    }
    if (curves) memcpy(curves, out, sizeof(double) * 9 * size);
    free(buf);
    free(out);
    return loglike;
}

/* mcmc.h:197-214 */
double orc_smooth_eval(const double* f, int n, double Ls, double L) {
    double dx = L / n;
    double loglike = 0;
    for (int i = 1; i + 1 < n; ++i) {
        double dd = f[i + 1] + f[i - 1] - 2 * f[i];
        loglike -= dd * dd / dx / dx / dx;
    }
    loglike *= Ls * Ls / L;
    return loglike;
}

static double orc_cos(const orc_covid_model* m, double x) { return m->use_libm ? cos(x) : mcmc_cos(x); }
static double orc_exp(int use_libm, double x) { return use_libm ? exp(x) : mcmc_exp(x); }

/* covid.h:499-622 (everything before newstate->eval) */
void orc_covid_propose(const orc_covid_model* m, orc_pcg32* rnd, const double* x,
                       double stepsize, double* xp) {
    orc_covid_propose_part(m, rnd, x, stepsize, xp, 1);
}
/* with_betas = 0 (multi-region extension): a region that does not own the shared betas skips the
 * betas' branch altogether, gate draw included */
void orc_covid_propose_part(const orc_covid_model* m, orc_pcg32* rnd, const double* x,
                            double stepsize, double* xp, int with_betas) {
    const int nB = orc_nb(m), nDisc = m->n_disc, dim = orc_dim(m);
    memcpy(xp, x, sizeof(double) * dim);                            /* covid.h:501 copy() */
    double* behavior = xp;
    double* disc = xp + nB;
    double* betaMild = xp + nB + nDisc;
    double* betaHigh = xp + nB + nDisc + 1;
    double* delay = xp + nB + nDisc + 2;
    const double sfmin = fmin(stepsize, 1.0);

    int big1 = orc_pcg32_next_float(rnd) < 0.6f;                    /* covid.h:503-507 */
    int big2 = orc_pcg32_next_float(rnd) < 0.6f;
    int big3 = orc_pcg32_next_float(rnd) < 0.6f;
    int big4 = orc_pcg32_next_float(rnd) < 0.6f;
    int big5 = orc_pcg32_next_float(rnd) < 0.6f;

    if (orc_pcg32_next_float(rnd) < 0.3f) {                         /* covid.h:510-515 */
        *delay += (orc_pcg32_next_double(rnd) - 0.5) * 0.5 * 0.1 * sfmin;
        if (big1) *delay += (orc_pcg32_next_double(rnd) - 0.5) * sfmin;
        *delay = orc_bound(*delay, 5.0, (double)m->max_delay);
    }
    if (with_betas && orc_pcg32_next_float(rnd) < 0.3f) {           /* covid.h:518-532 */
        if (!big2) *betaMild += (orc_pcg32_next_double(rnd) - 0.5) * 0.1 * sfmin;
        else       *betaMild += (orc_pcg32_next_double(rnd) - 0.5) * sfmin;
        *betaMild = orc_bound(*betaMild, 0.0, 100.0);
        if (!big3) *betaHigh += (orc_pcg32_next_double(rnd) - 0.5) * 0.1 * sfmin;
        else       *betaHigh += (orc_pcg32_next_double(rnd) - 0.5) * sfmin;
        *betaHigh = orc_bound(*betaHigh, 0.0, 100.0);
        if (*betaHigh < *betaMild) { double t = *betaHigh; *betaHigh = *betaMild; *betaMild = t; }
    }
    if (orc_pcg32_next_float(rnd) < 0.9f) {                         /* covid.h:557-598 */
        float nNotFixed = 0;                                        /* counts the FIXED ones */
        for (int i = 0; i < nDisc; ++i) if (m->disc_fixed[i]) nNotFixed += 1;
        int start = 0, stop = nDisc, incr = 1;
        if (orc_pcg32_next_double(rnd) < 0.5) { start = nDisc - 1; stop = -1; incr = -1; }
        for (int i = start; i != stop; i += incr) {
            if (m->disc_fixed[i]) continue;
            float xx = orc_pcg32_next_float(rnd);
            if (xx < 2 / nNotFixed) {                               /* float division, +inf if 0 */
                if (!big4) disc[i] += (orc_pcg32_next_double(rnd) - 0.5) * 0.1 * sfmin;
                else       disc[i] += (orc_pcg32_next_double(rnd) - 0.5) * sfmin;
                double lower = 0, upper = 1;
                if (i >= 1) upper = disc[i - 1];
                if (i < nDisc - 1) lower = disc[i + 1];
                disc[i] = orc_bound(disc[i], lower, upper);
            }
        }
    }
    if (orc_pcg32_next_float(rnd) < 0.3f) {                         /* covid.h:601-622 */
        double T = (double)m->n_data;
        double xx = orc_pcg32_next_double(rnd);
        double omega = 2 * 3.1415 / T * 5 * xx * xx;
        double A = sfmin * 0.1 * (orc_pcg32_next_double(rnd) - 0.5);
        if (big5) A += sfmin * (orc_pcg32_next_double(rnd) - 0.5);
        for (int i = 0; i < nB; ++i) {
            behavior[i] += A * (orc_cos(m, omega * i) - 1);
            behavior[i] = orc_bound(behavior[i], 0.0, 2.0);
        }
    }
}

/* covid.h:629-646 */
void orc_covid_force_bounds(const orc_covid_model* m, double* x) {
    const int nB = orc_nb(m), nDisc = m->n_disc;
    double* disc = x + nB;
    x[nB + nDisc + 2] = orc_bound(x[nB + nDisc + 2], 5.0, (double)m->max_delay);
    double lastVal = 1;
    for (int i = 0; i < nDisc; ++i) {
        if (m->disc_fixed[i]) continue;
        disc[i] = orc_bound(disc[i], 0.0, lastVal);
        lastVal = disc[i];
    }
    for (int i = 0; i < nB; ++i) x[i] = orc_bound(x[i], 0.0, 2.0);
    x[nB + nDisc] = orc_bound(x[nB + nDisc], 0.0, 100.0);
    x[nB + nDisc + 1] = orc_bound(x[nB + nDisc + 1], 0.0, 100.0);
}

/* mcmc.h:742-744, 806-808, 829-831 */
double orc_target_beta(const orc_target* t, double time) {
    switch (t->kind) {
        case ORC_TARGET_TEMP: return 1 / t->T;      /* bet(1/T), mcmc.h:801 */
        case ORC_TARGET_COOLING: return exp(time * t->slope) / t->Tinit;
        default: return 1;
    }
}

/* --------------------------------------------------------------- the chain ---- */
void orc_chain_init(const orc_covid_model* m, orc_chain* c, int chain_id, const double* x0,
                    double stepsize) {
    const int dim = orc_dim(m);
    memset(c, 0, sizeof(*c));
    orc_pcg32_seed(&c->rng, 0, (uint64_t)chain_id);                 /* mcmc.h:999 */
    c->x = (double*)malloc(sizeof(double) * dim);
    c->xp = (double*)malloc(sizeof(double) * dim);
    memcpy(c->x, x0, sizeof(double) * dim);
    c->stepsize = stepsize;
    if (m->faithful && m->grid) {                                   /* traj held by value, covid.h:290 */
        c->grid_own = (double*)malloc(sizeof(double) * ORC_NCURVES * ORC_NGRID);
        memcpy(c->grid_own, m->grid, sizeof(double) * ORC_NCURVES * ORC_NGRID);
    }
    /* State::init/eval (mcmc.h:385-393): deterministic, so one sweep equals ten */
    c->ll_model = orc_covid_eval(m, c->x, NULL, 0);
    c->ll_prior = m->has_prior ? orc_smooth_eval(c->x, orc_nb(m), m->smooth_Ls, m->smooth_L) : 0.0;
}
void orc_chain_free(orc_chain* c) { free(c->x); free(c->xp); free(c->grid_own); c->x = c->xp = c->grid_own = NULL; }

/* Target::step (mcmc.h:750-783) on State=[DiseaseSpread(,SmoothnessPrior)] */
int orc_chain_step(const orc_covid_model* m, const orc_target* t, orc_chain* c, double time,
                   int random_subspace) {
    const int nState = m->has_prior ? 2 : 1;
    if (random_subspace) {                                          /* mcmc.h:494-500 */
        size_t head;
        do {
            float f = orc_pcg32_next_float(&c->rng) * nState;
            head = (size_t)f;
            if (head > (size_t)(nState - 1)) head = nState - 1;
            /* SmoothnessPrior has no coords: coords.size()==nDerived==0 -> redraw */
        } while (head != 0);
    }
    /* head is always 0 (given-subspace steps reuse rec.whichHead of the same iteration) */
    orc_covid_propose(m, &c->rng, c->x, c->stepsize, c->xp);        /* mcmc.h:513 */
    /* faithful: the proposal is a copy() of the whole DiseaseSpread (covid.h:501, HAS_STEP
     * mcmc.h:55-56), trajectory arrays included: 11 fresh vectors of nGrid doubles, which the
     * proposal's eval then reads; the loser of the accept test is freed */
    orc_covid_model mm;
    const orc_covid_model* me = m;
    double* grid_prop = NULL;
    if (c->grid_own) {
        grid_prop = (double*)malloc(sizeof(double) * ORC_NCURVES * ORC_NGRID);
        for (int cv = 0; cv < ORC_NCURVES; ++cv)
            memcpy(grid_prop + (size_t)cv * ORC_NGRID, c->grid_own + (size_t)cv * ORC_NGRID, sizeof(double) * ORC_NGRID);
        mm = *m; mm.grid = grid_prop; me = &mm;
    }
    double llm = orc_covid_eval(me, c->xp, NULL, 0);                /* covid.h:624 */
    double llp = 0.0;
    double delta = 0;                                               /* mcmc.h:604-606 */
    delta += -c->ll_model + llm;
    if (m->has_prior) {                                             /* schedule [0 -> 1] */
        llp = orc_smooth_eval(c->xp, orc_nb(m), m->smooth_Ls, m->smooth_L);
        delta += -c->ll_prior + llp;
    }
    delta += (log(1.0) - log(1.0));                                 /* mcmc.h:765, weight()==1 */
    delta *= orc_target_beta(t, time);                              /* mcmc.h:766 */
    c->last_delta = delta;
    c->n_steps++;
    int acc = 0;
    if ((double)orc_pcg32_next_float(&c->rng) < 1.0 * orc_exp(m->use_libm, delta)) { /* mcmc.h:776 */
        double* tmp = c->x; c->x = c->xp; c->xp = tmp;              /* State::accept */
        c->ll_model = llm;
        c->ll_prior = llp;
        c->n_accept++;
        acc = 1;
        if (grid_prop) { double* g = c->grid_own; c->grid_own = grid_prop; grid_prop = g; }
    }
    free(grid_prop);
    c->last_accept = acc;
    return acc;
}

int orc_n_saved(int nSamples, int nBurnin, int nAdjust, int thinning) {
    int n = 0;
    for (int i = 0; i < nSamples + nAdjust; ++i)
        if ((i >= nBurnin) && (i % thinning) == 0) ++n;
    return n;
}

/* mcmc.h:1110-1112 */
static double rate2corr(double x) {
    return (1 + 0.726484 * x * x * x * x) / (0.82051 + 0.0427315 / (x + 0.0001));
}

/* MetropolisChain::run (mcmc.h:1005-1138) */
int orc_chain_run(const orc_covid_model* m, const orc_target* t, orc_chain* c, int nSamples,
                  int nBurnin, int nAdjust, int thinning, double* loglikes_out,
                  double* samples_out, double* mean_out) {
    const int dim = orc_dim(m);
    int n_saved = 0;
    c->n_accept = 0;                                                /* mcmc.h:1007 */
    if (mean_out) memcpy(mean_out, c->x, sizeof(double) * dim);     /* mcmc.h:1002 */
    for (int i = 0; i < nSamples + nAdjust; ++i) {                  /* mcmc.h:1023 */
        if (i == nAdjust) c->n_accept = 0;                          /* mcmc.h:1032 */
        if ((i >= nBurnin) && (i % thinning) == 0) {                /* mcmc.h:1037-1062 */
            double time = (i < nAdjust) ? (double)i / nAdjust : (double)(i - nAdjust) / nSamples;
            double ll = 0;
            ll += c->ll_model;                                      /* State::loglikelihood */
            if (m->has_prior) ll += c->ll_prior;
            if (loglikes_out) loglikes_out[n_saved] = log(1.0) + orc_target_beta(t, time) * ll;
            if (samples_out) memcpy(samples_out + (size_t)n_saved * dim, c->x, sizeof(double) * dim);
            ++n_saved;
        }
        if ((i > nBurnin) && mean_out)                              /* mcmc.h:1064-1065 */
            for (int k = 0; k < dim; ++k) mean_out[k] += c->x[k];
        if (i < nAdjust) {                                          /* mcmc.h:1083-1119 */
            double time = (double)i / nAdjust;
            c->n_accept = 0;
            orc_chain_step(m, t, c, time, 1);
            const int nRepeat = 20;
            for (int j = 0; j < nRepeat - 1; ++j) orc_chain_step(m, t, c, time, 0);
            double acceptRate = (double)c->n_accept / nRepeat;
            c->stepsize *= rate2corr(acceptRate);                   /* mcmc.h:1114, 663-665 */
        } else {
            orc_chain_step(m, t, c, (double)(i - nAdjust) / nSamples, 1); /* mcmc.h:1126 */
        }
    }
    if (mean_out)                                                   /* mcmc.h:1133-1134 */
        for (int k = 0; k < dim; ++k) mean_out[k] *= 1.0 / (nAdjust + nSamples - nBurnin);
    return n_saved;
}

/* ------------------------------------------------------- multi-region model ---- */
/* EXTENSION (BASELINE.json configs[4]; the reference has one region only, covid.h:284-321).
 * A chain's State is R region blocks [DiseaseSpread_r (, SmoothnessPrior_r)], each a complete
 * covid.h model with data, population and hospital capacity of its own, all sharing betaMild and
 * betaHigh, which region 0 owns.  Parameter vector: R blocks of the single-region layout; the beta
 * slots of every block mirror region 0's.  One MH step moves ONE region, the head, chosen by
 * systematic scan -- head = (number of random-subspace steps so far) mod R, a given-subspace step
 * reuses the head -- so the head is the same for every chain and a wavefront never diverges on it:
 *   - the subspace draw consumes what the single-region State's does (mcmc.h:494-500);
 *   - DiseaseSpread::step's draws (covid.h:499-622) on the head's block, the betas' branch only for
 *     region 0;
 *   - the head is evaluated; if the betas changed, every other region is re-evaluated with them
 *     (its dependents in the sense of mcmc.h:539-602);
 *   - delta = sum over the evaluated likelihoods in State order (DS_0, P_0, DS_1, P_1, ...) of
 *     (-old + new) (mcmc.h:604-606), accept as in Target::step (mcmc.h:762-781);
 *   - the adjust phase corrects the head's own stepsizeCorrectionFac (mcmc.h:1107-1114).
 * With R = 1 this is orc_chain_step / orc_chain_run exactly. */
void orc_multi_init(const orc_covid_model* regs, int R, orc_multi_chain* c, int chain_id, const double* x0,
                    double stepsize) {
    const int dimR = orc_dim(&regs[0]), nB = orc_nb(&regs[0]), nDisc = regs[0].n_disc;
    memset(c, 0, sizeof(*c));
    orc_pcg32_seed(&c->rng, 0, (uint64_t)chain_id);
    c->x = (double*)malloc(sizeof(double) * dimR * R);
    c->xp = (double*)malloc(sizeof(double) * dimR);
    c->ll = (double*)calloc((size_t)2 * R, sizeof(double));
    c->llp = (double*)calloc((size_t)R, sizeof(double));
    c->stepsize = (double*)malloc(sizeof(double) * R);
    memcpy(c->x, x0, sizeof(double) * dimR * R);
    for (int r = 0; r < R; ++r) {
        c->stepsize[r] = stepsize;
        c->x[r * dimR + nB + nDisc] = c->x[nB + nDisc];             /* the shared betas are region 0's */
        c->x[r * dimR + nB + nDisc + 1] = c->x[nB + nDisc + 1];
        c->ll[2 * r] = orc_covid_eval(&regs[r], c->x + r * dimR, NULL, 0);
        c->ll[2 * r + 1] = regs[r].has_prior ? orc_smooth_eval(c->x + r * dimR, nB, regs[r].smooth_Ls, regs[r].smooth_L) : 0.0;
    }
}
void orc_multi_free(orc_multi_chain* c) {
    free(c->x); free(c->xp); free(c->ll); free(c->llp); free(c->stepsize);
    memset(c, 0, sizeof(*c));
}

int orc_multi_step(const orc_covid_model* regs, int R, const orc_target* t, orc_multi_chain* c, double time,
                   int random_subspace) {
    const orc_covid_model* m0 = &regs[0];
    const int dimR = orc_dim(m0), nB = orc_nb(m0), nDisc = m0->n_disc;
    const int nState = m0->has_prior ? 2 : 1;
    if (random_subspace) {
        size_t head;
        do {                                                        /* mcmc.h:494-500, as orc_chain_step */
            float f = orc_pcg32_next_float(&c->rng) * nState;
            head = (size_t)f;
            if (head > (size_t)(nState - 1)) head = nState - 1;
        } while (head != 0);
        c->head = (int)(c->head_counter % (uint64_t)R);             /* systematic scan over the regions */
        c->head_counter++;
    }
    const int h = c->head;
    double* xh = c->x + (size_t)h * dimR;
    orc_covid_propose_part(&regs[h], &c->rng, xh, c->stepsize[h], c->xp, h == 0);
    const double bm = c->xp[nB + nDisc], bh = c->xp[nB + nDisc + 1];
    const int all = (bm != xh[nB + nDisc]) || (bh != xh[nB + nDisc + 1]);
    double llpH = 0.0, delta = 0;
    double* tmp = (double*)malloc(sizeof(double) * dimR);
    for (int q = 0; q < R; ++q) {                                   /* State order */
        if (q != h && !all) continue;
        double llq;
        if (q == h) {
            llq = orc_covid_eval(&regs[q], c->xp, NULL, 0);
        } else {                                                    /* a dependent: its own coords, the new betas */
            memcpy(tmp, c->x + (size_t)q * dimR, sizeof(double) * dimR);
            tmp[nB + nDisc] = bm; tmp[nB + nDisc + 1] = bh;
            llq = orc_covid_eval(&regs[q], tmp, NULL, 0);
        }
        c->llp[q] = llq;
        delta += -c->ll[2 * q] + llq;
        if (q == h && m0->has_prior) {
            llpH = orc_smooth_eval(c->xp, nB, regs[h].smooth_Ls, regs[h].smooth_L);
            delta += -c->ll[2 * h + 1] + llpH;
        }
    }
    free(tmp);
    delta += (log(1.0) - log(1.0));
    delta *= orc_target_beta(t, time);
    c->n_steps++;
    int acc = 0;
    if ((double)orc_pcg32_next_float(&c->rng) < 1.0 * orc_exp(m0->use_libm, delta)) {
        memcpy(xh, c->xp, sizeof(double) * dimR);
        for (int q = 0; q < R; ++q) {
            if (q != h && !all) continue;
            c->ll[2 * q] = c->llp[q];
            c->x[(size_t)q * dimR + nB + nDisc] = bm;
            c->x[(size_t)q * dimR + nB + nDisc + 1] = bh;
        }
        if (m0->has_prior) c->ll[2 * h + 1] = llpH;
        c->n_accept++;
        acc = 1;
    }
    return acc;
}

int orc_multi_run(const orc_covid_model* regs, int R, const orc_target* t, orc_multi_chain* c, int nSamples,
                  int nBurnin, int nAdjust, int thinning, double* loglikes_out, double* samples_out, double* mean_out) {
    const int dim = orc_dim(&regs[0]) * R;
    int n_saved = 0;
    c->n_accept = 0;
    if (mean_out) memcpy(mean_out, c->x, sizeof(double) * dim);
    for (int i = 0; i < nSamples + nAdjust; ++i) {                  /* mcmc.h:1023, as orc_chain_run */
        if (i == nAdjust) c->n_accept = 0;
        if ((i >= nBurnin) && (i % thinning) == 0) {
            double time = (i < nAdjust) ? (double)i / nAdjust : (double)(i - nAdjust) / nSamples;
            double ll = 0;
            for (int q = 0; q < R; ++q) { ll += c->ll[2 * q]; if (regs[0].has_prior) ll += c->ll[2 * q + 1]; }
            if (loglikes_out) loglikes_out[n_saved] = log(1.0) + orc_target_beta(t, time) * ll;
            if (samples_out) memcpy(samples_out + (size_t)n_saved * dim, c->x, sizeof(double) * dim);
            ++n_saved;
        }
        if ((i > nBurnin) && mean_out)
            for (int k = 0; k < dim; ++k) mean_out[k] += c->x[k];
        if (i < nAdjust) {
            double time = (double)i / nAdjust;
            c->n_accept = 0;
            orc_multi_step(regs, R, t, c, time, 1);
            const int nRepeat = 20;
            for (int j = 0; j < nRepeat - 1; ++j) orc_multi_step(regs, R, t, c, time, 0);
            c->stepsize[c->head] *= rate2corr((double)c->n_accept / nRepeat);
        } else {
            orc_multi_step(regs, R, t, c, (double)(i - nAdjust) / nSamples, 1);
        }
    }
    if (mean_out)
        for (int k = 0; k < dim; ++k) mean_out[k] *= 1.0 / (nAdjust + nSamples - nBurnin);
    return n_saved;
}

/* ------------------------------------------------ ChainManager::run_all_adjust ---- */
typedef struct {
    const orc_covid_model* m; const orc_target* t;
    int n_chains, first_id; double* x; double* stepsizes;
    int nSamples, nBurnin, nAdjust, thinning;
    double* final_ll; int* n_accept; uint64_t* rng_state; double* loglikes; int n_saved;
    volatile int next; pthread_mutex_t mu;
} pool_t;

static void* pool_worker(void* arg) {
    pool_t* p = (pool_t*)arg;
    const int dim = orc_dim(p->m);
    for (;;) {
        pthread_mutex_lock(&p->mu);
        int i = p->next++;
        pthread_mutex_unlock(&p->mu);
        if (i >= p->n_chains) break;
        orc_chain c;
        orc_chain_init(p->m, &c, p->first_id + i, p->x + (size_t)i * dim, p->stepsizes[i]);
        orc_chain_run(p->m, p->t, &c, p->nSamples, p->nBurnin, p->nAdjust, p->thinning,
                      p->loglikes ? p->loglikes + (size_t)i * p->n_saved : NULL, NULL, NULL);
        memcpy(p->x + (size_t)i * dim, c.x, sizeof(double) * dim);
        p->stepsizes[i] = c.stepsize;
        if (p->final_ll) { p->final_ll[2 * i] = c.ll_model; p->final_ll[2 * i + 1] = c.ll_prior; }
        if (p->n_accept) p->n_accept[i] = c.n_accept;
        if (p->rng_state) { p->rng_state[2 * i] = c.rng.state; p->rng_state[2 * i + 1] = c.rng.inc; }
        orc_chain_free(&c);
    }
    return NULL;
}

double orc_run_chains(const orc_covid_model* m, const orc_target* t, int n_chains,
                      int first_chain_id, double* x, double* stepsizes, int nSamples,
                      int nBurnin, int nAdjust, int thinning, int n_threads, double* final_ll,
                      int* n_accept, uint64_t* rng_state, double* loglikes_out) {
    pool_t p;
    memset(&p, 0, sizeof(p));
    p.m = m; p.t = t; p.n_chains = n_chains; p.first_id = first_chain_id; p.x = x;
    p.stepsizes = stepsizes; p.nSamples = nSamples; p.nBurnin = nBurnin; p.nAdjust = nAdjust;
    p.thinning = thinning; p.final_ll = final_ll; p.n_accept = n_accept; p.rng_state = rng_state;
    p.loglikes = loglikes_out; p.n_saved = orc_n_saved(nSamples, nBurnin, nAdjust, thinning);
    pthread_mutex_init(&p.mu, NULL);
    if (n_threads < 1) n_threads = 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * n_threads);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int k = 0; k < n_threads; ++k) pthread_create(&th[k], NULL, pool_worker, &p);
    for (int k = 0; k < n_threads; ++k) pthread_join(th[k], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th);
    pthread_mutex_destroy(&p.mu);
    return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}

/* ------------------------------------------------------------ toy targets ---- */
/* Generic single-SubspaceState run loop (mcmc.h:1005-1138 with State.size()==1, where
 * step_random_subspace still consumes one nextFloat: mcmc.h:498). */
typedef double (*toy_eval_fn)(const double* x, const void* ud);
typedef void (*toy_prop_fn)(orc_pcg32* r, const double* x, double sf, double* xp, const void* ud);

static int toy_run(int dimx, toy_eval_fn ev, toy_prop_fn pr, const void* ud, double beta,
                   int chain_id, double* x, int nSamples, int nBurnin, int nAdjust, int thinning,
                   double* loglikes_out, double* samples_out, double* stepsize_out, int* n_accept_out,
                   double* mean_out /* [dimx + with_max] or NULL */, int with_max, double stepsize0) {
    orc_pcg32 rng;
    orc_pcg32_seed(&rng, 0, (uint64_t)chain_id);
    double xp[8], mean[9];
    double ll = ev(x, ud), sf = stepsize0;
    int nAccept = 0, n_saved = 0;
    /* mean = deep copy of the evaluated state (mcmc.h:1002); MyLike1's eval also sets the
     * coordinate "max" (test.h:37), which addCoordsOf sums like any other (mcmc.h:273-281) */
    memcpy(mean, x, sizeof(double) * dimx);
    if (with_max) mean[dimx] = fmax(fmax(x[0], x[1]), x[2]);
    for (int i = 0; i < nSamples + nAdjust; ++i) {
        if (i == nAdjust) nAccept = 0;
        if ((i >= nBurnin) && (i % thinning) == 0) {
            double l = 0; l += ll;
            if (loglikes_out) loglikes_out[n_saved] = log(1.0) + beta * l;
            if (samples_out) memcpy(samples_out + (size_t)n_saved * dimx, x, sizeof(double) * dimx);
            ++n_saved;
        }
        if (i > nBurnin && mean_out) {                              /* mcmc.h:1064-1065 */
            for (int k = 0; k < dimx; ++k) mean[k] += x[k];
            if (with_max) mean[dimx] += fmax(fmax(x[0], x[1]), x[2]);
        }
        int reps = 1;
        if (i < nAdjust) { nAccept = 0; reps = 20; }
        for (int j = 0; j < reps; ++j) {
            if (j == 0) {                                           /* random subspace */
                size_t head;
                do { head = (size_t)(orc_pcg32_next_float(&rng) * 1); if (head > 0) head = 0; } while (0);
            }
            pr(&rng, x, sf, xp, ud);
            double llp = ev(xp, ud);
            double delta = 0; delta += -ll + llp;
            delta += (log(1.0) - log(1.0));
            delta *= beta;
            if ((double)orc_pcg32_next_float(&rng) < 1.0 * mcmc_exp(delta)) {
                memcpy(x, xp, sizeof(double) * dimx); ll = llp; ++nAccept;
            }
        }
        if (i < nAdjust) sf *= rate2corr((double)nAccept / 20);
    }
    if (stepsize_out) *stepsize_out = sf;
    if (n_accept_out) *n_accept_out = nAccept;
    if (mean_out) {                                                 /* mcmc.h:1133-1134 */
        const double f = 1.0 / (double)(nAdjust + nSamples - nBurnin);
        for (int k = 0; k < dimx + (with_max ? 1 : 0); ++k) mean_out[k] = mean[k] * f;
    }
    return n_saved;
}

/* MyLike1 (test.h:14-67) */
static double mylike1_eval(const double* x, const void* ud) {
    (void)ud;
    static const double dp[2][3] = {{2, 2, 0}, {-2, -2, 0}};
    double loglike = 0;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) loglike -= (dp[i][j] - x[j]) * (dp[i][j] - x[j]) * 0.5;
    return loglike;
}
static void mylike1_prop(orc_pcg32* r, const double* x, double sf, double* xp, const void* ud) {
    (void)ud;
    for (int j = 0; j < 3; ++j) xp[j] = x[j] + 10 * sf * (orc_pcg32_next_float(r) - 0.5); /* test.h:50 */
}
int orc_mylike1_run(int chain_id, int nSamples, int nBurnin, int nAdjust, int thinning,
                    double* loglikes_out, double* samples_out, double* stepsize_out, int* n_accept_out,
                    double* mean_out /* [4]: position, max */) {
    double x[3] = {1, 1, 1};                                        /* test.h:20 */
    return toy_run(3, mylike1_eval, mylike1_prop, NULL, 1.0, chain_id, x, nSamples, nBurnin,
                   nAdjust, thinning, loglikes_out, samples_out, stepsize_out, n_accept_out, mean_out, 1, 1.0);
}

/* FourGaussians (test.h:69-136) */
static double fourg_eval(const double* p, const void* ud) {
    double difficulty = *(const double*)ud;
    double x = p[0], y = p[1];
    double loglike;
    /* std::exp / std::log -> include/mcmc_detmath.h (<= 1 ulp from glibc), as in the HIP path */
    loglike  = mcmc_exp(-difficulty * ((x + 0.5) * (x + 0.5) / (2 * 0.5 * 0.5) + (y + 1) * (y + 1) / (2 * 0.2 * 0.2)));
    loglike += mcmc_exp(-difficulty * ((x - 1) * (x - 1) / (2 * 0.2 * 0.2) + (y - 0.5) * (y - 0.5) / (2 * 0.5 * 0.5)));
    loglike += mcmc_exp(-difficulty * ((x + y) * (x + y) / (2 * 1) + (x - y - 2) * (x - y - 2) / (2 * 0.2 * 0.2)));
    loglike += mcmc_exp(-difficulty * ((x + y) * (x + y) / (2 * 1) + (x - y + 2) * (x - y + 2) / (2 * 0.1)));
    return mcmc_log(loglike);
}
static void fourg_prop(orc_pcg32* r, const double* x, double sf, double* xp, const void* ud) {
    (void)ud;
    for (int j = 0; j < 2; ++j) {
        xp[j] = x[j] + sf * (orc_pcg32_next_float(r) - 0.5);        /* test.h:105 */
        xp[j] = orc_bound(xp[j], -5.0, 5.0);
    }
}
int orc_fourgauss_run(double difficulty, double T, int chain_id, const double* x0, int nSamples,
                      int nBurnin, int nAdjust, int thinning, double* loglikes_out,
                      double* samples_out, double* stepsize_out, int* n_accept_out, double* mean_out /* [2] */,
                      double stepsize0) {
    double x[2] = {x0 ? x0[0] : 0.0, x0 ? x0[1] : 0.0};             /* test.h:75 */
    return toy_run(2, fourg_eval, fourg_prop, &difficulty, 1 / T, chain_id, x, nSamples, nBurnin,
                   nAdjust, thinning, loglikes_out, samples_out, stepsize_out, n_accept_out, mean_out, 0, stepsize0);
}

/* Rosenbrock (EXTENSION, BASELINE configs[0]; no counterpart in the reference): FourGaussians'
 * pattern (test.h:69-136) with loglike = -((1-x)^2 + difficulty (y-x^2)^2) / 20 and an unbounded
 * random-walk step (test.h:105 without the bound() of test.h:106) */
static double rosen_eval(const double* p, const void* ud) {
    double difficulty = *(const double*)ud;
    double u = 1 - p[0], v = p[1] - p[0] * p[0];
    return -(u * u + difficulty * (v * v)) / 20;
}
static void rosen_prop(orc_pcg32* r, const double* x, double sf, double* xp, const void* ud) {
    (void)ud;
    for (int j = 0; j < 2; ++j) xp[j] = x[j] + sf * (orc_pcg32_next_float(r) - 0.5);
}
int orc_rosenbrock_run(double difficulty, double T, int chain_id, const double* x0, int nSamples,
                       int nBurnin, int nAdjust, int thinning, double* loglikes_out,
                       double* samples_out, double* stepsize_out, int* n_accept_out, double* mean_out /* [2] */,
                       double stepsize0) {
    double x[2] = {x0 ? x0[0] : 0.0, x0 ? x0[1] : 0.0};
    return toy_run(2, rosen_eval, rosen_prop, &difficulty, 1 / T, chain_id, x, nSamples, nBurnin,
                   nAdjust, thinning, loglikes_out, samples_out, stepsize_out, n_accept_out, mean_out, 0, stepsize0);
}
double orc_rosenbrock_eval(double difficulty, const double* x) { return rosen_eval(x, &difficulty); }

/* mcmc_detmath.h entry points, exported so that tests can compare them with libm */
double orc_det_cos(double x) { return mcmc_cos(x); }
double orc_det_exp(double x) { return mcmc_exp(x); }
double orc_det_log(double x) { return mcmc_log(x); }
/* toy log-likelihoods alone, for batched-evaluation checks */
double orc_mylike1_eval(const double* x) { return mylike1_eval(x, NULL); }
double orc_fourgauss_eval(double difficulty, const double* x) { return fourg_eval(x, &difficulty); }
