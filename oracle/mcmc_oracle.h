/*
 * mcmc_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the Metropolis-Hastings hot path of AndreasFinke/mcmc
 * (reference @ /root/reference; citations are file:line into it).  It exists only so
 * that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check / time
 * the HIP path against an independent formulation.  Nothing under mcmc_amd/ may link,
 * import or call it.
 *
 * PINNING STATUS.  The reference cannot be compiled in this image (all five submodules
 * -- pcg32, enoki, tbb, pybind11, eigen -- are empty directories and covid.h is not part
 * of the reference's own build) and it has no tests or golden vectors.  What is pinned:
 *   - pcg32 seed/nextUInt: the published PCG32 known-answer vector pcg32(42,54)
 *     (tests/golden/pcg32_kat.json);
 *   - the Metropolis-Hastings machinery, against OUTPUTS OF THE REFERENCE ITSELF: the
 *     author saved notebooks/test2.ipynb with its outputs, and the chain samples printed
 *     there (FourGaussians(6) under TempTarget(10) after 5.2 million steps; chain 0 of a
 *     100 000-candidate reservoir ChainManager) are reproduced to every printed digit
 *     (tests/golden/test2_ipynb_outputs.json + README, tests/test_oracle_cpu.py::
 *     test_reference_notebook_outputs).  That pins pcg32(0, id) seeding, nextFloat and
 *     nextDouble, the one float step_random_subspace consumes per step, bound(), the
 *     accept rule u < exp(beta*delta), TempTarget's beta, the 20-step adjust groups with
 *     rate2corr, burn-in / thinning, sampleInitialConditions, the bootstrap and chain
 *     ids i+1.
 * What is NOT pinned by any reference artefact ("parity unpinned"):
 *   - the covid model (DiseaseSpread::step / eval, SmoothnessPrior): the only recorded
 *     outputs (notebooks/covid*.ipynb) come from an older revision of covid.h with a
 *     different interface and capacity formula.  Restated expression by expression.  The
 *     reference pins no compiler, so two arithmetic modes exist ("Which
 *     arithmetic"): contraction off = the C++ source semantics (also what the author's 2020
 *     clang produced), and the `contract` flag = the projection statement covid.h:384 as one
 *     fused multiply-add, the dominant effect of a contracting build (gcc's default, clang
 *     >= 14) -- not any one compiler's exact output, since covid.h:368,462,491 stay unfused;
 *   - enoki::erfinv (restated as M. Giles' published single-precision polynomial); only
 *     used for the one-off response-table precompute, which is an explicit INPUT to both
 *     this oracle and the HIP path, so it cannot cause a mismatch;
 *   - std::cos / std::exp / std::log: replaced by include/mcmc_detmath.h (<= 1 ulp from
 *     glibc); `use_libm` switches the oracle back to libm to measure the effect.  (The
 *     notebook pin above passes with them, i.e. no accept decision in 5.2 million steps
 *     falls inside that last ulp.)
 */
#ifndef MCMC_ORACLE_H
#define MCMC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- pcg32 (wjakob/pcg32 submodule; call sites mcmc.h:999, covid.h:102,503-607) ---- */
typedef struct { uint64_t state, inc; } orc_pcg32;
void     orc_pcg32_default(orc_pcg32* r);                       /* pcg32()            */
void     orc_pcg32_seed(orc_pcg32* r, uint64_t initstate, uint64_t initseq); /* pcg32(a,b) */
uint32_t orc_pcg32_next_uint(orc_pcg32* r);
float    orc_pcg32_next_float(orc_pcg32* r);
double   orc_pcg32_next_double(orc_pcg32* r);

/* ---- helpers (mcmc.h:45-53) ---- */
double orc_bound(double val, double lower, double upper);        /* mcmc.h:49-53 */
double orc_erfinv(double x);                                     /* enoki::erfinv restated */
double orc_to_normal(double u);                                  /* mcmc.h:45-47 */

/* ---- DiseaseParams (covid.h:59-93) ---- */
typedef struct {
    double timeIncub, timeIncubSigma, probAsymp, probSerious, probLethal,
           probLethalDailyWhenSeriousUntreated, probICUIfSerious,
           timeMildDuration, timeMildDurationSigma, timeMildToSerious,
           timeMildToSeriousSigma, timeSeriousToRec, timeSeriousToRecSigma,
           timeSeriousToDeath, timeSeriousToDeathSigma;
} orc_disease_params;
void orc_disease_params_default(orc_disease_params* p);          /* covid.h:75-90 */

/* ---- AvgDiseaseTrajectory (covid.h:96-282) ----
 * grid: [11][nGrid=100000] in the order
 *   0 incubating 1 asymptomatic 2 mild 3 infectiousMild 4 infectiousHigh 5 serious
 *   6 dead 7 recovered 8 seriousFromSerious 9 recoveredFromSerious 10 deadFromSerious
 * n_traj = 10000000 in the reference (covid.h:207). */
#define ORC_NGRID 100000
#define ORC_NDAYS 500
#define ORC_NCURVES 11
void   orc_traj_generate(const orc_disease_params* p, uint64_t n_traj, double* grid);
double orc_traj_get_day(const double* curve /*[nGrid]*/, size_t day);  /* covid.h:221-235 */
/* R[c][d] = running sum of getDay(0..d) exactly as project() forms it (covid.h:377-384) */
void   orc_traj_response(const double* grid, int n_lag, double* R /*[11][n_lag]*/);

/* ---- the covid model: DiseaseData + DiseaseSpread + SmoothnessPrior + Target ---- */
typedef struct {
    int n_data;                 /* data.deathsPerDay.size()                      */
    int max_delay;              /* maxDelayDaysTilData                           */
    int n_predict;              /* nPredictDays                                  */
    int n_disc;                 /* discontinuousDays.size()                      */
    int fix_behavior_in_advance;/* data.fixBehaviorInAdvance                     */
    int pop_size;               /* int popSize (covid.h:292)                     */
    double prob_icu_if_serious, prob_lethal_daily_untreated; /* params used by eval */
    double cap0, cap_incr_rate;
    int has_prior;              /* State = [DiseaseSpread, SmoothnessPrior] if 1 */
    double smooth_Ls, smooth_L; /* SmoothnessPrior(name, smoothnessScale, scale) */
    const double* deaths;       /* [n_data]                                      */
    const double* sigma;        /* [n_data]                                      */
    const int*    disc_days;    /* [n_disc]                                      */
    const unsigned char* disc_fixed; /* [n_disc]                                 */
    const double* R;            /* [11][n_lag] response table (orc_traj_response) */
    int n_lag;                  /* columns of R (>= min(500, size))              */
    const double* grid;         /* optional raw [11][nGrid]; used when faithful=1 */
    int contract;               /* 0: strict C++ semantics; 1: projection accumulations
                                   use fma() as gcc -ffp-contract=fast would       */
    int use_libm;               /* 0: mcmc_detmath cos/exp; 1: libm cos/exp       */
    int faithful;               /* 1: recompute cumsum from grid via getDay per project()
                                   call like covid.h:376-386, AND give every proposal a fresh
                                   copy of the 8.8 MB trajectory arrays, as copy() at covid.h:501
                                   does through `AvgDiseaseTrajectory traj` held by value
                                   (covid.h:290); slow -- the CPU "faithful" leg of bench.py */
} orc_covid_model;

static inline int orc_size(const orc_covid_model* m) { return m->max_delay + m->n_predict + m->n_data; }
static inline int orc_nb(const orc_covid_model* m)   { return m->n_data - m->fix_behavior_in_advance; }
static inline int orc_dim(const orc_covid_model* m)  { return orc_nb(m) + m->n_disc + 3; }
/* parameter vector layout (reference coords order, covid.h:294,299-301):
 *   [0,nB) behavior | [nB,nB+nDisc) discontinuousVals | betaMild | betaHigh | delay */

/* DiseaseSpread::eval (covid.h:325-496).  curves (optional) = [9][size]:
 * mildlyInfectious, highlyInfectious, incubating, asymptomatic, mild, serious,
 * recovered, dead, totalBehavior (blended output coords).  sim_days <= 0 means `size`;
 * a smaller value truncates the day loop (used to validate the HIP path's LL-only
 * horizon max_delay+n_data: days beyond it cannot influence the log-likelihood). */
double orc_covid_eval(const orc_covid_model* m, const double* x, double* curves, int sim_days);
/* SmoothnessPrior::eval (mcmc.h:197-214) on f[0..n) */
double orc_smooth_eval(const double* f, int n, double Ls, double L);
/* DiseaseSpread::step proposal part (covid.h:499-622): x -> xp, draws from rng */
void   orc_covid_propose(const orc_covid_model* m, orc_pcg32* rng, const double* x,
                         double stepsize, double* xp);
void   orc_covid_propose_part(const orc_covid_model* m, orc_pcg32* rng, const double* x,
                              double stepsize, double* xp, int with_betas);
/* DiseaseSpread::force_bounds (covid.h:629-646) */
void   orc_covid_force_bounds(const orc_covid_model* m, double* x);

/* Target kinds (mcmc.h:705-841) */
enum { ORC_TARGET_PLAIN = 0, ORC_TARGET_TEMP = 1, ORC_TARGET_COOLING = 2 };
typedef struct { int kind; double T; double slope; double Tinit; } orc_target;
double orc_target_beta(const orc_target* t, double time);

/* one chain = MetropolisChain + its deep-copied Target/State (mcmc.h:984-1138) */
typedef struct {
    orc_pcg32 rng;
    double* x;          /* [dim] current parameters                               */
    double* xp;         /* [dim] scratch proposal                                 */
    double ll_model;    /* state[0]->loglike                                      */
    double ll_prior;    /* state[1]->loglike (0 if !has_prior)                    */
    double stepsize;    /* state[0]->stepsizeCorrectionFac                        */
    int    n_accept;    /* the run()-local nAccept                                */
    uint64_t n_steps;   /* Target::step calls so far                              */
    double last_delta;  /* rec.deltaloglike of the last step (after *beta)        */
    int    last_accept;
    double* grid_own;   /* faithful=1: this chain's own copy of the 11 x nGrid trajectory arrays
                           (DiseaseSpread holds AvgDiseaseTrajectory BY VALUE, covid.h:290)  */
} orc_chain;

void orc_chain_init(const orc_covid_model* m, orc_chain* c, int chain_id,
                    const double* x0, double stepsize);      /* mcmc.h:999 + State::eval */
void orc_chain_free(orc_chain* c);
/* Target::step (mcmc.h:750-783) incl. State::step_random/given_subspace */
int  orc_chain_step(const orc_covid_model* m, const orc_target* t, orc_chain* c,
                    double time, int random_subspace);
/* MetropolisChain::run (mcmc.h:1005-1138).
 * loglikes_out [n_saved]; samples_out [n_saved][dim] (may be NULL);
 * mean_out [dim] (may be NULL => computeMean=false).  Returns n_saved. */
int  orc_chain_run(const orc_covid_model* m, const orc_target* t, orc_chain* c,
                   int nSamples, int nBurnin, int nAdjust, int thinning,
                   double* loglikes_out, double* samples_out, double* mean_out);
int  orc_n_saved(int nSamples, int nBurnin, int nAdjust, int thinning);

/* ---- multi-region model (EXTENSION, BASELINE.json configs[4]; definition in mcmc_oracle.c) ----
 * regs [R]: complete models that differ in deaths, sigma, pop_size, cap0, cap_incr_rate only.
 * x [R][dimRegion] blocks of the single-region layout; ll [R][2] = (model, prior) per region. */
typedef struct {
    orc_pcg32 rng;
    double* x;          /* [R * dimRegion] */
    double* xp;         /* [dimRegion] the head's proposal */
    double* ll;         /* [R][2] */
    double* llp;        /* [R] scratch: proposed model log-likelihoods */
    double* stepsize;   /* [R] stepsizeCorrectionFac of every region's DiseaseSpread */
    int n_accept;
    uint64_t n_steps, head_counter;
    int head;
} orc_multi_chain;
void orc_multi_init(const orc_covid_model* regs, int R, orc_multi_chain* c, int chain_id, const double* x0, double stepsize);
void orc_multi_free(orc_multi_chain* c);
int  orc_multi_step(const orc_covid_model* regs, int R, const orc_target* t, orc_multi_chain* c, double time, int random_subspace);
int  orc_multi_run(const orc_covid_model* regs, int R, const orc_target* t, orc_multi_chain* c, int nSamples, int nBurnin,
                   int nAdjust, int thinning, double* loglikes_out, double* samples_out, double* mean_out);

/* many chains on a thread pool (ChainManager::run_all_adjust, mcmc.h:1513-1553);
 * chain i uses pcg32(0, i+1) (mcmc.h:1527).  x is [n_chains][dim] in/out; returns wall s
 * of the run loop only. final_ll [n_chains][2], stepsizes [n_chains], n_accept [n_chains]. */
double orc_run_chains(const orc_covid_model* m, const orc_target* t, int n_chains,
                      int first_chain_id, double* x, double* stepsizes,
                      int nSamples, int nBurnin, int nAdjust, int thinning, int n_threads,
                      double* final_ll, int* n_accept, uint64_t* rng_state,
                      double* loglikes_out /*[n_chains][n_saved] or NULL*/);

/* ---- toy targets of test.h (config 0: CLI run test.cpp:27-40) ---- */
/* MyLike1 (test.h:14-67), 3-D Gaussian, chain id, run(nSamples,nBurnin,nAdjust,thinning);
 * returns n_saved; samples_out [n_saved][3], loglikes_out [n_saved], mean_out [4] (position and
 * the coordinate "max", mcmc.h:1002,1064-1065,1133-1134; may be NULL) */
int orc_mylike1_run(int chain_id, int nSamples, int nBurnin, int nAdjust, int thinning,
                    double* loglikes_out, double* samples_out, double* stepsize_out,
                    int* n_accept_out, double* mean_out);
/* FourGaussians (test.h:69-136) under TempTarget(T), from x0 with stepsizeCorrectionFac stepsize0 */
int orc_fourgauss_run(double difficulty, double T, int chain_id, const double* x0,
                      int nSamples, int nBurnin, int nAdjust, int thinning,
                      double* loglikes_out, double* samples_out, double* stepsize_out,
                      int* n_accept_out, double* mean_out /* [2] or NULL */, double stepsize0);

/* Rosenbrock (EXTENSION, BASELINE configs[0]): see mcmc_oracle.c; same calling convention */
int orc_rosenbrock_run(double difficulty, double T, int chain_id, const double* x0,
                       int nSamples, int nBurnin, int nAdjust, int thinning,
                       double* loglikes_out, double* samples_out, double* stepsize_out,
                       int* n_accept_out, double* mean_out /* [2] or NULL */, double stepsize0);
double orc_rosenbrock_eval(double difficulty, const double* x);

/* include/mcmc_detmath.h, exported for tests */
double orc_det_cos(double x);
double orc_det_exp(double x);
double orc_det_log(double x);
double orc_mylike1_eval(const double* x);
double orc_fourgauss_eval(double difficulty, const double* x);

#ifdef __cplusplus
}
#endif
#endif
