/*
 * mcmc_hip.h -- C ABI of the MI355X Metropolis-Hastings hot path (libmcmc_hip.so).
 *
 * This is the drop-in boundary: the entry points a binding of AndreasFinke/mcmc would
 * call in place of running `MetropolisChain::run` (mcmc.h:1005-1138) on every chain
 * inside `ChainManager::run_all*`'s tbb::parallel_for (mcmc.h:1494-1508, 1534-1548).
 * Plain pointers and sizes only; the library owns all device memory behind the opaque
 * context; the caller owns every host buffer.  Every function returns 0 on success or
 * a negative MCMC_HIP_E* code; mcmc_hip_last_error() gives the message (thread-local).
 * No exception crosses the boundary.  A context is single-owner and not thread-safe
 * (the reference's chains share nothing mutable either, mcmc.h:1479,1520).
 *
 * All reference citations are file:line under /root/reference.
 */
#ifndef MCMC_HIP_H
#define MCMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCMC_HIP_OK            0
#define MCMC_HIP_EINVAL       -1   /* bad argument / inconsistent sizes              */
#define MCMC_HIP_ESTATE       -2   /* call order violated (e.g. run before set_*)    */
#define MCMC_HIP_EHIP         -3   /* a HIP runtime call failed                      */
#define MCMC_HIP_ENOMEM       -4
#define MCMC_HIP_ENODEVICE    -5   /* no usable gfx950 device                        */

/* number of response curves the model reads, in this order (covid.h:403-410,426-428):
 * infectiousMild, infectiousHigh, incubating, mild, serious, recovered, dead,
 * seriousFromSerious, recoveredFromSerious, deadFromSerious.  (`asymptomatic` is
 * projected with traj.mild, covid.h:406, so it needs no curve of its own.) */
#define MCMC_HIP_NCURVES 10
#define MCMC_HIP_TRAJ_NDAYS 500           /* AvgDiseaseTrajectory::nDays, covid.h:206 */

enum mcmc_target_kind {                    /* mcmc.h:705-841 */
    MCMC_TARGET_PLAIN = 0,                 /* Target:        beta = 1                   */
    MCMC_TARGET_TEMP = 1,                  /* TempTarget:    beta = 1/T                 */
    MCMC_TARGET_COOLING = 2                /* CoolingTarget: beta = exp(t*slope)/Tinit  */
};

/* DiseaseParams (covid.h:59-93): the 15 public fields, same names. */
typedef struct mcmc_disease_params {
    double timeIncub, timeIncubSigma, probAsymp, probSerious, probLethal,
           probLethalDailyWhenSeriousUntreated, probICUIfSerious,
           timeMildDuration, timeMildDurationSigma, timeMildToSerious,
           timeMildToSeriousSigma, timeSeriousToRec, timeSeriousToRecSigma,
           timeSeriousToDeath, timeSeriousToDeathSigma;
} mcmc_disease_params;

/* Everything DiseaseSpread's ctor (covid.h:294-321), DiseaseData (covid.h:16-57),
 * SmoothnessPrior's ctor (mcmc.h:188) and the Target (mcmc.h:797-841) fix per run. */
typedef struct mcmc_covid_config {
    int32_t n_data;                  /* data.deathsPerDay.size()                       */
    int32_t max_delay;               /* maxDelayDaysTilData                            */
    int32_t n_predict;               /* nPredictDays                                   */
    int32_t n_disc;                  /* data.discontinuousDays.size(), 1..8            */
    int32_t fix_behavior_in_advance; /* data.fixBehaviorInAdvance                      */
    int32_t pop_size;                /* int popSize (covid.h:292)                      */
    double  prob_icu_if_serious;     /* params.probICUIfSerious                        */
    double  prob_lethal_daily_untreated; /* params.probLethalDailyWhenSeriousUntreated */
    double  cap0, cap_incr_rate;     /* capacity[i] = cap0 + max(rate*(i-maxDelay),0)  */
    int32_t has_prior;               /* 1: State = [DiseaseSpread, SmoothnessPrior("behavior")] */
    double  smooth_Ls, smooth_L;     /* SmoothnessPrior(name, smoothnessScale, scale)  */
    int32_t target_kind;             /* enum mcmc_target_kind                          */
    double  target_T, target_slope, target_Tinit;
    int32_t contract;                /* 0: every projection term is mul then add (the C++
                                        source semantics); 1: projection terms are fused
                                        multiply-adds, as gcc -ffp-contract=fast compiles
                                        `dest[d+k] += n*cumsum` (covid.h:384)          */
    const int32_t* disc_days;        /* [n_disc] data.discontinuousDays                */
    const uint8_t* disc_fixed;       /* [n_disc] data.discontinuousValsFixed           */
} mcmc_covid_config;

typedef struct mcmc_hip_ctx mcmc_hip_ctx;

const char* mcmc_hip_last_error(void);
/* gfx950 devices visible to this process (HIP_VISIBLE_DEVICES applies). */
int mcmc_hip_device_count(void);

/* --- host-side precompute (AvgDiseaseTrajectory, covid.h:96-282; runs on the CPU as in
 * the reference, once per model) ------------------------------------------------------
 * Fills R[MCMC_HIP_NCURVES][n_lag], R[c][d] = the running sum `cumsum` that project()
 * (covid.h:377-384) forms from getDay(0..d) of curve c.  n_traj = 10000000 reproduces
 * AvgDiseaseTrajectory::nTrajectories.  n_lag <= MCMC_HIP_TRAJ_NDAYS. */
int mcmc_host_make_response_tables(const mcmc_disease_params* p, uint64_t n_traj,
                                   int32_t n_lag, double* R);
void mcmc_host_default_disease_params(mcmc_disease_params* p);          /* covid.h:75-90 */
/* Synthetic initial conditions: chain c, coordinate k (chain-major draw order from one
 * pcg32(initstate, initseq)): out[c*dim+k] = x0[k] + (nextFloat()-.5f)*x0[k]*a, the
 * per-coordinate rule of State::perturb (mcmc.h:306-314). */
int mcmc_host_perturb_ics(const double* x0, int32_t dim, int64_t n_chains, double a,
                          uint64_t initstate, uint64_t initseq, double* out);
/* The chains [first_chain, first_chain + n_chains) of that same sequence without drawing the ones
 * before them: the generator is moved first_chain * dim draws ahead in O(log) steps, so a rank that
 * owns a chain-id range far into the sequence sets up in time proportional to ITS chains.  (No
 * counterpart in the reference, whose managers draw every chain's start serially, mcmc.h:1319-1346.) */
int mcmc_host_perturb_ics_from(const double* x0, int32_t dim, int64_t first_chain, int64_t n_chains,
                               double a, uint64_t initstate, uint64_t initseq, double* out);

/* --- context ------------------------------------------------------------------------
 * Replaces constructing DiseaseSpread/SmoothnessPrior/State/Target (covid.h:294,
 * mcmc.h:188,268,709).  `device` is the HIP ordinal this context (and its process) uses. */
int mcmc_hip_create(const mcmc_covid_config* cfg, int device, mcmc_hip_ctx** out);
int mcmc_hip_destroy(mcmc_hip_ctx* ctx);

/* AvgDiseaseTrajectory as the model reads it: R[MCMC_HIP_NCURVES][n_lag] with
 * n_lag >= min(500, size); replaces `AvgDiseaseTrajectory traj` (covid.h:290). */
int mcmc_hip_set_tables(mcmc_hip_ctx* ctx, const double* R, int32_t n_lag);
/* data.deathsPerDay / data.deathsSigma (covid.h:47), each [n_data]. */
int mcmc_hip_set_data(mcmc_hip_ctx* ctx, const double* deaths, const double* sigma);

/* --- multi-region model (EXTENSION: BASELINE.json configs[4]; the reference models one region,
 * covid.h:284-321).  After create + set_tables, INSTEAD of set_data: the chain's State becomes
 * n_regions region blocks [DiseaseSpread_r (, SmoothnessPrior_r)] -- each the complete model of
 * covid.h:325-496 with its own observations deaths/sigma [n_regions][n_data], population and
 * hospital capacity (cap0, cap_incr_rate), everything else as in cfg -- sharing betaMild and
 * betaHigh, which region 0 owns.  mcmc_hip_dim becomes n_regions * (nB + nDisc + 3): a parameter
 * vector is n_regions blocks of the single-region layout, the beta slots of every block mirroring
 * region 0's (set_chains / eval copy region 0's into them).  One MH step moves ONE region, the
 * head, by systematic scan -- head = (random-subspace steps since set_chains) mod n_regions, the
 * same for every chain, so wavefronts do not diverge; a given-subspace step reuses the head (as
 * rec.whichHead does, mcmc.h:510).  The step draws what DiseaseSpread::step draws (covid.h:499-622)
 * on the head's block, the betas' branch only for region 0; changed betas make every other region a
 * dependent that is re-evaluated (mcmc.h:539-602); delta sums (-old + new) in State order
 * (mcmc.h:604-606); the adjust phase corrects the head's own step size (mcmc.h:1107-1114).
 * n_regions = 1 is the reference's State, bit for bit.  get_loglike then returns per chain (sum
 * of the regions' model log-likelihoods, sum of their priors); the per-region values and step
 * sizes: the two getters below.  set_state's loglike is [n][n_regions][2] here.  Not combined with
 * the tempering ladder, the fast mode, eval_static or derive (derive a region through a
 * single-region context). */
int mcmc_hip_set_regions(mcmc_hip_ctx* ctx, int32_t n_regions, const int32_t* pop_size, const double* cap0,
                         const double* cap_incr_rate, const double* deaths, const double* sigma);
int32_t mcmc_hip_n_regions(const mcmc_hip_ctx* ctx);
int mcmc_hip_get_region_loglike(mcmc_hip_ctx* ctx, double* out /*[n_chains][n_regions][2]*/);
int mcmc_hip_get_region_stepsizes(mcmc_hip_ctx* ctx, double* out /*[n_chains][n_regions]*/);
/* What a multi-region chain carries beyond mcmc_hip_set_state's four arrays (checkpoint / resume; same
 * role as there, mcmc.h:1141-1199,1376-1418): the position of the systematic scan -- the number of
 * random-subspace steps since set_chains; the next head is scan_position mod n_regions, a
 * given-subspace step reuses (scan_position - 1) mod n_regions -- and every region's own
 * stepsizeCorrectionFac, [n_chains][n_regions] as get_region_stepsizes returns it (set_chains takes one
 * step size per chain and gives it to every region; NULL keeps that).  Call after set_chains (which
 * resets the scan to 0).  On a single-region context the position is kept and has no effect. */
int mcmc_hip_get_scan_position(mcmc_hip_ctx* ctx, int64_t* out);
int mcmc_hip_set_region_state(mcmc_hip_ctx* ctx, int64_t scan_position, const double* region_stepsizes);

/* Arithmetic of the model time-stepper.  No counterpart in the reference, whose only
 * arithmetic is `using Float = double` (mcmc.h:43) = MCMC_HIP_PRECISION_F64, the default and
 * the only mode held to bit-exact parity.  MCMC_HIP_PRECISION_F32_MODEL ("fast mode", the FP32
 * figure north_star asks for) runs the renewal convolution of DiseaseSpread::eval
 * (covid.h:376-410,426-428) in single precision and everything else -- proposal,
 * capacity test, feedback, likelihood sum, prior, accept -- in double; it needs
 * cfg.contract = 1 and is held to a statistical bar (tests/test_gpu_fast_mode.py).
 * Call after set_tables and before set_chains; eval/derive: eval follows the mode, derive is
 * always double. */
#define MCMC_HIP_PRECISION_F64        0
#define MCMC_HIP_PRECISION_F32_MODEL  1   /* the fast mode: convolution on the FP32 matrix cores (v_mfma_f32_32x32x2_f32) */
#define MCMC_HIP_PRECISION_F32_PACKED 2   /* its first form, packed FP32 vector multiply-adds; kept for comparison */
int mcmc_hip_set_precision(mcmc_hip_ctx* ctx, int32_t mode);

/* How the chains are laid onto the GPU by the step kernels.  NO COUNTERPART IN THE REFERENCE (one TBB task per
 * chain, mcmc.h:1494-1508) and no effect on any result: both mappings execute the same operations in the same
 * order and give the same bits (tests/test_gpu_wave_mapping.py).
 *   LANE  one chain per lane, 64 chains per wavefront in lockstep: north_star's mapping, the one that fills the
 *         chip from 64 k chains on and is the faster one from about 17 k (200 days) to 24 k (365 days).
 *   WAVE  one chain per WAVEFRONT (mcmc_wave_kernel.hip.h): every lane owns one day of a 32-day block of one of
 *         the two integer-shift passes; a chain's state lives in LDS for the length of a launch.  For small
 *         ensembles -- BASELINE.json configs[1]'s 1 024 chains are 16 wavefronts under LANE and 1 024 under
 *         WAVE.  Both arithmetic modes (cfg.contract).  Needs the double-precision mode, the single-region State
 *         and a series that fits a wave's LDS slice (about 1 200 data days); set_mapping(WAVE) fails otherwise.
 *   AUTO  (default) WAVE where possible for ensembles of at most 16 384 chains (8 192 / 4 096 where LDS holds only
 *         two / one chain per workgroup: series beyond about 380 / 710 data days), else LANE.
 * May be called at any time; it takes effect from the next launch.  get_mapping: what the next launch uses. */
#define MCMC_HIP_MAPPING_AUTO 0
#define MCMC_HIP_MAPPING_LANE 1
#define MCMC_HIP_MAPPING_WAVE 2
int mcmc_hip_set_mapping(mcmc_hip_ctx* ctx, int32_t mapping);
int32_t mcmc_hip_get_mapping(const mcmc_hip_ctx* ctx);

/* Replaces ChainManager's per-chain setup loop (mcmc.h:1517-1532): n chains, chain i has
 * RNG pcg32(0, chain_ids[i]) (mcmc.h:999,1527) and parameters params[i*dim .. i*dim+dim)
 * in the reference's coords order: behavior[nB], discontinuousVals[nDisc], betaMild,
 * betaHigh, delay (covid.h:294,299-301); dim = nB+nDisc+3.  stepsizes[i] =
 * stepsizeCorrectionFac of the DiseaseSpread sub-state (NULL: all 1).  chain_ids NULL:
 * first_chain_id+i.  Evaluates the initial log-likelihoods on the device (State::eval,
 * mcmc.h:385-393).  Chains are independent: a caller sharding over GPUs gives each
 * context its own id range (mcmc.h:1487). */
int mcmc_hip_set_chains(mcmc_hip_ctx* ctx, int64_t n_chains, const double* params,
                        const double* stepsizes, const uint64_t* chain_ids,
                        uint64_t first_chain_id);

/* Continuing chains in another context or process (checkpoint / resume; the reference's own
 * substitutes are the recorded `ics` + reevaluate(), mcmc.h:1141-1199, and seeding a manager from a
 * finished chain, mcmc.h:1376-1418).  After set_chains (parameters, step sizes), this overrides
 * what set_chains derives from scratch, each array chain-major as its getter returns it and each
 * optional (NULL keeps set_chains' value): rng_state [n][2] = pcg32 (state, inc) of
 * get_rng_state -- a MetropolisChain seeds `rnd` ONCE in its constructor and keeps drawing from it
 * across run() calls (mcmc.h:994-999) --; loglike [n][2] of get_loglike; mean [n][dim] = the
 * running `mean` a second run() keeps adding to (mcmc.h:1002,1064-1065,1133-1134; get_mean);
 * accept_counts [n] = run()'s nAccept.  params + stepsizes + these four are a chain's whole state
 * in the reference's single-region State; a multi-region chain (set_regions) has loglike
 * [n][n_regions][2] here and ALSO needs mcmc_hip_set_region_state (scan position, per-region step
 * sizes) to continue bit for bit. */
int mcmc_hip_set_state(mcmc_hip_ctx* ctx, const uint64_t* rng_state, const double* loglike,
                       const double* mean, const int32_t* accept_counts);

/* Replaces `chains[i].run(nSamples, nBurnin, nAdjust, thinning)` for ALL chains
 * (mcmc.h:1005-1138 under mcmc.h:1494-1508 / 1534-1548).  record_samples mirrors
 * MetropolisChain::recordSamples, compute_mean mirrors computeMean.  May be called again:
 * like a second chain.run() it continues from the current state and RNG, and replaces
 * the recorded loglikes/samples. */
int mcmc_hip_run(mcmc_hip_ctx* ctx, int32_t n_samples, int32_t n_burnin, int32_t n_adjust,
                 int32_t thinning, int32_t record_samples, int32_t compute_mean);
/* Finer-grained driver used by tests and by callers that interleave their own work:
 * advance every chain by n_steps Target::step calls (mcmc.h:750-783) at inverse
 * temperature beta; random_subspace as the 4th argument of Target::step. */
int mcmc_hip_step(mcmc_hip_ctx* ctx, int32_t n_steps, double beta, int32_t random_subspace);

/* number of entries MetropolisChain::loglikes would hold (mcmc.h:1037) */
int32_t mcmc_hip_n_saved(int32_t n_samples, int32_t n_burnin, int32_t n_adjust, int32_t thinning);

/* --- results (all copy out; layouts are chain-major = what per-chain getters return) -- */
int mcmc_hip_dim(const mcmc_hip_ctx* ctx);           /* nB + nDisc + 3                     */
int mcmc_hip_size(const mcmc_hip_ctx* ctx);          /* nDaysTotal (covid.h:297)           */
int64_t mcmc_hip_n_chains(const mcmc_hip_ctx* ctx);
int mcmc_hip_get_params(mcmc_hip_ctx* ctx, double* out /*[n_chains][dim]*/);
int mcmc_hip_get_loglike(mcmc_hip_ctx* ctx, double* out /*[n_chains][2]: model, prior*/);
int mcmc_hip_get_stepsizes(mcmc_hip_ctx* ctx, double* out /*[n_chains]*/);
int mcmc_hip_get_accept_counts(mcmc_hip_ctx* ctx, int32_t* out /*[n_chains] run()'s nAccept*/);
int mcmc_hip_get_rng_state(mcmc_hip_ctx* ctx, uint64_t* out /*[n_chains][2]: state, inc*/);
/* MetropolisChain::get_loglikes (mcmc.h:1286-1301) for every chain */
int mcmc_hip_get_loglikes(mcmc_hip_ctx* ctx, double* out /*[n_chains][n_saved]*/);
/* MetropolisChain::get_samples (mcmc.h:1249-1268) restricted to sampled coordinates
 * [coord_first, coord_first+coord_count) of one chain: out[n_saved][coord_count]. */
int mcmc_hip_get_samples(mcmc_hip_ctx* ctx, int64_t chain, int32_t coord_first,
                         int32_t coord_count, double* out);
/* every chain, one saved sample: out[n_chains][dim] */
int mcmc_hip_get_sample_all(mcmc_hip_ctx* ctx, int32_t saved_index, double* out);
/* MetropolisChain::get_mean (mcmc.h:1236-1248): out[n_chains][dim] */
int mcmc_hip_get_mean(mcmc_hip_ctx* ctx, double* out);

/* Parity hook / batched likelihood (DiseaseSpread::eval + SmoothnessPrior::eval,
 * covid.h:325-496, mcmc.h:197-214) of n parameter vectors, without touching chain state:
 * ll_out[n][2].  Also what ChainManager's reservoir constructors need (mcmc.h:1356-1371). */
int mcmc_hip_eval(mcmc_hip_ctx* ctx, int64_t n, const double* params /*[n][dim]*/, double* ll_out);
/* The derived output coords of DiseaseSpread::eval (covid.h:467-485) for n parameter
 * vectors: curves_out[n][9][size] in the order mildlyInfectious, highlyInfectious,
 * incubating, asymptomatic, mild, serious, recovered, dead, totalBehavior -- computed
 * only on request (the reference's compute_derived_late idea, mcmc.h:100-104). */
int mcmc_hip_derive(mcmc_hip_ctx* ctx, int64_t n, const double* params, double* curves_out);

/* mcmc_hip_eval for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE (its chains
 * allocate freely, e.g. copy() at covid.h:501): this variant exists for callers that must not
 * reach hipMalloc -- tight serving loops, harnesses that forbid allocation after set-up.  (It is not
 * itself capturable into a HIP graph: it runs on the context's own stream and synchronises it.)
 * Same kernels and results as mcmc_hip_eval (bit for bit), but params_dev [n][dim] and ll_out_dev
 * [n][2] are DEVICE pointers on the context's GPU, and the kernels' scratch (tiled parameters,
 * per-lane histories, weighted deaths, results) is a file-scope device array of 16 MiB that every
 * GPU the library runs on holds until the process ends.  The call allocates and frees nothing;
 * it returns after its kernels have finished (it synchronises the context's stream).  n is
 * limited by that array: mcmc_hip_eval_static_max(ctx) vectors for this model (960 at 365 data
 * days); a larger n is rejected with MCMC_HIP_EINVAL.  Every call on one GPU shares that GPU's
 * array, from any context: within one process the library serialises them (a per-device lock held
 * until the call's kernels have finished); calls from different PROCESSES do not share the array
 * (each process image has its own copy). */
int mcmc_hip_eval_static(mcmc_hip_ctx* ctx, int64_t n, const double* params_dev, double* ll_out_dev);
int64_t mcmc_hip_eval_static_max(const mcmc_hip_ctx* ctx);

/* Cross-chain diagnostics of the current state, reduced on the device with wavefront
 * shuffle reductions: out[0]=sum of run()'s nAccept, out[1]=number of Target::step calls
 * per chain since set_chains/run start, out[2]=sum loglike, out[3]=sum loglike^2,
 * then per coordinate k<dim: out[4+2k]=sum x_k, out[5+2k]=sum x_k^2.  Length 4+2*dim.
 * A multi-GPU caller all-reduces this vector (it is additive over shards). */
int mcmc_hip_diagnostics(mcmc_hip_ctx* ctx, double* out);

/* wall time (s) the last mcmc_hip_run spent in its sampling phase and adjust phase,
 * measured with hipEvents on the library's stream, plus the kernel-only time of the
 * step kernel and the number of its launches (for roofline accounting). */
int mcmc_hip_last_timing(mcmc_hip_ctx* ctx, double* adjust_s, double* sampling_s,
                         double* step_kernel_s, int64_t* step_kernel_launches);
/* algorithmic (day,lag) pairs of the evaluations the step kernel made since the counter was
 * reset, summed over chains: what the reference performs over its full horizon, one multiply-add
 * per compartment and pair (covid.h:384), lags cut at traj.nDays (covid.h:378). */
int mcmc_hip_pair_count(mcmc_hip_ctx* ctx, double* pairs_out, int32_t reset);
/* (day,lag) pairs the step kernel actually ran for the chains since the reset: only the days
 * the log-likelihood can see (maxDelay+nData of the reference's maxDelay+nPredict+nData,
 * covid.h:297,488-492), seven accumulator families each (asymptomatic == mild, covid.h:406).
 * Either call with reset != 0 clears BOTH counters. */
int mcmc_hip_executed_pair_count(mcmc_hip_ctx* ctx, double* pairs_out, int32_t reset);

/* --- tempering ladder with swap exchange (EXTENSION: BASELINE.json configs[3]; the reference has
 * no replica exchange -- its TempTarget, mcmc.h:797-808, only fixes one temperature per run).
 * After set_chains, the chains [L*n_temps, (L+1)*n_temps) form ladder L and chain i starts on
 * rung i % n_temps, i.e. under TempTarget(temps[rung]): every step's beta is multiplied by
 * 1/temps[rung].  mcmc_hip_swap does one exchange round between the rungs (r, r+1),
 * r = parity, parity+2, ...: accept iff double(u) < exp((beta_r - beta_{r+1}) (l_{r+1} - l_r)),
 * l = State::loglikelihood (mcmc.h:612-618), u = nextFloat() of pcg32(swap_seed, first_ladder+L),
 * one draw per pair.  Accepted pairs exchange their temperatures, not their states.  Ladders are
 * independent, so several GPUs take whole ladders each (first_ladder = the shard's first ladder
 * id) and no exchange crosses a device.  n_temps = 0 removes the ladder. */
int mcmc_hip_set_ladder(mcmc_hip_ctx* ctx, int32_t n_temps, const double* temps, uint64_t swap_seed,
                        uint64_t first_ladder);
int mcmc_hip_swap(mcmc_hip_ctx* ctx, int32_t parity);
/* n_rounds x (steps_per_round Target::step calls with a random subspace, then one swap round of
 * alternating parity 0,1,0,...) */
int mcmc_hip_run_tempered(mcmc_hip_ctx* ctx, int32_t n_rounds, int32_t steps_per_round);
int mcmc_hip_get_rungs(mcmc_hip_ctx* ctx, int32_t* out /*[n_chains]*/);
/* per rung pair r = 0..n_temps-2: accepted and proposed exchanges since set_ladder */
int mcmc_hip_get_swap_counts(mcmc_hip_ctx* ctx, int64_t* accepted, int64_t* proposed);

/* --- toy targets (test.h:14-136): MyLike1 (3-D Gaussian, the CLI run of test.cpp:27-40) and
 * FourGaussians(difficulty), each as the single likelihood of a State, under
 * MetropolisChain::run (mcmc.h:1005-1138) with Target / TempTarget / CoolingTarget.  One chain
 * per lane, the whole run in one launch; chain c draws pcg32(0, chain_ids[c]) (NULL:
 * first_chain_id + c).  x [n][dim] (position / pos) and stepsizes [n] are read and overwritten
 * with the end state.  A saved point and the running mean (mcmc.h:1002,1064-1065,1133-1134)
 * hold ALL coords of the likelihood, n_out = dim + 1 for MyLike1 (its coordinate "max",
 * test.h:37) and dim for FourGaussians: loglikes_out [n][n_saved], samples_out
 * [n][n_saved][n_out], mean_out [n][n_out] and n_accept_out [n] may each be NULL.
 * A later run() of the same chains (the reference seeds `rnd` once per chain and keeps its `mean`,
 * mcmc.h:994-1003): rng_state [n][2] non-NULL = start from these pcg32 (state, inc) instead of
 * seeding, and receive the end state; mean_continue = 1: mean_out holds the previous run's mean
 * on entry and the run adds to it (mcmc.h:1064-1065,1133-1134). */
/* MCMC_TOY_ROSENBROCK (EXTENSION: BASELINE.json configs[0] names a Rosenbrock target, the reference
 * has none) follows FourGaussians' pattern (test.h:69-136): coordinate "pos" (x, y), step
 * pos[k] += stepsizeCorrectionFac * (nextFloat() - 0.5) without bounds, and
 * loglike = -((1 - x)^2 + difficulty * (y - x^2)^2) / 20; difficulty = 100 is the usual density. */
enum mcmc_toy_kind { MCMC_TOY_MYLIKE1 = 0, MCMC_TOY_FOURGAUSS = 1, MCMC_TOY_ROSENBROCK = 2 };
int mcmc_hip_toy_dim(int32_t kind);                                   /* 3, 2, 2 */
int mcmc_hip_toy_n_out(int32_t kind);                                 /* 4, 2, 2 */
int mcmc_hip_toy_eval(int device, int32_t kind, double difficulty, int64_t n, const double* x, double* ll_out);
int mcmc_hip_toy_run(int device, int32_t kind, double difficulty, int32_t target_kind, double target_T,
                     double target_slope, double target_Tinit, int64_t n_chains, const uint64_t* chain_ids,
                     uint64_t first_chain_id, double* x, double* stepsizes, int32_t n_samples,
                     int32_t n_burnin, int32_t n_adjust, int32_t thinning, double* loglikes_out,
                     double* samples_out, double* mean_out, int32_t* n_accept_out,
                     uint64_t* rng_state, int32_t mean_continue);

/* Profiling aid: shader cycles summed over all waves per phase of the MH step since the last
 * reset, out[16] in the order propose, eval set-up, lag chunks, lag tail, day loop,
 * log-likelihood, prior, accept, record, lag chunks with e-terms.  All zeros unless the library was built with
 * -DMCMC_STAMPS (the stamps cost a few per cent). */
int mcmc_hip_debug_phase_cycles(mcmc_hip_ctx* ctx, double* out, int32_t reset);
/* Diagnostic (no counterpart in the reference; -DMCMC_STAMPS builds): per tile of the last launch, the
 * wave's shader cycles, (XCC_ID << 32) | HW_ID of where it ran, its lag chunks with e-terms, the days on which
 * any of its chains had a capacity correction and the length of the day range spanning them (summed over
 * the launch's evaluations); n <= 8192. */
int mcmc_hip_debug_wave_log(mcmc_hip_ctx* ctx, double* cycles, uint64_t* hwid, double* e_chunks, double* e_days,
                            double* e_hull, int64_t n);
/* Diagnostic (no counterpart in the reference; -DMCMC_F32_CHECK builds of the fast mode): out[4] = kind of the
 * first row range that fell outside its buffer (0: none), its low and high row and the buffer's rows. */
int mcmc_hip_debug_f32_check(mcmc_hip_ctx* ctx, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* MCMC_HIP_H */
