/*
 * mcmc_host.hpp -- C++17 host side of the MI355X Metropolis-Hastings hot path.
 *
 * Mirrors, for the Metropolis-Hastings path, the interface of AndreasFinke/mcmc that a user program or
 * the pybind11 module touches: SubspaceState / SmoothnessPrior / DiseaseData /
 * DiseaseParams / DiseaseSpread / State / Target / TempTarget / CoolingTarget /
 * MetropolisChain / ChainManager (mcmc.h:62-228, 241-700, 705-841, 984-1312, 1314-1655;
 * covid.h:16-93, 284-321, 629-646).  Same names, argument order and error behaviour
 * (messages on std::cout and an early return, as the reference does).  What differs is
 * where the work runs: nothing here evaluates the model on the CPU.  State::eval,
 * MetropolisChain::run and ChainManager::run_all* hand the parameters to libmcmc_hip.so
 * through the C ABI of include/mcmc_hip.h, and ALL chains of a manager advance in one
 * launch sequence on the GPU (the reference runs one TBB task per chain,
 * mcmc.h:1494-1508, 1534-1548).  There is no CPU fallback: without the library or a GPU
 * the calls report the C ABI's error.
 *
 * Supported State compositions: the covid notebooks' [ DiseaseSpread ] or
 * [ DiseaseSpread, SmoothnessPrior("behavior", Ls, L) ], and the single toy likelihoods of
 * test.h, [ MyLike1 ] (= MyState, the CLI run of test.cpp:27-40) and [ FourGaussians(d) ].
 * Everything is header-only and uses nothing but the C ABI, so a C++ program can use it
 * without Python; mcmc_amd/pymod/mcmc_module.cpp wraps it for Python.
 */
#pragma once

#include <algorithm>
#include <numeric>
#include <cmath>
#include <cstdint>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <optional>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <unordered_set>
#include <vector>

#include "mcmc_hip.h"

namespace mcmc_amd {

using Float = double;                      /* mcmc.h:43 */

/* mcmc.h:49-53 (host copy: force_bounds runs where the reference runs it) */
inline void bound(Float& val, Float lower, Float upper) {
    val = 0.5 * (val - lower) / (upper - lower);
    val = 2 * std::abs(val - std::round(val));
    val = lower + val * (upper - lower);
}

/* wjakob/pcg32 as the reference seeds and draws it (mcmc.h:1319,1625) */
struct pcg32 {
    uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;
    pcg32() = default;
    pcg32(uint64_t initstate, uint64_t initseq) {
        state = 0; inc = (initseq << 1u) | 1u; nextUInt(); state += initstate; nextUInt();
    }
    uint32_t nextUInt() {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t x = (uint32_t)(((old >> 18u) ^ old) >> 27u), r = (uint32_t)(old >> 59u);
        return (x >> r) | (x << ((~r + 1u) & 31));
    }
    float nextFloat() {
        union { uint32_t u; float f; } x;
        x.u = (nextUInt() >> 9) | 0x3f800000u;
        return x.f - 1.0f;
    }
    double nextDouble() {
        union { uint64_t u; double d; } x;
        x.u = ((uint64_t)nextUInt() << 20) | 0x3ff0000000000000ULL;
        return x.d - 1.0;
    }
};

struct HipError : std::runtime_error {
    int code;
    HipError(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};
inline void hip_check(int rc, const char* where) {
    if (rc != 0) throw HipError(rc, std::string(where) + ": " + mcmc_hip_last_error());
}

/* mcmc.h:55-58 */
#define HAS_STEP std::shared_ptr<SubspaceState> copy() const override {\
    return std::make_shared<std::remove_const_t<std::remove_reference_t<decltype(*this)>>>(*this);}
using SharedParams = std::map<std::string, std::vector<Float>>;

/* ------------------------------------------------------------------ SubspaceState ---- */
/* mcmc.h:62-183: the plugin interface -- named ragged coords, loglike, stepsizeCorrectionFac,
 * fixed set, and the virtuals a model overrides (eval, step, the initial-condition trio,
 * compute_derived_late, force_bounds), with the reference's signatures and default bodies, so
 * that a likelihood written against the reference compiles against this header.  The device path
 * never calls a user's eval/step: it runs the compositions State::supported() lists, and any
 * other composition is refused there, loudly, instead of being stepped on the CPU. */
class SubspaceState {
public:
    using Proposal = std::pair<std::shared_ptr<SubspaceState>, Float>;         /* mcmc.h:92 */
    virtual void eval(const SharedParams&) { loglike = 0; }                    /* mcmc.h:96-98 */
    virtual void compute_derived_late(const SharedParams&) {}                  /* mcmc.h:103-104 */
    virtual Proposal step(pcg32&, const SharedParams&) const { return Proposal{copy(), 1}; } /* mcmc.h:107-110 */
    virtual std::vector<Float> sampleInitialConditions(pcg32&) { return {}; }  /* mcmc.h:112-114 */
    virtual std::vector<Float> getInitialConditions() { return {}; }           /* mcmc.h:116-118 */
    virtual void setInitialConditions(const std::vector<Float>&) {}            /* mcmc.h:120-121 */
    bool isFixed(std::pair<size_t, size_t> index) const { return fixed.find(index) != fixed.end(); } /* mcmc.h:158-160 */

    Float loglike = 0;
    Float stepsizeCorrectionFac = 1;
    bool derivedOnShared = false;
    std::set<std::pair<size_t, size_t>> fixed = {};
    int dim = 0;

    SubspaceState(std::vector<std::string>&& coordNames, bool derivedParamsDependOnSharedParams = false,
                  size_t nDerived = 0)
        : derivedOnShared(derivedParamsDependOnSharedParams), nDerived(nDerived) {
        for (size_t i = 0; i < coordNames.size(); ++i) names[coordNames[i]] = i;
    }
    virtual ~SubspaceState() {}
    virtual std::shared_ptr<SubspaceState> copy() const = 0;
    virtual void force_bounds() {}

    void setCoords(std::vector<std::vector<Float>> init) {           /* mcmc.h:81-88 */
        coords = init;
        dim = 0;
        for (size_t i = 0; i + nDerived < init.size(); ++i) dim += (int)init[i].size();
    }
    auto& getCoords() { return coords; }
    const auto& getCoords() const { return coords; }
    std::optional<std::vector<Float>> getCoords(const std::string& name) const {
        auto it = names.find(name);
        if (it != names.end()) return coords[it->second];
        return {};
    }
    std::vector<Float>& getCoordsAt(const std::string& name) { return coords[names.at(name)]; }
    std::map<std::string, std::vector<Float>> getAll() const {
        std::map<std::string, std::vector<Float>> ret;
        for (auto& [name, idx] : names) ret[name] = coords[idx];
        return ret;
    }
    bool isDerived(const std::string& name) const {
        auto it = names.find(name);
        return it != names.end() && it->second >= names.size() - nDerived;
    }
    std::map<std::string, size_t> get_names() const { return names; }
    size_t numDerived() const { return nDerived; }

protected:
    std::vector<std::vector<Float>> coords = {};
    std::map<std::string, size_t> names = {};
    std::vector<std::string> requestedSharedNames;                /* mcmc.h:175 */
    std::unordered_set<std::string> offeredSharedNames;           /* mcmc.h:178 */
    size_t nDerived = 0;
};

/* mcmc.h:185-228 */
class SmoothnessPrior : public SubspaceState {
public:
    SmoothnessPrior(const std::string& functionName, Float smoothnessScale, Float scale)
        : SubspaceState({}), L(scale), Lsmooth(smoothnessScale), functionName(functionName) {
        requestedSharedNames.push_back(functionName);                 /* mcmc.h:190 */
    }
    Float L, Lsmooth;
    std::string functionName;
    std::shared_ptr<SubspaceState> copy() const override { return std::make_shared<SmoothnessPrior>(*this); }
};

/* covid.h:16-57; the arrays are [2][nData] (deaths, sigma) and [3][nDisc] (day, value,
 * flag: >= 0 sampled, < 0 fixed), row-major */
class DiseaseData {
public:
    DiseaseData(const Float* deathsAndSigma, size_t nData, const Float* daysValsFlags, size_t nDisc) {
        deathsPerDay.assign(deathsAndSigma, deathsAndSigma + nData);
        deathsSigma.assign(deathsAndSigma + nData, deathsAndSigma + 2 * nData);
        discontinuousDays.resize(nDisc);
        discontinuousVals.resize(nDisc);
        discontinuousValsFixed.resize(nDisc);
        for (size_t i = 0; i < nDisc; ++i) {
            discontinuousDays[i] = (int)std::round(daysValsFlags[i]);
            discontinuousVals[i] = daysValsFlags[nDisc + i];
            discontinuousValsFixed[i] = !(daysValsFlags[2 * nDisc + i] >= 0);
        }
    }
    std::vector<Float> deathsPerDay, deathsSigma;
    std::vector<int> discontinuousDays;
    std::vector<Float> discontinuousVals;
    std::vector<bool> discontinuousValsFixed;
    Float initialBetaMild = 2;
    Float initialBetaHigh = 7;
    Float initialDelay = 14;
    int fixBehaviorInAdvance = 14;
};

/* covid.h:59-93 */
class DiseaseParams {
public:
    Float timeIncub = 4, timeIncubSigma = 2, probAsymp = 0.1, probSerious = 0.08, probLethal = 0.004,
          probLethalDailyWhenSeriousUntreated = 0.2, probICUIfSerious = 0.3, timeMildDuration = 10,
          timeMildDurationSigma = 5, timeMildToSerious = 8, timeMildToSeriousSigma = 3, timeSeriousToRec = 10,
          timeSeriousToRecSigma = 5, timeSeriousToDeath = 9, timeSeriousToDeathSigma = 5;
    mcmc_disease_params c_struct() const {
        return mcmc_disease_params{timeIncub, timeIncubSigma, probAsymp, probSerious, probLethal,
                                   probLethalDailyWhenSeriousUntreated, probICUIfSerious, timeMildDuration,
                                   timeMildDurationSigma, timeMildToSerious, timeMildToSeriousSigma,
                                   timeSeriousToRec, timeSeriousToRecSigma, timeSeriousToDeath,
                                   timeSeriousToDeathSigma};
    }
};

/* covid.h:284-321, 629-646.  The constructor runs the AvgDiseaseTrajectory precompute
 * (covid.h:100-203, host, as in the reference) and keeps the 10 x 500 response table the
 * device kernels read; the 8.8 MB trajectory grid itself is not kept, so copies are cheap
 * (the reference copies it on every proposal, covid.h:290,501). */
class DiseaseSpread : public SubspaceState {
public:
    const DiseaseData& data;
    const DiseaseParams& params;
    int nPredictDays, maxDelayDaysTilData;
    int popSize;
    double cap0, capIncrRate;
    std::shared_ptr<const std::vector<double>> responseTable;      /* [10][500] */
    /* Extension, no counterpart in the reference: run the renewal convolution of eval in packed
     * FP32 on the device (MCMC_HIP_PRECISION_F32_MODEL, include/mcmc_hip.h).  Off = `Float` is
     * double everywhere, as in the reference (mcmc.h:43). */
    bool fastMode = false;
    /* Extension, no counterpart in the reference and off by default: the reference's DiseaseSpread
     * does not override sampleInitialConditions (the base returns {}, mcmc.h:112-114), so
     * ChainManager(target, nReservoir, nChain) weighs the same point nReservoir times and seeds ONE
     * chain.  With initialSpread = a > 0 a draw is the point the likelihood stood at when the first
     * draw was made, every sampled coordinate moved by State::perturb's rule (mcmc.h:306-314),
     *     x = x0 + (rnd.nextFloat() - .5f) * x0 * a,
     * in the coords' order (behavior, discontinuousVals, betaMild, betaHigh, delay), then
     * force_bounds.  The draw density is constant over its box, so the manager's importance
     * weights (target density / draw density, mcmc.h:1331-1338) stay what they are up to a
     * common factor; points that force_bounds moved to an edge carry the edge's weight. */
    double initialSpread = 0;
    std::vector<Float> initialCentre;      /* x0 above, taken at the first draw */

    static constexpr uint64_t kTrajectories = 10000000;             /* covid.h:207 */

    DiseaseSpread(const DiseaseData& data, const DiseaseParams& params, int popSize, double cap0,
                  double capIncrRate, int maxDelayDaysTilData, size_t nPredictDays,
                  uint64_t nTrajectories = kTrajectories)
        : SubspaceState({"behavior", "discontinuousVals", "betaMild", "betaHigh", "delay", "mildlyInfectious",
                         "highlyInfectious", "incubating", "asymptomatic", "mild", "serious", "recovered", "dead",
                         "capacity", "totalBehavior"}, false, 10),
          data(data), params(params), nPredictDays((int)nPredictDays), maxDelayDaysTilData(maxDelayDaysTilData),
          popSize(popSize), cap0(cap0), capIncrRate(capIncrRate) {
        const size_t nData = data.deathsPerDay.size();
        const size_t nDaysTotal = maxDelayDaysTilData + nPredictDays + nData;          /* covid.h:297 */
        if (nData <= (size_t)data.fixBehaviorInAdvance)
            throw std::invalid_argument("DiseaseSpread: fixBehaviorInAdvance must be smaller than the number of data days");
        std::vector<std::vector<Float>> init = {
            std::vector<Float>(nData - data.fixBehaviorInAdvance, Float(1)), data.discontinuousVals,
            {data.initialBetaMild}, {data.initialBetaHigh}, {data.initialDelay}};
        for (int k = 0; k < 9; ++k) init.push_back(std::vector<Float>(nDaysTotal, Float(0)));
        init.push_back(std::vector<Float>(nDaysTotal, Float(1)));
        setCoords(init);
        for (size_t i = 0; i < nDaysTotal; ++i)
            getCoordsAt("capacity")[i] = cap0 + std::max(capIncrRate * ((double)i - maxDelayDaysTilData), Float(0.0));
        for (size_t i = 0; i < data.discontinuousValsFixed.size(); ++i)
            if (data.discontinuousValsFixed[i]) fixed.insert(std::make_pair(size_t(1), i));
        auto tab = std::make_shared<std::vector<double>>((size_t)MCMC_HIP_NCURVES * MCMC_HIP_TRAJ_NDAYS);
        const mcmc_disease_params p = params.c_struct();
        hip_check(mcmc_host_make_response_tables(&p, nTrajectories, MCMC_HIP_TRAJ_NDAYS, tab->data()),
                  "AvgDiseaseTrajectory");
        responseTable = tab;
    }

    void force_bounds() override {                                    /* covid.h:629-646 */
        bound(getCoordsAt("delay")[0], Float(5), Float(maxDelayDaysTilData));
        Float lastVal = 1;
        auto& dv = getCoordsAt("discontinuousVals");
        for (size_t i = 0; i < dv.size(); ++i) {
            if (data.discontinuousValsFixed[i]) continue;
            bound(dv[i], Float(0), lastVal);
            lastVal = dv[i];
        }
        for (auto& b : getCoordsAt("behavior")) bound(b, Float(0), Float(2));
        bound(getCoordsAt("betaMild")[0], Float(0), Float(100));
        bound(getCoordsAt("betaHigh")[0], Float(0), Float(100));
    }
    std::shared_ptr<SubspaceState> copy() const override { return std::make_shared<DiseaseSpread>(*this); }

    /* one draw of the initialSpread extension above; false = off, nothing drawn, nothing moved */
    bool sampleInitialPoint(pcg32& rnd) {
        if (!(initialSpread > 0)) { initialCentre.clear(); return false; }   /* switched off: forget the centre */
        if (initialCentre.empty()) initialCentre = pack();
        std::vector<Float> x = initialCentre;
        for (auto& v : x) v += (rnd.nextFloat() - .5f) * v * Float(initialSpread);
        unpack(x.data());
        for (const auto& f : fixed) coords[f.first][f.second] = initialCentre_at(f.first, f.second);
        force_bounds();
        return true;
    }

    size_t nDaysTotal() const { return maxDelayDaysTilData + nPredictDays + data.deathsPerDay.size(); }
    size_t nBehavior() const { return data.deathsPerDay.size() - data.fixBehaviorInAdvance; }
    size_t nSampled() const { return nBehavior() + data.discontinuousVals.size() + 3; }

    /* the sampled coordinates in the C ABI's order (= the coords order, covid.h:294) */
    std::vector<Float> pack() const {
        std::vector<Float> x;
        for (int k = 0; k < 5; ++k) x.insert(x.end(), coords[k].begin(), coords[k].end());
        return x;
    }
    void unpack(const Float* x) {
        for (int k = 0; k < 5; ++k)
            for (auto& v : coords[k]) v = *x++;
    }
    Float initialCentre_at(size_t k, size_t i) const {
        size_t off = 0;
        for (size_t c = 0; c < k; ++c) off += coords[c].size();
        return initialCentre[off + i];
    }
    /* derived output coords from mcmc_hip_derive's [9][size] block (covid.h:467-485) */
    void set_derived(const Float* curves) {
        static const char* order[9] = {"mildlyInfectious", "highlyInfectious", "incubating", "asymptomatic", "mild",
                                       "serious", "recovered", "dead", "totalBehavior"};
        const size_t n = nDaysTotal();
        for (int c = 0; c < 9; ++c) getCoordsAt(order[c]).assign(curves + c * n, curves + (c + 1) * n);
    }
};

/* --------------------------------------------------------------- toy likelihoods ---- */
/* test.h:14-136.  Both pass (names, 1, false) to a constructor whose order is
 * (names, derivedOnShared, nDerived) (mcmc.h:75), i.e. derivedOnShared = true and
 * nDerived = 0: MyLike1's coordinate "max" counts as a coordinate like "position". */
class ToyLikelihood : public SubspaceState {
public:
    ToyLikelihood(std::vector<std::string>&& coordNames, int kind, Float difficulty)
        : SubspaceState(std::move(coordNames), true, 0), toyKind(kind), difficulty(difficulty) {}
    int toyKind;
    Float difficulty;
    size_t nSampled() const { return coords[0].size(); }                  /* what step() moves */
    size_t nOut() const { size_t n = 0; for (auto& c : coords) n += c.size(); return n; }
    std::vector<Float> pack() const { return coords[0]; }
    /* coords that eval() sets besides loglike */
    virtual void refresh_derived() {}
    void unpack(const Float* x) { for (auto& v : coords[0]) v = *x++; refresh_derived(); }
};
class MyLike1 : public ToyLikelihood {
public:
    MyLike1() : ToyLikelihood({"position", "max"}, MCMC_TOY_MYLIKE1, 1) { setCoords({{1, 1, 1}, {1}}); }   /* test.h:18-22 */
    void refresh_derived() override { coords[1][0] = std::max(std::max(coords[0][0], coords[0][1]), coords[0][2]); } /* test.h:37 */
    std::shared_ptr<SubspaceState> copy() const override { return std::make_shared<MyLike1>(*this); }
};
class FourGaussians : public ToyLikelihood {
public:
    FourGaussians(Float difficulty) : ToyLikelihood({"pos"}, MCMC_TOY_FOURGAUSS, difficulty) { setCoords({{0, 0}}); } /* test.h:73-77 */
    std::vector<Float> sampleInitialConditions(pcg32& rnd) override {    /* test.h:116-123 */
        std::vector<Float> ret = {-5 + 10 * rnd.nextDouble()};
        ret.push_back(-5 + 10 * rnd.nextDouble());
        unpack(ret.data());
        return ret;
    }
    std::vector<Float> getInitialConditions() override { return coords[0]; }                  /* test.h:124-126 */
    void setInitialConditions(const std::vector<Float>& ics) override { if (ics.size() >= 2) unpack(ics.data()); } /* test.h:127-130 */
    std::shared_ptr<SubspaceState> copy() const override { return std::make_shared<FourGaussians>(*this); }
};

/* EXTENSION (BASELINE configs[0] names a Rosenbrock target; the reference has none): FourGaussians'
 * pattern with loglike = -((1-x)^2 + difficulty (y-x^2)^2) / 20 and an unbounded random-walk step. */
class Rosenbrock : public ToyLikelihood {
public:
    Rosenbrock(Float difficulty = 100) : ToyLikelihood({"pos"}, MCMC_TOY_ROSENBROCK, difficulty) { setCoords({{0, 0}}); }
    std::vector<Float> sampleInitialConditions(pcg32& rnd) override {    /* as test.h:116-123 */
        std::vector<Float> ret = {-5 + 10 * rnd.nextDouble()};
        ret.push_back(-5 + 10 * rnd.nextDouble());
        unpack(ret.data());
        return ret;
    }
    std::vector<Float> getInitialConditions() override { return coords[0]; }
    void setInitialConditions(const std::vector<Float>& ics) override { if (ics.size() >= 2) unpack(ics.data()); }
    std::shared_ptr<SubspaceState> copy() const override { return std::make_shared<Rosenbrock>(*this); }
};

/* ---------------------------------------------------------------- device engine ---- */
enum class TargetKind { Plain = MCMC_TARGET_PLAIN, Temp = MCMC_TARGET_TEMP, Cooling = MCMC_TARGET_COOLING };
struct TargetSpec { TargetKind kind = TargetKind::Plain; Float T = 1, slope = 0, Tinit = 1; };

/* RAII owner of one mcmc_hip_ctx for a given model + target */
class Engine {
public:
    Engine(const DiseaseSpread& ds, const SmoothnessPrior* prior, const TargetSpec& t, int device = 0,
           int contract = default_contract()) {
        std::vector<int32_t> days(ds.data.discontinuousDays.begin(), ds.data.discontinuousDays.end());
        std::vector<uint8_t> fixedv(ds.data.discontinuousValsFixed.size());
        for (size_t i = 0; i < fixedv.size(); ++i) fixedv[i] = ds.data.discontinuousValsFixed[i] ? 1 : 0;
        mcmc_covid_config cfg{};
        cfg.n_data = (int32_t)ds.data.deathsPerDay.size();
        cfg.max_delay = ds.maxDelayDaysTilData;
        cfg.n_predict = ds.nPredictDays;
        cfg.n_disc = (int32_t)days.size();
        cfg.fix_behavior_in_advance = ds.data.fixBehaviorInAdvance;
        cfg.pop_size = ds.popSize;
        cfg.prob_icu_if_serious = ds.params.probICUIfSerious;
        cfg.prob_lethal_daily_untreated = ds.params.probLethalDailyWhenSeriousUntreated;
        cfg.cap0 = ds.cap0;
        cfg.cap_incr_rate = ds.capIncrRate;
        cfg.has_prior = prior ? 1 : 0;
        cfg.smooth_Ls = prior ? prior->Lsmooth : 0;
        cfg.smooth_L = prior ? prior->L : 1;
        cfg.target_kind = (int32_t)t.kind;
        cfg.target_T = t.T; cfg.target_slope = t.slope; cfg.target_Tinit = t.Tinit;
        cfg.contract = contract;
        cfg.disc_days = days.data();
        cfg.disc_fixed = fixedv.data();
        hip_check(mcmc_hip_create(&cfg, device, &ctx), "mcmc_hip_create");
        try {
            hip_check(mcmc_hip_set_tables(ctx, ds.responseTable->data(), MCMC_HIP_TRAJ_NDAYS), "mcmc_hip_set_tables");
            hip_check(mcmc_hip_set_data(ctx, ds.data.deathsPerDay.data(), ds.data.deathsSigma.data()), "mcmc_hip_set_data");
            if (ds.fastMode) hip_check(mcmc_hip_set_precision(ctx, MCMC_HIP_PRECISION_F32_MODEL), "mcmc_hip_set_precision");
        } catch (...) { mcmc_hip_destroy(ctx); ctx = nullptr; throw; }
        dim = mcmc_hip_dim(ctx);
        size = mcmc_hip_size(ctx);
    }
    ~Engine() { if (ctx) mcmc_hip_destroy(ctx); }
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;

    /* FMA contraction of the projection statement (covid.h:384): what the reference's own
     * build flags (-march=native, CMakeLists.txt:98) give; MCMC_AMD_CONTRACT=0 selects the
     * strict two-rounding form */
    static int default_contract() {
        const char* e = std::getenv("MCMC_AMD_CONTRACT");
        return e ? std::atoi(e) : 1;
    }
    mcmc_hip_ctx* ctx = nullptr;
    int dim = 0, size = 0;
};

/* What one batched device run recorded, chain-major.  A saved point / mean row has nOut
 * columns: the sampled coordinates (covid) or all coords of the toy likelihood. */
struct BatchResults {
    size_t n = 0, dim = 0, nOut = 0;
    int nSaved = 0;
    bool recorded = false;                               /* the saved points stayed on the device / in the backend */
    std::vector<Float> lls, params, mean, fll, fss;      /* [n][nSaved] [n][dim] [n][nOut] (empty: no mean) [n][2] [n] */
    std::vector<int32_t> acc;
    std::vector<uint64_t> rng;                           /* [n][2] pcg32 (state, inc) at the end of the run */
    std::vector<Float> diagnostics;                      /* mcmc_hip_diagnostics (covid) */
};

/* What a run records and where it starts from.  recordSamples: keep the saved points (the
 * reference's `ics`, mcmc.h:1043, + `samples`, mcmc.h:1060-1061); computeMean: mcmc.h:1064-1065;
 * rng [n][2] / meanIn [n][nOut]: continue these generator streams / running means instead of
 * seeding pcg32(0, id) and starting the mean at the current point -- a MetropolisChain keeps both
 * across run() calls (mcmc.h:994-1003). */
struct RunOptions {
    bool recordSamples = true;
    bool computeMean = false;
    const uint64_t* rng = nullptr;
    const Float* meanIn = nullptr;
};

/* The device work behind one supported State composition. */
class Backend {
public:
    virtual ~Backend() {}
    virtual size_t dim() const = 0;
    virtual size_t n_out() const = 0;
    virtual size_t curve_len() const { return 0; }
    /* ll [n][2]: (likelihood, prior) */
    virtual void eval(size_t n, const Float* x, Float* ll) = 0;
    /* derived output curves of the covid model, [n][9][curve_len] */
    virtual void derive(size_t, const Float*, Float*) { throw std::logic_error("derive: not a covid state"); }
    /* MetropolisChain::run for n chains with ids firstId.. */
    virtual void run(BatchResults& R, size_t n, const Float* x, const Float* ss, uint64_t firstId, int nSamples,
                     int nBurnin, int nAdjust, int thinning, const RunOptions& opt) = 0;
    /* the saved points of the last run (recordSamples): chain i as [nSaved][nOut], saved point s of
     * every chain as [n][nOut] -- fetched on request, not with every run */
    virtual std::vector<Float> chain_samples(size_t i) = 0;
    virtual std::vector<Float> sample_all(int s) = 0;
    /* Tempering ladders (extension, no counterpart in the reference): chains [L*K, (L+1)*K) form ladder L on the
     * temperatures `temps`; nRounds x (stepsPerRound MH steps, one exchange round).  rungs[i] = the rung chain i
     * ends on; swapAcc/swapProp [K-1] per neighbouring pair. */
    virtual void run_tempered(BatchResults&, size_t, const Float*, const Float*, uint64_t, const std::vector<Float>&,
                              uint64_t, int, int, std::vector<int32_t>&, std::vector<int64_t>&, std::vector<int64_t>&) {
        throw std::logic_error("run_tempered: only the covid state has a tempering ladder");
    }
};

class CovidBackend : public Backend {
public:
    CovidBackend(const DiseaseSpread& ds, const SmoothnessPrior* prior, const TargetSpec& t) : eng(ds, prior, t) {}
    size_t dim() const override { return (size_t)eng.dim; }
    size_t n_out() const override { return (size_t)eng.dim; }
    size_t curve_len() const override { return (size_t)eng.size; }
    void eval(size_t n, const Float* x, Float* ll) override {
        hip_check(mcmc_hip_eval(eng.ctx, (int64_t)n, x, ll), "mcmc_hip_eval");
    }
    void derive(size_t n, const Float* x, Float* curves) override {
        hip_check(mcmc_hip_derive(eng.ctx, (int64_t)n, x, curves), "mcmc_hip_derive");
    }
    void run(BatchResults& R, size_t n, const Float* x, const Float* ss, uint64_t firstId, int nSamples, int nBurnin,
             int nAdjust, int thinning, const RunOptions& opt) override {
        const size_t d = dim();
        hip_check(mcmc_hip_set_chains(eng.ctx, (int64_t)n, x, ss, nullptr, firstId), "mcmc_hip_set_chains");
        if (opt.rng || opt.meanIn)
            hip_check(mcmc_hip_set_state(eng.ctx, opt.rng, nullptr, opt.meanIn, nullptr), "mcmc_hip_set_state");
        hip_check(mcmc_hip_run(eng.ctx, nSamples, nBurnin, nAdjust, thinning, opt.recordSamples ? 1 : 0,
                               opt.computeMean ? 1 : 0), "mcmc_hip_run");
        const int nSaved = mcmc_hip_n_saved(nSamples, nBurnin, nAdjust, thinning);
        R = BatchResults{};
        R.lls.assign(n * (size_t)std::max(nSaved, 1), 0); R.params.assign(n * d, 0);
        R.fll.assign(n * 2, 0); R.fss.assign(n, 0); R.acc.assign(n, 0); R.rng.assign(n * 2, 0);
        if (nSaved > 0) hip_check(mcmc_hip_get_loglikes(eng.ctx, R.lls.data()), "mcmc_hip_get_loglikes");
        hip_check(mcmc_hip_get_params(eng.ctx, R.params.data()), "mcmc_hip_get_params");
        if (opt.computeMean) {
            R.mean.assign(n * d, 0);
            hip_check(mcmc_hip_get_mean(eng.ctx, R.mean.data()), "mcmc_hip_get_mean");
        }
        hip_check(mcmc_hip_get_loglike(eng.ctx, R.fll.data()), "mcmc_hip_get_loglike");
        hip_check(mcmc_hip_get_stepsizes(eng.ctx, R.fss.data()), "mcmc_hip_get_stepsizes");
        hip_check(mcmc_hip_get_accept_counts(eng.ctx, R.acc.data()), "mcmc_hip_get_accept_counts");
        hip_check(mcmc_hip_get_rng_state(eng.ctx, R.rng.data()), "mcmc_hip_get_rng_state");
        R.diagnostics.assign(4 + 2 * d, 0);
        hip_check(mcmc_hip_diagnostics(eng.ctx, R.diagnostics.data()), "mcmc_hip_diagnostics");
        R.n = n; R.dim = d; R.nOut = d; R.nSaved = nSaved; R.recorded = opt.recordSamples && nSaved > 0;
        lastSaved = R.recorded ? nSaved : 0; lastN = n;
    }
    std::vector<Float> chain_samples(size_t i) override {
        std::vector<Float> out((size_t)lastSaved * dim());
        if (lastSaved > 0) hip_check(mcmc_hip_get_samples(eng.ctx, (int64_t)i, 0, (int32_t)dim(), out.data()), "mcmc_hip_get_samples");
        return out;
    }
    std::vector<Float> sample_all(int s) override {
        if (s < 0 || s >= lastSaved) throw std::out_of_range("sample_all: no such saved point");
        std::vector<Float> out(lastN * dim());
        hip_check(mcmc_hip_get_sample_all(eng.ctx, s, out.data()), "mcmc_hip_get_sample_all");
        return out;
    }
    void run_tempered(BatchResults& R, size_t n, const Float* x, const Float* ss, uint64_t firstId,
                      const std::vector<Float>& temps, uint64_t swapSeed, int nRounds, int stepsPerRound,
                      std::vector<int32_t>& rungs, std::vector<int64_t>& swapAcc, std::vector<int64_t>& swapProp) override {
        const size_t d = dim(), K = temps.size();
        hip_check(mcmc_hip_set_chains(eng.ctx, (int64_t)n, x, ss, nullptr, firstId), "mcmc_hip_set_chains");
        hip_check(mcmc_hip_set_ladder(eng.ctx, (int32_t)K, temps.data(), swapSeed, 0), "mcmc_hip_set_ladder");
        hip_check(mcmc_hip_run_tempered(eng.ctx, nRounds, stepsPerRound), "mcmc_hip_run_tempered");
        R = BatchResults{};
        R.params.assign(n * d, 0); R.mean.assign(n * d, 0); R.fll.assign(n * 2, 0); R.fss.assign(n, 0); R.acc.assign(n, 0);
        hip_check(mcmc_hip_get_params(eng.ctx, R.params.data()), "mcmc_hip_get_params");
        R.mean = R.params;                                   /* no running mean in a tempered run */
        hip_check(mcmc_hip_get_loglike(eng.ctx, R.fll.data()), "mcmc_hip_get_loglike");
        hip_check(mcmc_hip_get_stepsizes(eng.ctx, R.fss.data()), "mcmc_hip_get_stepsizes");
        hip_check(mcmc_hip_get_accept_counts(eng.ctx, R.acc.data()), "mcmc_hip_get_accept_counts");
        R.rng.assign(n * 2, 0);
        hip_check(mcmc_hip_get_rng_state(eng.ctx, R.rng.data()), "mcmc_hip_get_rng_state");
        rungs.assign(n, 0); swapAcc.assign(K - 1, 0); swapProp.assign(K - 1, 0);
        hip_check(mcmc_hip_get_rungs(eng.ctx, rungs.data()), "mcmc_hip_get_rungs");
        hip_check(mcmc_hip_get_swap_counts(eng.ctx, swapAcc.data(), swapProp.data()), "mcmc_hip_get_swap_counts");
        R.diagnostics.assign(4 + 2 * d, 0);
        hip_check(mcmc_hip_diagnostics(eng.ctx, R.diagnostics.data()), "mcmc_hip_diagnostics");
        R.n = n; R.dim = d; R.nOut = d; R.nSaved = 0;
        lastSaved = 0; lastN = n;
    }
    Engine eng;
private:
    int lastSaved = 0;
    size_t lastN = 0;
};

class ToyBackend : public Backend {
public:
    ToyBackend(const ToyLikelihood& t, const TargetSpec& spec, int device = 0)
        : kind(t.toyKind), difficulty(t.difficulty), spec(spec), device(device) {
        d = mcmc_hip_toy_dim(kind); no = mcmc_hip_toy_n_out(kind);
        if (d < 0 || no < 0) throw HipError(d, std::string("ToyBackend: ") + mcmc_hip_last_error());
    }
    size_t dim() const override { return (size_t)d; }
    size_t n_out() const override { return (size_t)no; }
    void eval(size_t n, const Float* x, Float* ll) override {
        std::vector<Float> l(n);
        hip_check(mcmc_hip_toy_eval(device, kind, difficulty, (int64_t)n, x, l.data()), "mcmc_hip_toy_eval");
        for (size_t i = 0; i < n; ++i) { ll[2 * i] = l[i]; ll[2 * i + 1] = 0; }
    }
    void run(BatchResults& R, size_t n, const Float* x, const Float* ss, uint64_t firstId, int nSamples, int nBurnin,
             int nAdjust, int thinning, const RunOptions& opt) override {
        const int nSaved = mcmc_hip_n_saved(nSamples, nBurnin, nAdjust, thinning);
        R = BatchResults{};
        R.params.assign(x, x + n * d);
        R.fss.assign(ss, ss + n);
        R.lls.assign(n * (size_t)std::max(nSaved, 1), 0); R.acc.assign(n, 0);
        R.rng.assign(n * 2, 0);
        for (size_t c = 0; c < n; ++c) {                 /* continue the given streams, else rnd(0, id), mcmc.h:999 */
            if (opt.rng) { R.rng[2 * c] = opt.rng[2 * c]; R.rng[2 * c + 1] = opt.rng[2 * c + 1]; }
            else { pcg32 r(0, firstId + c); R.rng[2 * c] = r.state; R.rng[2 * c + 1] = r.inc; }
        }
        if (opt.computeMean) {
            R.mean.assign(n * no, 0);
            if (opt.meanIn) R.mean.assign(opt.meanIn, opt.meanIn + n * no);
        }
        samples.assign(opt.recordSamples ? n * (size_t)std::max(nSaved, 1) * no : 0, 0);
        hip_check(mcmc_hip_toy_run(device, kind, difficulty, (int32_t)spec.kind, spec.T, spec.slope, spec.Tinit, (int64_t)n,
                                   nullptr, firstId, R.params.data(), R.fss.data(), nSamples, nBurnin, nAdjust, thinning,
                                   R.lls.data(), opt.recordSamples ? samples.data() : nullptr,
                                   opt.computeMean ? R.mean.data() : nullptr, R.acc.data(), R.rng.data(),
                                   (opt.computeMean && opt.meanIn) ? 1 : 0), "mcmc_hip_toy_run");
        R.fll.assign(n * 2, 0);
        eval(n, R.params.data(), R.fll.data());
        R.n = n; R.dim = d; R.nOut = no; R.nSaved = nSaved; R.recorded = opt.recordSamples && nSaved > 0;
        lastSaved = R.recorded ? nSaved : 0; lastN = n;
    }
    std::vector<Float> chain_samples(size_t i) override {
        return std::vector<Float>(samples.begin() + i * lastSaved * no, samples.begin() + (i + 1) * lastSaved * no);
    }
    std::vector<Float> sample_all(int s) override {
        if (s < 0 || s >= lastSaved) throw std::out_of_range("sample_all: no such saved point");
        std::vector<Float> out(lastN * no);
        for (size_t c = 0; c < lastN; ++c)
            std::copy(samples.begin() + (c * lastSaved + s) * no, samples.begin() + (c * lastSaved + s + 1) * no, out.begin() + c * no);
        return out;
    }
private:
    int kind; Float difficulty; TargetSpec spec; int device; int d = 0, no = 0;
    std::vector<Float> samples;                          /* [n][nSaved][nOut] of the last recorded run */
    int lastSaved = 0;
    size_t lastN = 0;
};

/* ------------------------------------------------------------------------- State ---- */
/* mcmc.h:241-700, restricted to the compositions the device path runs:
 *     [ DiseaseSpread ],  [ DiseaseSpread, SmoothnessPrior("behavior", Ls, L) ],
 *     [ MyLike1 ],  [ FourGaussians(difficulty) ],  [ Rosenbrock(difficulty) ] */
class State {
public:
    std::vector<std::shared_ptr<SubspaceState>> state;
    std::vector<std::vector<Float>> initialConditions;
    Float weight = 1;
    int dim = 0;
    bool isInitialized = false;
    bool isEvaluated = false;
    int sharedDependencyMaxDepth = 10;

    virtual ~State() {}
    void add(const std::shared_ptr<SubspaceState>& s) { state.push_back(s); dim += s->dim; }

    DiseaseSpread* disease() const {
        return state.empty() ? nullptr : dynamic_cast<DiseaseSpread*>(state[0].get());
    }
    SmoothnessPrior* prior() const {
        return state.size() < 2 ? nullptr : dynamic_cast<SmoothnessPrior*>(state[1].get());
    }
    ToyLikelihood* toy() const {
        return state.size() != 1 ? nullptr : dynamic_cast<ToyLikelihood*>(state[0].get());
    }
    bool supported(std::string* why = nullptr) const {
        auto fail = [&](const char* m) { if (why) *why = m; return false; };
        if (toy()) return true;
        if (state.empty() || !disease())
            return fail("the first likelihood must be a DiseaseSpread (or the state a single MyLike1 / FourGaussians)");
        if (state.size() > 2) return fail("at most [DiseaseSpread, SmoothnessPrior] is supported on the GPU path");
        if (state.size() == 2 && (!prior() || prior()->functionName != "behavior"))
            return fail("the second likelihood must be SmoothnessPrior(\"behavior\", ...)");
        return true;
    }
    /* the device work for this composition under a target */
    std::unique_ptr<Backend> backend(const TargetSpec& t) const {
        std::string why;
        if (!supported(&why)) throw std::runtime_error("State: " + why);
        if (ToyLikelihood* tl = toy()) return std::make_unique<ToyBackend>(*tl, t);
        return std::make_unique<CovidBackend>(*disease(), prior(), t);
    }
    /* the coordinates a chain moves, in the C ABI's order */
    std::vector<Float> packed() const { return toy() ? toy()->pack() : disease() ? disease()->pack() : std::vector<Float>{}; }
    void set_packed(const Float* x) { if (toy()) toy()->unpack(x); else if (disease()) disease()->unpack(x); }
    /* columns [off, off+len) of a saved point / mean row that hold coordinate `name` */
    bool out_range(const std::string& name, size_t& off, size_t& len) const {
        if (state.empty()) return false;
        const SubspaceState& s = *state[0];
        const size_t nc = toy() ? s.getCoords().size() : 5;
        const std::map<std::string, size_t> names = s.get_names();
        const auto it = names.find(name);
        if (it == names.end() || it->second >= nc) return false;
        off = 0;
        for (size_t k = 0; k < it->second; ++k) off += s.getCoords()[k].size();
        len = s.getCoords()[it->second].size();
        return true;
    }

    void init() {                                                      /* mcmc.h:401-492 */
        std::cout << "\nResolving dependencies for state with " << state.size() << " likelihoods \n";
        std::string why;
        if (!supported(&why)) {
            std::cout << "State::init: " << why << ".\n";
            throw std::string("Missing shared param");                  /* mcmc.h:460 */
        }
        if (prior()) std::cout << "Searching for behavior...found in likelihood 0\n";
        isInitialized = true;
        eval();
    }

    /* State::eval (mcmc.h:385-393) on the device: log-likelihoods and the derived curves */
    void eval() {
        std::unique_ptr<Backend> be = backend(TargetSpec{});
        const std::vector<Float> x = packed();
        Float ll[2] = {0, 0};
        be->eval(1, x.data(), ll);
        state[0]->loglike = ll[0];
        if (DiseaseSpread* ds = disease()) {
            std::vector<Float> curves((size_t)9 * be->curve_len());
            be->derive(1, x.data(), curves.data());
            ds->set_derived(curves.data());
            if (prior()) prior()->loglike = ll[1];
        } else {
            toy()->refresh_derived();
        }
        isEvaluated = true;
    }

    Float loglikelihood() const {                                      /* mcmc.h:612-618 */
        Float l = 0;
        for (auto& s : state) l += s->loglike;
        return l;
    }
    void force_bounds() { for (auto& s : state) s->force_bounds(); }
    std::map<std::string, std::vector<Float>> get_all() const {
        std::map<std::string, std::vector<Float>> ret;
        for (auto& s : state) { auto r = s->getAll(); ret.insert(r.begin(), r.end()); }
        return ret;
    }
    std::optional<std::vector<Float>> getCoords(const std::string& name) const {
        for (auto& s : state) if (auto c = s->getCoords(name)) return c;
        return {};
    }
    void set_stepsizes(const std::vector<Float>& sz) {
        for (size_t i = 0; i < state.size() && i < sz.size(); ++i) state[i]->stepsizeCorrectionFac = sz[i];
    }
    std::vector<Float> get_stepsizes() const {
        std::vector<Float> r;
        for (auto& s : state) r.push_back(s->stepsizeCorrectionFac);
        return r;
    }
    std::shared_ptr<State> deep_copy() const {
        State ret(*this);
        for (auto& s : ret.state) s = s->copy();
        return std::make_shared<State>(ret);
    }
    /* mcmc.h:317-325.  Only FourGaussians draws something (test.h:116-123); the other
     * likelihoods keep the base's empty answer (mcmc.h:112-114) and stay where they are. */
    std::vector<std::vector<Float>> sampleInitialConditions(pcg32& rnd) {
        if (ToyLikelihood* t = toy()) t->sampleInitialConditions(rnd);
        else if (DiseaseSpread* ds = disease()) ds->sampleInitialPoint(rnd);     /* opt-in extension */
        isEvaluated = false;
        eval();
        return getInitialConditions();
    }
    /* DiseaseSpread and MyLike1 have no getInitialConditions (the base returns {},
     * mcmc.h:116-118): here a state's initial conditions are the coordinates a chain moves */
    std::vector<std::vector<Float>> getInitialConditions() {
        initialConditions = {packed()};
        for (size_t i = 1; i < state.size(); ++i) initialConditions.push_back({});
        return initialConditions;
    }
    void setInitialConditions(const std::vector<std::vector<Float>>& ics) {
        if (ics.size() != state.size()) throw std::invalid_argument("setInitialConditions: wrong number of likelihoods");
        initialConditions = ics;
        if (ics[0].size() >= packed().size() && !ics[0].empty()) set_packed(ics[0].data());
        isEvaluated = false;
        eval();
    }
};

/* test.h:296-310 */
class MyState : public State {
public:
    MyState() { state.push_back(std::make_shared<MyLike1>()); init(); }
};

/* ------------------------------------------------------------------------ Target ---- */
/* mcmc.h:705-841 */
class Target {
public:
    virtual ~Target() {}
    void set_posterior(std::shared_ptr<State> s) {
        try {
            if (!s->isInitialized) s->init();
            if (!s->isEvaluated) {
                s->eval();
                std::cout << "Target::set_posterior received unevaluated state and forced evaluation.\n";
            }
            state = s;
            state->weight = weight();
        } catch (...) {
            std::cout << "set_posterior failed. Fix Your setup and try again." << std::endl;
        }
    }
    virtual Float weight() const { return 1; }
    virtual Float beta(Float) { return 1; }
    virtual TargetSpec spec() const { return TargetSpec{}; }
    Float logprobability(Float time) { return std::log(weight()) + beta(time) * state->loglikelihood(); }
    virtual std::shared_ptr<Target> deep_copy() {
        Target t(*this);
        if (state) t.state = state->deep_copy();
        return std::make_shared<Target>(t);
    }
    std::shared_ptr<State> state = nullptr;
};
class TempTarget : public Target {
public:
    TempTarget(Float T = 1) : bet(1 / T), T(T) {}
    Float bet, T;
    Float beta(Float) override { return bet; }
    TargetSpec spec() const override { return TargetSpec{TargetKind::Temp, T, 0, 1}; }
    std::shared_ptr<Target> deep_copy() override {
        TempTarget t(*this);
        if (state) t.state = state->deep_copy();
        return std::make_shared<TempTarget>(t);
    }
};
class CoolingTarget : public Target {
public:
    Float slope, Tinit;
    CoolingTarget(Float coolingSlope, Float Tinit) : slope(coolingSlope), Tinit(Tinit) {}
    Float beta(Float time) override { return std::exp(time * slope) / Tinit; }
    TargetSpec spec() const override { return TargetSpec{TargetKind::Cooling, 1, slope, Tinit}; }
    std::shared_ptr<Target> deep_copy() override {
        CoolingTarget t(*this);
        if (state) t.state = state->deep_copy();
        return std::make_shared<CoolingTarget>(t);
    }
};

/* --------------------------------------------------------------- MetropolisChain ---- */
/* What one chain recorded (MetropolisChain::loglikes / ics / mean, mcmc.h:1307-1310). */
struct ChainRecord {
    std::vector<Float> loglikes;                       /* [nSaved]                              */
    std::vector<std::vector<Float>> ics;               /* [nSaved][nOut] rows of saved points   */
    std::vector<Float> mean;                           /* [nOut]                                */
    std::vector<Float> finalParams;                    /* [dim]                                 */
    Float finalLL[2] = {0, 0};
    Float stepsize = 1;
    int nAccept = 0;

    /* chain `i` of a batched run: its loglikes and saved points are APPENDED, as run() push_backs
     * them (mcmc.h:1039-1043); the mean and the end state replace what was there */
    void take(const BatchResults& R, size_t i, Backend& be) {
        const size_t ns = (size_t)R.nSaved;
        loglikes.insert(loglikes.end(), R.lls.begin() + i * ns, R.lls.begin() + (i + 1) * ns);
        if (R.recorded) {
            const std::vector<Float> pts = be.chain_samples(i);
            for (size_t s = 0; s < ns; ++s) ics.emplace_back(pts.begin() + s * R.nOut, pts.begin() + (s + 1) * R.nOut);
        }
        if (!R.mean.empty()) mean.assign(R.mean.begin() + i * R.nOut, R.mean.begin() + (i + 1) * R.nOut);
        finalParams.assign(R.params.begin() + i * R.dim, R.params.begin() + (i + 1) * R.dim);
        finalLL[0] = R.fll[2 * i]; finalLL[1] = R.fll[2 * i + 1];
        stepsize = R.fss[i];
        nAccept = R.acc[i];
    }
    /* makes `st` the chain's end state */
    void apply_to(State& st) const {
        st.set_packed(finalParams.data());
        st.state[0]->loglike = finalLL[0];
        st.state[0]->stepsizeCorrectionFac = stepsize;
        if (st.prior()) st.prior()->loglike = finalLL[1];
    }
};

/* mcmc.h:984-1312.  run() drives the device for this one chain; a chain handed out by
 * ChainManager::get_chain carries what the manager's batched run recorded for it. */
class MetropolisChain {
public:
    MetropolisChain(std::shared_ptr<Target> target, int chainid) : rnd(0, (uint64_t)chainid), target(target), id(chainid) {}

    Float weight = 1;
    bool computeMean = false;                           /* mcmc.h:990 */
    bool recordSamples = true;
    bool writeSamplesToDisk = false;
    pcg32 rnd;                                          /* seeded ONCE, rnd(0, chainid) (mcmc.h:994,999): a later
                                                           run() goes on drawing where the last one stopped */
    std::shared_ptr<Target> target;
    int id = 0;
    ChainRecord rec;
    bool haveSamples = false;                           /* samples of the reference = ics + late derive */

    void run(int nSamples, int nBurnin, int nAdjust, int thinning) {
        if (!target || !target->state) { std::cout << "No proposal received...\n"; return; }   /* mcmc.h:752-755,1091 */
        std::string why;
        if (!target->state->supported(&why)) { std::cout << "MetropolisChain::run: " << why << ".\n"; return; }
        if (thinning < 1) { std::cout << "MetropolisChain::run: thinning must be at least 1.\n"; return; }
        State& st = *target->state;
        std::unique_ptr<Backend> be = st.backend(target->spec());
        const std::vector<Float> x = st.packed();
        const Float ss = st.state[0]->stepsizeCorrectionFac;
        std::cout << "Starting to adjust and sample.\n";
        BatchResults R;
        /* the saved points are the chain's `ics`, which the reference keeps whatever recordSamples says
         * (mcmc.h:1043); the generator stream and, with computeMean, the running mean carry over from
         * the previous run() of this chain */
        RunOptions opt;
        opt.recordSamples = true;
        opt.computeMean = computeMean;
        const uint64_t stream[2] = {rnd.state, rnd.inc};
        opt.rng = stream;
        if (computeMean && rec.mean.size() == be->n_out()) opt.meanIn = rec.mean.data();
        be->run(R, 1, x.data(), &ss, (uint64_t)id, nSamples, nBurnin, nAdjust, thinning, opt);
        rec.take(R, 0, *be);
        rnd.state = R.rng[0]; rnd.inc = R.rng[1];
        haveSamples = recordSamples || writeSamplesToDisk;
        rec.apply_to(st);
        if (st.disease()) {
            std::vector<Float> curves((size_t)9 * be->curve_len());
            be->derive(1, rec.finalParams.data(), curves.data());
            st.disease()->set_derived(curves.data());
        }
        if (writeSamplesToDisk) write_samples(*be, "chainweight ");                   /* mcmc.h:1018 */
        std::cout << "\nThe final acceptance rate was " << (double)rec.nAccept / std::max(nBurnin + nSamples, 1)
                  << "." << std::endl;                                                        /* mcmc.h:1136 */
    }

    /* mcmc.h:1141-1199: drop the first nBurnin recorded points and (re)build the samples */
    void reevaluate(std::shared_ptr<State> evalstate, int nBurnin) {
        if (!recordSamples && !writeSamplesToDisk) {
            std::cout << "Re-evaluation of chain " << id << " is not requested to record samples in chain nor write them to disk. Aborting.\n";
            return;
        }
        if (rec.ics.size() == 0) {
            std::cout << "Re-evaluation of chain " << id << " impossible: No initial conditions have been recorded.\n";
            return;
        }
        if (haveSamples) std::cout << "Re-evaluation of chain " << id << " despite saved samples. Overwriting... \n";
        if (nBurnin < 0 || (size_t)nBurnin >= rec.ics.size()) {
            std::cout << "Requested re-evaluation with burnin as big as original run or bigger. Aborting.\n";
            return;
        }
        rec.ics.erase(rec.ics.begin(), rec.ics.begin() + nBurnin);
        rec.loglikes.erase(rec.loglikes.begin(), rec.loglikes.begin() + nBurnin);
        if (evalstate && evalstate->supported()) evalState = evalstate;
        haveSamples = true;
        if (writeSamplesToDisk) {
            std::unique_ptr<Backend> be = eval_state()->backend(target->spec());
            write_samples(*be, "chainweight = ");                                            /* mcmc.h:1167 */
        }
    }

    /* getters of the pybind surface (mcmc.h:1236-1301), as plain vectors */
    std::vector<Float> get_mean(const std::string& coordName) {
        if (!computeMean) std::cout << "Requested mean, but MetropolisChain::computeMean is false\n";
        const State* st = eval_state();
        size_t off, len;
        if (!st || rec.mean.empty() || !st->out_range(coordName, off, len)) return {};
        return std::vector<Float>(rec.mean.begin() + off, rec.mean.begin() + off + len);
    }
    /* returns row-major [nSamples][len]; len through *lenOut */
    std::vector<Float> get_samples(const std::string& coordName, size_t* lenOut) {
        *lenOut = 0;
        if (!haveSamples || rec.ics.empty()) { std::cout << "Requested samples, but none are present\n"; return {}; }
        State* st = eval_state();
        size_t off, len;
        std::vector<Float> out;
        if (st->out_range(coordName, off, len)) {
            for (auto& x : rec.ics) out.insert(out.end(), x.begin() + off, x.begin() + off + len);
            *lenOut = len;
            return out;
        }
        const DiseaseSpread* ds = st->disease();
        if (ds) {
            static const char* order[9] = {"mildlyInfectious", "highlyInfectious", "incubating", "asymptomatic", "mild",
                                           "serious", "recovered", "dead", "totalBehavior"};
            const size_t size = ds->nDaysTotal();
            if (coordName == "capacity") {
                auto cap = ds->getCoords("capacity");
                for (size_t i = 0; i < rec.ics.size(); ++i) out.insert(out.end(), cap->begin(), cap->end());
                *lenOut = size;
                return out;
            }
            for (int c = 0; c < 9; ++c) {
                if (coordName != order[c]) continue;
                /* derived output coords: computed now, for the saved samples only (the
                 * compute_derived_late idea, mcmc.h:100-104) */
                std::unique_ptr<Backend> be = st->backend(target->spec());
                const size_t n = rec.ics.size();
                std::vector<Float> flat;
                for (auto& x : rec.ics) flat.insert(flat.end(), x.begin(), x.end());
                std::vector<Float> curves(n * 9 * size);
                be->derive(n, flat.data(), curves.data());
                for (size_t i = 0; i < n; ++i)
                    out.insert(out.end(), curves.begin() + (i * 9 + c) * size, curves.begin() + (i * 9 + c + 1) * size);
                *lenOut = size;
                return out;
            }
        }
        std::cout << "Requested samples of unknown coordinate " << coordName << "\n";
        return {};
    }
    std::vector<Float> get_weights() {
        if (!haveSamples || rec.ics.empty()) { std::cout << "Requested weights, but none are present\n"; return {}; }
        return std::vector<Float>(rec.ics.size(), 1.0);              /* 1/samples[i]->weight, weight() == 1 */
    }
    std::vector<Float> get_loglikes() {
        if (rec.loglikes.empty()) { std::cout << "Requested loglikes, but none are present\n"; return {}; }
        return rec.loglikes;
    }

private:
    std::shared_ptr<State> evalState;

    State* eval_state() const { return evalState ? evalState.get() : (target ? target->state.get() : nullptr); }

    /* text dump in the reference's format (mcmc.h:1013-1019, 1047-1059, 1163-1184) */
    void write_samples(Backend& be, const char* head) {
        std::stringstream filename;
        filename << "samples" << id << ".txt";
        std::ofstream file(filename.str());
        file << head << weight << "\n\n";
        std::shared_ptr<State> work = eval_state()->deep_copy();
        const size_t n = rec.ics.size(), d = be.dim(), no = be.n_out();
        if (n == 0) return;
        std::vector<Float> flat;
        for (auto& x : rec.ics) flat.insert(flat.end(), x.begin(), x.begin() + d);
        std::vector<Float> ll(n * 2), curves;
        be.eval(n, flat.data(), ll.data());
        DiseaseSpread* ds = work->disease();
        if (ds) { curves.resize(n * 9 * be.curve_len()); be.derive(n, flat.data(), curves.data()); }
        (void)no;
        for (size_t i = 0; i < n; ++i) {
            work->set_packed(rec.ics[i].data());
            if (ds) ds->set_derived(curves.data() + i * 9 * be.curve_len());
            file << "weight " << 1.0 << " loglike " << ll[2 * i] + ll[2 * i + 1] << "\n";
            for (auto& [name, vec] : work->get_all()) {
                file << name << " ";
                for (Float x : vec) file << x << " ";
                file << "\n";
            }
            file << "\n";
        }
    }
};

/* ------------------------------------------------------------------ ChainManager ---- */
/* mcmc.h:1314-1655 for ChainType = MetropolisChain.  The per-chain set-up loop and the
 * tbb::parallel_for of run_all / run_all_adjust become one batched device run over all
 * chains; chain i keeps id i+1 (mcmc.h:1487,1527).  The reservoir constructors evaluate
 * their candidates in batches on the device. */
class ChainManager {
public:
    using ICs = std::vector<std::vector<Float>>;

    /* mcmc.h:1317-1347: draw initial conditions until nChain have a finite log probability.
     * (DiseaseSpread and MyLike1 do not implement sampleInitialConditions, so every draw
     * is the state's current point.) */
    ChainManager(std::shared_ptr<Target> target, size_t nChain) : target(target), nChain(nChain) {
        pcg32 rnd(0, 0);
        std::cout << "\nConstructing " << nChain << " random inital states. " << std::endl;
        target->set_posterior(target->state);
        std::vector<std::pair<ICs, Float>> got;
        draw_candidates(rnd, nChain, got);
        for (size_t i = 0; i < got.size(); ++i) {
            std::cout << "Logprobabilty of state " << i << " is " << got[i].second << "\n";
            chainICs.push_back(std::make_pair(got[i].first, 1));
        }
        stepsizes = std::vector<Float>(target->state->state.size(), Float(1));
    }
    /* mcmc.h:1348-1374 */
    ChainManager(std::shared_ptr<Target> target, size_t nChainReservoir, size_t nChain) : target(target), nChain(nChain) {
        pcg32 rnd(0, 0);
        std::cout << "\nConstructing a discrete distribution of " << nChainReservoir << " random inital states. " << std::endl;
        if (!target->state->isInitialized) target->state->init();
        std::vector<std::pair<ICs, Float>> trial;
        draw_candidates(rnd, nChainReservoir, trial);
        stepsizes = std::vector<Float>(target->state->state.size(), Float(1));
        bootstrap(rnd, trial);
    }
    /* mcmc.h:1376-1418: seed the chains from the recorded points of a generator chain,
     * re-weighted to the new target */
    ChainManager(std::shared_ptr<MetropolisChain> generator, std::shared_ptr<Target> targetRHS, size_t nChain)
        : target(targetRHS), nChain(nChain) {
        pcg32 rnd(0, 0);
        if (!generator->recordSamples)
            std::cout << "Re-evaluating chain samples for their likelihood given the current target. This is as many times quicker as the original chain as the thinning.\n";
        if (generator->rec.ics.empty()) throw std::runtime_error("ChainManager: the generator chain holds no recorded points");
        if (!target->state->isInitialized) target->state->init();
        const size_t d = target->state->packed().size();
        std::vector<ICs> cand;
        for (auto& x : generator->rec.ics) {
            if (x.size() < d) throw std::invalid_argument("ChainManager: generator point with wrong dimension");
            ICs ic = {std::vector<Float>(x.begin(), x.begin() + d)};
            for (size_t i = 1; i < target->state->state.size(); ++i) ic.push_back({});
            cand.push_back(ic);
        }
        std::vector<std::pair<ICs, Float>> trial;
        weigh_candidates(cand, generator->rec.loglikes, trial);
        if (trial.empty()) throw std::runtime_error("ChainManager: no candidate has a finite log probability");
        stepsizes = generator->target->state->get_stepsizes();
        bootstrap(rnd, trial);
    }

    std::vector<Float> stepsizes;

    /* user-supplied initial conditions [nChain][dim], each with weight 1: what a caller
     * does who diversifies the chains himself (DiseaseSpread has no way to, see above) */
    void set_initial_conditions(const std::vector<std::vector<Float>>& x) {
        chainICs.clear();
        for (auto& v : x) {
            ICs ic = {v};
            for (size_t i = 1; i < target->state->state.size(); ++i) ic.push_back({});
            chainICs.push_back(std::make_pair(ic, 1));
        }
        nChain = chainICs.size();
    }

    /* mcmc.h:1468-1512 */
    void run_all(size_t nSteps, size_t thinning, Float stepsizefac, int /*nThread*/) {
        for (auto& s : stepsizes) s *= stepsizefac;
        run_impl((int)nSteps, 0, (int)thinning, true);
    }
    /* mcmc.h:1513-1553 */
    void run_all_adjust(size_t nSteps, size_t nAdjust, size_t thinning, int /*nThread*/) {
        run_impl((int)nSteps, (int)nAdjust, (int)thinning, false);
        if (nRun > 0) stepsizes = {R.fss[0]};
        while (stepsizes.size() < target->state->state.size()) stepsizes.push_back(1);
    }
    /* Extension (BASELINE configs[3]; the reference has no replica exchange): the manager's chains as
     * ladders of temps.size() temperatures -- chains [L*K, (L+1)*K) are ladder L, rung r starts on
     * TempTarget(temps[r]) -- advanced by nRounds x (stepsPerRound MH steps + one exchange round, even and
     * odd neighbour pairs alternating).  Exchanges swap temperatures, not states; tempered_rungs() tells
     * which rung every chain ended on (rung 0 of temps[0] = 1 holds the posterior's samples).  Step sizes as
     * in run_all.  One batched device run, like run_all. */
    void run_tempered(const std::vector<Float>& temps, size_t nRounds, size_t stepsPerRound, uint64_t swapSeed) {
        chains.clear();
        nRun = 0;
        lastBackend.reset();
        std::string why;
        if (!target->state || !target->state->supported(&why)) { std::cout << "ChainManager: " << why << ".\n"; return; }
        const size_t n = chainICs.size(), K = temps.size();
        if (K < 2 || n == 0 || n % K != 0) {
            std::cout << "ChainManager::run_tempered: need at least 2 temperatures and a chain count that is a multiple of their number.\n";
            return;
        }
        State& st = *target->state;
        std::unique_ptr<Backend> be = st.backend(target->spec());
        const size_t dim = be->dim();
        std::vector<Float> flat(n * dim);
        for (size_t i = 0; i < n; ++i) {
            if (chainICs[i].first[0].size() != dim) { std::cout << "ChainManager: initial condition " << i << " has the wrong dimension.\n"; return; }
            std::copy(chainICs[i].first[0].begin(), chainICs[i].first[0].end(), flat.begin() + i * dim);
        }
        std::vector<Float> ss(n, stepsizes.empty() ? 1.0 : stepsizes[0]);
        be->run_tempered(R, n, flat.data(), ss.data(), 1, temps, swapSeed, (int)nRounds, (int)stepsPerRound, rungs,
                         swapAccepted, swapProposed);
        last_diagnostics = R.diagnostics;
        lastBackend = std::move(be);
        nRun = n;
        std::cout << "Tempered chain runs completed." << std::endl << std::endl;
    }
    const std::vector<int32_t>& tempered_rungs() const { return rungs; }
    const std::vector<int64_t>& swaps_accepted() const { return swapAccepted; }
    const std::vector<int64_t>& swaps_proposed() const { return swapProposed; }

    /* mcmc.h:1555-1565 (the reference's bare `throw;` terminates; an exception is kinder) */
    MetropolisChain& get_chain(size_t i) {
        if (nRun == 0) {
            std::cout << "ChainManager::get_chain: Requested chain " << i << "of the manager, but no valid chain is present. call ChainManager::run_chains first!\n";
            throw std::out_of_range("ChainManager::get_chain: no chains");
        }
        if (nRun <= i) {
            std::cout << "ChainManager::get_chain: Requested chain " << i << "of the manager, but only up to " << nRun << " chains are present. Valid indices up to " << nRun - 1 << " only!\n";
            throw std::out_of_range("ChainManager::get_chain: index");
        }
        return materialize(i);
    }
    /* mcmc.h:1574-1596 */
    void reevaluate_all(std::shared_ptr<State> state, int nBurnin, bool recordSamples, bool writeSamplesToDisk, int /*nThread*/) {
        for (size_t i = 0; i < nRun; ++i) {
            std::cout << "Reevaluating chain with id = " << i << "." << std::endl;
            MetropolisChain& c = materialize(i);
            c.recordSamples = recordSamples;
            c.writeSamplesToDisk = writeSamplesToDisk;
            c.reevaluate(state->deep_copy(), nBurnin);
        }
        std::cout << "Reevaluation completed." << std::endl << std::endl;
    }
    size_t n_chains() const { return chainICs.size(); }
    /* Every chain keeps the points it saves (its `ics`, mcmc.h:1043) on the device, and they are only
     * copied to the host for a chain that is asked for (get_chain, reevaluate_all, saved_sample).
     * false: record the log-likelihood trace only -- for very large ensembles whose points nobody reads
     * (no reevaluate_all afterwards). */
    bool recordInitialConditions = true;
    /* bulk views of the last run, chain-major: loglikes [n][nSaved], saved sample s [n][nOut] */
    const std::vector<Float>& all_loglikes() const { return R.lls; }
    const std::vector<Float>& all_final_params() const { return R.params; }
    /* saved point s of every chain, fetched from the device on request */
    std::vector<Float> saved_sample(int s) const {
        if (!lastBackend || !R.recorded) throw std::out_of_range("ChainManager::saved_sample: the last run recorded no points");
        return lastBackend->sample_all(s);
    }
    int n_saved() const { return R.nSaved; }
    int n_dim() const { return (int)R.nOut; }
    int chain_weight(size_t i) const { return chainICs.at(i).second; }
    /* acceptance counts, sums ... of the last run (mcmc_hip_diagnostics), covid states only */
    std::vector<Float> last_diagnostics;

protected:
    std::shared_ptr<Target> target;
    size_t nChain = 1;
    std::vector<std::pair<ICs, int>> chainICs;

    /* log p_target(candidate) - oldlogprob for a batch of candidates, finite ones kept in order */
    void weigh_candidates(const std::vector<ICs>& cand, const std::vector<Float>& oldlogprob,
                          std::vector<std::pair<ICs, Float>>& out) {
        std::unique_ptr<Backend> be = target->state->backend(target->spec());
        const size_t n = cand.size(), dim = be->dim();
        std::vector<Float> flat(n * dim), ll(n * 2);
        for (size_t i = 0; i < n; ++i) {
            if (cand[i][0].size() != dim) throw std::invalid_argument("ChainManager: candidate with wrong dimension");
            std::copy(cand[i][0].begin(), cand[i][0].end(), flat.begin() + i * dim);
        }
        be->eval(n, flat.data(), ll.data());
        for (size_t i = 0; i < n; ++i) {
            const Float prob = std::log(target->weight()) + target->beta(0) * (ll[2 * i] + ll[2 * i + 1]);
            if (!std::isnan(prob) && !std::isinf(prob)) out.push_back(std::make_pair(cand[i], prob - oldlogprob[i]));
        }
    }
    /* the `while (size < want) { sampleInitialConditions; eval; keep if finite }` loops of
     * mcmc.h:1324-1343,1356-1371: the draws come from rnd in the reference's order and are
     * evaluated in batches */
    void draw_candidates(pcg32& rnd, size_t want, std::vector<std::pair<ICs, Float>>& out) {
        State& st = *target->state;
        ToyLikelihood* t = st.toy();
        DiseaseSpread* ds = st.disease();
        const bool draws = t || (ds && ds->initialSpread > 0);
        size_t rounds = 0;
        while (out.size() < want) {
            const size_t need = want - out.size();
            std::vector<ICs> cand;
            for (size_t i = 0; i < need; ++i) {
                if (t) t->sampleInitialConditions(rnd);
                else if (ds) ds->sampleInitialPoint(rnd);
                ICs ic = {st.packed()};
                for (size_t k = 1; k < st.state.size(); ++k) ic.push_back({});
                cand.push_back(ic);
            }
            const size_t before = out.size();
            weigh_candidates(cand, std::vector<Float>(need, 0.0), out);
            /* a likelihood that cannot draw gives the same point every time: the
             * reference would loop forever on a non-finite one */
            if (out.size() == before && (!draws || ++rounds > 1000))
                throw std::runtime_error("ChainManager: the state's log probability is not finite, no initial state can be drawn");
        }
        if (draws) st.setInitialConditions(out.back().first);
    }

    /* mcmc.h:1599-1650: importance resampling of the candidates.  Candidate k carries the log weight
     * lp_k (log target density minus the density it was drawn from); nChain DISTINCT candidates are
     * drawn with probability proportional to w_k = exp(lp_k - max lp) -- the best one at w = 1, so that
     * log weights of order -1e7 do not all underflow -- and a candidate drawn again raises the integer
     * weight of the chain it already seeds.  Draw order, arithmetic and messages follow the reference:
     * one nextDouble() per draw, scaled by the running total, located in the running sums by
     * upper_bound, the last sum lifted to 1.000001 x total against rounding. */
    void bootstrap(pcg32& rnd, std::vector<std::pair<ICs, Float>>& cand) {
        const size_t m = cand.size();
        Float top = -1e20;                                   /* the reference's floor for the maximum */
        for (const auto& c : cand) top = std::max(top, c.second);
        std::cout << "Selecting " << nChain << " chains according to the target density." << std::endl;
        std::cout << "Highest log probability in set: " << top << "\n";
        std::vector<double> w(m), upTo(m);
        for (size_t k = 0; k < m; ++k) w[k] = std::exp(cand[k].second - top);
        const double total = std::accumulate(w.begin(), w.end(), 0.0);          /* left to right, like the reference */
        std::partial_sum(w.begin(), w.end(), upTo.begin());
        upTo.back() = 1.000001 * total;
        std::vector<long> seeds(m, -1);                      /* candidate -> the chain it seeds, -1: none yet */
        /* the reference draws until nChain distinct candidates are found, which never ends when fewer
         * exist: stop after a bounded number of draws instead */
        const size_t maxDraws = 1000 * std::max<size_t>(nChain, 1) + 1000;
        size_t found = 0;
        for (size_t draws = 0; found < nChain && draws < maxDraws; ++draws) {
            const double x = rnd.nextDouble() * total;
            const size_t k = std::min<size_t>(std::upper_bound(upTo.begin(), upTo.end(), x) - upTo.begin(), m - 1);
            std::cout << "Relative loglikelihood of chain drawn is " << std::log(w[k]) << "\n";
            if (seeds[k] < 0) {
                seeds[k] = (long)chainICs.size();
                chainICs.emplace_back(cand[k].first, 1);
                ++found;
            } else {
                auto& chain = chainICs[(size_t)seeds[k]];
                std::cout << "Note: same chain " << seeds[k] << " that has been picked " << chain.second
                          << " times was picked again. New chain weight: " << chain.second + 1 << std::endl;
                ++chain.second;
            }
        }
        std::cout << "Selection completed." << std::endl << std::endl;
    }

    /* what the last batched run recorded, for all chains; a MetropolisChain object (with its
     * own deep-copied Target, mcmc.h:1479,1520) is only built when somebody asks for it */
    BatchResults R;
    std::unique_ptr<Backend> lastBackend;                /* holds the saved points of the last run */
    size_t nRun = 0;
    std::map<size_t, MetropolisChain> chains;
    std::vector<int32_t> rungs;
    std::vector<int64_t> swapAccepted, swapProposed;

    MetropolisChain& materialize(size_t i) {
        auto it = chains.find(i);
        if (it != chains.end()) return it->second;
        std::shared_ptr<Target> tar = target->deep_copy();
        MetropolisChain c(tar, (int)i + 1);
        c.recordSamples = false;
        c.weight = chainICs[i].second;
        c.rec.take(R, i, *lastBackend);
        if (R.rng.size() >= 2 * (i + 1)) { c.rnd.state = R.rng[2 * i]; c.rnd.inc = R.rng[2 * i + 1]; }
        c.rec.apply_to(*tar->state);
        return chains.emplace(i, std::move(c)).first->second;
    }

    void run_impl(int nSteps, int nAdjust, int thinning, bool useStepsizes) {
        chains.clear();
        nRun = 0;
        lastBackend.reset();
        std::string why;
        if (!target->state || !target->state->supported(&why)) { std::cout << "ChainManager: " << why << ".\n"; return; }
        if (thinning < 1) { std::cout << "ChainManager: thinning must be at least 1.\n"; return; }
        const size_t n = chainICs.size();
        if (n == 0) { std::cout << "ChainManager: no initial conditions.\n"; return; }
        State& st = *target->state;
        std::unique_ptr<Backend> be = st.backend(target->spec());
        const size_t dim = be->dim();
        std::vector<Float> flat(n * dim);
        for (size_t i = 0; i < n; ++i) {
            if (chainICs[i].first[0].size() != dim) { std::cout << "ChainManager: initial condition " << i << " has the wrong dimension.\n"; return; }
            std::copy(chainICs[i].first[0].begin(), chainICs[i].first[0].end(), flat.begin() + i * dim);
        }
        /* run_all applies the manager's step sizes, run_all_adjust starts from the target
         * state's own (mcmc.h:1483 vs 1523) */
        const Float ss0 = useStepsizes ? (stepsizes.empty() ? 1.0 : stepsizes[0]) : st.state[0]->stepsizeCorrectionFac;
        std::vector<Float> ss(n, ss0);
        {
            std::vector<Float> ll(n * 2);
            be->eval(n, flat.data(), ll.data());
            for (size_t i = 0; i < n; ++i)
                std::cout << "Initialized chain " << i << " with true loglikelihood = " << ll[2 * i] + ll[2 * i + 1] << "\n";
        }
        /* the manager's chains: recordSamples = false (mcmc.h:1490,1530), computeMean at its default
         * false (mcmc.h:990), each a fresh MetropolisChain(tar, i+1) seeding rnd(0, i+1) (mcmc.h:1487,1527) */
        RunOptions opt;
        opt.recordSamples = recordInitialConditions;
        opt.computeMean = false;
        be->run(R, n, flat.data(), ss.data(), 1, nSteps, 0, nAdjust, thinning, opt);
        lastBackend = std::move(be);
        last_diagnostics = R.diagnostics;
        nRun = n;
        std::cout << "Chain runs completed." << std::endl << std::endl;
    }
};

}  // namespace mcmc_amd
