/*
 * mcmc_module.cpp -- pybind11 module `mcmc`: the Python surface of AndreasFinke/mcmc for
 * the covid path (test.cpp:46-141 plus the covid classes the notebooks use,
 * notebooks/covid.ipynb cells 3-4), over the C++ host side of mcmc_amd/host/mcmc_host.hpp,
 * which calls the HIP kernels through the C ABI (include/mcmc_hip.h).
 *
 * Names, argument orders and quirks are the reference's: `Chain` is MetropolisChain,
 * `ChainManager.run_all` is bound to run_all_adjust and `run_all_adjust` to run_all
 * (test.cpp:124-125), `SubspaceState.getCoords` is getCoordsAt (test.cpp:51).
 * The toy targets of test.h that BASELINE configs[0] names are here too: FourGaussians
 * (test.cpp:63-64), MyState (test.cpp:79-81) and its MyLike1.  Classes of the reference off
 * that path (A-D, the distfind.h PDFs, AdvCoolingTarget, GradientDecent) are not part of
 * this module.
 */
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "mcmc_host.hpp"

namespace py = pybind11;
using namespace mcmc_amd;

static py::array_t<Float> to_array(const std::vector<Float>& v) {
    py::array_t<Float> a(v.size());
    std::copy(v.begin(), v.end(), a.mutable_data());
    return a;
}

PYBIND11_MODULE(mcmc, m) {
    m.doc() = "MI355X-native Metropolis-Hastings sampler with the Python surface of AndreasFinke/mcmc (covid path)";
    py::register_exception<HipError>(m, "McmcHipError");

    py::class_<SubspaceState, std::shared_ptr<SubspaceState>>(m, "SubspaceState")
        .def("get_names", &SubspaceState::get_names)
        .def("setCoords", &SubspaceState::setCoords)
        .def("getCoords", [](SubspaceState& s, const std::string& n) { return s.getCoordsAt(n); })
        .def_readwrite("loglike", &SubspaceState::loglike)
        .def_readwrite("stepsizeCorrectionFac", &SubspaceState::stepsizeCorrectionFac);
    py::class_<SmoothnessPrior, SubspaceState, std::shared_ptr<SmoothnessPrior>>(m, "SmoothnessPrior")
        .def(py::init<const std::string&, Float, Float>());

    py::class_<MyLike1, SubspaceState, std::shared_ptr<MyLike1>>(m, "MyLike1")
        .def(py::init<>());
    py::class_<FourGaussians, SubspaceState, std::shared_ptr<FourGaussians>>(m, "FourGaussians")
        .def(py::init<Float>());
    py::class_<Rosenbrock, SubspaceState, std::shared_ptr<Rosenbrock>>(m, "Rosenbrock")     /* extension, configs[0] */
        .def(py::init<Float>(), py::arg("difficulty") = 100.0);

    py::class_<DiseaseParams>(m, "DiseaseParams")
        .def(py::init<>())
        .def_readwrite("timeIncub", &DiseaseParams::timeIncub)
        .def_readwrite("timeIncubSigma", &DiseaseParams::timeIncubSigma)
        .def_readwrite("probAsymp", &DiseaseParams::probAsymp)
        .def_readwrite("probSerious", &DiseaseParams::probSerious)
        .def_readwrite("probLethal", &DiseaseParams::probLethal)
        .def_readwrite("probLethalDailyWhenSeriousUntreated", &DiseaseParams::probLethalDailyWhenSeriousUntreated)
        .def_readwrite("probICUIfSerious", &DiseaseParams::probICUIfSerious)
        .def_readwrite("timeMildDuration", &DiseaseParams::timeMildDuration)
        .def_readwrite("timeMildDurationSigma", &DiseaseParams::timeMildDurationSigma)
        .def_readwrite("timeMildToSerious", &DiseaseParams::timeMildToSerious)
        .def_readwrite("timeMildToSeriousSigma", &DiseaseParams::timeMildToSeriousSigma)
        .def_readwrite("timeSeriousToRec", &DiseaseParams::timeSeriousToRec)
        .def_readwrite("timeSeriousToRecSigma", &DiseaseParams::timeSeriousToRecSigma)
        /* notebooks/covid.ipynb cell 3 spells these two `timeSeriousToRecovered[Sigma]` */
        .def_readwrite("timeSeriousToRecovered", &DiseaseParams::timeSeriousToRec)
        .def_readwrite("timeSeriousToRecoveredSigma", &DiseaseParams::timeSeriousToRecSigma)
        .def_readwrite("timeSeriousToDeath", &DiseaseParams::timeSeriousToDeath)
        .def_readwrite("timeSeriousToDeathSigma", &DiseaseParams::timeSeriousToDeathSigma);

    py::class_<DiseaseData>(m, "DiseaseData")
        .def(py::init([](py::array_t<Float, py::array::c_style | py::array::forcecast> deaths,
                         py::array_t<Float, py::array::c_style | py::array::forcecast> dis) {
                 if (deaths.ndim() != 2 || deaths.shape(0) != 2)
                     throw std::invalid_argument("DiseaseData: deathsPerDayAndSigma must have shape [2, nData]");
                 if (dis.ndim() != 2 || dis.shape(0) != 3)
                     throw std::invalid_argument("DiseaseData: discontinuousDaysAndVals must have shape [3, nDisc]");
                 return new DiseaseData(deaths.data(), (size_t)deaths.shape(1), dis.data(), (size_t)dis.shape(1));
             }))
        .def_readwrite("initialBetaMild", &DiseaseData::initialBetaMild)
        .def_readwrite("initialBetaHigh", &DiseaseData::initialBetaHigh)
        .def_readwrite("initialDelay", &DiseaseData::initialDelay)
        .def_readwrite("fixBehaviorInAdvance", &DiseaseData::fixBehaviorInAdvance);

    /* DiseaseSpread holds references to its data and params (covid.h:288-289): keep them alive */
    py::class_<DiseaseSpread, SubspaceState, std::shared_ptr<DiseaseSpread>>(m, "DiseaseSpread")
        .def(py::init<const DiseaseData&, const DiseaseParams&, int, double, double, int, size_t>(),
             py::keep_alive<1, 2>(), py::keep_alive<1, 3>())
        .def(py::init<const DiseaseData&, const DiseaseParams&, int, double, double, int, size_t, uint64_t>(),
             py::keep_alive<1, 2>(), py::keep_alive<1, 3>())
        .def_readwrite("fastMode", &DiseaseSpread::fastMode)       /* extension: FP32 convolution on the device */
        .def_readwrite("initialSpread", &DiseaseSpread::initialSpread)   /* extension: 0 = the reference (one point) */
        .def("response_table", [](const DiseaseSpread& d) {
            py::array_t<Float> a({(size_t)MCMC_HIP_NCURVES, (size_t)MCMC_HIP_TRAJ_NDAYS});
            std::copy(d.responseTable->begin(), d.responseTable->end(), a.mutable_data());
            return a;
        })
        .def("set_response_table", [](DiseaseSpread& d, py::array_t<Float, py::array::c_style | py::array::forcecast> t) {
            if (t.ndim() != 2 || t.shape(0) != MCMC_HIP_NCURVES || t.shape(1) != MCMC_HIP_TRAJ_NDAYS)
                throw std::invalid_argument("set_response_table: need shape [10, 500]");
            d.responseTable = std::make_shared<std::vector<double>>(t.data(), t.data() + t.size());
        });

    py::class_<State, std::shared_ptr<State>>(m, "State")
        .def(py::init<>())
        .def("add", &State::add)
        .def("init", &State::init)
        .def("loglike", &State::loglikelihood)
        .def("force_bounds", &State::force_bounds)
        .def("get_all", &State::get_all)
        .def_readwrite("sharedDependencyMaxDepth", &State::sharedDependencyMaxDepth);

    py::class_<MyState, State, std::shared_ptr<MyState>>(m, "MyState")
        .def(py::init<>());

    py::class_<Target, std::shared_ptr<Target>>(m, "Target")
        .def(py::init<>())
        .def("set_posterior", &Target::set_posterior);
    py::class_<TempTarget, Target, std::shared_ptr<TempTarget>>(m, "TempTarget")
        .def(py::init<Float>());
    py::class_<CoolingTarget, Target, std::shared_ptr<CoolingTarget>>(m, "CoolingTarget")
        .def(py::init<Float, Float>());

    py::class_<MetropolisChain, std::shared_ptr<MetropolisChain>>(m, "Chain")
        .def(py::init<std::shared_ptr<Target>, int>())
        .def("run", &MetropolisChain::run, py::call_guard<py::gil_scoped_release>())
        .def("get_mean", [](MetropolisChain& c, const std::string& n) { return to_array(c.get_mean(n)); })
        .def("get_samples", [](MetropolisChain& c, const std::string& n) {
            size_t len = 0;
            std::vector<Float> v = c.get_samples(n, &len);
            if (len == 0) return py::array_t<Float>();
            py::array_t<Float> a({v.size() / len, len});
            std::copy(v.begin(), v.end(), a.mutable_data());
            return a;
        })
        .def("get_weights", [](MetropolisChain& c) { return to_array(c.get_weights()); })
        .def("get_loglikes", [](MetropolisChain& c) { return to_array(c.get_loglikes()); })
        .def("reevaluate", &MetropolisChain::reevaluate)
        .def_readwrite("weight", &MetropolisChain::weight)
        .def_readwrite("computeMean", &MetropolisChain::computeMean)
        .def_readwrite("recordSamples", &MetropolisChain::recordSamples)
        .def_readwrite("writeSamplesToDisk", &MetropolisChain::writeSamplesToDisk);

    py::class_<ChainManager>(m, "ChainManager")
        .def(py::init<std::shared_ptr<Target>, size_t, size_t>())
        .def(py::init<std::shared_ptr<MetropolisChain>, std::shared_ptr<Target>, size_t>())
        .def(py::init<std::shared_ptr<Target>, size_t>())
        /* cross-wired exactly as test.cpp:124-125 */
        .def("run_all", &ChainManager::run_all_adjust, py::call_guard<py::gil_scoped_release>())
        .def("run_all_adjust", &ChainManager::run_all, py::call_guard<py::gil_scoped_release>())
        .def("reevaluate_all", &ChainManager::reevaluate_all)
        .def("get_chain", &ChainManager::get_chain, py::return_value_policy::reference_internal)
        /* not in the reference: bulk access for many chains and caller-chosen starts */
        .def("set_initial_conditions", &ChainManager::set_initial_conditions)
        /* tempering ladders (extension): temps, rounds, MH steps per round, seed of the exchange draws */
        .def("run_tempered", &ChainManager::run_tempered, py::call_guard<py::gil_scoped_release>(),
             py::arg("temps"), py::arg("n_rounds"), py::arg("steps_per_round"), py::arg("swap_seed") = 1)
        .def("tempered_rungs", [](const ChainManager& c) { return std::vector<int>(c.tempered_rungs().begin(), c.tempered_rungs().end()); })
        .def("swap_counts", [](const ChainManager& c) {
            return std::make_pair(std::vector<long long>(c.swaps_accepted().begin(), c.swaps_accepted().end()),
                                  std::vector<long long>(c.swaps_proposed().begin(), c.swaps_proposed().end()));
        })
        .def("all_final_params", [](const ChainManager& c) {
            py::array_t<Float> a({c.n_chains(), (size_t)c.n_dim()});
            std::copy(c.all_final_params().begin(), c.all_final_params().end(), a.mutable_data());
            return a;
        })
        .def("n_chains", &ChainManager::n_chains)
        .def("chain_weight", &ChainManager::chain_weight)
        .def("all_loglikes", [](const ChainManager& c) {
            const size_t n = c.n_chains();
            py::array_t<Float> a({n, (size_t)c.n_saved()});
            std::copy(c.all_loglikes().begin(), c.all_loglikes().begin() + n * c.n_saved(), a.mutable_data());
            return a;
        })
        .def("saved_sample", [](const ChainManager& c, int s) {
            py::array_t<Float> a({c.n_chains(), (size_t)c.n_dim()});
            const std::vector<Float> pts = c.saved_sample(s);
            std::copy(pts.begin(), pts.end(), a.mutable_data());
            return a;
        })
        .def_readwrite("recordInitialConditions", &ChainManager::recordInitialConditions)
        .def_readwrite("stepsizes", &ChainManager::stepsizes)
        .def_readonly("last_diagnostics", &ChainManager::last_diagnostics);
}
