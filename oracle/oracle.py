"""ctypes binding of the CPU ORACLE (oracle/mcmc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under mcmc_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmcmc_oracle.so")

NGRID, NDAYS, NCURVES = 100000, 500, 11
CURVE_NAMES = ["incubating", "asymptomatic", "mild", "infectiousMild", "infectiousHigh",
               "serious", "dead", "recovered", "seriousFromSerious", "recoveredFromSerious",
               "deadFromSerious"]
OUT_CURVES = ["mildlyInfectious", "highlyInfectious", "incubating", "asymptomatic", "mild",
              "serious", "recovered", "dead", "totalBehavior"]


def build(force=False):
    src = os.path.join(_HERE, "mcmc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class Pcg32(C.Structure):
    _fields_ = [("state", C.c_uint64), ("inc", C.c_uint64)]


class DiseaseParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "timeIncub", "timeIncubSigma", "probAsymp", "probSerious", "probLethal",
        "probLethalDailyWhenSeriousUntreated", "probICUIfSerious", "timeMildDuration",
        "timeMildDurationSigma", "timeMildToSerious", "timeMildToSeriousSigma",
        "timeSeriousToRec", "timeSeriousToRecSigma", "timeSeriousToDeath",
        "timeSeriousToDeathSigma")]


class CovidModel(C.Structure):
    _fields_ = [
        ("n_data", C.c_int), ("max_delay", C.c_int), ("n_predict", C.c_int), ("n_disc", C.c_int),
        ("fix_behavior_in_advance", C.c_int), ("pop_size", C.c_int),
        ("prob_icu_if_serious", C.c_double), ("prob_lethal_daily_untreated", C.c_double),
        ("cap0", C.c_double), ("cap_incr_rate", C.c_double),
        ("has_prior", C.c_int), ("smooth_Ls", C.c_double), ("smooth_L", C.c_double),
        ("deaths", C.POINTER(C.c_double)), ("sigma", C.POINTER(C.c_double)),
        ("disc_days", C.POINTER(C.c_int)), ("disc_fixed", C.POINTER(C.c_ubyte)),
        ("R", C.POINTER(C.c_double)), ("n_lag", C.c_int),
        ("grid", C.POINTER(C.c_double)),
        ("contract", C.c_int), ("use_libm", C.c_int), ("faithful", C.c_int),
    ]


class Target(C.Structure):
    _fields_ = [("kind", C.c_int), ("T", C.c_double), ("slope", C.c_double), ("Tinit", C.c_double)]


class Chain(C.Structure):
    _fields_ = [("rng", Pcg32), ("x", C.POINTER(C.c_double)), ("xp", C.POINTER(C.c_double)),
                ("ll_model", C.c_double), ("ll_prior", C.c_double), ("stepsize", C.c_double),
                ("n_accept", C.c_int), ("n_steps", C.c_uint64), ("last_delta", C.c_double),
                ("last_accept", C.c_int), ("grid_own", C.POINTER(C.c_double))]


class MultiChainC(C.Structure):
    _fields_ = [("rng", Pcg32), ("x", C.POINTER(C.c_double)), ("xp", C.POINTER(C.c_double)),
                ("ll", C.POINTER(C.c_double)), ("llp", C.POINTER(C.c_double)), ("stepsize", C.POINTER(C.c_double)),
                ("n_accept", C.c_int), ("n_steps", C.c_uint64), ("head_counter", C.c_uint64), ("head", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    dp = C.POINTER(C.c_double)
    L.orc_pcg32_default.argtypes = [C.POINTER(Pcg32)]
    L.orc_pcg32_seed.argtypes = [C.POINTER(Pcg32), C.c_uint64, C.c_uint64]
    L.orc_pcg32_next_uint.argtypes = [C.POINTER(Pcg32)]
    L.orc_pcg32_next_uint.restype = C.c_uint32
    L.orc_pcg32_next_float.argtypes = [C.POINTER(Pcg32)]
    L.orc_pcg32_next_float.restype = C.c_float
    L.orc_pcg32_next_double.argtypes = [C.POINTER(Pcg32)]
    L.orc_pcg32_next_double.restype = C.c_double
    L.orc_bound.argtypes = [C.c_double] * 3
    L.orc_bound.restype = C.c_double
    L.orc_erfinv.argtypes = [C.c_double]
    L.orc_erfinv.restype = C.c_double
    L.orc_to_normal.argtypes = [C.c_double]
    L.orc_to_normal.restype = C.c_double
    L.orc_disease_params_default.argtypes = [C.POINTER(DiseaseParams)]
    L.orc_traj_generate.argtypes = [C.POINTER(DiseaseParams), C.c_uint64, dp]
    L.orc_traj_get_day.argtypes = [dp, C.c_size_t]
    L.orc_traj_get_day.restype = C.c_double
    L.orc_traj_response.argtypes = [dp, C.c_int, dp]
    L.orc_covid_eval.argtypes = [C.POINTER(CovidModel), dp, dp, C.c_int]
    L.orc_covid_eval.restype = C.c_double
    L.orc_smooth_eval.argtypes = [dp, C.c_int, C.c_double, C.c_double]
    L.orc_smooth_eval.restype = C.c_double
    L.orc_covid_propose.argtypes = [C.POINTER(CovidModel), C.POINTER(Pcg32), dp, C.c_double, dp]
    L.orc_covid_force_bounds.argtypes = [C.POINTER(CovidModel), dp]
    L.orc_target_beta.argtypes = [C.POINTER(Target), C.c_double]
    L.orc_target_beta.restype = C.c_double
    L.orc_chain_init.argtypes = [C.POINTER(CovidModel), C.POINTER(Chain), C.c_int, dp, C.c_double]
    L.orc_chain_free.argtypes = [C.POINTER(Chain)]
    L.orc_chain_step.argtypes = [C.POINTER(CovidModel), C.POINTER(Target), C.POINTER(Chain),
                                 C.c_double, C.c_int]
    L.orc_chain_step.restype = C.c_int
    L.orc_chain_run.argtypes = [C.POINTER(CovidModel), C.POINTER(Target), C.POINTER(Chain),
                                C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, dp]
    L.orc_chain_run.restype = C.c_int
    L.orc_n_saved.argtypes = [C.c_int] * 4
    L.orc_n_saved.restype = C.c_int
    L.orc_run_chains.argtypes = [C.POINTER(CovidModel), C.POINTER(Target), C.c_int, C.c_int, dp, dp,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp,
                                 C.POINTER(C.c_int), C.POINTER(C.c_uint64), dp]
    L.orc_run_chains.restype = C.c_double
    L.orc_multi_init.argtypes = [C.POINTER(CovidModel), C.c_int, C.POINTER(MultiChainC), C.c_int, dp, C.c_double]
    L.orc_multi_free.argtypes = [C.POINTER(MultiChainC)]
    L.orc_multi_step.argtypes = [C.POINTER(CovidModel), C.c_int, C.POINTER(Target), C.POINTER(MultiChainC), C.c_double, C.c_int]
    L.orc_multi_step.restype = C.c_int
    L.orc_multi_run.argtypes = [C.POINTER(CovidModel), C.c_int, C.POINTER(Target), C.POINTER(MultiChainC)] + [C.c_int] * 4 + [dp, dp, dp]
    L.orc_multi_run.restype = C.c_int
    L.orc_mylike1_run.argtypes = [C.c_int] * 5 + [dp, dp, dp, C.POINTER(C.c_int), dp]
    L.orc_mylike1_run.restype = C.c_int
    L.orc_fourgauss_run.argtypes = [C.c_double, C.c_double, C.c_int, dp] + [C.c_int] * 4 + \
                                   [dp, dp, dp, C.POINTER(C.c_int), dp, C.c_double]
    L.orc_fourgauss_run.restype = C.c_int
    L.orc_rosenbrock_run.argtypes = L.orc_fourgauss_run.argtypes
    L.orc_rosenbrock_run.restype = C.c_int
    L.orc_rosenbrock_eval.argtypes = [C.c_double, dp]
    L.orc_rosenbrock_eval.restype = C.c_double
    L.orc_det_cos.argtypes = [C.c_double]
    L.orc_det_cos.restype = C.c_double
    L.orc_det_exp.argtypes = [C.c_double]
    L.orc_det_exp.restype = C.c_double
    L.orc_det_log.argtypes = [C.c_double]
    L.orc_det_log.restype = C.c_double
    L.orc_mylike1_eval.argtypes = [dp]
    L.orc_mylike1_eval.restype = C.c_double
    L.orc_fourgauss_eval.argtypes = [C.c_double, dp]
    L.orc_fourgauss_eval.restype = C.c_double
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


class Rng:
    """pcg32 as the reference constructs it: Rng() default, Rng(initstate, initseq)."""

    def __init__(self, initstate=None, initseq=1):
        self.r = Pcg32()
        if initstate is None:
            lib().orc_pcg32_default(C.byref(self.r))
        else:
            lib().orc_pcg32_seed(C.byref(self.r), initstate, initseq)

    def next_uint(self):
        return lib().orc_pcg32_next_uint(C.byref(self.r))

    def next_float(self):
        return lib().orc_pcg32_next_float(C.byref(self.r))

    def next_double(self):
        return lib().orc_pcg32_next_double(C.byref(self.r))

    @property
    def state(self):
        return (self.r.state, self.r.inc)


def traj_generate(params: DiseaseParams, n_traj=10_000_000):
    grid = np.zeros((NCURVES, NGRID), dtype=np.float64)
    lib().orc_traj_generate(C.byref(params), n_traj, _dp(grid))
    return grid


def traj_response(grid, n_lag=NDAYS):
    R = np.zeros((NCURVES, n_lag), dtype=np.float64)
    lib().orc_traj_response(_dp(np.ascontiguousarray(grid)), n_lag, _dp(R))
    return R


class Model:
    """Owns the numpy buffers a CovidModel points into."""

    def __init__(self, cfg, R, grid=None, contract=0, use_libm=0, faithful=0):
        """cfg: a dict as produced by mcmc_amd.covid_config (plain host data)."""
        self.cfg = cfg
        self.deaths = np.ascontiguousarray(cfg["deaths"], dtype=np.float64)
        self.sigma = np.ascontiguousarray(cfg["sigma"], dtype=np.float64)
        self.disc_days = np.ascontiguousarray(cfg["disc_days"], dtype=np.int32)
        self.disc_fixed = np.ascontiguousarray(cfg["disc_fixed"], dtype=np.uint8)
        self.R = np.ascontiguousarray(R, dtype=np.float64)
        assert self.R.shape[0] == NCURVES
        self.grid = None if grid is None else np.ascontiguousarray(grid, dtype=np.float64)
        m = CovidModel()
        m.n_data = int(cfg["n_data"]); m.max_delay = int(cfg["max_delay"])
        m.n_predict = int(cfg["n_predict"]); m.n_disc = len(self.disc_days)
        m.fix_behavior_in_advance = int(cfg["fix_behavior_in_advance"])
        m.pop_size = int(cfg["pop_size"])
        m.prob_icu_if_serious = float(cfg["prob_icu_if_serious"])
        m.prob_lethal_daily_untreated = float(cfg["prob_lethal_daily_untreated"])
        m.cap0 = float(cfg["cap0"]); m.cap_incr_rate = float(cfg["cap_incr_rate"])
        m.has_prior = int(cfg.get("has_prior", 1))
        m.smooth_Ls = float(cfg.get("smooth_Ls", 50.0))
        m.smooth_L = float(cfg.get("smooth_L", m.n_data - m.fix_behavior_in_advance))
        m.deaths = _dp(self.deaths); m.sigma = _dp(self.sigma)
        m.disc_days = self.disc_days.ctypes.data_as(C.POINTER(C.c_int))
        m.disc_fixed = self.disc_fixed.ctypes.data_as(C.POINTER(C.c_ubyte))
        m.R = _dp(self.R); m.n_lag = self.R.shape[1]
        m.grid = _dp(self.grid) if self.grid is not None else None
        m.contract = contract; m.use_libm = use_libm; m.faithful = faithful
        self.m = m
        self.size = m.max_delay + m.n_predict + m.n_data
        self.nB = m.n_data - m.fix_behavior_in_advance
        self.dim = self.nB + m.n_disc + 3

    def eval(self, x, curves=False, sim_days=0):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.shape == (self.dim,)
        out = np.zeros((9, self.size)) if curves else None
        ll = lib().orc_covid_eval(C.byref(self.m), _dp(x), _dp(out), sim_days)
        return (ll, out) if curves else ll

    def prior(self, x):
        f = np.ascontiguousarray(x[:self.nB], dtype=np.float64)
        return lib().orc_smooth_eval(_dp(f), self.nB, self.m.smooth_Ls, self.m.smooth_L)

    def propose(self, rng: Rng, x, stepsize):
        x = np.ascontiguousarray(x, dtype=np.float64)
        xp = np.empty_like(x)
        lib().orc_covid_propose(C.byref(self.m), C.byref(rng.r), _dp(x), stepsize, _dp(xp))
        return xp

    def force_bounds(self, x):
        x = np.array(x, dtype=np.float64)
        lib().orc_covid_force_bounds(C.byref(self.m), _dp(x))
        return x


def make_target(kind="plain", T=1.0, slope=0.0, Tinit=1.0):
    return Target({"plain": 0, "temp": 1, "cooling": 2}[kind], T, slope, Tinit)


class OracleChain:
    def __init__(self, model: Model, chain_id, x0, stepsize=1.0):
        self.model = model
        self.c = Chain()
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        lib().orc_chain_init(C.byref(model.m), C.byref(self.c), chain_id, _dp(x0), stepsize)

    def step(self, target, time=0.0, random_subspace=True):
        return lib().orc_chain_step(C.byref(self.model.m), C.byref(target), C.byref(self.c), time,
                                    1 if random_subspace else 0)

    def run(self, target, nSamples, nBurnin, nAdjust, thinning, record=True, mean=False):
        n = lib().orc_n_saved(nSamples, nBurnin, nAdjust, thinning)
        ll = np.zeros(n)
        samples = np.zeros((n, self.model.dim)) if record else None
        mu = np.zeros(self.model.dim) if mean else None
        got = lib().orc_chain_run(C.byref(self.model.m), C.byref(target), C.byref(self.c), nSamples,
                                  nBurnin, nAdjust, thinning, _dp(ll), _dp(samples), _dp(mu))
        assert got == n
        return ll, samples, mu

    @property
    def x(self):
        return np.ctypeslib.as_array(self.c.x, shape=(self.model.dim,)).copy()

    @property
    def ll(self):
        return (self.c.ll_model, self.c.ll_prior)

    @property
    def rng_state(self):
        return (self.c.rng.state, self.c.rng.inc)

    def __del__(self):
        try:
            lib().orc_chain_free(C.byref(self.c))
        except Exception:
            pass


def run_chains(model: Model, target, x0, stepsizes, nSamples, nBurnin, nAdjust, thinning,
               n_threads=1, first_chain_id=1, want_loglikes=False):
    """ChainManager::run_all_adjust equivalent.  x0 [n][dim].  Returns dict."""
    x = np.array(x0, dtype=np.float64, order="C")
    n = x.shape[0]
    ss = np.array(np.broadcast_to(stepsizes, (n,)), dtype=np.float64)
    final_ll = np.zeros((n, 2)); nacc = np.zeros(n, dtype=np.int32)
    rng = np.zeros((n, 2), dtype=np.uint64)
    ns = lib().orc_n_saved(nSamples, nBurnin, nAdjust, thinning)
    lls = np.zeros((n, ns)) if want_loglikes else None
    wall = lib().orc_run_chains(C.byref(model.m), C.byref(target), n, first_chain_id, _dp(x), _dp(ss),
                                nSamples, nBurnin, nAdjust, thinning, n_threads, _dp(final_ll),
                                nacc.ctypes.data_as(C.POINTER(C.c_int)),
                                rng.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(lls))
    return dict(x=x, stepsizes=ss, ll=final_ll, n_accept=nacc, rng=rng, wall=wall, loglikes=lls,
                steps=n * (20 * nAdjust + nSamples))


def mylike1_run(chain_id, nSamples, nBurnin, nAdjust, thinning, with_mean=False):
    n = lib().orc_n_saved(nSamples, nBurnin, nAdjust, thinning)
    ll = np.zeros(n); s = np.zeros((n, 3)); sf = C.c_double(); na = C.c_int(); mean = np.zeros(4)
    lib().orc_mylike1_run(chain_id, nSamples, nBurnin, nAdjust, thinning, _dp(ll), _dp(s),
                          C.byref(sf), C.byref(na), _dp(mean))
    return (ll, s, sf.value, na.value, mean) if with_mean else (ll, s, sf.value, na.value)


def fourgauss_run(difficulty, T, chain_id, x0, nSamples, nBurnin, nAdjust, thinning, with_mean=False, stepsize=1.0):
    n = lib().orc_n_saved(nSamples, nBurnin, nAdjust, thinning)
    ll = np.zeros(n); s = np.zeros((n, 2)); sf = C.c_double(); na = C.c_int(); mean = np.zeros(2)
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    lib().orc_fourgauss_run(difficulty, T, chain_id, _dp(x0), nSamples, nBurnin, nAdjust, thinning,
                            _dp(ll), _dp(s), C.byref(sf), C.byref(na), _dp(mean), stepsize)
    return (ll, s, sf.value, na.value, mean) if with_mean else (ll, s, sf.value, na.value)


def rosenbrock_run(difficulty, T, chain_id, x0, nSamples, nBurnin, nAdjust, thinning, with_mean=False, stepsize=1.0):
    n = lib().orc_n_saved(nSamples, nBurnin, nAdjust, thinning)
    ll = np.zeros(n); s = np.zeros((n, 2)); sf = C.c_double(); na = C.c_int(); mean = np.zeros(2)
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    lib().orc_rosenbrock_run(difficulty, T, chain_id, _dp(x0), nSamples, nBurnin, nAdjust, thinning,
                             _dp(ll), _dp(s), C.byref(sf), C.byref(na), _dp(mean), stepsize)
    return (ll, s, sf.value, na.value, mean) if with_mean else (ll, s, sf.value, na.value)


class MultiModel:
    """R region models (extension, BASELINE configs[4]): cfgs is a list of per-region config dicts that differ
    in deaths, sigma, pop_size, cap0, cap_incr_rate only."""

    def __init__(self, cfgs, R11, contract=0):
        self.models = [Model(c, R11, contract=contract) for c in cfgs]
        self.R = len(cfgs)
        self.arr = (CovidModel * self.R)(*[m.m for m in self.models])
        self.dim_region = self.models[0].dim
        self.dim = self.dim_region * self.R


class MultiChain:
    def __init__(self, mm: MultiModel, chain_id, x0, stepsize=1.0):
        self.mm = mm
        self.c = MultiChainC()
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        assert x0.shape == (mm.dim,)
        lib().orc_multi_init(mm.arr, mm.R, C.byref(self.c), chain_id, _dp(x0), stepsize)

    def step(self, target, time=0.0, random_subspace=True):
        return lib().orc_multi_step(self.mm.arr, self.mm.R, C.byref(target), C.byref(self.c), time, 1 if random_subspace else 0)

    def run(self, target, nSamples, nBurnin, nAdjust, thinning, record=True, mean=False):
        n = lib().orc_n_saved(nSamples, nBurnin, nAdjust, thinning)
        ll = np.zeros(n)
        samples = np.zeros((n, self.mm.dim)) if record else None
        mu = np.zeros(self.mm.dim) if mean else None
        got = lib().orc_multi_run(self.mm.arr, self.mm.R, C.byref(target), C.byref(self.c), nSamples, nBurnin, nAdjust,
                                  thinning, _dp(ll), _dp(samples), _dp(mu))
        assert got == n
        return ll, samples, mu

    @property
    def x(self):
        return np.ctypeslib.as_array(self.c.x, shape=(self.mm.dim,)).copy()

    @property
    def ll(self):
        return np.ctypeslib.as_array(self.c.ll, shape=(self.mm.R, 2)).copy()

    @property
    def stepsizes(self):
        return np.ctypeslib.as_array(self.c.stepsize, shape=(self.mm.R,)).copy()

    @property
    def rng_state(self):
        return (self.c.rng.state, self.c.rng.inc)

    def __del__(self):
        try:
            lib().orc_multi_free(C.byref(self.c))
        except Exception:
            pass
