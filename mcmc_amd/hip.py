"""ctypes view of the C ABI in include/mcmc_hip.h (libmcmc_hip.so).

This is the only way Python reaches the hot path; there is no CPU fallback.  If the
library is missing or no GPU is visible the calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import covid_config as CC

_HERE = os.path.dirname(os.path.abspath(__file__))
# MCMC_HIP_LIB selects another build of the same library (kernel-variant experiments)
LIB_PATH = os.environ.get("MCMC_HIP_LIB") or os.path.join(_HERE, "libmcmc_hip.so")

NCURVES = 10
OUT_CURVES = ["mildlyInfectious", "highlyInfectious", "incubating", "asymptomatic", "mild",
              "serious", "recovered", "dead", "totalBehavior"]

# every symbol include/mcmc_hip.h declares
SYMBOLS = [
    "mcmc_hip_last_error", "mcmc_hip_device_count", "mcmc_host_make_response_tables",
    "mcmc_host_default_disease_params", "mcmc_host_perturb_ics", "mcmc_host_perturb_ics_from", "mcmc_hip_create",
    "mcmc_hip_destroy", "mcmc_hip_set_tables", "mcmc_hip_set_data", "mcmc_hip_set_precision", "mcmc_hip_set_mapping", "mcmc_hip_get_mapping", "mcmc_hip_set_chains",
    "mcmc_hip_set_state", "mcmc_hip_set_regions", "mcmc_hip_n_regions", "mcmc_hip_get_region_loglike",
    "mcmc_hip_get_region_stepsizes", "mcmc_hip_get_scan_position", "mcmc_hip_set_region_state",
    "mcmc_hip_run", "mcmc_hip_step", "mcmc_hip_n_saved", "mcmc_hip_dim", "mcmc_hip_size",
    "mcmc_hip_n_chains", "mcmc_hip_get_params", "mcmc_hip_get_loglike", "mcmc_hip_get_stepsizes",
    "mcmc_hip_get_accept_counts", "mcmc_hip_get_rng_state", "mcmc_hip_get_loglikes",
    "mcmc_hip_get_samples", "mcmc_hip_get_sample_all", "mcmc_hip_get_mean", "mcmc_hip_eval",
    "mcmc_hip_derive", "mcmc_hip_diagnostics", "mcmc_hip_last_timing", "mcmc_hip_pair_count",
    "mcmc_hip_executed_pair_count", "mcmc_hip_eval_static", "mcmc_hip_eval_static_max",
    "mcmc_hip_debug_phase_cycles", "mcmc_hip_debug_wave_log", "mcmc_hip_debug_f32_check", "mcmc_hip_set_ladder", "mcmc_hip_swap", "mcmc_hip_run_tempered",
    "mcmc_hip_get_rungs", "mcmc_hip_get_swap_counts", "mcmc_hip_toy_dim", "mcmc_hip_toy_n_out", "mcmc_hip_toy_eval", "mcmc_hip_toy_run",
]


class McmcHipError(RuntimeError):
    pass


class DiseaseParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in CC.PARAM_FIELDS]


class CovidConfig(C.Structure):
    _fields_ = [
        ("n_data", C.c_int32), ("max_delay", C.c_int32), ("n_predict", C.c_int32),
        ("n_disc", C.c_int32), ("fix_behavior_in_advance", C.c_int32), ("pop_size", C.c_int32),
        ("prob_icu_if_serious", C.c_double), ("prob_lethal_daily_untreated", C.c_double),
        ("cap0", C.c_double), ("cap_incr_rate", C.c_double),
        ("has_prior", C.c_int32), ("smooth_Ls", C.c_double), ("smooth_L", C.c_double),
        ("target_kind", C.c_int32), ("target_T", C.c_double), ("target_slope", C.c_double),
        ("target_Tinit", C.c_double), ("contract", C.c_int32),
        ("disc_days", C.POINTER(C.c_int32)), ("disc_fixed", C.POINTER(C.c_uint8)),
    ]


_lib = None


def lib():
    """Load libmcmc_hip.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise McmcHipError(f"{LIB_PATH} not found: build it with `make -C mcmc_amd/csrc` "
                           "(or __graft_entry__.build()); the hot path has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    dp = C.POINTER(C.c_double)
    vp = C.c_void_p
    L.mcmc_hip_last_error.restype = C.c_char_p
    L.mcmc_hip_device_count.restype = C.c_int
    L.mcmc_host_make_response_tables.argtypes = [C.POINTER(DiseaseParams), C.c_uint64, C.c_int32, dp]
    L.mcmc_host_default_disease_params.argtypes = [C.POINTER(DiseaseParams)]
    L.mcmc_host_default_disease_params.restype = None
    L.mcmc_host_perturb_ics.argtypes = [dp, C.c_int32, C.c_int64, C.c_double, C.c_uint64, C.c_uint64, dp]
    L.mcmc_host_perturb_ics_from.argtypes = [dp, C.c_int32, C.c_int64, C.c_int64, C.c_double, C.c_uint64, C.c_uint64, dp]
    L.mcmc_hip_create.argtypes = [C.POINTER(CovidConfig), C.c_int, C.POINTER(vp)]
    L.mcmc_hip_destroy.argtypes = [vp]
    L.mcmc_hip_set_tables.argtypes = [vp, dp, C.c_int32]
    L.mcmc_hip_set_data.argtypes = [vp, dp, dp]
    L.mcmc_hip_set_precision.argtypes = [vp, C.c_int32]
    L.mcmc_hip_set_mapping.argtypes = [vp, C.c_int32]
    L.mcmc_hip_get_mapping.argtypes = [vp]
    L.mcmc_hip_set_chains.argtypes = [vp, C.c_int64, dp, dp, C.POINTER(C.c_uint64), C.c_uint64]
    L.mcmc_hip_set_regions.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32), dp, dp, dp, dp]
    L.mcmc_hip_n_regions.argtypes = [vp]
    L.mcmc_hip_get_region_loglike.argtypes = [vp, dp]
    L.mcmc_hip_get_region_stepsizes.argtypes = [vp, dp]
    L.mcmc_hip_get_scan_position.argtypes = [vp, C.POINTER(C.c_int64)]
    L.mcmc_hip_set_region_state.argtypes = [vp, C.c_int64, dp]
    L.mcmc_hip_set_state.argtypes = [vp, C.POINTER(C.c_uint64), dp, dp, C.POINTER(C.c_int32)]
    L.mcmc_hip_run.argtypes = [vp] + [C.c_int32] * 6
    L.mcmc_hip_step.argtypes = [vp, C.c_int32, C.c_double, C.c_int32]
    L.mcmc_hip_n_saved.argtypes = [C.c_int32] * 4
    L.mcmc_hip_n_saved.restype = C.c_int32
    L.mcmc_hip_dim.argtypes = [vp]
    L.mcmc_hip_size.argtypes = [vp]
    L.mcmc_hip_n_chains.argtypes = [vp]
    L.mcmc_hip_n_chains.restype = C.c_int64
    for name in ("get_params", "get_loglike", "get_stepsizes", "get_loglikes", "get_mean", "diagnostics"):
        getattr(L, "mcmc_hip_" + name).argtypes = [vp, dp]
    L.mcmc_hip_get_accept_counts.argtypes = [vp, C.POINTER(C.c_int32)]
    L.mcmc_hip_get_rng_state.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.mcmc_hip_get_samples.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, dp]
    L.mcmc_hip_get_sample_all.argtypes = [vp, C.c_int32, dp]
    L.mcmc_hip_eval.argtypes = [vp, C.c_int64, dp, dp]
    L.mcmc_hip_derive.argtypes = [vp, C.c_int64, dp, dp]
    L.mcmc_hip_eval_static.argtypes = [vp, C.c_int64, C.c_void_p, C.c_void_p]      # DEVICE pointers
    L.mcmc_hip_eval_static_max.argtypes = [vp]
    L.mcmc_hip_eval_static_max.restype = C.c_int64
    L.mcmc_hip_last_timing.argtypes = [vp, dp, dp, dp, C.POINTER(C.c_int64)]
    L.mcmc_hip_pair_count.argtypes = [vp, dp, C.c_int32]
    L.mcmc_hip_executed_pair_count.argtypes = [vp, dp, C.c_int32]
    L.mcmc_hip_debug_phase_cycles.argtypes = [vp, dp, C.c_int32]
    L.mcmc_hip_debug_wave_log.argtypes = [vp, dp, C.POINTER(C.c_uint64), dp, dp, dp, C.c_int64]
    L.mcmc_hip_debug_f32_check.argtypes = [vp, C.POINTER(C.c_int32)]
    L.mcmc_hip_set_ladder.argtypes = [vp, C.c_int32, dp, C.c_uint64, C.c_uint64]
    L.mcmc_hip_swap.argtypes = [vp, C.c_int32]
    L.mcmc_hip_run_tempered.argtypes = [vp, C.c_int32, C.c_int32]
    L.mcmc_hip_get_rungs.argtypes = [vp, C.POINTER(C.c_int32)]
    L.mcmc_hip_get_swap_counts.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.mcmc_hip_toy_dim.argtypes = [C.c_int32]
    L.mcmc_hip_toy_n_out.argtypes = [C.c_int32]
    L.mcmc_hip_toy_eval.argtypes = [C.c_int, C.c_int32, C.c_double, C.c_int64, dp, dp]
    L.mcmc_hip_toy_run.argtypes = [C.c_int, C.c_int32, C.c_double, C.c_int32, C.c_double, C.c_double, C.c_double,
                                   C.c_int64, C.POINTER(C.c_uint64), C.c_uint64, dp, dp, C.c_int32, C.c_int32,
                                   C.c_int32, C.c_int32, dp, dp, dp, C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.c_int32]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise McmcHipError(f"mcmc_hip error {rc}: {lib().mcmc_hip_last_error().decode()}")


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


_KIND = {"plain": 0, "temp": 1, "cooling": 2}


def make_response_tables(disease_params: dict, n_traj=10_000_000, n_lag=500):
    """AvgDiseaseTrajectory (covid.h:96-282) -> R[10][n_lag]; host precompute as in the
    reference."""
    p = DiseaseParams(**{k: float(disease_params[k]) for k in CC.PARAM_FIELDS})
    R = np.zeros((NCURVES, n_lag), dtype=np.float64)
    _check(lib().mcmc_host_make_response_tables(C.byref(p), n_traj, n_lag, _dp(R)))
    return R


def perturb_ics(x0, n_chains, a=0.01, initstate=0, initseq=0, first_chain=0):
    """chains [first_chain, first_chain + n_chains) of ONE global sequence of starting points (chain i has
    the same start on any sharding); first_chain > 0 jumps the generator ahead instead of drawing the
    earlier chains"""
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    out = np.zeros((n_chains, x0.size), dtype=np.float64)
    if first_chain:
        _check(lib().mcmc_host_perturb_ics_from(_dp(x0), x0.size, first_chain, n_chains, a, initstate, initseq, _dp(out)))
    else:
        _check(lib().mcmc_host_perturb_ics(_dp(x0), x0.size, n_chains, a, initstate, initseq, _dp(out)))
    return out


TOY = {"mylike1": 0, "fourgauss": 1, "rosenbrock": 2}


def toy_eval(kind, x, difficulty=1.0, device=0):
    """log-likelihood of MyLike1 / FourGaussians (test.h:26-37, 87-100) for n positions."""
    x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
    out = np.zeros(x.shape[0])
    _check(lib().mcmc_hip_toy_eval(device, TOY[kind], difficulty, x.shape[0], _dp(x), _dp(out)))
    return out


def toy_run(kind, x0, n_samples, n_burnin, n_adjust, thinning, difficulty=1.0, target="plain", T=1.0, slope=0.0,
            Tinit=1.0, stepsizes=1.0, chain_ids=None, first_chain_id=0, device=0, rng_state=None, mean_in=None,
            record=True):
    """MetropolisChain::run for n toy chains on the GPU (test.cpp:27-40 is kind='mylike1',
    one chain, id 0, x0 = (1,1,1), run(100000, 10, 500, 5)).  Returns a dict; `samples` and
    `mean` have n_out columns (MyLike1: position + its coordinate "max")."""
    L = lib()
    x = np.array(np.atleast_2d(x0), dtype=np.float64, order="C")
    n, dim = x.shape
    assert dim == L.mcmc_hip_toy_dim(TOY[kind])
    ss = np.array(np.broadcast_to(stepsizes, (n,)), dtype=np.float64)
    ns = L.mcmc_hip_n_saved(n_samples, n_burnin, n_adjust, thinning)
    n_out = L.mcmc_hip_toy_n_out(TOY[kind])
    ll = np.zeros((n, ns)) if record else None                 # record=False: end state, mean and accept counts only
    samples = np.zeros((n, ns, n_out)) if record else None
    mean = np.zeros((n, n_out)) if mean_in is None else np.array(mean_in, dtype=np.float64, order="C").reshape(n, n_out)
    rng = None if rng_state is None else np.array(rng_state, dtype=np.uint64, order="C").reshape(n, 2)
    acc = np.zeros(n, dtype=np.int32)
    ids = None if chain_ids is None else np.ascontiguousarray(chain_ids, dtype=np.uint64)
    _check(L.mcmc_hip_toy_run(device, TOY[kind], difficulty, _KIND[target], T, slope, Tinit, n,
                              ids.ctypes.data_as(C.POINTER(C.c_uint64)) if ids is not None else None,
                              first_chain_id, _dp(x), _dp(ss), n_samples, n_burnin, n_adjust, thinning,
                              _dp(ll), _dp(samples), _dp(mean), acc.ctypes.data_as(C.POINTER(C.c_int32)),
                              rng.ctypes.data_as(C.POINTER(C.c_uint64)) if rng is not None else None,
                              0 if mean_in is None else 1))
    return dict(x=x, stepsizes=ss, loglikes=ll, samples=samples, mean=mean, n_accept=acc, rng_state=rng)


class CovidSampler:
    """Owns one mcmc_hip_ctx: the GPU-resident replacement of
    ChainManager<MetropolisChain> + DiseaseSpread + SmoothnessPrior + Target for the covid
    model (mcmc.h:1314-1655, covid.h:284-662)."""

    PRECISIONS = {"f64": 0, "f32_model": 1, "f32_packed": 2}          # MCMC_HIP_PRECISION_*

    def __init__(self, cfg: dict, R, device=0, precision="f64"):
        L = lib()
        self.cfg = cfg
        self._disc_days = np.ascontiguousarray(cfg["disc_days"], dtype=np.int32)
        self._disc_fixed = np.ascontiguousarray(cfg["disc_fixed"], dtype=np.uint8)
        c = CovidConfig()
        for k in ("n_data", "max_delay", "n_predict", "n_disc", "fix_behavior_in_advance", "pop_size",
                  "has_prior", "contract"):
            setattr(c, k, int(cfg[k]))
        for k in ("prob_icu_if_serious", "prob_lethal_daily_untreated", "cap0", "cap_incr_rate",
                  "smooth_Ls", "smooth_L", "target_T", "target_slope", "target_Tinit"):
            setattr(c, k, float(cfg[k]))
        c.target_kind = _KIND[cfg["target_kind"]]
        c.disc_days = self._disc_days.ctypes.data_as(C.POINTER(C.c_int32))
        c.disc_fixed = self._disc_fixed.ctypes.data_as(C.POINTER(C.c_uint8))
        self._h = C.c_void_p()
        _check(L.mcmc_hip_create(C.byref(c), device, C.byref(self._h)))
        R = np.ascontiguousarray(R, dtype=np.float64)
        assert R.ndim == 2 and R.shape[0] == NCURVES
        _check(L.mcmc_hip_set_tables(self._h, _dp(R), R.shape[1]))
        deaths = np.ascontiguousarray(cfg["deaths"], dtype=np.float64)
        sigma = np.ascontiguousarray(cfg["sigma"], dtype=np.float64)
        _check(L.mcmc_hip_set_data(self._h, _dp(deaths), _dp(sigma)))
        if precision != "f64":
            self.set_precision(precision)
        self.dim = L.mcmc_hip_dim(self._h)
        self.size = L.mcmc_hip_size(self._h)
        self.n_chains = 0
        self.n_saved = 0
        self.n_regions = 1

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().mcmc_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- chains -------------------------------------------------------------------------
    def set_precision(self, precision):
        """"f64" (the reference's arithmetic, default) or "f32_model" (fast mode: the renewal
        convolution in packed FP32; statistical bar, see include/mcmc_hip.h).  Before set_chains."""
        _check(lib().mcmc_hip_set_precision(self._h, self.PRECISIONS[precision]))
        self.n_chains = 0

    MAPPINGS = {"auto": 0, "lane": 1, "wave": 2}                        # MCMC_HIP_MAPPING_*

    def set_mapping(self, mapping):
        """"lane" (one chain per lane), "wave" (one chain per wavefront: small ensembles) or "auto"; results do
        not depend on it (include/mcmc_hip.h at mcmc_hip_set_mapping)."""
        _check(lib().mcmc_hip_set_mapping(self._h, self.MAPPINGS[mapping]))

    def mapping(self):
        return {1: "lane", 2: "wave"}[lib().mcmc_hip_get_mapping(self._h)]

    def set_regions(self, cfgs):
        """multi-region model (extension, BASELINE configs[4]): cfgs = per-region config dicts that differ in
        deaths, sigma, pop_size, cap0, cap_incr_rate; replaces set_data."""
        R = len(cfgs)
        pop = np.ascontiguousarray([int(c["pop_size"]) for c in cfgs], dtype=np.int32)
        cap0 = np.ascontiguousarray([float(c["cap0"]) for c in cfgs], dtype=np.float64)
        rate = np.ascontiguousarray([float(c["cap_incr_rate"]) for c in cfgs], dtype=np.float64)
        deaths = np.ascontiguousarray([c["deaths"] for c in cfgs], dtype=np.float64)
        sigma = np.ascontiguousarray([c["sigma"] for c in cfgs], dtype=np.float64)
        _check(lib().mcmc_hip_set_regions(self._h, R, pop.ctypes.data_as(C.POINTER(C.c_int32)), _dp(cap0), _dp(rate),
                                          _dp(deaths), _dp(sigma)))
        self.n_regions = R
        self.dim = lib().mcmc_hip_dim(self._h)
        self.n_chains = 0

    def region_loglike(self):
        return self._get("get_region_loglike", (self.n_chains, self.n_regions, 2))

    def region_stepsizes(self):
        return self._get("get_region_stepsizes", (self.n_chains, self.n_regions))

    def scan_position(self):
        v = C.c_int64()
        _check(lib().mcmc_hip_get_scan_position(self._h, C.byref(v)))
        return v.value

    def set_region_state(self, scan_position, region_stepsizes=None):
        """mcmc_hip_set_region_state: the scan position and per-region step sizes of resumed multi-region chains"""
        ss = None if region_stepsizes is None else np.ascontiguousarray(region_stepsizes, dtype=np.float64)
        assert ss is None or ss.shape == (self.n_chains, self.n_regions)
        _check(lib().mcmc_hip_set_region_state(self._h, scan_position, _dp(ss)))

    def set_chains(self, params, stepsizes=None, chain_ids=None, first_chain_id=1):
        params = np.ascontiguousarray(params, dtype=np.float64)
        assert params.ndim == 2 and params.shape[1] == self.dim, (params.shape, self.dim)
        n = params.shape[0]
        ss = None if stepsizes is None else np.ascontiguousarray(
            np.broadcast_to(stepsizes, (n,)), dtype=np.float64)
        ids = None if chain_ids is None else np.ascontiguousarray(chain_ids, dtype=np.uint64)
        _check(lib().mcmc_hip_set_chains(
            self._h, n, _dp(params), _dp(ss),
            ids.ctypes.data_as(C.POINTER(C.c_uint64)) if ids is not None else None, first_chain_id))
        self.n_chains = n

    def set_state(self, rng_state=None, loglike=None, mean=None, accept_counts=None):
        """mcmc_hip_set_state: continue chains that another context / process advanced."""
        r = None if rng_state is None else np.ascontiguousarray(rng_state, dtype=np.uint64)
        l = None if loglike is None else np.ascontiguousarray(loglike, dtype=np.float64)
        m = None if mean is None else np.ascontiguousarray(mean, dtype=np.float64)
        a = None if accept_counts is None else np.ascontiguousarray(accept_counts, dtype=np.int32)
        _check(lib().mcmc_hip_set_state(
            self._h, r.ctypes.data_as(C.POINTER(C.c_uint64)) if r is not None else None, _dp(l), _dp(m),
            a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None))

    def run(self, n_samples, n_burnin, n_adjust, thinning, record_samples=False, compute_mean=False):
        _check(lib().mcmc_hip_run(self._h, n_samples, n_burnin, n_adjust, thinning,
                                  int(record_samples), int(compute_mean)))
        self.n_saved = lib().mcmc_hip_n_saved(n_samples, n_burnin, n_adjust, thinning)

    # --- tempering ladder (extension, BASELINE configs[3]) ----------------------------------
    def set_ladder(self, temps, swap_seed=1, first_ladder=0):
        t = np.ascontiguousarray(temps, dtype=np.float64)
        _check(lib().mcmc_hip_set_ladder(self._h, t.size, _dp(t), swap_seed, first_ladder))
        self.n_temps = t.size

    def swap(self, parity):
        _check(lib().mcmc_hip_swap(self._h, parity))

    def run_tempered(self, n_rounds, steps_per_round):
        _check(lib().mcmc_hip_run_tempered(self._h, n_rounds, steps_per_round))

    def rungs(self):
        return self._get("get_rungs", (self.n_chains,), np.int32, C.c_int32)

    def swap_counts(self):
        a = np.zeros(self.n_temps - 1, dtype=np.int64); p = np.zeros(self.n_temps - 1, dtype=np.int64)
        _check(lib().mcmc_hip_get_swap_counts(self._h, a.ctypes.data_as(C.POINTER(C.c_int64)),
                                              p.ctypes.data_as(C.POINTER(C.c_int64))))
        return a, p

    def step(self, n_steps, beta=1.0, random_subspace=True):
        _check(lib().mcmc_hip_step(self._h, n_steps, beta, int(random_subspace)))

    # --- results ------------------------------------------------------------------------
    def _get(self, name, shape, dtype=np.float64, ctype=C.c_double):
        out = np.zeros(shape, dtype=dtype)
        _check(getattr(lib(), "mcmc_hip_" + name)(self._h, out.ctypes.data_as(C.POINTER(ctype))))
        return out

    def params(self):
        return self._get("get_params", (self.n_chains, self.dim))

    def loglike(self):
        return self._get("get_loglike", (self.n_chains, 2))

    def stepsizes(self):
        return self._get("get_stepsizes", (self.n_chains,))

    def accept_counts(self):
        return self._get("get_accept_counts", (self.n_chains,), np.int32, C.c_int32)

    def rng_state(self):
        return self._get("get_rng_state", (self.n_chains, 2), np.uint64, C.c_uint64)

    def loglikes(self):
        return self._get("get_loglikes", (self.n_chains, self.n_saved))

    def mean(self):
        return self._get("get_mean", (self.n_chains, self.dim))

    def samples(self, chain, coord_first=0, coord_count=None):
        coord_count = self.dim - coord_first if coord_count is None else coord_count
        out = np.zeros((self.n_saved, coord_count))
        _check(lib().mcmc_hip_get_samples(self._h, chain, coord_first, coord_count, _dp(out)))
        return out

    def sample_all(self, saved_index):
        out = np.zeros((self.n_chains, self.dim))
        _check(lib().mcmc_hip_get_sample_all(self._h, saved_index, _dp(out)))
        return out

    def eval(self, params):
        params = np.ascontiguousarray(np.atleast_2d(params), dtype=np.float64)
        out = np.zeros((params.shape[0], 2))
        _check(lib().mcmc_hip_eval(self._h, params.shape[0], _dp(params), _dp(out)))
        return out

    def eval_static(self, n, params_dev_ptr, ll_out_dev_ptr):
        """mcmc_hip_eval_static: device pointers (integers), no allocation inside the library."""
        _check(lib().mcmc_hip_eval_static(self._h, n, C.c_void_p(params_dev_ptr), C.c_void_p(ll_out_dev_ptr)))

    def eval_static_max(self):
        return int(lib().mcmc_hip_eval_static_max(self._h))

    def derive(self, params):
        params = np.ascontiguousarray(np.atleast_2d(params), dtype=np.float64)
        out = np.zeros((params.shape[0], 9, self.size))
        _check(lib().mcmc_hip_derive(self._h, params.shape[0], _dp(params), _dp(out)))
        return out

    def diagnostics(self):
        return self._get("diagnostics", (4 + 2 * self.dim,))

    def last_timing(self):
        a, s, k = C.c_double(), C.c_double(), C.c_double()
        n = C.c_int64()
        _check(lib().mcmc_hip_last_timing(self._h, C.byref(a), C.byref(s), C.byref(k), C.byref(n)))
        return dict(adjust_s=a.value, sampling_s=s.value, step_kernel_s=k.value, launches=n.value)

    PHASES = ["propose", "setup", "chunks", "tail", "days", "loglike", "prior", "accept", "record", "chunks_e", "n_chunks", "n_chunks_e", "wait", "e_days", "e_hull"]

    def phase_cycles(self, reset=False):
        out = np.zeros(16)
        _check(lib().mcmc_hip_debug_phase_cycles(self._h, _dp(out), int(reset)))
        d = dict(zip(self.PHASES, out[:len(self.PHASES)]))
        d["slowest_wave"] = out[15]        # max over waves of one launch's cycles
        return d

    def f32_check(self):
        out = (C.c_int32 * 4)()
        _check(lib().mcmc_hip_debug_f32_check(self._h, out))
        return list(out)

    def wave_log(self, n_tiles):
        cyc, hw = np.zeros(n_tiles), np.zeros(n_tiles, dtype=np.uint64)
        ech, eda, ehu = np.zeros(n_tiles), np.zeros(n_tiles), np.zeros(n_tiles)
        _check(lib().mcmc_hip_debug_wave_log(self._h, _dp(cyc), hw.ctypes.data_as(C.POINTER(C.c_uint64)),
                                             _dp(ech), _dp(eda), _dp(ehu), n_tiles))
        return cyc, hw, ech, eda, ehu

    def pair_count(self, reset=False):
        v = C.c_double()
        _check(lib().mcmc_hip_pair_count(self._h, C.byref(v), int(reset)))
        return v.value

    def executed_pair_count(self, reset=False):
        v = C.c_double()
        _check(lib().mcmc_hip_executed_pair_count(self._h, C.byref(v), int(reset)))
        return v.value
