"""Host-side description of the covid model run: plain data, no device code.

Mirrors what a user of the reference sets up in notebooks/covid.ipynb cells 3-4
(`DiseaseParams`, `DiseaseData`, `DiseaseSpread`, `SmoothnessPrior`, `TempTarget`) as a
dict that both the C-ABI wrapper (mcmc_amd.hip) and the test oracle consume.
"""
import numpy as np

# notebooks/covid.ipynb cell 3: cumulative deaths in Italy from 21 Feb 2020 (input data)
ITALY_DEATHS = np.array([1, 2, 3, 7, 10, 12, 17, 21, 29, 34, 52, 79, 107, 148, 197, 233, 366,
                         463, 631, 827, 827, 1266, 1441, 1809, 2158, 2503, 2978, 3405, 4032,
                         4825, 5476, 6077, 6820, 7503, 8215, 9134, 10023, 10779, 11591, 12428,
                         13155, 13915, 14681, 15362], dtype=np.float64)

# covid.h:75-90 defaults
DISEASE_PARAMS_DEFAULT = dict(
    timeIncub=4.0, timeIncubSigma=2.0, probAsymp=0.1, probSerious=0.08, probLethal=0.004,
    probLethalDailyWhenSeriousUntreated=0.2, probICUIfSerious=0.3, timeMildDuration=10.0,
    timeMildDurationSigma=5.0, timeMildToSerious=8.0, timeMildToSeriousSigma=3.0,
    timeSeriousToRec=10.0, timeSeriousToRecSigma=5.0, timeSeriousToDeath=9.0,
    timeSeriousToDeathSigma=5.0)
PARAM_FIELDS = list(DISEASE_PARAMS_DEFAULT.keys())


def ipynb_disease_params():
    """DiseaseParams after the overrides of notebooks/covid.ipynb cell 3.  The notebook
    sets `timeSeriousToRecovered[Sigma]`, a name the class of covid.h:87-88 spells
    `timeSeriousToRec[Sigma]`; mapped to that field."""
    p = dict(DISEASE_PARAMS_DEFAULT)
    p.update(probLethalDailyWhenSeriousUntreated=0.3, probSerious=0.024, probAsymp=0.3,
             timeIncub=6.0, timeIncubSigma=4.0, timeMildToSerious=7.1,
             timeMildToSeriousSigma=13.0, timeMildDuration=7.4, timeMildDurationSigma=2.0,
             timeSeriousToDeath=10.0, timeSeriousToDeathSigma=6.0, timeSeriousToRec=8.0,
             timeSeriousToRecSigma=4.5)
    p["probLethal"] = p["probSerious"] / 6
    return p


def sigma_recipe(deaths):
    """covid.ipynb cell 3: shot noise sqrt(deaths) + cumsum of 10 unnoticed/day for the
    first 14 days.  sqrt(max(deaths,1)) keeps sigma > 0 for synthetic series."""
    deaths = np.asarray(deaths, dtype=np.float64)
    unnoticed = 10 * np.ones(deaths.shape)
    unnoticed[14:] = 0
    return np.cumsum(unnoticed) + np.sqrt(np.maximum(deaths, 1.0))


def make_config(deaths, sigma, *, disease_params=None, disc_days=(1, 9, 12, 29),
                disc_vals=(1.0, 0.6, 0.5, 0.4), disc_fixed=None, pop_size=60_000_000,
                cap0=7500.0, cap_incr_rate=100.0, max_delay=40, n_predict=60,
                fix_behavior_in_advance=14, initial_beta_mild=2.0, initial_beta_high=10.0,
                initial_delay=23.0, smooth_Ls=50.0, smooth_L=None, has_prior=1,
                target_kind="temp", target_T=1.0, target_slope=0.0, target_Tinit=1.0,
                contract=0):
    """All per-run constants.  Defaults are covid.ipynb cells 3-4 (fixBehaviorInAdvance
    keeps the class default 14, covid.h:56)."""
    deaths = np.ascontiguousarray(deaths, dtype=np.float64)
    sigma = np.ascontiguousarray(sigma, dtype=np.float64)
    assert deaths.shape == sigma.shape and deaths.ndim == 1
    dp = dict(disease_params or ipynb_disease_params())
    n_data = deaths.size
    disc_days = np.asarray(disc_days, dtype=np.int32)
    if disc_fixed is None:
        disc_fixed = np.zeros(disc_days.size, dtype=np.uint8)
    nB = n_data - fix_behavior_in_advance
    cfg = dict(
        n_data=n_data, max_delay=int(max_delay), n_predict=int(n_predict),
        n_disc=int(disc_days.size), fix_behavior_in_advance=int(fix_behavior_in_advance),
        pop_size=int(pop_size), prob_icu_if_serious=dp["probICUIfSerious"],
        prob_lethal_daily_untreated=dp["probLethalDailyWhenSeriousUntreated"],
        cap0=float(cap0), cap_incr_rate=float(cap_incr_rate), has_prior=int(has_prior),
        smooth_Ls=float(smooth_Ls), smooth_L=float(nB if smooth_L is None else smooth_L),
        target_kind=target_kind, target_T=float(target_T), target_slope=float(target_slope),
        target_Tinit=float(target_Tinit), contract=int(contract),
        deaths=deaths, sigma=sigma, disc_days=disc_days,
        disc_fixed=np.asarray(disc_fixed, dtype=np.uint8), disease_params=dp,
        disc_vals=np.asarray(disc_vals, dtype=np.float64),
        initial_beta_mild=float(initial_beta_mild), initial_beta_high=float(initial_beta_high),
        initial_delay=float(initial_delay))
    cfg["size"] = cfg["max_delay"] + cfg["n_predict"] + n_data       # covid.h:297
    cfg["nB"] = nB                                                    # covid.h:299
    cfg["dim"] = nB + cfg["n_disc"] + 3
    return cfg


def initial_params(cfg):
    """DiseaseSpread's initial coords (covid.h:299-301): behavior=1, discontinuousVals,
    betaMild, betaHigh, delay."""
    return np.concatenate([np.ones(cfg["nB"]), cfg["disc_vals"],
                           [cfg["initial_beta_mild"], cfg["initial_beta_high"],
                            cfg["initial_delay"]]]).astype(np.float64)


def synthetic_config(n_data, forward, **kw):
    """SURVEY 8(d) synthetic inputs: run the forward model at the initial parameters
    (behavior == 1) and take deaths[i] = dead[i + maxDelay]; sigma by sigma_recipe.
    `forward(cfg, x) -> dead[size]` is supplied by the caller (the HIP derive entry point
    in bench.py, the oracle in CPU tests)."""
    probe = make_config(np.zeros(n_data), np.ones(n_data), **kw)
    dead = np.asarray(forward(probe, initial_params(probe)))
    md = probe["max_delay"]
    deaths = dead[md:md + n_data].copy()
    return make_config(deaths, sigma_recipe(deaths), **kw)


def target_beta(cfg, time):
    """Target::beta (mcmc.h:742-744, 806-808, 829-831)."""
    k = cfg["target_kind"]
    if k == "temp":
        return 1.0 / cfg["target_T"]
    if k == "cooling":
        import math
        return math.exp(time * cfg["target_slope"]) / cfg["target_Tinit"]
    return 1.0
