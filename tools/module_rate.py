#!/usr/bin/env python3
"""The reference's Python surface at scale: notebooks/covid.ipynb's flow (DiseaseData, DiseaseParams, DiseaseSpread,
SmoothnessPrior, State, TempTarget, ChainManager.run_all) through the pybind module `mcmc` with 65 536 chains on a
365-day series.  Prints the wall time of each call.   python tools/module_rate.py [chains=65536] [steps=400]"""
import contextlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mcmc_amd", "pymod"))
import numpy as np
import mcmc
import bench
from mcmc_amd import covid_config as CC

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400


@contextlib.contextmanager
def quiet():                                             # the manager prints one line per chain, like the reference
    sys.stdout.flush()
    saved, null = os.dup(1), os.open(os.devnull, os.O_WRONLY)
    os.dup2(null, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved); os.close(null)


def timed(label, f):
    t0 = time.perf_counter()
    with quiet():
        r = f()
    print(f"{label}: {time.perf_counter() - t0:.3f} s", flush=True)
    return r


# the bench's synthetic series (365 days) through the module's own classes
cfg0, R, xs, s0 = bench.build_workload(0, 64, 1, 1)
deaths = np.asarray(cfg0["deaths"], dtype=float)
s0.close()
sigma = CC.sigma_recipe(deaths)
DISC = np.array([[1, 9, 12, 29], [1, .6, .5, .4], [1, 1, 1, 1]], dtype=float)
data = mcmc.DiseaseData(np.array([deaths, sigma]), DISC)
data.initialBetaMild, data.initialBetaHigh, data.initialDelay = 0.8, 2.5, 23.0
p = mcmc.DiseaseParams()
for k, v in CC.ipynb_disease_params().items():
    setattr(p, k, v)
ds = timed("DiseaseSpread (10 M trajectories on the host)", lambda: mcmc.DiseaseSpread(data, p, 60000000, 7500.0, 100.0, 40, 60, 10000000))
ds.initialSpread = 0.02                                  # extension: distinct starting points
state = mcmc.State()
state.add(ds)
state.add(mcmc.SmoothnessPrior("behavior", 50.0, float(len(deaths) - data.fixBehaviorInAdvance)))
timed("State.init", state.init)
target = mcmc.TempTarget(1.0)
target.set_posterior(state)
cm = timed(f"ChainManager(target, {n})", lambda: mcmc.ChainManager(target, n))
t0 = time.perf_counter()
with quiet():
    cm.run_all(steps, 10, 10, 4)                         # bound to run_all_adjust(nSteps, nAdjust, thinning, nThread)
dt = time.perf_counter() - t0
total = steps + 200
print(f"run_all({steps}, nAdjust 10, thinning 10): {dt:.3f} s for {total} steps of {n} chains = {n * total / dt / 1e6:.2f} M steps/s", flush=True)
ll = timed("all_loglikes", cm.all_loglikes)
print("saved points per chain:", ll.shape[1], " finite:", bool(np.all(np.isfinite(ll))), " distinct starts:", len(np.unique(ll[:, 0])))
c = timed("get_chain(12345).get_loglikes", lambda: np.asarray(cm.get_chain(min(12345, n - 1)).get_loglikes()))
