#!/usr/bin/env python3
"""How long does the reference's sampler need to mix on the headline workload?  (bench.py's ESS leg
says `converged: false` after 6 000 steps; this probe runs the same chains for a long stretch and
follows R-hat, ESS and the posterior mean of betaMild / betaHigh / delay.)

  python tools/mixing_probe.py [chains=16384] [steps=200000] [every=100] [adjust=200]

The chains are bench.py's (same synthetic series, same +-1 % start, ids from 1), adapted for
`adjust` iterations of 20 steps, then stepped `every` steps at a time with the three scalars and
the log-likelihood downloaded after each stretch.  Statistics are over the SECOND HALF of what has
been collected so far (Gelman-Rubin over all chains; Geyer ESS per chain), printed every ~10 % of
the run, one JSON object per line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402
from mcmc_amd.sharding import ess_geyer                        # noqa: E402


def rhat(x):
    """x: [chains][m] -> Gelman-Rubin potential scale reduction"""
    m = x.shape[1]
    w = x.var(axis=1, ddof=1).mean()
    b_over_m = x.mean(axis=1).var(ddof=1)
    return float(np.sqrt(((m - 1) / m * w + b_over_m) / w)) if w > 0 else float("inf")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    adjust = int(sys.argv[4]) if len(sys.argv) > 4 else 200
    precision = os.environ.get("MCMC_PRECISION", "f64")      # f32_model: the same chains in the fast mode
    cfg, R, xs, s = bench.build_workload(0, n, 1, 1, precision=precision)
    dim = cfg["dim"]
    names = ["betaMild", "betaHigh", "delay"]
    t0 = time.perf_counter()
    s.run(0, 0, adjust, 1000000)
    print(json.dumps({"precision": precision, "adapted": adjust * 20, "seconds": time.perf_counter() - t0,
                      "stepsize_mean": float(np.mean(s.stepsizes()))}), flush=True)
    n_fetch = steps // every
    trace = np.empty((n, n_fetch, 4))
    report = max(1, n_fetch // 10)
    last_print = time.perf_counter()
    for i in range(n_fetch):
        s.step(every, 1.0, True)
        trace[:, i, :3] = s.params()[:, dim - 3:]
        trace[:, i, 3] = np.asarray(s.loglike()).reshape(n, -1).sum(axis=1)
        now = time.perf_counter()
        if (i + 1) % report == 0 or i + 1 == n_fetch:
            half = trace[:, (i + 1) // 2:i + 1]
            out = {"steps": (i + 1) * every, "seconds": now - t0, "window_points": half.shape[1],
                   "rhat": {nm: rhat(half[:, :, j]) for j, nm in enumerate(names)},
                   "rhat_loglike": rhat(half[:, :, 3]),
                   "mean": {nm: float(half[:, :, j].mean()) for j, nm in enumerate(names)},
                   "sd": {nm: float(half[:, :, j].std()) for j, nm in enumerate(names)},
                   "loglike_mean": float(half[:, :, 3].mean())}
            sub = half[:min(n, 2048)]
            out["ess_per_step"] = {nm: float(ess_geyer(sub[:, :, j]).mean() / (half.shape[1] * every))
                                   for j, nm in enumerate(names)}
            print(json.dumps(out), flush=True)
            last_print = now
        elif now - last_print > 45:
            print(json.dumps({"steps": (i + 1) * every, "seconds": now - t0}), flush=True)
            last_print = now
    acc = s.accept_counts().mean()
    print(json.dumps({"accepted_per_chain_since_start": float(acc), "truth": {"betaMild": 0.8, "betaHigh": 2.5, "delay": 23.0}}),
          flush=True)
    s.close()


if __name__ == "__main__":
    main()
