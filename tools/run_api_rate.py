#!/usr/bin/env python3
"""Does a whole MetropolisChain::run schedule through the C ABI (mcmc_hip_run: adjust groups, burn-in, thinning,
recording) reach bench.py's stepping rate?   python tools/run_api_rate.py [chains=65536]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cfg, R, xs, s = bench.build_workload(0, n, 1, 1)
s.run(0, 0, 5, 1000000)                                   # warm
for label, kw in [("adjust only: 50 groups of 20 steps", dict(n_samples=0, n_burnin=0, n_adjust=50, thinning=1000000)),
                  ("1000 samples, thinning 10, log-likelihood trace only", dict(n_samples=1000, n_burnin=0, n_adjust=0, thinning=10)),
                  ("1000 samples, thinning 50, points recorded + mean", dict(n_samples=1000, n_burnin=100, n_adjust=0, thinning=50, record_samples=True, compute_mean=True))]:
    t0 = time.perf_counter()
    s.run(**kw)
    dt = time.perf_counter() - t0
    steps = kw["n_samples"] + 20 * kw["n_adjust"]
    print(f"{label}: {steps} steps of {n} chains in {dt:.3f} s = {n * steps / dt / 1e6:.2f} M steps/s ({1e3 * dt / steps:.3f} ms per step)", flush=True)
s.close()
