#!/usr/bin/env python3
"""Per-phase cycle breakdown of the MH step (needs a library built with -DMCMC_STAMPS):
   MCMC_HIP_LIB=.../libmcmc_hip_stamps.so python tools/phase_profile.py [chains] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
adjust = int(sys.argv[3]) if len(sys.argv) > 3 else 25
precision = os.environ.get("MCMC_PRECISION", "f64")        # or f32_model
regions = int(os.environ.get("MCMC_REGIONS", "0"))         # > 0: the multi-region workload (region days: MCMC_REGION_DAYS)
if regions:
    cfgs, R, xs, s = bench.build_region_workload(0, n, 1, 1, regions, int(os.environ.get("MCMC_REGION_DAYS", "500")))
else:
    cfg, R, xs, s = bench.build_workload(0, n, 1, 1, n_data=int(os.environ.get("MCMC_NDATA", "365")), precision=precision)
    if os.environ.get("MCMC_MAPPING"):
        s.set_mapping(os.environ["MCMC_MAPPING"])
    print("mapping:", s.mapping())
if adjust:
    s.run(0, 0, adjust, 1000000)
s.step(16, 1.0, True)
s.phase_cycles(reset=True)
import time
t0 = time.perf_counter()
s.step(steps, 1.0, True)
dt = time.perf_counter() - t0
ph = s.phase_cycles()
print(f"{n * steps / dt / 1e6:.2f} M steps/s (stamped build), acceptance {s.accept_counts().mean():.1f} per chain since reset")
counts = {k: ph.pop(k) for k in ("n_chunks", "n_chunks_e", "e_days", "e_hull")}
slowest = ph.pop("slowest_wave")
tot = sum(ph.values())
waves = n if s.mapping() == "wave" else (n + 63) // 64
for k, v in ph.items():
    print(f"{k:8s} {v / waves / steps:12.0f} cycles/wave/step  {100 * v / max(tot, 1):5.1f} %")
print(f"total    {tot / waves / steps:12.0f} cycles/wave/step")
print(f"slowest wave of a launch {slowest / steps:12.0f} cycles/step = {slowest / steps / (tot / waves / steps):.3f} x the mean (launches of {steps} steps)")
for k, v in counts.items():
    print(f"{k:10s} {v / waves / steps:10.1f} per wave per step")
