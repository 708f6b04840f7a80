#!/usr/bin/env python3
"""Headline benchmark: Metropolis-Hastings steps/s on the covid model.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one Target::step (mcmc.h:750-783) for EVERY chain of the workload: proposal,
compartment-model forward simulation, likelihood, prior, accept/reject.  Workload =
BASELINE.json configs[2]: covid.h model, 65536 independent chains PER GPU, 365-day synthetic
series, chain-sharded by chain-id range (weak scaling, no collective on the data path).
`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
(mcmc_amd/launcher.py), before this process touches HIP or torch; under torch.distributed.run the
process is one of the ranks.  The chains first run `--adjust` iterations of the reference's
step-size adaptation (untimed, like the adjust phase that precedes sampling in
MetropolisChain::run), then W warm-up steps.  All inputs are resident in HBM before the timed
region.  Arithmetic is FP64 as in the reference (`using Float = double`, mcmc.h:43).

Other BASELINE.json configurations through the same line (`--config`): c1 = configs[1] (1 024 chains x 200
days: a small ensemble, 16 waves on a chip of 1 024 SIMDs -- `config.occupancy` says so), c2 = the default,
c3 = configs[3] (64-rung ladders), c4 = configs[4] (50 regions x 500 days x 131 072 chains).  `--total-chains N`
splits a FIXED total over the ranks (strong scaling; the default gives every GPU 65 536 chains = weak).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mcmc_amd import covid_config as CC  # noqa: E402
from mcmc_amd import hip as H  # noqa: E402   (importing it loads nothing; H.lib() does)
from mcmc_amd import launcher  # noqa: E402

PEAK_FP64_VECTOR_TFLOPS = 78.6      # MI355X FP64 vector (FMA = 2 flop): 256 CU x 4 SIMD x 16 lanes/clk x 2 x 2.4 GHz
PEAK_FP32_VECTOR_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (vector)"
N_DATA = 365
CHAINS_PER_GPU = 65536
STEPS_PER_LAUNCH = 16               # MCMC_STEPS_PER_LAUNCH of the library
# SURVEY 8d names initial betas (2, 10) (covid.ipynb); with the current covid.h and the notebook's
# constants that epidemic infects the whole population within the series and the forward model
# overflows to NaN, so the synthetic counts are generated at (0.8, 2.5): DESIGN.md section 5.
BETA_MILD, BETA_HIGH = 0.8, 2.5


def build_workload(device, n_chains, first_chain_id, contract, n_data=N_DATA, precision="f64", spread=0.01):
    """Synthetic inputs per SURVEY 8(d): response table by the host precompute, counts by the
    forward model at the initial parameters, sigma by the notebook's recipe, chains = initial
    parameters perturbed by +-1 % (pcg32(0,0)), chain ids first_chain_id.."""
    dp = CC.ipynb_disease_params()
    R = H.make_response_tables(dp)                       # host, 10M trajectories, a few seconds

    def forward(cfg, x):
        s = H.CovidSampler(cfg, R, device)
        dead = s.derive(x[None, :])[0, 7]
        s.close()
        return dead

    cfg = CC.synthetic_config(n_data, forward, disease_params=dp, initial_beta_mild=BETA_MILD,
                              initial_beta_high=BETA_HIGH, contract=contract)
    x0 = CC.initial_params(cfg)
    # one global IC stream so that chain i has the same start on any sharding; a rank jumps the generator to
    # its first chain instead of drawing everybody else's (tests/test_host_cpu.py: jumped == sliced)
    xs = H.perturb_ics(x0, n_chains, spread, first_chain=first_chain_id - 1)
    sampler = H.CovidSampler(cfg, R, device, precision=precision)
    sampler.set_chains(xs, first_chain_id=first_chain_id)
    return cfg, R, xs, sampler


def build_region_workload(device, n_chains, first_chain_id, contract, n_regions, n_data, spread=0.01):
    """BASELINE.json configs[4] (extension, include/mcmc_hip.h at mcmc_hip_set_regions): n_regions regions per
    chain over a long horizon (n_data days; maxDelay 40 + nPredict 60 on top: beyond the 500-day response
    table), each a complete DiseaseSpread block -- three behaviour nodes, one breakpoint value, its delay --
    plus the two shared betas.  Regions differ in population and hospital capacity; their counts come from
    the forward model at their own constants."""
    dp = CC.ipynb_disease_params()
    R = H.make_response_tables(dp)
    kw = dict(disease_params=dp, initial_beta_mild=BETA_MILD, initial_beta_high=BETA_HIGH, contract=contract,
              fix_behavior_in_advance=n_data - 3, disc_days=(20,), disc_vals=(0.6,))

    def forward(cfg, x):
        s = H.CovidSampler(cfg, R, device)
        dead = s.derive(x[None, :])[0, 7]
        s.close()
        return dead

    cfgs = [CC.synthetic_config(n_data, forward, pop_size=int(6e7 / (1 + 0.25 * (r % 8))), cap0=7500.0 / (1 + r % 5),
                                cap_incr_rate=100.0 - r % 50, **kw) for r in range(n_regions)]
    x0 = np.concatenate([CC.initial_params(c) for c in cfgs])
    xs = H.perturb_ics(x0, n_chains, spread, first_chain=first_chain_id - 1)
    sampler = H.CovidSampler(cfgs[0], R, device)
    sampler.set_regions(cfgs)
    sampler.set_chains(xs, first_chain_id=first_chain_id)
    return cfgs, R, xs, sampler


def host_threads():
    """Threads of the CPU leg: the cores this process may run on, capped by the box's per-GPU CPU
    share (the GPU box exports OMP_NUM_THREADS = 16 for a one-GPU slot of a much larger machine and
    asks worker pools to stay within it)."""
    visible = len(os.sched_getaffinity(0))
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    used = min(visible, share) if share > 0 else visible
    return max(1, min(used, 256)), visible


def cpu_baseline(cfg, R10, xs, stepsizes, contract, seconds=12.0):
    """The oracle (a CPU port of the same path, oracle/mcmc_oracle.c) timed on this box's host
    cores on a bounded sample of the same workload, from the same adapted state.  Two figures
    (SURVEY 8d): LEAN (response table shared by pointer -- the faster one, `value`) and FAITHFUL
    (every proposal copies the 8.8 MB trajectory arrays and re-forms the running sums from them,
    as DiseaseSpread::step's copy() does, covid.h:290,501,376-386).  Reported, not the target."""
    from oracle import oracle as O
    # ABI curve order -> oracle's 11-curve order (asymptomatic unused by the model)
    R11 = np.zeros((O.NCURVES, R10.shape[1]))
    for k, c in enumerate([3, 4, 0, 2, 5, 7, 6, 8, 9, 10]):
        R11[c] = R10[k]
    threads, visible = host_threads()
    tgt = O.make_target("temp", 1.0)
    n_chains = min(4 * threads, xs.shape[0])
    ss = stepsizes[:n_chains]

    def timed(model, label):
        # calibrate, then one run sized for `seconds`; a run that still came out under 10 s (the calibration
        # carries the pool's start-up) is repeated once with the step count scaled up
        cal = O.run_chains(model, tgt, xs[:n_chains], ss, 4, 0, 0, 1000, n_threads=threads)
        n_samples = int(max(4, min(40000, seconds / max(cal["wall"] / 4, 1e-6))))
        res = O.run_chains(model, tgt, xs[:n_chains], ss, n_samples, 0, 0, 1000, n_threads=threads)
        if res["wall"] < 10.0:
            n_samples = int(min(80000, n_samples * seconds / max(res["wall"], 1e-3))) + 1
            res = O.run_chains(model, tgt, xs[:n_chains], ss, n_samples, 0, 0, 1000, n_threads=threads)
        return res["steps"] / res["wall"], f"{n_chains} chains x {n_samples} steps, {res['wall']:.1f} s ({label})"

    lean, lean_s = timed(O.Model(cfg, R11, contract=contract), "lean: table shared by pointer")
    out = {"value": lean, "unit": "MH steps/s", "cores": threads, "cores_visible": visible, "kind": "port",
           "sample": f"{lean_s} of the same {cfg['n_data']}-day workload from the same adapted state "
                     f"(oracle/mcmc_oracle.c, -O3 -mavx2 -mfma, {threads} pthreads = the box's CPU share per GPU)"}
    try:
        grid = O.traj_generate(O.DiseaseParams(**{k: float(v) for k, v in CC.ipynb_disease_params().items()}))
        faithful, faithful_s = timed(O.Model(cfg, O.traj_response(grid, R11.shape[1]), grid=grid, contract=contract,
                                             faithful=1), "faithful: 8.8 MB copy per proposal + getDay sums")
        out["faithful"] = {"value": faithful, "unit": "MH steps/s", "sample": faithful_s}
    except Exception as e:                                # the faithful leg is extra information, never fatal
        out["faithful"] = {"value": None, "error": repr(e)}
    return out


def ess_leg(cfg, R, device, contract, xs, stepsizes, first_id, n_steps, thinning):
    """ESS per MH step of delay / betaMild / betaHigh and R-hat from a recorded run of a
    subset of this rank's chains, continued from the adapted state.  Returns the ADDITIVE pieces
    (one vector, summed over ranks by one all-reduce) -- see ess_finish.  Chains are bit-identical on
    the CPU path, so ESS/s on either side = ESS per step x that side's steps/s."""
    from mcmc_amd.sharding import ess_geyer, moment_sums
    n = xs.shape[0]
    s = H.CovidSampler(cfg, R, device)
    s.set_chains(xs, stepsizes=stepsizes, first_chain_id=first_id)
    s.run(n_steps, 0, 0, thinning, record_samples=True, compute_mean=False)
    n_saved = s.n_saved
    dim = cfg["dim"]
    tail = np.stack([s.sample_all(k)[:, dim - 3:] for k in range(n_saved)], axis=1)    # [n][n_saved][3]
    acc = float(s.accept_counts().sum())
    s.close()
    ess = [float(ess_geyer(tail[:, :, j]).sum()) for j in range(3)]                  # Geyer, per chain, summed
    return np.concatenate([[float(n), acc], ess, moment_sums(tail)]), n_saved


def ess_finish(tot, n_saved, n_steps, thinning, ranks):
    from mcmc_amd.sharding import rhat_ess
    names = ["betaMild", "betaHigh", "delay"]
    n = tot[0]
    rhat, _ = rhat_ess(tot[5:], 3, n_saved)                                          # Gelman-Rubin over all chains
    converged = bool(np.all(np.isfinite(rhat)) and float(np.max(rhat)) < 1.1)
    out = {"converged": converged, "acceptance": float(tot[1] / (n * n_steps)), "rhat": dict(zip(names, map(float, rhat))),
           # Geyer's estimate inside each chain's own trace, averaged over chains: how fast a chain decorrelates
           # around the level it is at.  It is an ESS of the posterior only if `converged`.
           "ess_per_step_within_chain": {nm: float(tot[2 + j] / (n * n_steps)) for j, nm in enumerate(names)},
           "sample": f"{int(n)} chains ({ranks} rank(s)) x {n_steps} steps after the adaptation, every {thinning}th point, "
                     "Geyer initial positive sequence per chain, summed over chains; R-hat over all of them"}
    if converged:
        out["ess_per_step"] = dict(out["ess_per_step_within_chain"])
    else:
        out["note"] = ("R-hat >= 1.1: the chains have not reached stationarity in this stretch (a random-walk sampler in "
                       "this many dimensions needs far more than a benchmark's steps), so the within-chain figure is NOT "
                       "a posterior ESS and no covid ESS/s is quoted; `ess_toy` holds a converged ESS/s (GPU and CPU) on "
                       "the targets of BASELINE configs[0], where the reference's sampler mixes")
    return out


def toy_ess_leg(device, n_chains, threads):
    """ESS/s -- the second half of BASELINE.json's metric -- on workloads where the reference's sampler reaches
    stationarity within a benchmark: the toy targets of BASELINE configs[0] under MetropolisChain::run
    (mcmc.h:1005-1138) through mcmc_hip_toy_run, n_chains chains in one launch; the CPU oracle runs the same
    chains (bit-identical, tests/test_gpu_toy.py) on the box's host threads.  ESS = Geyer per chain over the
    recorded post-burn-in points, summed over chains; ESS/s = that sum / wall time of the WHOLE run (adjust
    phase included, upload and download included on the GPU side)."""
    from concurrent.futures import ThreadPoolExecutor
    from mcmc_amd.sharding import ess_geyer, moment_sums, rhat_ess
    from oracle import oracle as O
    legs = {}
    n_rec = min(4096, n_chains)
    rs = np.random.default_rng(11)
    # Rosenbrock: a random walk explores the curved ridge slowly, so its chains start from EXACT draws of the target
    # (x ~ N(1, 10), y | x ~ N(x^2, 10 / difficulty)) and are stationary from the first step; MyLike1 starts every
    # chain at (1, 1, 1) like test.cpp:36-38 and discards a burn-in
    xr = 1 + np.sqrt(10.0) * rs.standard_normal(n_chains)
    X_ros = np.stack([xr, xr ** 2 + np.sqrt(0.1) * rs.standard_normal(n_chains)], axis=1)
    # name, starts, (nSamples, nBurnin, nAdjust, thinning), difficulty, scalars of the ESS
    for name, X, sched, diff, k_use in [("mylike1", np.ones((n_chains, 3)), (10000, 2000, 500, 5), 1.0, 3),
                                         ("rosenbrock", X_ros, (600000, 100000, 500, 200), 100.0, 2)]:
        ns, nb, na, th = sched
        steps_per_chain = 20 * na + ns
        H.toy_run(name, X[:64], 10, 0, 1, 1, difficulty=diff, first_chain_id=1, device=device, record=False)   # module load
        gpu_wall = 1e30
        for _ in range(3):                                                   # best of three: the runs are milliseconds
            t0 = time.perf_counter()
            H.toy_run(name, X, ns, nb, na, th, difficulty=diff, first_chain_id=1, device=device, record=False)
            gpu_wall = min(gpu_wall, time.perf_counter() - t0)
        rec = H.toy_run(name, X[:n_rec], ns, nb, na, th, difficulty=diff, first_chain_id=1, device=device)
        pts = rec["samples"][:, :, :k_use]                                   # [n_rec][n_saved][k]
        n_saved = pts.shape[1]
        ess = np.array([ess_geyer(pts[:, :, j]).mean() for j in range(k_use)])   # per chain
        rhat, _ = rhat_ess(moment_sums(pts), k_use, n_saved)
        # CPU: the oracle's run of the same schedule, one chain per call, `threads` at a time; the chain count is
        # sized so that the timed sample lasts about two seconds
        fn = {"mylike1": lambda cid: O.mylike1_run(cid, ns, nb, na, th),
              "rosenbrock": lambda cid: O.rosenbrock_run(diff, 1.0, cid, X[cid - 1], ns, nb, na, th)}[name]

        def cpu_run(n_cpu):
            t0 = time.perf_counter()
            with ThreadPoolExecutor(threads) as ex:
                res = list(ex.map(fn, range(1, n_cpu + 1)))
            return res, time.perf_counter() - t0
        cpu, cpu_wall = cpu_run(4 * threads)
        n_cpu = int(min(n_chains, max(4 * threads, 4 * threads * 2.0 / max(cpu_wall, 1e-4))))
        if n_cpu > 4 * threads:
            cpu, cpu_wall = cpu_run(n_cpu)
        same = all(np.array_equal(cpu[c][1][:, :k_use], pts[c]) for c in range(min(n_cpu, n_rec, 256)))
        worst = float(ess.min())
        legs[name] = {"converged": bool(np.all(np.isfinite(rhat)) and float(np.max(rhat)) < 1.1),
                      "rhat_max": float(np.max(rhat)), "ess_per_chain_worst_coordinate": worst,
                      "ess_per_step": worst / steps_per_chain,
                      "gpu": {"chains": n_chains, "wall_s": gpu_wall, "steps_per_s": n_chains * steps_per_chain / gpu_wall,
                              "ess_per_s": worst * n_chains / gpu_wall},
                      "cpu": {"chains": n_cpu, "threads": threads, "wall_s": cpu_wall,
                              "steps_per_s": n_cpu * steps_per_chain / cpu_wall, "ess_per_s": worst * n_cpu / cpu_wall},
                      "gpu_over_cpu_ess_per_s": (n_chains / gpu_wall) / (n_cpu / cpu_wall),
                      "cpu_chains_bit_identical_to_gpu": bool(same),
                      "schedule": f"run({ns}, {nb}, {na}, {th}): {steps_per_chain} MH steps per chain, {n_saved} recorded points; "
                                  f"ESS (Geyer, per chain, worst coordinate, mean over chains) from the first {n_rec} chains; "
                                  "GPU wall = best of three whole mcmc_hip_toy_run calls (upload, one launch, download of the end state)"}
    return legs


def side_leg(cfg, R, device, start, first_id, args, precision, contract):
    """The same workload under another arithmetic, from the main leg's adapted state (parameters and
    step sizes; chain ids as there): warm-up and timed steps of its own.  Reported beside the
    headline, never as `value`."""
    cfg2 = dict(cfg)
    cfg2["contract"] = contract
    s = H.CovidSampler(cfg2, R, device, precision=precision)
    if precision == "f64":
        s.set_mapping(args.mapping)                  # the main leg's; the fast mode runs one chain per lane only
    s.set_chains(start[0], stepsizes=start[1], first_chain_id=first_id)
    steps = max(STEPS_PER_LAUNCH, args.steps // 2 // STEPS_PER_LAUNCH * STEPS_PER_LAUNCH)
    s.step(args.warmup, 1.0, True)
    s.pair_count(reset=True)
    t0 = s.last_timing()
    w0 = time.perf_counter()
    s.step(steps, 1.0, True)
    wall = time.perf_counter() - w0
    t1 = s.last_timing()
    pairs, pairs_exec = s.pair_count(), s.executed_pair_count()
    n = start[0].shape[0]
    mapping = s.mapping()
    s.close()
    ksec, launches = t1["step_kernel_s"] - t0["step_kernel_s"], t1["launches"] - t0["launches"]
    flops = 16.0 * pairs + (10.0 * cfg["n_data"] + 7.0 * cfg["nB"]) * n * steps
    return {"value": n * steps / wall, "unit": "MH steps/s", "steps": steps, "ms_per_step": 1e3 * wall / steps,
            "kernel_ms_per_launch": 1e3 * ksec / max(launches, 1), "achieved": flops / ksec / 1e12,
            "executed_tflops": 14.0 * pairs_exec / ksec / 1e12, "unit_roofline": "TFLOP/s", "mapping": mapping}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--chains-per-gpu", type=int, default=None, help=f"weak scaling (default {CHAINS_PER_GPU} per GPU)")
    ap.add_argument("--total-chains", type=int, default=None,
                    help="STRONG scaling: this many chains in total, split over the ranks by contiguous chain-id range")
    ap.add_argument("--n-data", type=int, default=None, help=f"data days of the single-region series (default {N_DATA})")
    ap.add_argument("--config", default=None, choices=["c1", "c2", "c3", "c4"],
                    help="BASELINE.json configs[1..4]: c1 = 1024 chains x 200 days, c2 = the default (65536 x 365), "
                         "c3 = 64-rung ladders, c4 = 50 regions x 500 days x 131072 chains; explicit flags win")
    ap.add_argument("--contract", type=int, default=int(os.environ.get("MCMC_BENCH_CONTRACT", "1")))
    ap.add_argument("--adjust", type=int, default=int(os.environ.get("MCMC_BENCH_ADJUST", "25")),
                    help="adjust-phase iterations (20 MH steps each, mcmc.h:1083-1119) run untimed before the warm-up")
    ap.add_argument("--start-spread", type=float, default=None,
                    help="relative spread of the chains' starting points around the generating parameters "
                         "(State::perturb's rule, mcmc.h:306-314); default 0.01 as SURVEY 8d, 0.001 with --regions")
    ap.add_argument("--mapping", default="auto", choices=["auto", "lane", "wave"],
                    help="chains onto the GPU: one per lane, one per wavefront (small ensembles), or the library's choice "
                         "(wave up to 16384 chains); no effect on results (include/mcmc_hip.h, mcmc_hip_set_mapping)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fast-leg", action="store_true", help="skip the FP32 fast-mode leg (N=1 only)")
    ap.add_argument("--no-strict-leg", action="store_true", help="skip the strict (no FMA contraction) leg (N=1 only)")
    ap.add_argument("--precision", default="f64", choices=["f64", "f32_model", "f32_packed"],
                    help="arithmetic of the MAIN leg; f32_model (fast mode) is for profiling that kernel only: "
                         "the line then says dtype f32+f64 and is not the headline metric")
    ap.add_argument("--ess-chains", type=int, default=4096, help="chains of the ESS leg (0: skip it)")
    ap.add_argument("--no-toy-ess", action="store_true", help="skip the converged-ESS/s leg on the toy targets (N=1 only)")
    ap.add_argument("--ess-steps", type=int, default=2000)
    # BASELINE.json configs[3] (extension): ladders of K temperatures, whole ladders per GPU
    ap.add_argument("--ladder", type=int, default=0, help="temperatures per ladder (0: no tempering)")
    ap.add_argument("--ladder-tmax", type=float, default=16.0)
    ap.add_argument("--swap-every", type=int, default=8, help="MH steps between exchange rounds")
    # BASELINE.json configs[4] (extension): multi-region model, e.g. --regions 50 --chains-per-gpu 131072
    ap.add_argument("--regions", type=int, default=0, help="regions per chain (0: the single-region headline workload)")
    ap.add_argument("--region-days", type=int, default=500, help="data days of the multi-region workload")
    # rehearsal knobs: several ranks on ONE card need the gloo backend and a fixed device
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--device", type=int, default=None, help="HIP ordinal for every rank (default: LOCAL_RANK)")
    args = ap.parse_args()
    preset = {"c1": dict(chains_per_gpu=1024, n_data=200), "c2": {}, "c3": dict(ladder=64),
              "c4": dict(regions=50, chains_per_gpu=131072)}.get(args.config or "c2")
    for k, v in preset.items():
        if getattr(args, k) in (None, 0):
            setattr(args, k, v)
    if args.total_chains is not None and args.chains_per_gpu is not None:
        raise SystemExit("bench.py: --total-chains (strong scaling) and --chains-per-gpu (weak) exclude each other")
    if args.chains_per_gpu is None:
        args.chains_per_gpu = CHAINS_PER_GPU
    if args.n_data is None:
        args.n_data = N_DATA
    if args.start_spread is None:
        # a region's 600-day epidemic answers a 1 % change of the betas with a log-likelihood of -5e7: started
        # there the timed steps are a wild transient (acceptance 0.17); 0.1 % starts the chains near the mode
        args.start_spread = 0.001 if args.regions else 0.01

    # ---- one process per GPU: become the launcher if this is a plain `--gpus N` call (nothing in
    # this process has touched HIP or torch yet) ----
    overlays = launcher.plan_launch(args.gpus, os.environ)
    if overlays is not None:
        raise SystemExit(launcher.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], overlays))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    import torch
    device = args.device if args.device is not None else (local_rank if world > 1 else 0)
    n_dev = H.lib().mcmc_hip_device_count()
    if n_dev < 1:
        raise SystemExit("bench.py: no GPU visible; the hot path has no CPU fallback")
    if device >= n_dev:
        raise SystemExit(f"bench.py: rank {rank} wants HIP device {device} but only {n_dev} are visible "
                         f"(--gpus {args.gpus}); use --device to rehearse several ranks on one card")
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(device)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")

    from mcmc_amd.sharding import shard_range
    if args.total_chains is not None:                    # strong scaling: a fixed total, mcmc.h:1487's ids i+1
        lo, hi = shard_range(args.total_chains, rank, world)
        if args.ladder and (lo % args.ladder or hi % args.ladder):
            raise SystemExit("bench.py: --total-chains must give every rank whole ladders")
        n_chains, first_id, n_total = hi - lo, 1 + lo, args.total_chains
        if n_chains < 1:
            raise SystemExit(f"bench.py: rank {rank} of {world} gets no chain out of {args.total_chains}")
    else:
        n_chains = args.chains_per_gpu
        first_id = 1 + rank * n_chains                   # contiguous chain-id ranges
        n_total = world * n_chains
    n_data = args.n_data
    if args.regions:
        if args.ladder or args.precision != "f64":
            raise SystemExit("bench.py: --regions runs without --ladder, in f64")
        n_data = args.region_days
        cfgs, R, xs, s = build_region_workload(device, n_chains, first_id, args.contract, args.regions, n_data, args.start_spread)
        cfg = dict(cfgs[0])
        cfg["dim"] = s.dim
    else:
        cfg, R, xs, s = build_workload(device, n_chains, first_id, args.contract, n_data=n_data, precision=args.precision, spread=args.start_spread)
        if args.mapping != "auto":
            s.set_mapping(args.mapping)
    fast_main = args.precision != "f64"
    peak = PEAK_FP32_VECTOR_TFLOPS if fast_main else PEAK_FP64_VECTOR_TFLOPS

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if args.adjust > 0:
        s.run(0, 0, args.adjust, 1000000)                # step-size adaptation only, nothing recorded
    if args.ladder:
        temps = np.geomspace(1.0, args.ladder_tmax, args.ladder)
        s.set_ladder(temps, swap_seed=1, first_ladder=(first_id - 1) // args.ladder)
    s.step(args.warmup, 1.0, True)
    adapted = (s.params(), s.stepsizes()) if not args.regions else None   # start of the CPU leg and the side legs
    s.pair_count(reset=True)
    t_before = s.last_timing()
    diag_before = s.diagnostics()                        # additive sums of this shard (untimed)
    sync()
    t0 = time.perf_counter()
    if args.ladder:
        s.run_tempered(args.steps // args.swap_every, args.swap_every)
        if args.steps % args.swap_every:
            s.step(args.steps % args.swap_every, 1.0, True)
    else:
        s.step(args.steps, 1.0, True)                    # the library syncs its stream per launch
    sync()
    dt = time.perf_counter() - t0
    t_after = s.last_timing()
    if dist is not None:
        tt = torch.tensor([dt], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # cross-chain diagnostics: the additive vector of every shard, summed over the
    # ranks with ONE all-reduce (RCCL under nccl) -- outside the timed region, not on the data path
    from mcmc_amd.sharding import allreduce_sum
    diag_after = s.diagnostics()
    local = np.concatenate([[diag_after[0] - diag_before[0], float(n_chains)], diag_after[2:]])
    tot = allreduce_sum(local)

    # ESS / R-hat leg on every rank (its first --ess-chains chains), pieces summed by one more all-reduce
    ess_tot = None
    if args.ess_chains > 0 and not args.ladder and not fast_main and not args.regions:
        ne = min(args.ess_chains, n_chains)
        pieces, ess_saved = ess_leg(cfg, R, device, args.contract, adapted[0][:ne], adapted[1][:ne], first_id,
                                    args.ess_steps, 5)
        ess_tot = allreduce_sum(pieces)

    if rank == 0:
        total_steps = n_total * args.steps
        value = total_steps / dt
        n_simd = 4 * torch.cuda.get_device_properties(device).multi_processor_count
        n_waves = n_chains if s.mapping() == "wave" else (n_chains + 63) // 64
        # roofline of the dominant kernel (mh_step_kernel) on this rank
        ksec = t_after["step_kernel_s"] - t_before["step_kernel_s"]
        launches = t_after["launches"] - t_before["launches"]
        pairs = s.pair_count()                            # sum over chains/steps, SURVEY 8d work unit
        pairs_exec = s.executed_pair_count()
        nB = cfg["nB"]
        flops = 16.0 * pairs + (10.0 * n_data + 7.0 * nB) * n_chains * args.steps   # the ALGORITHMIC count
        flops_exec = 14.0 * pairs_exec                    # the multiply-adds the kernel ran: 7 families, visible days
        achieved = flops / ksec / 1e12
        # HBM-side traffic: not measured in this run (PMC counters need rocprofv3); taken from the
        # committed profile of the same kernel and scaled per chain-step, with its source named
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_fast_mode.json" if fast_main else "traffic_latest.json")
        if os.path.exists(tpath) and n_data == N_DATA and not args.regions and s.mapping() == "lane":     # bytes per chain-step depend on the series length
            try:
                tj = json.load(open(tpath))
                per_chain_step = tj["hbm_bytes_per_launch"] / (tj.get("chains", CHAINS_PER_GPU) * tj.get("steps_per_launch", STEPS_PER_LAUNCH))
                traffic = per_chain_step * n_chains * args.steps / max(launches, 1)
                traffic_src = {"from_profile": os.path.relpath(tpath, ROOT), "tag": tj.get("tag"),
                               "profiled_launch": f"{tj.get('steps_per_launch', STEPS_PER_LAUNCH)} steps x {tj.get('chains', CHAINS_PER_GPU)} chains",
                               "scaled_to": f"{args.steps / max(launches, 1):g} steps x {n_chains} chains per launch (bytes are per chain-step)"}
            except Exception:
                traffic, traffic_src = None, None
        out = {
            "metric": "MH steps/sec (whole node) on covid model, 64k chains; ESS/sec vs CPU TBB",   # BASELINE.json's string; ESS/s is in "ess"
            "value": value, "unit": "MH steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "strong" if args.total_chains is not None else "weak", "vs_baseline": None,
            "dtype": ("f32 %s (renewal convolution) + f64 -- FAST MODE, not the headline metric" % ("MFMA" if args.precision == "f32_model" else "packed FMA")) if fast_main else "f64",
            "data": "synthetic",
            "config": {"workload": (f"multi-region covid.h model (extension), {args.regions} regions x {n_data} data days, "
                                    f"{n_chains} independent chains per GPU, systematic scan over the regions, chain-sharded "
                                    "(BASELINE.json configs[4]) -- NOT the headline workload") if args.regions else
                                   (f"covid.h model, {n_chains} independent chains per GPU, {n_data}-day synthetic "
                                    "series, chain-sharded (BASELINE.json configs[%s])" %
                                    ("1" if (n_chains, n_data) == (1024, 200) else "3" if args.ladder else
                                     "2" if n_data == N_DATA else "2 shape, other length")),
                       **({"regions": args.regions, "dim_per_region": cfgs[0]["dim"]} if args.regions else {}),
                       "mapping": s.mapping() + (" (one chain per wavefront, mcmc_wave_kernel.hip.h)" if s.mapping() == "wave" else " (one chain per lane)"),
                       "chains_per_gpu": n_chains, "chains_total": n_total, "n_data": n_data, "dim": cfg["dim"],
                       # one chain per lane: how much of the chip this ensemble can occupy at all
                       "occupancy": {"waves_per_gpu": n_waves, "simds_per_gpu": n_simd, "frac": min(1.0, n_waves / n_simd)},
                       "size": cfg["size"], "fp_contract": "fma" if args.contract else "strict",
                       "fp_contract_models": ("the dominant effect of a contracting build of the reference (-march=native via "
                                              "enoki_set_native_flags, CMakeLists.txt:98; gcc's default, clang >= 14): project()'s "
                                              "`dest[d+k] += n*cumsum` (covid.h:384) is one fused multiply-add; every other expression is a "
                                              "separate mul / add (covid.h:368,462,491 stay unfused: not any one compiler's exact output)") if args.contract
                                             else "the C++ source semantics: every projection term a multiply, then an add",
                       "adjust_iterations_before_timing": args.adjust, "start_spread": args.start_spread,
                       **({"ladder": {"temperatures": args.ladder, "t_max": args.ladder_tmax,
                                      "swap_every": args.swap_every}} if args.ladder else {}),
                       "parallelism": f"chain-shard x{world}"},
            "roofline": {"bound": "valu", "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "mh_step_wave_kernel" if s.mapping() == "wave" else "mh_step_kernel",
                         "kernel_ms_per_launch": 1e3 * ksec / max(launches, 1), "launches": launches,
                         "steps_per_launch": args.steps / max(launches, 1),
                         "algorithmic_flops_per_launch": flops / max(launches, 1),
                         "frac_of_fp32_vector_peak": achieved / PEAK_FP32_VECTOR_TFLOPS,
                         # `achieved` / `frac` follow the contract's definition (SURVEY 8d's algorithmic FLOP per step x the
                         # steps of a launch / the launch's duration); what the FMA pipe itself executes is smaller:
                         "frac_counts": "algorithmic FLOP of the reference's formulation (8 compartments, full horizon), SURVEY 8d",
                         "achieved_executed": flops_exec / ksec / 1e12, "frac_executed": flops_exec / ksec / 1e12 / peak,
                         # what the kernel's FMA pipe really did: the reference-algorithmic figure above counts 8
                         # compartments over maxDelay+nPredict+nData days, the kernel runs 7 over maxDelay+nData
                         "executed": {"tflops": flops_exec / ksec / 1e12, "frac": flops_exec / ksec / 1e12 / peak,
                                      "flops_per_launch": flops_exec / max(launches, 1),
                                      "what": "14 flop x (day, lag) pairs actually run (7 accumulator families, days "
                                              "the log-likelihood can see); capacity corrections, proposal and the "
                                              "sequential day loop not counted"}},
        }
        if args.regions:
            out["metric"] = "MH steps/sec on the multi-region covid model (extension leg, BASELINE.json configs[4])"
            out["roofline"]["traffic"] = None
            out["roofline"]["traffic_source"] = None
        n_all, dim = tot[1], cfg["dim"]

        def mean_sd(k):                                   # cross-chain mean / sd from (sum x, sum x^2)
            mu = tot[k] / n_all
            return [float(mu), float(np.sqrt(max(tot[k + 1] / n_all - mu * mu, 0.0)))]
        out["diagnostics"] = {"chains": int(n_all), "ranks": world, "acceptance_timed_steps": float(tot[0] / (n_all * args.steps)),
                              "loglike_mean_sd": mean_sd(2),
                              **{nm: mean_sd(4 + 2 * (dim - 3 + j)) for j, nm in enumerate(["betaMild_mean_sd", "betaHigh_mean_sd", "delay_mean_sd"])},
                              "how": f"mcmc_hip_diagnostics per rank, one all-reduce(SUM) of {len(local)} doubles "
                                     f"({'RCCL' if (dist is not None and args.backend == 'nccl') else 'single rank' if dist is None else 'gloo'})"}
        n_keep = max(256, min(4096, n_chains))
        if world == 1 and not args.no_cpu_baseline and not args.regions:
            cb = cpu_baseline(cfg, R, adapted[0][:n_keep], adapted[1][:n_keep], args.contract)
            out["cpu_baseline"] = cb
            out["gpu_over_cpu"] = value / cb["value"]
        side = world == 1 and not args.ladder and not fast_main and not args.regions
        if side and args.contract and not args.no_strict_leg:
            # the C++ source semantics: every projection term a multiply and an add (cfg.contract = 0)
            out["strict"] = side_leg(cfg, R, device, adapted, first_id, args, "f64", 0)
            out["strict"].update({"fp_contract": "strict", "dtype": "f64", "peak": PEAK_FP64_VECTOR_TFLOPS,
                                  "frac": out["strict"]["achieved"] / PEAK_FP64_VECTOR_TFLOPS,
                                  "slowdown_vs_fma": value / out["strict"]["value"],
                                  "parity": "bit-exact against the oracle built with contract=0 (tests/test_gpu_parity.py)"})
        if side and args.contract and not args.no_fast_leg:
            out["fp32_fast_mode"] = side_leg(cfg, R, device, adapted, first_id, args, "f32_model", 1)
            out["fp32_fast_mode"].update({"dtype": "f32 MFMA v_mfma_f32_32x32x2_f32 (renewal convolution) + f64 (everything else)",
                                          "peak": PEAK_FP32_VECTOR_TFLOPS,
                                          "frac": out["fp32_fast_mode"]["achieved"] / PEAK_FP32_VECTOR_TFLOPS,
                                          "speedup_over_f64": out["fp32_fast_mode"]["value"] / value,
                                          "parity": "statistical (tests/test_gpu_fast_mode.py), not bit-exact: never the headline value"})
        if ess_tot is not None:
            e = ess_finish(ess_tot, ess_saved, args.ess_steps, 5, world)
            if e["converged"]:
                worst = min(e["ess_per_step"].values())
                e["gpu_ess_per_s"] = worst * value        # slowest-mixing of the three coordinates, whole job
                if "cpu_baseline" in out:
                    e["cpu_ess_per_s"] = worst * out["cpu_baseline"]["value"]
            out["ess"] = e
        if world == 1 and not args.no_toy_ess and not args.no_cpu_baseline and side:
            out["ess_toy"] = toy_ess_leg(device, 65536, host_threads()[0])
        print(json.dumps(out), flush=True)
    s.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
