"""bench.py's output contract, checked on small workloads: ONE JSON line from rank 0 with the driver's keys, the
`roofline` and `cpu_baseline` objects of the measurement section, the strict / fast-mode / ESS legs, and the
self-launched two-rank path (two ranks rehearsed on one card over gloo).  Not a performance test."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, timeout=600):
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    # exactly one JSON line (gloo itself may print a connection notice on stdout in the rehearsal mode)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_the_whole_contract():
    d = _bench("--chains-per-gpu", "4096", "--steps", "16", "--warmup", "16", "--adjust", "2", "--mapping", "lane",
               "--ess-chains", "512", "--ess-steps", "200")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 16 and d["warmup"] == 16 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and d["config"]["chains_per_gpu"] == 4096
    assert d["value"] > 0 and abs(d["value"] - 4096 * 16 / (d["ms_per_step"] * 16e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms_per_launch", "executed"):
        assert k in rf, k
    assert rf["unit"] == "TFLOP/s" and rf["peak"] == 78.6 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert 0 < rf["executed"]["frac"] < rf["frac"] < 1            # executed work is a subset of the algorithmic count
    assert rf["traffic"] is None or "from_profile" in rf["traffic_source"]
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "faithful"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > cb["faithful"]["value"] > 0
    assert d["strict"]["fp_contract"] == "strict" and 0 < d["strict"]["value"] < d["value"]
    assert d["fp32_fast_mode"]["peak"] == 157.3 and d["fp32_fast_mode"]["value"] > 0
    assert "converged" in d["ess"] and (d["ess"]["converged"] or "gpu_ess_per_s" not in d["ess"])
    assert set(d["ess"]["ess_per_step_within_chain"]) == {"betaMild", "betaHigh", "delay"}
    assert d["config"]["occupancy"] == {"waves_per_gpu": 64, "simds_per_gpu": 1024, "frac": 64 / 1024}
    assert rf["frac_executed"] == rf["executed"]["frac"] and "algorithmic" in rf["frac_counts"]
    # the metric's second half on targets where the reference's sampler mixes (BASELINE configs[0])
    toy = d["ess_toy"]
    assert toy["mylike1"]["converged"] and toy["mylike1"]["cpu_chains_bit_identical_to_gpu"]
    assert toy["rosenbrock"]["cpu_chains_bit_identical_to_gpu"]
    for leg in toy.values():
        assert leg["gpu"]["ess_per_s"] > leg["cpu"]["ess_per_s"] > 0 and leg["gpu"]["chains"] == 65536


def test_config_c1_line_names_its_occupancy():
    """BASELINE configs[1]: 1 024 chains x 200 days on one GPU -- 16 waves for 1 024 SIMDs."""
    d = _bench("--config", "c1", "--steps", "16", "--warmup", "16", "--adjust", "2", "--ess-chains", "0",
               "--no-fast-leg", "--no-strict-leg", "--no-toy-ess")
    c = d["config"]
    assert c["chains_per_gpu"] == 1024 and c["n_data"] == 200 and c["size"] == 300 and c["dim"] == 193
    # AUTO gives a small ensemble one chain per WAVEFRONT (mcmc_hip_set_mapping): 1 024 waves instead of 16
    assert "configs[1]" in c["workload"] and c["mapping"].startswith("wave") and c["occupancy"]["waves_per_gpu"] == 1024
    assert c["occupancy"]["frac"] == 1.0 and d["roofline"]["kernel"] == "mh_step_wave_kernel"
    lane = _bench("--config", "c1", "--mapping", "lane", "--steps", "16", "--warmup", "16", "--adjust", "2", "--ess-chains", "0",
                  "--no-fast-leg", "--no-strict-leg", "--no-toy-ess", "--no-cpu-baseline")
    assert lane["config"]["mapping"].startswith("lane") and lane["config"]["occupancy"]["waves_per_gpu"] == 16
    assert abs(lane["config"]["occupancy"]["frac"] - 16 / 1024) < 1e-12
    assert d["value"] > 1.8 * lane["value"]                      # VERDICT r02 #2's bar for a small-ensemble mapping
    assert d["roofline"]["traffic"] is None and "200-day" in d["cpu_baseline"]["sample"]


def test_total_chains_is_strong_scaling_and_does_not_change_the_chains():
    """--total-chains splits a FIXED total by contiguous chain-id range: one rank and two ranks (rehearsed on one
    card over gloo) run the same 4 096 chains -- same starts (jumped generator), same streams -- so the
    cross-chain diagnostics of the two jobs agree to summation order."""
    common = ("--total-chains", "4096", "--steps", "16", "--warmup", "16", "--adjust", "2", "--no-cpu-baseline",
              "--ess-chains", "0", "--no-fast-leg", "--no-strict-leg", "--no-toy-ess")
    one = _bench(*common)
    two = _bench("--gpus", "2", "--backend", "gloo", "--device", "0", *common)
    assert one["scaling"] == two["scaling"] == "strong"
    assert one["config"]["chains_total"] == two["config"]["chains_total"] == 4096
    assert one["config"]["chains_per_gpu"] == 4096 and two["config"]["chains_per_gpu"] == 2048 and two["n_gpus"] == 2
    d1, d2 = one["diagnostics"], two["diagnostics"]
    assert d1["chains"] == d2["chains"] == 4096 and d1["acceptance_timed_steps"] == d2["acceptance_timed_steps"]
    for k in ("loglike_mean_sd", "betaMild_mean_sd", "betaHigh_mean_sd", "delay_mean_sd"):
        assert abs(d1[k][0] - d2[k][0]) <= 1e-12 * abs(d1[k][0]) and abs(d1[k][1] - d2[k][1]) <= 1e-9 * abs(d1[k][1]), k


def test_gpus_flag_launches_its_ranks():
    d = _bench("--gpus", "2", "--backend", "gloo", "--device", "0", "--chains-per-gpu", "2048", "--steps", "16",
               "--warmup", "16", "--adjust", "2", "--no-cpu-baseline", "--ess-chains", "0")
    assert d["n_gpus"] == 2 and d["diagnostics"]["ranks"] == 2 and d["diagnostics"]["chains"] == 4096
    assert d["config"]["parallelism"].endswith("x2") and "gloo" in d["diagnostics"]["how"]
    assert d["value"] > 0 and "cpu_baseline" not in d
