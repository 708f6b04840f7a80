"""GPU parity tests: the HIP path, called through the C ABI (libmcmc_hip.so via ctypes),
against the CPU oracle on the same seeded inputs.  Bar: BIT-EXACT for every quantity
(parameters, log-likelihoods, RNG state, accept counts, step sizes, recorded samples) --
the arithmetic is FP64 in a fixed operation order on both sides (DESIGN.md "Parity").
Run with:  python -u -m pytest tests -m gpu -x -v --timeout 120 --timeout-method thread
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, make_cfg, perturbed
from mcmc_amd import covid_config as CC, hip as H
from oracle import oracle as O

pytestmark = pytest.mark.gpu

SMALL = dict(max_delay=12, n_predict=5, fix_behavior_in_advance=5, initial_delay=7.0)


def _oracle_after_steps(m, tgt, chain_id, x0, n_steps, time=0.0, stepsize=1.0, random=True):
    oc = O.OracleChain(m, chain_id, x0, stepsize)
    for _ in range(n_steps):
        oc.step(tgt, time, random)
    return oc


def _assert_chain_equal(oc, s_state, c):
    P, LL, RNG, ACC = s_state
    assert np.array_equal(oc.x, P[c]), f"chain {c}: parameters differ"
    assert oc.ll == tuple(LL[c]), f"chain {c}: loglike differs {oc.ll} vs {LL[c]}"
    assert oc.rng_state == tuple(int(v) for v in RNG[c]), f"chain {c}: RNG state differs"
    assert oc.c.n_accept == ACC[c], f"chain {c}: accept count differs"


def _state(s):
    return s.params(), s.loglike(), s.rng_state(), s.accept_counts()


# ------------------------------------------------------------------- eval / derive ----
@pytest.mark.parametrize("contract", [0, 1])
@pytest.mark.parametrize("n_data", [40, 200])
def test_eval_and_derive_bitexact(R11, R10, contract, n_data):
    kw = SMALL if n_data == 40 else {}
    cfg = make_cfg(n_data, R11, contract=contract, **kw)
    m = O.Model(cfg, R11, contract=contract)
    xs = perturbed(cfg, 70, 0.08)                 # 70: one full wave + a ragged one
    s = H.CovidSampler(cfg, R10)
    ll = s.eval(xs)
    ref = np.array([[m.eval(x), m.prior(x)] for x in xs])
    assert np.array_equal(ll, ref)
    cur = s.derive(xs[:5])
    refc = np.array([m.eval(x, curves=True)[1] for x in xs[:5]])
    assert np.array_equal(cur, refc)
    s.close()


@pytest.mark.parametrize("contract", [0, 1])
def test_eval_capacity_overflow_branch(R11, R10, contract):
    """Hospital overflow (covid.h:413-429): extraDeaths > 0 on many days, on only SOME lanes
    of a wave (capacity differs per chain through the betas)."""
    cfg = make_cfg(60, R11, contract=contract, cap0=3.0, cap_incr_rate=0.5, initial_beta_mild=1.5,
                   initial_beta_high=4.0, **SMALL)
    m = O.Model(cfg, R11, contract=contract)
    xs = perturbed(cfg, 64, 0.3)
    xs[::3, -3] *= 0.2                            # a third of the lanes barely grow: no overflow
    xs[::3, -2] *= 0.2
    s = H.CovidSampler(cfg, R10)
    ll = s.eval(xs)
    ref = np.array([[m.eval(x), m.prior(x)] for x in xs])
    assert np.array_equal(ll, ref)
    cur = s.derive(xs[:8])
    refc = np.array([m.eval(x, curves=True)[1] for x in xs[:8]])
    assert np.array_equal(cur, refc)
    # the branch really ran on some lanes and not on others
    big = make_cfg(60, R11, contract=contract, cap0=1e15, initial_beta_mild=1.5, initial_beta_high=4.0, **SMALL)
    big = {**big, "deaths": cfg["deaths"], "sigma": cfg["sigma"]}
    diff = [m.eval(x) != O.Model(big, R11, contract=contract).eval(x) for x in xs[:12]]
    assert any(diff) and not all(diff)
    s.close()


def test_eval_delay_edge_cases(R11, R10):
    """delay at both bounds, integer delays (frac == 0: one shift has zero weight), lanes of
    one wave starting on very different days."""
    cfg = make_cfg(40, R11, **SMALL)
    m = O.Model(cfg, R11)
    xs = perturbed(cfg, 64, 0.05)
    delays = [5.0, 12.0, 7.0, 6.0, 11.999999, 5.000001, 8.5, 9.25]
    for i in range(64):
        xs[i, -1] = delays[i % len(delays)]
    s = H.CovidSampler(cfg, R10)
    ll = s.eval(xs)
    ref = np.array([[m.eval(x), m.prior(x)] for x in xs])
    assert np.array_equal(ll, ref)
    cur = s.derive(xs[:8])
    refc = np.array([m.eval(x, curves=True)[1] for x in xs[:8]])
    assert np.array_equal(cur, refc)
    s.close()


@pytest.mark.parametrize("n_data", [45, 48])
def test_no_prediction_days(R11, R10, n_data):
    """nPredictDays = 0 (covid.h:294 takes a size_t): the simulated horizon is the whole series, so the
    last 8-day block overhangs the arrays' logical end by up to 6 days (n_data = 45: 57 days = 7 blocks
    + 1) -- eval, derive and MH steps against the oracle, delays spread over both bounds."""
    kw = dict(SMALL, n_predict=0)
    cfg = make_cfg(n_data, R11, **kw)
    assert cfg["size"] == 12 + n_data
    m = O.Model(cfg, R11)
    xs = perturbed(cfg, 70, 0.05)
    delays = [5.0, 12.0, 7.0, 6.0, 11.999999, 5.000001, 8.5, 9.25]
    for i in range(70):
        xs[i, -1] = delays[i % len(delays)]
    s = H.CovidSampler(cfg, R10)
    ll = s.eval(xs)
    ref = np.array([[m.eval(x), m.prior(x)] for x in xs])
    assert np.array_equal(ll, ref)
    cur = s.derive(xs[:4])
    refc = np.array([m.eval(x, curves=True)[1] for x in xs[:4]])
    assert np.array_equal(cur, refc)
    tgt = O.make_target("temp", 1.0)
    s.set_chains(xs)
    s.step(6, 1.0, True)
    st = _state(s)
    for c in (0, 63, 69):
        _assert_chain_equal(_oracle_after_steps(m, tgt, c + 1, xs[c], 6), st, c)
    s.close()


def test_long_horizon_beyond_the_trajectory_table(R11, R10):
    """size = 40 + 60 + 520 = 620 days > traj.nDays = 500: lags past the table contribute
    nothing (the `d < nDays` cut of project(), covid.h:378), whole chunks of them are skipped
    and the lag window starts later -- eval, derive and MH steps against the oracle."""
    cfg = make_cfg(520, R11, initial_beta_mild=0.5, initial_beta_high=1.5)
    assert cfg["size"] == 620
    m = O.Model(cfg, R11)
    xs = perturbed(cfg, 66, 0.05)
    s = H.CovidSampler(cfg, R10)
    ll = s.eval(xs)
    ref = np.array([[m.eval(x), m.prior(x)] for x in xs])
    assert np.array_equal(ll, ref) and np.all(np.isfinite(ll))
    cur = s.derive(xs[:3])
    refc = np.array([m.eval(x, curves=True)[1] for x in xs[:3]])
    assert np.array_equal(cur, refc)
    tgt = O.make_target("temp", 1.0)
    s.set_chains(xs)
    s.step(5, 1.0, True)
    st = _state(s)
    for c in (0, 63, 65):
        _assert_chain_equal(_oracle_after_steps(m, tgt, c + 1, xs[c], 5), st, c)
    s.close()


def test_long_horizon_with_capacity_corrections(R11, R10):
    """The flat add() of the extra deaths stops after traj.nDays = 500 days just like project()
    (covid.h:387-391): with corrections from day ~49 on and 560 simulated days, the lags 500..510
    of those source days must contribute nothing -- not even the `+ e`."""
    cfg = make_cfg(520, R11, cap0=3.0, cap_incr_rate=0.5, initial_beta_mild=1.5, initial_beta_high=4.0)
    m = O.Model(cfg, R11)
    xs = perturbed(cfg, 64, 0.05)
    s = H.CovidSampler(cfg, R10)
    ll = s.eval(xs)
    ref = np.array([[m.eval(x), m.prior(x)] for x in xs[:8]])
    assert np.array_equal(ll[:8], ref) and np.all(np.isfinite(ll))
    cur = s.derive(xs[:2])
    refc = np.array([m.eval(x, curves=True)[1] for x in xs[:2]])
    assert np.array_equal(cur, refc)
    s.close()


@pytest.mark.parametrize("scale", [1.0, 0.25, 8.0, 1.0 / 3.0, 37.3])
def test_smoothness_prior_scales(R11, R10, scale):
    """dx = L/n of SmoothnessPrior (mcmc.h:201): the kernels multiply by the exact reciprocal when
    dx is a power of two (1, 1/4, 8) and divide otherwise (1/3, 37.3/n) -- same bits either way."""
    base = make_cfg(40, R11, **SMALL)
    L = 37.3 if scale == 37.3 else base["nB"] * scale
    cfg = make_cfg(40, R11, smooth_L=L, smooth_Ls=17.0, **SMALL)
    m = O.Model(cfg, R11)
    xs = perturbed(cfg, 64, 0.1)
    s = H.CovidSampler(cfg, R10)
    ll = s.eval(xs)
    assert np.array_equal(ll[:, 1], np.array([m.prior(x) for x in xs]))
    tgt = O.make_target("temp", 1.0)
    s.set_chains(xs)
    s.step(12, 1.0, True)
    st = _state(s)
    for c in (0, 31, 63):
        _assert_chain_equal(_oracle_after_steps(m, tgt, c + 1, xs[c], 12), st, c)
    s.close()


def test_eval_nonfinite_parameters_do_not_fault(R11, R10):
    """NaN/Inf parameters (the reference rejects such initial states, mcmc.h:1334,1365): the
    kernel must stay in bounds and report a non-finite or oracle-equal loglike."""
    cfg = make_cfg(40, R11, **SMALL)
    m = O.Model(cfg, R11)
    xs = perturbed(cfg, 8, 0.05)
    xs[0, -1] = np.nan
    xs[1, -1] = 1e300
    xs[2, -1] = -1e300
    xs[3, -2] = np.inf
    xs[4, 3] = np.nan
    s = H.CovidSampler(cfg, R10)
    ll = s.eval(xs)
    for i in (5, 6, 7):
        assert ll[i, 0] == m.eval(xs[i])
    assert not np.isfinite(ll[3, 0]) and not np.isfinite(ll[4, 0])
    s.close()


# ----------------------------------------------------------------------- MH steps ----
@pytest.mark.parametrize("contract", [0, 1])
def test_golden_traces_step_by_step(R11, R10, contract):
    """The committed 40-step traces (tests/golden/covid200_traces.json): accept bit, RNG
    state after every step, final loglike and parameter hash."""
    gold = json.load(open(os.path.join(GOLDEN, "covid200_traces.json")))[f"contract{contract}"]
    cfg = make_cfg(200, R11, contract=contract)
    assert [float(v).hex() for v in cfg["deaths"]] == gold["deaths_hex"]
    xs = perturbed(cfg, 4, 0.05)
    s = H.CovidSampler(cfg, R10)
    s.set_chains(xs, first_chain_id=1)
    ll0 = s.loglike()
    for c in range(4):
        assert [ll0[c, 0].hex(), ll0[c, 1].hex()] == gold["chains"][c]["ll0"]
    prev_acc = np.zeros(4, dtype=np.int64)
    for k in range(40):
        s.step(1, 1.0, True)
        rng, acc = s.rng_state(), s.accept_counts()
        for c in range(4):
            tr = gold["chains"][c]
            assert [int(rng[c, 0]), int(rng[c, 1])] == tr["rng"][k], f"step {k} chain {c}"
            assert int(acc[c] - prev_acc[c]) == tr["accept"][k], f"step {k} chain {c}"
        prev_acc = acc.astype(np.int64)
    P, LL = s.params(), s.loglike()
    for c in range(4):
        tr = gold["chains"][c]
        assert [LL[c, 0].hex(), LL[c, 1].hex()] == tr["ll_end"]
        assert hashlib.sha256(np.ascontiguousarray(P[c]).tobytes()).hexdigest() == tr["x_end_sha"]
    s.close()


@pytest.mark.parametrize("contract", [0, 1])
@pytest.mark.parametrize("n_chains", [1, 65, 130])
def test_steps_match_oracle_ragged_chain_counts(R11, R10, contract, n_chains):
    cfg = make_cfg(40, R11, contract=contract, **SMALL)
    m = O.Model(cfg, R11, contract=contract)
    xs = perturbed(cfg, n_chains, 0.05)
    s = H.CovidSampler(cfg, R10)
    s.set_chains(xs, first_chain_id=1)
    s.step(25, 1.0, True)
    st = _state(s)
    tgt = O.make_target("temp", 1.0)
    for c in range(n_chains):
        _assert_chain_equal(_oracle_after_steps(m, tgt, c + 1, xs[c], 25), st, c)
    s.close()


def test_steps_given_subspace_and_tempered(R11, R10):
    """random_subspace=False skips the head draw (mcmc.h:1096-1105); beta != 1 (TempTarget)."""
    cfg = make_cfg(40, R11, target_T=3.0, **SMALL)
    m = O.Model(cfg, R11)
    xs = perturbed(cfg, 64, 0.05)
    s = H.CovidSampler(cfg, R10)
    s.set_chains(xs, stepsizes=0.37, first_chain_id=11)
    s.step(10, 1.0 / 3.0, False)
    st = _state(s)
    tgt = O.make_target("temp", 3.0)
    for c in range(64):
        _assert_chain_equal(_oracle_after_steps(m, tgt, c + 11, xs[c], 10, stepsize=0.37, random=False), st, c)
    s.close()


def test_steps_with_overflow_fixed_breakpoints_and_no_prior(R11, R10):
    """capacity branch inside the sampler, some discontinuousVals fixed (covid.h:560-581:
    the threshold 2/nNotFixed counts the FIXED entries), State without SmoothnessPrior."""
    cfg = make_cfg(50, R11, cap0=3.0, cap_incr_rate=0.5, initial_beta_mild=1.5, initial_beta_high=4.0,
                   disc_fixed=[0, 1, 0, 1], has_prior=0, **SMALL)
    m = O.Model(cfg, R11)
    xs = perturbed(cfg, 64, 0.2)
    s = H.CovidSampler(cfg, R10)
    s.set_chains(xs, first_chain_id=1)
    s.step(30, 1.0, True)
    st = _state(s)
    tgt = O.make_target("plain")
    for c in range(0, 64, 3):
        _assert_chain_equal(_oracle_after_steps(m, tgt, c + 1, xs[c], 30), st, c)
    assert np.all(st[1][:, 1] == 0.0)
    s.close()


def test_many_breakpoints(R11, R10):
    cfg = make_cfg(60, R11, disc_days=(1, 4, 9, 14, 20, 27, 33, 41), disc_vals=(1, .9, .8, .7, .6, .5, .4, .3),
                   disc_fixed=[1, 0, 0, 0, 0, 0, 0, 0], **SMALL)
    m = O.Model(cfg, R11)
    xs = perturbed(cfg, 64, 0.02)
    s = H.CovidSampler(cfg, R10)
    s.set_chains(xs, first_chain_id=1)
    s.step(30, 1.0, True)
    st = _state(s)
    tgt = O.make_target("temp", 1.0)
    for c in range(0, 64, 5):
        _assert_chain_equal(_oracle_after_steps(m, tgt, c + 1, xs[c], 30), st, c)
    s.close()


def test_single_breakpoint_and_behavior_over_all_data_days(R11, R10):
    """One breakpoint (days before it have no piecewise factor, covid.h:432-444) and
    fixBehaviorInAdvance = 0 (the behaviour function spans every data day, covid.h:299)."""
    kw = dict(SMALL, fix_behavior_in_advance=0)
    cfg = make_cfg(50, R11, disc_days=(9,), disc_vals=(.7,), disc_fixed=[0], **kw)
    assert cfg["n_disc"] == 1 and cfg["nB"] == 50
    m = O.Model(cfg, R11)
    xs = perturbed(cfg, 64, 0.05)
    s = H.CovidSampler(cfg, R10)
    ll = s.eval(xs)
    assert np.array_equal(ll, np.array([[m.eval(x), m.prior(x)] for x in xs]))
    s.set_chains(xs)
    s.step(25, 1.0, True)
    st = _state(s)
    tgt = O.make_target("temp", 1.0)
    for c in (0, 17, 63):
        _assert_chain_equal(_oracle_after_steps(m, tgt, c + 1, xs[c], 25), st, c)
    s.close()


# ------------------------------------------------------------- MetropolisChain::run ----
@pytest.mark.parametrize("target", [("temp", 1.0, 0.0, 1.0), ("cooling", 1.0, 2.5, 8.0), ("plain", 1.0, 0.0, 1.0)])
def test_run_matches_oracle_run(R11, R10, target):
    """mcmc_hip_run == MetropolisChain::run (mcmc.h:1005-1138) for every chain: adjust phase
    (20 steps per iteration, step-size correction), burn-in, thinning, recorded loglikes and
    samples, running mean, final step sizes and accept counts."""
    kind, T, slope, Tinit = target
    cfg = make_cfg(40, R11, target_kind=kind, target_T=T, target_slope=slope, target_Tinit=Tinit, **SMALL)
    m = O.Model(cfg, R11)
    n = 66
    xs = perturbed(cfg, n, 0.05)
    s = H.CovidSampler(cfg, R10)
    s.set_chains(xs, first_chain_id=1)
    ns, nb, na, th = 37, 3, 4, 3
    s.run(ns, nb, na, th, record_samples=True, compute_mean=True)
    assert s.n_saved == O.lib().orc_n_saved(ns, nb, na, th)
    st = _state(s)
    LLS, SS, MEAN = s.loglikes(), s.stepsizes(), s.mean()
    tgt = O.make_target(kind, T, slope, Tinit)
    for c in list(range(0, n, 7)) + [n - 1]:
        oc = O.OracleChain(m, c + 1, xs[c])
        ll, samples, mean = oc.run(tgt, ns, nb, na, th, record=True, mean=True)
        _assert_chain_equal(oc, st, c)
        assert np.array_equal(ll, LLS[c])
        assert oc.c.stepsize == SS[c]
        assert np.array_equal(samples, s.samples(c))
        assert np.array_equal(samples[:, -3:], s.samples(c, s.dim - 3, 3))
        assert np.array_equal(mean, MEAN[c])
    # every chain, one saved sample
    assert np.array_equal(s.sample_all(2)[n - 1], s.samples(n - 1)[2])
    # a second run() continues from the current state (like calling chain.run() again)
    s.run(5, 0, 0, 1)
    oc = O.OracleChain(m, 1, xs[0])
    oc.run(tgt, ns, nb, na, th)
    ll2, _, _ = oc.run(tgt, 5, 0, 0, 1, record=False)
    assert np.array_equal(ll2, s.loglikes()[0])
    s.close()


def test_run_without_recording_and_zero_lengths(R11, R10):
    cfg = make_cfg(40, R11, **SMALL)
    xs = perturbed(cfg, 64, 0.05)
    s = H.CovidSampler(cfg, R10)
    s.set_chains(xs)
    before = _state(s)
    s.run(0, 0, 0, 1)                              # nothing to do
    after = _state(s)
    assert all(np.array_equal(a, b) for a, b in zip(before, after))
    with pytest.raises(H.McmcHipError, match="none are present"):
        s.loglikes()
    s.run(10, 0, 0, 2, record_samples=False)
    with pytest.raises(H.McmcHipError, match="none are present"):
        s.samples(0)
    assert s.loglikes().shape == (64, 5)
    s.close()


def test_error_paths(R11, R10):
    cfg = make_cfg(40, R11, **SMALL)
    s = H.CovidSampler(cfg, R10)
    with pytest.raises(H.McmcHipError, match="set_chains"):
        s.step(1)
    with pytest.raises(H.McmcHipError, match="set_chains"):
        s.run(1, 0, 0, 1)
    L = H.lib()
    assert L.mcmc_hip_set_chains(s._h, 0, None, None, None, 1) == -1
    xs = perturbed(cfg, 3, 0.05)
    s.set_chains(xs)
    assert L.mcmc_hip_run(s._h, 5, 0, 0, 0, 0, 0) == -1           # thinning 0
    out = np.zeros(3)
    assert L.mcmc_hip_get_samples(s._h, 7, 0, 1, out.ctypes.data_as(C.POINTER(C.c_double))) != 0
    short = np.zeros((10, 10))
    assert L.mcmc_hip_set_tables(s._h, short.ctypes.data_as(C.POINTER(C.c_double)), 10) == -1
    s.close()


# ------------------------------------------------------------ sharding / diagnostics ----
def test_chain_results_do_not_depend_on_sharding(R11, R10):
    """Chain i draws pcg32(0, i+1) wherever it runs (mcmc.h:999,1527): running the id range
    in two contexts ("two GPUs") gives the same per-chain results as one context."""
    cfg = make_cfg(40, R11, **SMALL)
    xs = perturbed(cfg, 150, 0.05)
    whole = H.CovidSampler(cfg, R10)
    whole.set_chains(xs, first_chain_id=1)
    whole.run(12, 0, 2, 4)
    a, b = H.CovidSampler(cfg, R10), H.CovidSampler(cfg, R10)
    a.set_chains(xs[:75], first_chain_id=1)
    b.set_chains(xs[75:], first_chain_id=76)
    a.run(12, 0, 2, 4)
    b.run(12, 0, 2, 4)
    for f in ("params", "loglike", "rng_state", "accept_counts", "stepsizes", "loglikes"):
        got = np.concatenate([getattr(a, f)(), getattr(b, f)()])
        assert np.array_equal(got, getattr(whole, f)()), f
    # explicit id list == id range
    c = H.CovidSampler(cfg, R10)
    c.set_chains(xs[75:], chain_ids=np.arange(76, 151))
    c.run(12, 0, 2, 4)
    assert np.array_equal(c.params(), b.params())
    # the additive diagnostics vector of the shards sums to the whole's
    dw, da, db = whole.diagnostics(), a.diagnostics(), b.diagnostics()
    assert dw[1] == da[1] == db[1] == 2 * 20 + 12
    assert np.allclose(np.delete(da + db, 1), np.delete(dw, 1), rtol=1e-12)
    for s in (whole, a, b, c):
        s.close()


def test_diagnostics_vector(R11, R10):
    cfg = make_cfg(40, R11, **SMALL)
    xs = perturbed(cfg, 100, 0.05)
    s = H.CovidSampler(cfg, R10)
    s.set_chains(xs)
    s.run(20, 0, 0, 5)
    d = s.diagnostics()
    P, LL, ACC = s.params(), s.loglike(), s.accept_counts()
    l = LL[:, 0] + LL[:, 1]
    assert d[0] == ACC.sum() and d[1] == 20
    assert np.isclose(d[2], l.sum(), rtol=1e-12) and np.isclose(d[3], (l * l).sum(), rtol=1e-12)
    assert np.allclose(d[4::2], P.sum(axis=0), rtol=1e-12)
    assert np.allclose(d[5::2], (P * P).sum(axis=0), rtol=1e-12)
    s.close()


# ------------------------------------------- full-size properties (BASELINE configs) ----
@pytest.mark.parametrize("n", [16384, 65536, 131072])
def test_full_size_batch_independence_and_determinism(R11, R10, n):
    """BASELINE configs[2] shape (365-day series, dim 358) at 16384 chains, at the headline's 65536
    (1024 waves = one per SIMD, the launch shape bench.py times) and at 131072 (the per-GPU share of
    configs[4]; twice the chip's worth of tiles): the oracle is too slow to follow every chain, so use
    size-independent properties: (1) chain i's trajectory does not depend on which other chains share
    its wave/batch, (2) a repeat run is bit-identical, (3) the first and the last chain equal the
    oracle."""
    cfg = make_cfg(365, R11, contract=1)
    assert cfg["size"] == 465 and cfg["dim"] == 358
    xs = H.perturb_ics(CC.initial_params(cfg), n, 0.01)
    big = H.CovidSampler(cfg, R10)
    big.set_chains(xs, first_chain_id=1)
    big.step(3, 1.0, True)
    P, LL, RNG, ACC = _state(big)
    assert np.all(np.isfinite(LL))
    # (1) a scattered subset run alone, with the same ids
    idx = np.array([0, 1, 63, 64, 4097, 9999, n // 2 + 77, n - 65, n - 1])
    sub = H.CovidSampler(cfg, R10)
    sub.set_chains(xs[idx], chain_ids=idx + 1)
    sub.step(3, 1.0, True)
    Ps, LLs, RNGs, ACCs = _state(sub)
    assert np.array_equal(Ps, P[idx]) and np.array_equal(LLs, LL[idx])
    assert np.array_equal(RNGs, RNG[idx]) and np.array_equal(ACCs, ACC[idx])
    # (2) determinism
    big.set_chains(xs, first_chain_id=1)
    big.step(3, 1.0, True)
    P2, LL2, RNG2, ACC2 = _state(big)
    assert np.array_equal(P, P2) and np.array_equal(LL, LL2) and np.array_equal(RNG, RNG2)
    # (3) oracle on two chains
    m = O.Model(cfg, R11, contract=1)
    tgt = O.make_target("temp", 1.0)
    for c in (0, n - 1):
        _assert_chain_equal(_oracle_after_steps(m, tgt, c + 1, xs[c], 3), (P, LL, RNG, ACC), c)
    # every lane consumed draws: the stream position moved, the increment did not
    assert np.all(RNG[:, 1] == (2 * (np.arange(n, dtype=np.uint64) + 1) + 1))
    big.close()
    sub.close()


def test_set_state_resumes_chains_in_a_new_context(R11, R10):
    """Checkpoint / resume across contexts (mcmc_hip_set_state): chains advanced in one context, their
    whole state read out (params, step sizes, rng, loglike, mean, accept counts) and planted in a fresh
    context, go on exactly like the chains that never left -- the second run()'s loglikes, samples,
    mean and end state are bit-identical."""
    cfg = make_cfg(40, R11, contract=1, **SMALL)
    xs = perturbed(cfg, 70, 0.05)
    stay = H.CovidSampler(cfg, R10)
    stay.set_chains(xs, first_chain_id=1)
    stay.run(15, 0, 2, 3, record_samples=True, compute_mean=True)
    saved = dict(params=stay.params(), ss=stay.stepsizes(), rng=stay.rng_state(), ll=stay.loglike(), mean=stay.mean(),
                 acc=stay.accept_counts())
    stay.run(11, 2, 1, 2, record_samples=True, compute_mean=True)
    moved = H.CovidSampler(cfg, R10)
    moved.set_chains(saved["params"], stepsizes=saved["ss"], first_chain_id=1000)      # ids do not matter once the rng is planted
    moved.set_state(rng_state=saved["rng"], loglike=saved["ll"], mean=saved["mean"], accept_counts=saved["acc"])
    assert np.array_equal(moved.rng_state(), saved["rng"]) and np.array_equal(moved.loglike(), saved["ll"])
    assert np.array_equal(moved.accept_counts(), saved["acc"]) and np.array_equal(moved.mean(), saved["mean"])
    moved.run(11, 2, 1, 2, record_samples=True, compute_mean=True)
    for a, b in zip(_state(stay), _state(moved)):
        assert np.array_equal(a, b)
    assert np.array_equal(stay.loglikes(), moved.loglikes()) and np.array_equal(stay.mean(), moved.mean())
    assert np.array_equal(stay.stepsizes(), moved.stepsizes())
    for c in (0, 63, 64, 69):
        assert np.array_equal(stay.samples(c), moved.samples(c))
    # the oracle chain run twice is the same continued chain
    m = O.Model(cfg, R11, contract=1)
    tgt = O.make_target("temp", 1.0)
    oc = O.OracleChain(m, 1, xs[0])
    oc.run(tgt, 15, 0, 2, 3)
    ll2, s2, _ = oc.run(tgt, 11, 2, 1, 2)
    assert np.array_equal(moved.loglikes()[0], ll2) and np.array_equal(moved.samples(0), s2)
    # only rng given: the rest keeps what set_chains derived; a bad increment is refused
    moved.set_chains(saved["params"], stepsizes=saved["ss"], first_chain_id=1)
    moved.set_state(rng_state=saved["rng"])
    assert np.array_equal(moved.loglike(), saved["ll"])                 # the evaluation is deterministic
    bad = saved["rng"].copy(); bad[3, 1] &= ~np.uint64(1)
    with pytest.raises(H.McmcHipError, match="odd"):
        moved.set_state(rng_state=bad)
    stay.close(); moved.close()


def test_pair_count_is_the_survey_work_unit(R11, R10):
    """mcmc_hip_pair_count == sum over evaluations of m1(m1+1)/2 + m0(m0+1)/2 with
    m_s = size - start - s, whatever horizon the kernel actually simulates."""
    cfg = make_cfg(40, R11, **SMALL)
    xs = perturbed(cfg, 64, 0.0)
    s = H.CovidSampler(cfg, R10)
    s.set_chains(xs)
    s.pair_count(reset=True)
    s.step(1, 0.0, True)                 # beta 0: delta == 0, so whatever was proposed is counted once
    P = s.params()
    got = s.pair_count()
    # proposals with an unchanged delay have start = int(12 - 7) = 5
    size = cfg["size"]
    lo = 64 * sum((size - 5 - k) * (size - 5 - k + 1) // 2 for k in (0, 1))
    assert abs(got - lo) <= 64 * 4 * size          # a changed delay moves start by at most +-1
    # the executed count: the same sum over the days the log-likelihood can see (maxDelay + nData)
    sim = cfg["max_delay"] + cfg["n_data"]
    ex = 64 * sum((sim - 5 - k) * (sim - 5 - k + 1) // 2 for k in (0, 1))
    assert abs(s.executed_pair_count() - ex) <= 64 * 4 * sim
    assert s.executed_pair_count() < got
    s.executed_pair_count(reset=True)
    assert s.pair_count() == 0 and s.executed_pair_count() == 0
    s.close()


def test_posterior_summaries_equal_the_cpu_sampler(R11, R10):
    """north_star: "posterior means and quantiles within a stated tolerance of the reference
    CPU sampler".  The chains are bit-identical to the CPU oracle's, so the tolerance is ZERO:
    means and 5/50/95 % quantiles of delay, betaMild, betaHigh and the breakpoint values over
    32 chains x 300 saved points agree exactly."""
    cfg = make_cfg(40, R11, contract=1, **SMALL)
    m = O.Model(cfg, R11, contract=1)
    n = 32
    xs = perturbed(cfg, n, 0.05)
    s = H.CovidSampler(cfg, R10)
    s.set_chains(xs, first_chain_id=1)
    ns, nb, na, th = 600, 0, 10, 2
    s.run(ns, nb, na, th, record_samples=True)
    k0 = cfg["nB"]
    gpu = np.array([s.samples(c, k0, cfg["n_disc"] + 3) for c in range(n)])
    tgt = O.make_target("temp", 1.0)
    cpu = []
    for c in range(n):
        oc = O.OracleChain(m, c + 1, xs[c])
        _, samples, _ = oc.run(tgt, ns, nb, na, th, record=True)
        cpu.append(samples[:, k0:])
    cpu = np.array(cpu)
    assert gpu.shape == cpu.shape == (n, s.n_saved, cfg["n_disc"] + 3)
    post_g, post_c = gpu[:, 20:].reshape(-1, gpu.shape[2]), cpu[:, 20:].reshape(-1, cpu.shape[2])
    assert np.array_equal(post_g.mean(axis=0), post_c.mean(axis=0))
    for q in (5, 50, 95):
        assert np.array_equal(np.percentile(post_g, q, axis=0), np.percentile(post_c, q, axis=0))
    # the sampler moved: not a trivially constant trace
    assert np.all(post_g.std(axis=0) > 0)
    s.close()
