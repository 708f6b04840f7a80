"""GPU tests of the FAST arithmetic mode (MCMC_HIP_PRECISION_F32_MODEL, include/mcmc_hip.h):
the renewal convolution of DiseaseSpread::eval in FP32 -- on the matrix cores ("f32_model") or, in its first
form kept for comparison, with packed vector multiply-adds ("f32_packed") -- everything else in double.
Every test runs in both forms.

This mode is not the reference's arithmetic (`using Float = double`, mcmc.h:43), so its bar is
not bit-exactness but the one north_star states for reduced precision: log-likelihoods close
to the FP64 kernel's (tolerances below, relative to the bit-exact path that IS held to the
oracle) and posterior summaries of a chain ensemble within Monte-Carlo error.  What must stay
bit-identical is everything outside the model evaluation: prior, RNG use per step, pair counts.
"""
import numpy as np
import pytest

from conftest import make_cfg, perturbed
from mcmc_amd import hip as H

pytestmark = pytest.mark.gpu

SMALL = dict(max_delay=12, n_predict=5, fix_behavior_in_advance=5, initial_delay=7.0)
# |ll32 - ll64| <= REL * |ll64| + ABS.  FP32 rounding of the ~400-term sums behind every day's
# deaths is ~1e-6 relative; the likelihood squares (deaths - observed), hence the absolute part.
REL, ABS = 2e-5, 2e-2
# 560 days of saturating epidemic with capacity corrections switching on and off: rounding
# differences are amplified by the feedback n_{i+1} = f(buffers[i])
REL_LONG_OVERFLOW = 1e-3


FAST = ["f32_model", "f32_packed"]


def _pair(cfg, R10, fast="f32_model"):
    return H.CovidSampler(cfg, R10), H.CovidSampler(cfg, R10, precision=fast)


@pytest.mark.parametrize("fast", FAST)
@pytest.mark.parametrize("case", ["small", "no_prediction_days", "overflow", "c2_200_days", "long_horizon", "long_horizon_overflow"])
def test_fast_mode_loglike_close_to_f64(R11, R10, case, fast):
    if case == "small":
        cfg = make_cfg(60, R11, contract=1, **SMALL)
    elif case == "no_prediction_days":             # the last block overhangs the horizon by 6 days
        cfg = make_cfg(45, R11, contract=1, **dict(SMALL, n_predict=0))
    elif case == "overflow":                       # capacity corrections on some lanes only
        cfg = make_cfg(60, R11, contract=1, cap0=3.0, cap_incr_rate=0.5, initial_beta_mild=1.5,
                       initial_beta_high=4.0, **SMALL)
    elif case == "c2_200_days":                    # BASELINE configs[1] shape
        cfg = make_cfg(200, R11, contract=1)
    elif case == "long_horizon":                   # lags past the 500-day table: skipped chunks
        cfg = make_cfg(520, R11, contract=1, initial_beta_mild=0.5, initial_beta_high=1.5)
    else:                                          # ... + corrections over 500 days of feedback
        cfg = make_cfg(520, R11, contract=1, cap0=3.0, cap_incr_rate=0.5, initial_beta_mild=1.5,
                       initial_beta_high=4.0)
    xs = perturbed(cfg, 130, 0.1)                  # two full tiles and a ragged one
    if case == "overflow":
        xs[::3, -3] *= 0.2
        xs[::3, -2] *= 0.2
    s64, s32 = _pair(cfg, R10, fast)
    a, b = s64.eval(xs), s32.eval(xs)
    s64.close()
    s32.close()
    assert np.all(np.isfinite(b))
    assert np.array_equal(a[:, 1], b[:, 1])                        # the prior is untouched
    err = np.abs(a[:, 0] - b[:, 0])
    rel = REL_LONG_OVERFLOW if case == "long_horizon_overflow" else REL
    worst = int(np.argmax(err / (rel * np.abs(a[:, 0]) + ABS)))
    assert np.all(err <= rel * np.abs(a[:, 0]) + ABS), (worst, err[worst], a[worst, 0], b[worst, 0])
    assert np.any(a[:, 0] != b[:, 0])                              # and it really is another arithmetic


@pytest.mark.parametrize("fast", FAST)
def test_fast_mode_is_deterministic_and_independent_of_the_batch(R11, R10, fast):
    """A chain's fast-mode result depends on its own inputs only: the wave-uniform shortcuts (common
    first day, chunks with capacity corrections) add exact zeros for the lanes they do not concern.
    Same bits for any order / composition of the batch, run after run, steps included."""
    cfg = make_cfg(60, R11, contract=1, cap0=3.0, cap_incr_rate=0.5, initial_beta_mild=1.5, initial_beta_high=4.0, **SMALL)
    xs = perturbed(cfg, 130, 0.1)
    xs[::3, -3] *= 0.2                               # lanes with and without corrections in one wave
    xs[::3, -2] *= 0.2
    xs[::5, -1] = 11.5                               # and very different first days
    perm = np.random.default_rng(3).permutation(130)
    s = H.CovidSampler(cfg, R10, precision=fast)
    a = s.eval(xs)
    assert np.array_equal(a, s.eval(xs))
    assert np.array_equal(a[perm], s.eval(xs[perm]))
    assert np.array_equal(a[:7], s.eval(xs[:7]))
    ids = np.arange(1, 131, dtype=np.uint64)
    s.set_chains(xs, chain_ids=ids)
    s.step(12, 1.0, True)
    p1, r1 = s.params(), s.rng_state()
    s.set_chains(xs[perm], chain_ids=ids[perm])
    s.step(12, 1.0, True)
    assert np.array_equal(p1[perm], s.params()) and np.array_equal(r1[perm], s.rng_state())
    s.close()


@pytest.mark.parametrize("fast", FAST)
def test_fast_mode_batch_independence_on_a_long_horizon(R11, R10, fast):
    """560 simulated days: lags run past the 500-row response table, so whole blocks of early source days are
    skipped and (matrix-core form) the first source-day pair can start on an odd day.  A chain's bits still
    must not depend on the batch: alone, permuted, or next to chains that start weeks earlier or later."""
    cfg = make_cfg(520, R11, contract=1, cap0=3.0, cap_incr_rate=0.5, initial_beta_mild=1.5, initial_beta_high=4.0)
    xs = perturbed(cfg, 70, 0.05)
    xs[::4, -1] = 3.25                               # first days spread over the whole admissible range
    xs[1::4, -1] = 38.5
    xs[::3, -3] *= 0.3                               # lanes with and without capacity corrections
    s = H.CovidSampler(cfg, R10, precision=fast)
    a = s.eval(xs)
    assert np.all(np.isfinite(a[:, 0]))
    perm = np.random.default_rng(5).permutation(70)
    assert np.array_equal(a[perm], s.eval(xs[perm]))
    for i in (0, 1, 2, 5, 69):
        assert np.array_equal(a[i:i + 1], s.eval(xs[i:i + 1])), i
    assert np.array_equal(a[64:], s.eval(xs[64:]))
    s.close()


@pytest.mark.parametrize("fast", FAST)
def test_fast_mode_full_size_properties(R11, R10, fast):
    """BASELINE configs[2] shape (365 days, dim 358) at 16384 chains in the fast mode: a scattered
    subset run alone ends on the same bits, a repeat run is bit-identical, the log-likelihoods of the
    start points agree with the FP64 kernel's, and every chain consumed its own RNG stream."""
    from mcmc_amd import covid_config as CC
    cfg = make_cfg(365, R11, contract=1)
    n = 16384
    xs = H.perturb_ics(CC.initial_params(cfg), n, 0.01)
    big = H.CovidSampler(cfg, R10, precision=fast)
    big.set_chains(xs, first_chain_id=1)
    ll0 = big.loglike().copy()
    big.step(3, 1.0, True)
    P, LL, RNG = big.params(), big.loglike(), big.rng_state()
    assert np.all(np.isfinite(LL))
    idx = np.array([0, 1, 63, 64, 4097, 9999, n - 1])
    sub = H.CovidSampler(cfg, R10, precision=fast)
    sub.set_chains(xs[idx], chain_ids=idx + 1)
    sub.step(3, 1.0, True)
    assert np.array_equal(sub.params(), P[idx]) and np.array_equal(sub.loglike(), LL[idx])
    assert np.array_equal(sub.rng_state(), RNG[idx])
    big.set_chains(xs, first_chain_id=1)
    big.step(3, 1.0, True)
    assert np.array_equal(P, big.params()) and np.array_equal(LL, big.loglike()) and np.array_equal(RNG, big.rng_state())
    assert np.all(RNG[:, 1] == (2 * (np.arange(n, dtype=np.uint64) + 1) + 1))
    f64 = H.CovidSampler(cfg, R10)
    ref = f64.eval(xs[:256])
    assert np.all(np.abs(ref[:, 0] - ll0[:256, 0]) <= REL * np.abs(ref[:, 0]) + ABS)
    for s in (big, sub, f64):
        s.close()


@pytest.mark.parametrize("fast", FAST)
def test_fast_mode_chains_agree_statistically(R11, R10, fast):
    """1024 chains, same starts / chain ids / step schedule in both modes: the ensembles'
    acceptance and posterior summaries agree within Monte-Carlo error, per-step RNG use and
    pair counts are identical while the chains still coincide."""
    cfg = make_cfg(100, R11, contract=1, **SMALL)
    xs = perturbed(cfg, 1024, 0.02)
    out = {}
    for name, s in zip(("f64", "f32"), _pair(cfg, R10, fast)):
        s.set_chains(xs)
        s.step(1, 1.0, True)
        first = (s.rng_state().copy(), s.pair_count())
        s.run(400, 100, 10, 10, record_samples=True, compute_mean=True)
        tail = np.stack([s.sample_all(k)[:, -3:] for k in range(s.n_saved)], axis=1)   # [n][saved][3]
        out[name] = dict(first=first, acc=s.accept_counts().mean() / 400, tail=tail, mean=s.mean()[:, -3:],
                         ll=s.loglikes())
        s.close()
    a, b = out["f64"], out["f32"]
    # one step in: same draws consumed (the proposal does not depend on the model arithmetic)
    assert np.array_equal(a["first"][0][:, 1], b["first"][0][:, 1])
    same = np.mean(a["first"][0][:, 0] == b["first"][0][:, 0])
    assert same > 0.99 and a["first"][1] == b["first"][1]
    assert np.all(np.isfinite(b["ll"]))
    assert abs(a["acc"] - b["acc"]) < 0.02, (a["acc"], b["acc"])
    for j, nm in enumerate(["betaMild", "betaHigh", "delay"]):
        xa, xb = a["tail"][:, -10:, j].mean(axis=1), b["tail"][:, -10:, j].mean(axis=1)
        se = np.sqrt(xa.var(ddof=1) / len(xa) + xb.var(ddof=1) / len(xb))
        assert abs(xa.mean() - xb.mean()) < 4 * se + 1e-12, (nm, xa.mean(), xb.mean(), se)
        qa, qb = np.quantile(a["tail"][:, -1, j], [.05, .5, .95]), np.quantile(b["tail"][:, -1, j], [.05, .5, .95])
        spread = qa[2] - qa[0]
        assert np.all(np.abs(qa - qb) < 0.25 * spread), (nm, qa, qb)


def test_fast_mode_needs_fma_contraction_and_a_known_mode(R11, R10):
    s = H.CovidSampler(make_cfg(40, R11, contract=0, **SMALL), R10)
    for fast in FAST:
        with pytest.raises(RuntimeError, match="contract"):
            s.set_precision(fast)
    with pytest.raises(RuntimeError, match="unknown mode"):
        H._check(H.lib().mcmc_hip_set_precision(s._h, 7))
    s.close()


def test_zz_row_checks_of_a_diagnosis_build_are_clean(R11, R10):
    """Libraries built with -DMCMC_F32_CHECK (tools/build_variants.sh, e.g. for another MCMC_TF) compare
    every row range of the fast-mode evaluation with its buffer before touching it and record the first
    violation; the tests above have run in this process, so the record covers them.  The product build
    compiles the checks out and always reports zeros."""
    cfg = make_cfg(60, R11, contract=1, **SMALL)
    s = H.CovidSampler(cfg, R10, precision="f32_packed")
    s.eval(perturbed(cfg, 70, 0.1))
    kind, lo, hi, bound = s.f32_check()
    s.close()
    assert kind == 0, f"row range kind {kind}: rows [{lo}, {hi}] outside a buffer of {bound} rows"
