"""BASELINE.json's configurations AT THEIR STATED SHAPES, through the C ABI.

The other GPU test files hold the HIP path to the oracle on shapes the oracle can follow chain by chain.  The
shapes below were, until round 3, only run by bench.py legs:
  configs[4]  multi-region model: 50 regions x 500 data days (horizon 600 > the 500-row response table: lag
              chunks made of zero rows are skipped, the flat add() of covid.h:387-391 stops after 500 lags) x
              131 072 chains per GPU;
  configs[3]  tempering ladders of 64 rungs, 1 024 of them (65 536 chains);
  configs[2]  65 536 chains x 365 days in the fast arithmetic mode, and under mcmc_hip_run with recording.
Where the oracle is too slow for every chain the tests use what does not depend on size: a chain's trajectory
is independent of the batch it runs in, a repeat is bit-identical, sharding over contexts changes nothing,
every ladder keeps each rung once -- plus oracle-followed chains picked from both ends of the ensemble.
Bar: bit-exact (fast mode: bit-exact against itself, log-likelihood tolerance against FP64 as in
test_gpu_fast_mode.py).
"""
import numpy as np
import pytest

from conftest import make_cfg, perturbed
from mcmc_amd import covid_config as CC, hip as H
from oracle import oracle as O

pytestmark = pytest.mark.gpu

SMALL = dict(max_delay=12, n_predict=5, fix_behavior_in_advance=5, initial_delay=7.0)


def _state(s):
    return s.params(), s.loglike(), s.rng_state(), s.accept_counts()


# ------------------------------------------------------------------ configs[4]: regions x long horizon ----
def long_region_cfgs(R11, R, n_data, contract=1, **kw):
    """bench.py --regions' shape: every region three behaviour nodes, one breakpoint value and its delay (+ the
    two shared betas), populations and capacity lines that differ; counts from the forward model at each
    region's own constants.  Small capacities: the capacity-correction branch is on for weeks."""
    out = []
    for r in range(R):
        out.append(make_cfg(n_data, R11, contract=contract, fix_behavior_in_advance=n_data - 3, disc_days=(20,),
                            disc_vals=(0.6,), pop_size=int(6e7 / (1 + 0.25 * (r % 8))), cap0=7500.0 / (1 + r % 5),
                            cap_incr_rate=100.0 - r % 50, **kw))
    return out


def region_starts(cfgs, n, a=0.01):
    x0 = np.concatenate([CC.initial_params(c) for c in cfgs])
    return H.perturb_ics(x0, n, a)


@pytest.mark.parametrize("contract", [0, 1])
def test_three_regions_on_a_520_day_series_equal_the_oracle(R11, R10, contract):
    """R = 3 x 520 data days (size 620: 120 lags beyond the response table), capacity corrections on, against
    orc_multi_*: start, a whole run() with adjust groups, burn-in, thinning, recording and mean, then more
    steps -- bit for bit."""
    cfgs = long_region_cfgs(R11, 3, 520, contract=contract)
    assert cfgs[0]["size"] == 620 and cfgs[0]["dim"] == 7
    mm = O.MultiModel(cfgs, R11, contract=contract)
    n = 70
    xs = region_starts(cfgs, n, 0.03)
    xs[::5, 6] = 6.25                       # early and late first days in one wave: skipped chunks differ per block
    xs[1::5, 6] = 37.5
    s = H.CovidSampler(cfgs[0], R10)
    s.set_regions(cfgs)
    s.set_chains(xs, first_chain_id=1)
    tgt = O.make_target("temp", 1.0)
    check = (0, 1, 5, 64, 69)
    for c in check:
        oc = O.MultiChain(mm, c + 1, xs[c])
        assert np.array_equal(s.region_loglike()[c], oc.ll), f"chain {c}: start"
    # the corrections are on: some region of some checked chain has days above capacity (the derived curve
    # `serious` x probICU against the capacity line, through a single-region context)
    one = H.CovidSampler(cfgs[2], R10)
    cur = one.derive(xs[:1, 14:21])[0]
    cap = cfgs[2]["cap0"] + np.maximum(cfgs[2]["cap_incr_rate"] * (np.arange(620) - 40), 0)
    assert np.any(cur[5] * cfgs[2]["prob_icu_if_serious"] > cap)
    one.close()
    sched = (14, 2, 1, 3)
    s.run(*sched, record_samples=True, compute_mean=True)
    P, LLr, SSr, RNG, ACC, LLS, MEAN = (s.params(), s.region_loglike(), s.region_stepsizes(), s.rng_state(),
                                        s.accept_counts(), s.loglikes(), s.mean())
    for c in check:
        oc = O.MultiChain(mm, c + 1, xs[c])
        ll, smp, mean = oc.run(tgt, *sched, record=True, mean=True)
        assert np.array_equal(P[c], oc.x), f"chain {c}: parameters differ"
        assert np.array_equal(LLr[c], oc.ll) and np.array_equal(SSr[c], oc.stepsizes)
        assert oc.rng_state == tuple(int(v) for v in RNG[c]) and oc.c.n_accept == ACC[c]
        assert np.array_equal(LLS[c], ll) and np.array_equal(MEAN[c], mean) and np.array_equal(s.samples(c), smp)
    s.close()


def test_fifty_regions_500_days_131072_chains_by_properties(R11, R10):
    """configs[4] at its stated shape: 50 regions x 500 data days (horizon 600) x 131 072 chains in one context.
    52 steps = one whole scan of the regions and the next beta step.  (1) scattered chains run alone end on the
    same bits, (2) a repeat is bit-identical, (3) the first and the last chain equal the oracle's, (4) the
    shared betas stay mirrored in every block and every chain drew from its own stream."""
    R, n_data, n = 50, 500, 131072
    cfgs = long_region_cfgs(R11, R, n_data)
    dimR = cfgs[0]["dim"]
    xs = region_starts(cfgs, n)
    steps = 52
    big = H.CovidSampler(cfgs[0], R10)
    big.set_regions(cfgs)
    assert big.dim == R * dimR == 350
    big.set_chains(xs, first_chain_id=1)
    big.step(steps, 1.0, True)
    P, LLr, RNG, ACC = big.params(), big.region_loglike(), big.rng_state(), big.accept_counts()
    assert np.all(np.isfinite(LLr)) and ACC.sum() > 0 and big.scan_position() == steps
    idx = np.array([0, 63, 64, 4097, n // 2 + 77, n - 65, n - 1])
    sub = H.CovidSampler(cfgs[0], R10)
    sub.set_regions(cfgs)
    sub.set_chains(xs[idx], chain_ids=idx + 1)
    sub.step(steps, 1.0, True)
    assert np.array_equal(sub.params(), P[idx]) and np.array_equal(sub.region_loglike(), LLr[idx])
    assert np.array_equal(sub.rng_state(), RNG[idx]) and np.array_equal(sub.accept_counts(), ACC[idx])
    sub.close()
    big.set_chains(xs, first_chain_id=1)
    big.step(steps, 1.0, True)
    assert np.array_equal(P, big.params()) and np.array_equal(LLr, big.region_loglike()) and np.array_equal(RNG, big.rng_state())
    big.close()
    mm = O.MultiModel(cfgs, R11, contract=1)
    tgt = O.make_target("temp", 1.0)
    for c in (0, n - 1):
        oc = O.MultiChain(mm, c + 1, xs[c])
        for _ in range(steps):
            oc.step(tgt, 0.0, True)
        assert np.array_equal(P[c], oc.x) and np.array_equal(LLr[c], oc.ll), f"chain {c}"
        assert oc.rng_state == tuple(int(v) for v in RNG[c]) and oc.c.n_accept == ACC[c]
    blocks = P.reshape(n, R, dimR)
    iB = cfgs[0]["nB"] + cfgs[0]["n_disc"]
    assert np.all(blocks[:, :, iB] == blocks[:, :1, iB]) and np.all(blocks[:, :, iB + 1] == blocks[:, :1, iB + 1])
    assert np.all(RNG[:, 1] == (2 * (np.arange(n, dtype=np.uint64) + 1) + 1))


def test_multi_region_chains_resume_in_a_new_context(R11, R10):
    """Checkpoint / resume of MULTI-REGION chains: params + per-region step sizes + scan position
    (mcmc_hip_set_region_state) + rng, per-region log-likelihoods, mean, accept counts (mcmc_hip_set_state),
    saved after a run whose step count is NOT a multiple of R and whose adjust groups left the regions with
    different step sizes, planted in a fresh context: the second run() is bit-identical to the chains that
    never left -- and without the scan position it is not."""
    R = 3
    cfgs = long_region_cfgs(R11, R, 60, **{k: v for k, v in SMALL.items() if k != "fix_behavior_in_advance"})
    xs = region_starts(cfgs, 70, 0.05)
    stay = H.CovidSampler(cfgs[0], R10); stay.set_regions(cfgs); stay.set_chains(xs, first_chain_id=1)
    stay.run(11, 0, 5, 3, record_samples=True, compute_mean=True)          # 5 adjust groups + 11 steps: scan at 16
    assert stay.scan_position() == 16 and stay.scan_position() % R != 0
    saved = dict(params=stay.params(), ss=stay.region_stepsizes(), scan=stay.scan_position(), rng=stay.rng_state(),
                 ll=stay.region_loglike(), mean=stay.mean(), acc=stay.accept_counts())
    assert np.any(saved["ss"][:, 0] != saved["ss"][:, 1])                  # the regions adapted apart
    second = (13, 1, 2, 2)
    stay.run(*second, record_samples=True, compute_mean=True)
    moved = H.CovidSampler(cfgs[0], R10); moved.set_regions(cfgs)
    moved.set_chains(saved["params"], first_chain_id=500)
    moved.set_region_state(saved["scan"], saved["ss"])
    moved.set_state(rng_state=saved["rng"], loglike=saved["ll"], mean=saved["mean"], accept_counts=saved["acc"])
    assert moved.scan_position() == 16 and np.array_equal(moved.region_stepsizes(), saved["ss"])
    moved.run(*second, record_samples=True, compute_mean=True)
    for f in ("params", "region_loglike", "region_stepsizes", "rng_state", "accept_counts", "loglikes", "mean"):
        assert np.array_equal(getattr(stay, f)(), getattr(moved, f)()), f
    for c in (0, 64, 69):
        assert np.array_equal(stay.samples(c), moved.samples(c))
    # the scan position matters: the same plant without it restarts at region 0 and goes elsewhere
    wrong = H.CovidSampler(cfgs[0], R10); wrong.set_regions(cfgs)
    wrong.set_chains(saved["params"], first_chain_id=500)
    wrong.set_region_state(0, saved["ss"])
    wrong.set_state(rng_state=saved["rng"], loglike=saved["ll"], mean=saved["mean"], accept_counts=saved["acc"])
    wrong.run(*second, record_samples=True, compute_mean=True)
    assert not np.array_equal(wrong.params(), stay.params())
    # bad arguments are refused
    with pytest.raises(H.McmcHipError, match="negative"):
        moved.set_region_state(-1)
    bad = saved["ss"].copy(); bad[3, 1] = 0.0
    with pytest.raises(H.McmcHipError, match="positive"):
        moved.set_region_state(16, bad)
    for s in (stay, moved, wrong):
        s.close()


# ------------------------------------------------------------------ configs[3]: 64 rungs x 1024 ladders ----
def test_a_64_rung_ladder_equals_the_oracle(R11, R10):
    """One whole ladder of configs[3]'s height (64 temperatures = exactly one wavefront) + a second, against
    the oracle's chains under TempTarget(T_rung) and the exchange rule restated in test_gpu_tempering.py."""
    from test_gpu_tempering import OracleLadders, _check
    cfg = make_cfg(40, R11, **SMALL)
    m = O.Model(cfg, R11)
    temps = list(np.geomspace(1.0, 16.0, 64))
    xs = perturbed(cfg, 128, 0.05)
    s = H.CovidSampler(cfg, R10)
    s.set_chains(xs)
    s.set_ladder(temps, swap_seed=5)
    o = OracleLadders(m, xs, temps, 5)
    s.run_tempered(6, 2)
    o.run(6, 2)
    _check(s, o)
    acc, prop = s.swap_counts()
    assert list(acc) == o.accepted and sum(o.accepted) > 0
    assert list(prop) == [2 * 3] * 63                                   # 2 ladders x 3 rounds of either parity
    assert np.array_equal(np.sort(s.rungs().reshape(2, 64), axis=1), np.tile(np.arange(64), (2, 1)))
    s.close()


def test_1024_ladders_of_64_rungs_by_invariants(R11, R10):
    """configs[3] at its stated shape on the 365-day model: 65 536 chains = 1 024 ladders x 64 rungs.  Every
    ladder holds every rung exactly once after the exchanges; the swap counts and every chain's end state
    equal those of the same ladders split over two contexts (600 + 424 ladders, ids continuing); the hottest
    and coldest rung of the first and last ladder equal oracle chains that follow the recorded rung history."""
    cfg = make_cfg(365, R11, contract=1)
    K, nL = 64, 1024
    n = K * nL
    temps = np.geomspace(1.0, 16.0, K)
    xs = H.perturb_ics(CC.initial_params(cfg), n, 0.01)
    rounds, per = 4, 2
    one = H.CovidSampler(cfg, R10)
    one.set_chains(xs, first_chain_id=1)
    one.set_ladder(temps, swap_seed=9)
    hist = [one.rungs().copy()]
    for k in range(rounds):                                              # run_tempered, with the rungs read out per round
        one.step(per, 1.0, True)
        one.swap(k & 1)
        hist.append(one.rungs().copy())
    P, LL, RNG, ACC = _state(one)
    rungs = hist[-1]
    acc, prop = one.swap_counts()
    one.close()
    assert np.array_equal(np.sort(rungs.reshape(nL, K), axis=1), np.tile(np.arange(K), (nL, 1)))
    assert list(prop) == [nL * 2] * (K - 1) and 0 < acc.sum() < prop.sum()
    cut = 600 * K
    a, b = H.CovidSampler(cfg, R10), H.CovidSampler(cfg, R10)
    a.set_chains(xs[:cut], first_chain_id=1); a.set_ladder(temps, swap_seed=9, first_ladder=0)
    b.set_chains(xs[cut:], first_chain_id=cut + 1); b.set_ladder(temps, swap_seed=9, first_ladder=600)
    a.run_tempered(rounds, per); b.run_tempered(rounds, per)
    assert np.array_equal(rungs, np.concatenate([a.rungs(), b.rungs()]))
    assert np.array_equal(P, np.vstack([a.params(), b.params()])) and np.array_equal(RNG, np.vstack([a.rng_state(), b.rng_state()]))
    aa, _ = a.swap_counts(); ab, _ = b.swap_counts()
    assert np.array_equal(acc, aa + ab)
    a.close(); b.close()
    # oracle chains following the temperature each chain actually had in every round
    m = O.Model(cfg, R11, contract=1)
    for c in (0, K - 1, n - K, n - 1):
        oc = O.OracleChain(m, c + 1, xs[c])
        for k in range(rounds):
            tgt = O.make_target("temp", float(temps[hist[k][c]]))
            for _ in range(per):
                oc.step(tgt, 0.0, True)
        assert np.array_equal(oc.x, P[c]) and oc.ll == tuple(LL[c]), f"chain {c}"
        assert oc.rng_state == tuple(int(v) for v in RNG[c]) and oc.c.n_accept == ACC[c]


# ------------------------------------------------------------------ configs[2]: 65 536 chains x 365 days ----
def test_run_with_recording_at_65536_chains(R11, R10):
    """mcmc_hip_run -- adjust group, burn-in, thinning, recorded points, log-likelihood trace and running mean --
    at the headline shape (65 536 chains x 365 days, 1 024 waves): scattered chains run alone record the same
    bits, and the first and last chain equal the oracle's MetropolisChain::run, recorded points included."""
    cfg = make_cfg(365, R11, contract=1)
    n = 65536
    xs = H.perturb_ics(CC.initial_params(cfg), n, 0.01)
    sched = (8, 1, 1, 2)                       # 1 adjust group (20 steps) + 8 steps; points recorded at i = 2, 4, 6, 8
    big = H.CovidSampler(cfg, R10)
    big.set_chains(xs, first_chain_id=1)
    big.run(*sched, record_samples=True, compute_mean=True)
    assert big.n_saved == 4
    P, LL, RNG, ACC = _state(big)
    SS, LLS, MEAN = big.stepsizes(), big.loglikes(), big.mean()
    idx = np.array([0, 1, 63, 64, 4097, 9999, n // 2 + 77, n - 65, n - 1])
    pts = {int(c): big.samples(int(c)) for c in idx}
    last_all = big.sample_all(3)
    big.close()
    sub = H.CovidSampler(cfg, R10)
    sub.set_chains(xs[idx], chain_ids=idx + 1)
    sub.run(*sched, record_samples=True, compute_mean=True)
    assert np.array_equal(sub.params(), P[idx]) and np.array_equal(sub.loglike(), LL[idx])
    assert np.array_equal(sub.rng_state(), RNG[idx]) and np.array_equal(sub.accept_counts(), ACC[idx])
    assert np.array_equal(sub.stepsizes(), SS[idx]) and np.array_equal(sub.loglikes(), LLS[idx])
    assert np.array_equal(sub.mean(), MEAN[idx])
    for j, c in enumerate(idx):
        assert np.array_equal(sub.samples(j), pts[int(c)])
        assert np.array_equal(pts[int(c)][3], last_all[c])
    sub.close()
    m = O.Model(cfg, R11, contract=1)
    tgt = O.make_target("temp", 1.0)
    for c in (0, n - 1):
        oc = O.OracleChain(m, c + 1, xs[c])
        ll, smp, mean = oc.run(tgt, *sched, record=True, mean=True)
        assert np.array_equal(oc.x, P[c]) and oc.ll == tuple(LL[c]) and oc.c.n_accept == ACC[c]
        assert np.array_equal(LLS[c], ll) and np.array_equal(pts[c], smp) and np.array_equal(MEAN[c], mean)
        assert SS[c] == oc.c.stepsize


def test_fast_mode_at_65536_chains(R11, R10):
    """The fast mode (convolution on the FP32 matrix cores) at the shape bench.py times it on: 65 536 chains x
    365 days.  Bit-exact against itself -- scattered chains alone, a repeat --, every chain on its own stream,
    and the start log-likelihoods within the fast mode's tolerance of the FP64 kernel's."""
    from test_gpu_fast_mode import ABS, REL
    cfg = make_cfg(365, R11, contract=1)
    n = 65536
    xs = H.perturb_ics(CC.initial_params(cfg), n, 0.01)
    big = H.CovidSampler(cfg, R10, precision="f32_model")
    big.set_chains(xs, first_chain_id=1)
    ll0 = big.loglike().copy()
    big.step(3, 1.0, True)
    P, LL, RNG, ACC = _state(big)
    assert np.all(np.isfinite(LL))
    idx = np.array([0, 1, 63, 64, 4097, 9999, n // 2 + 77, n - 65, n - 1])
    sub = H.CovidSampler(cfg, R10, precision="f32_model")
    sub.set_chains(xs[idx], chain_ids=idx + 1)
    sub.step(3, 1.0, True)
    assert np.array_equal(sub.params(), P[idx]) and np.array_equal(sub.loglike(), LL[idx])
    assert np.array_equal(sub.rng_state(), RNG[idx]) and np.array_equal(sub.accept_counts(), ACC[idx])
    sub.close()
    big.set_chains(xs, first_chain_id=1)
    big.step(3, 1.0, True)
    assert np.array_equal(P, big.params()) and np.array_equal(LL, big.loglike()) and np.array_equal(RNG, big.rng_state())
    big.close()
    assert np.all(RNG[:, 1] == (2 * (np.arange(n, dtype=np.uint64) + 1) + 1))
    f64 = H.CovidSampler(cfg, R10)
    pick = np.r_[0:128, n - 128:n]
    ref = f64.eval(xs[pick])
    f64.close()
    assert np.all(np.abs(ref[:, 0] - ll0[pick, 0]) <= REL * np.abs(ref[:, 0]) + ABS)
    assert np.array_equal(ref[:, 1], ll0[pick, 1])                       # the prior is double in both modes
