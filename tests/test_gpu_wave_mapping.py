"""GPU parity of the SMALL-ENSEMBLE mapping (one chain per wavefront, mcmc_amd/csrc/mcmc_wave_kernel.hip.h;
include/mcmc_hip.h at mcmc_hip_set_mapping) through the C ABI.

The mapping is an extension beside north_star's one-chain-per-lane and must not change a single bit: every test
runs the same chains under "lane" and under "wave" and compares parameters, log-likelihoods, RNG states, accept
counts, step sizes, recorded traces, recorded points and means -- and holds chains of both to the CPU oracle.
Covered: run() schedules with adjust groups / burn-in / thinning / mean, given-subspace steps under other
temperatures, the hospital-capacity branch (source days with corrections take the one-lag form of the matrix
instruction), fixed breakpoints, no prior, many breakpoints, delays at both bounds, BASELINE configs[1]'s shape
(1 024 chains x 200 days) and the 365-day series, ragged ensembles, the tempering ladder, and what the mapping
refuses."""
import numpy as np
import pytest

from conftest import make_cfg, perturbed
from mcmc_amd import covid_config as CC, hip as H
from oracle import oracle as O

pytestmark = pytest.mark.gpu

SMALL = dict(max_delay=12, n_predict=5, fix_behavior_in_advance=5, initial_delay=7.0)
FIELDS = ("params", "loglike", "rng_state", "accept_counts", "stepsizes")


def _pair(cfg, R10, xs, **kw):
    out = []
    for mp in ("lane", "wave"):
        s = H.CovidSampler(cfg, R10)
        s.set_mapping(mp)
        s.set_chains(xs, **kw)
        assert s.mapping() == mp
        out.append(s)
    return out


def _same(a, b, fields=FIELDS):
    for f in fields:
        assert np.array_equal(getattr(a, f)(), getattr(b, f)()), f


def _oracle_equal(oc, s, c):
    assert np.array_equal(oc.x, s.params()[c]), f"chain {c}: parameters"
    assert oc.ll == tuple(s.loglike()[c]), f"chain {c}: loglike"
    assert oc.rng_state == tuple(int(v) for v in s.rng_state()[c]) and oc.c.n_accept == s.accept_counts()[c]


@pytest.mark.parametrize("contract", [1, 0])
def test_wave_mapping_runs_the_same_chains_as_lane_and_oracle(R11, R10, contract):
    cfg = make_cfg(40, R11, contract=contract, **SMALL)
    m = O.Model(cfg, R11, contract=contract)
    xs = perturbed(cfg, 70, 0.05)                      # a ragged ensemble: 70 wavefronts in 18 workgroups
    lane, wave = _pair(cfg, R10, xs, first_chain_id=1)
    sched = (40, 3, 4, 3)
    for s in (lane, wave):
        s.run(*sched, record_samples=True, compute_mean=True)
    _same(lane, wave, FIELDS + ("loglikes", "mean"))
    assert wave.accept_counts().sum() > 0
    tgt = O.make_target("temp", 1.0)
    for c in (0, 1, 63, 64, 69):
        assert np.array_equal(lane.samples(c), wave.samples(c))
        oc = O.OracleChain(m, c + 1, xs[c])
        ll, smp, mean = oc.run(tgt, *sched, record=True, mean=True)
        _oracle_equal(oc, wave, c)
        assert np.array_equal(wave.loglikes()[c], ll) and np.array_equal(wave.samples(c), smp) and np.array_equal(wave.mean()[c], mean)
        assert wave.stepsizes()[c] == oc.c.stepsize
    # given-subspace steps under another temperature, then the mapping switched mid-run: the chains go on
    for s in (lane, wave):
        s.step(5, 0.4, False)
    _same(lane, wave)
    wave.set_mapping("lane"); lane.set_mapping("wave")
    assert wave.mapping() == "lane" and lane.mapping() == "wave"
    for s in (lane, wave):
        s.step(7, 1.0, True)
    _same(lane, wave)
    assert lane.pair_count() == wave.pair_count() and lane.executed_pair_count() == wave.executed_pair_count()
    lane.close(); wave.close()


@pytest.mark.parametrize("case", ["overflow", "overflow_fixed_no_prior", "many_breakpoints", "delay_bounds", "plain_target",
                                  "tiny_behaviour"])
@pytest.mark.parametrize("contract", [1, 0])
def test_wave_mapping_on_the_models_branches(R11, R10, case, contract):
    kw, a, steps, target = dict(SMALL, contract=contract), 0.1, 30, ("temp", 1.0)
    if case == "overflow":                 # capacity corrections on many days (covid.h:413-429), switching on and off
        cfg = make_cfg(60, R11, cap0=3.0, cap_incr_rate=0.5, initial_beta_mild=1.5, initial_beta_high=4.0, **kw)
    elif case == "overflow_fixed_no_prior":
        cfg = make_cfg(50, R11, cap0=3.0, cap_incr_rate=0.5, initial_beta_mild=1.5, initial_beta_high=4.0,
                       disc_fixed=[0, 1, 0, 1], has_prior=0, **kw)
        a, target = 0.2, ("plain",)
    elif case == "many_breakpoints":
        cfg = make_cfg(60, R11, disc_days=(1, 4, 9, 14, 20, 27, 33, 41), disc_vals=(1, .9, .8, .7, .6, .5, .4, .3),
                       disc_fixed=[1, 0, 0, 0, 0, 0, 0, 0], **kw)
        a = 0.02
    elif case == "delay_bounds":
        cfg = make_cfg(45, R11, **kw)
    elif case == "tiny_behaviour":         # numerators of the day's division down in the denormals: the wave kernel's short
        cfg = make_cfg(44, R11, **kw)          # division hands the evaluation over to the compiler's (wave_div_pop)
    else:
        cfg = make_cfg(48, R11, target_kind="plain", **kw)
        target = ("plain",)
    xs = perturbed(cfg, 66, a)
    if case == "tiny_behaviour":
        xs[::2, 3] = 1e-305
        xs[1::4, 10] = 3e-300
        xs[2::4, 7] = 0.0
    if case == "delay_bounds":             # first days at both ends of [5, maxDelay]: start = 7 and 0, frac = 0 and not
        xs[::3, -1] = 5.0
        xs[1::3, -1] = 12.0
        xs[2::3, -1] = 11.999
    if case == "overflow":
        xs[::3, -3] *= 0.2                 # a third of the chains barely grow: no overflow at all
    lane, wave = _pair(cfg, R10, xs, first_chain_id=1)
    for s in (lane, wave):
        s.step(steps, 1.0, True)
    _same(lane, wave)
    m = O.Model(cfg, R11, contract=contract)
    tgt = O.make_target(*target)
    for c in range(0, 66, 7):
        oc = O.OracleChain(m, c + 1, xs[c])
        for _ in range(steps):
            oc.step(tgt, 0.0, True)
        _oracle_equal(oc, wave, c)
    if case.startswith("overflow"):        # the branch was taken: the derived `serious` curve crosses the capacity line
        cur = wave.derive(np.vstack([xs[:16], wave.params()[:16]]))
        cap = cfg["cap0"] + np.maximum(cfg["cap_incr_rate"] * (np.arange(cfg["size"]) - cfg["max_delay"]), 0)
        over = (cur[:, 5] * cfg["prob_icu_if_serious"] > cap).any(axis=1)
        assert over.any() and not over.all()
    lane.close(); wave.close()


@pytest.mark.parametrize("n_data", [8, 15, 33, 64, 129])
def test_wave_mapping_on_odd_series_lengths(R11, R10, n_data):
    """Series shorter than a block, one day more than a block, exactly two blocks, ...: block and group edges of the
    wave kernel (32-day blocks, 16-lag groups, 16-term groups of the ordered sums) against the lane kernel and the oracle."""
    cfg = make_cfg(n_data, R11, contract=1, cap0=40.0, cap_incr_rate=2.0, **SMALL)
    xs = perturbed(cfg, 5, 0.05)
    xs[1, -1] = 5.0                        # the latest first day
    xs[2, -1] = 12.0                       # the earliest
    lane, wave = _pair(cfg, R10, xs, first_chain_id=1)
    for s in (lane, wave):
        s.run(20, 2, 2, 2, record_samples=True, compute_mean=True)
    _same(lane, wave, FIELDS + ("loglikes", "mean"))
    m = O.Model(cfg, R11, contract=1)
    tgt = O.make_target("temp", 1.0)
    for c in range(5):
        oc = O.OracleChain(m, c + 1, xs[c])
        oc.run(tgt, 20, 2, 2, 2)
        _oracle_equal(oc, wave, c)
    lane.close(); wave.close()


def test_wave_mapping_at_baseline_config_1(R11, R10):
    """BASELINE configs[1]: 1 024 chains x 200 days.  AUTO picks the wave mapping for it; every chain's end state
    after an adjust group and sampling steps equals the lane mapping's, four chains equal the oracle's."""
    cfg = make_cfg(200, R11, contract=1)
    n = 1024
    xs = H.perturb_ics(CC.initial_params(cfg), n, 0.01)
    auto = H.CovidSampler(cfg, R10)
    auto.set_chains(xs, first_chain_id=1)
    assert auto.mapping() == "wave"
    lane = H.CovidSampler(cfg, R10)
    lane.set_mapping("lane")
    lane.set_chains(xs, first_chain_id=1)
    sched = (12, 0, 1, 4)
    for s in (auto, lane):
        s.run(*sched, record_samples=True, compute_mean=True)
    _same(auto, lane, FIELDS + ("loglikes", "mean"))
    assert np.array_equal(auto.sample_all(2), lane.sample_all(2))
    m = O.Model(cfg, R11, contract=1)
    tgt = O.make_target("temp", 1.0)
    for c in (0, 511, 1023):
        oc = O.OracleChain(m, c + 1, xs[c])
        oc.run(tgt, *sched)
        _oracle_equal(oc, auto, c)
    auto.close(); lane.close()


def test_wave_mapping_over_a_long_schedule(R11, R10):
    """300 steps with five adjust groups, burn-in, thinning and the running mean on the 200-day series: every recorded
    number of 256 chains under the wave mapping equals the lane mapping's (corrections switch on and off along the way)."""
    cfg = make_cfg(200, R11, contract=1)
    xs = H.perturb_ics(CC.initial_params(cfg), 256, 0.01)
    lane, wave = _pair(cfg, R10, xs, first_chain_id=1)
    for s in (lane, wave):
        s.run(200, 20, 5, 3, record_samples=True, compute_mean=True)
    _same(lane, wave, FIELDS + ("loglikes", "mean"))
    assert np.array_equal(lane.sample_all(0), wave.sample_all(0)) and np.array_equal(lane.sample_all(40), wave.sample_all(40))
    assert wave.accept_counts().sum() > 0 and np.isfinite(wave.loglike()).all()
    lane.close(); wave.close()


def test_wave_mapping_on_the_365_day_series_and_auto_threshold(R11, R10):
    cfg = make_cfg(365, R11, contract=1)
    xs = H.perturb_ics(CC.initial_params(cfg), 300, 0.01)
    lane, wave = _pair(cfg, R10, xs, first_chain_id=1)
    for s in (lane, wave):
        s.step(6, 1.0, True)
    _same(lane, wave)
    assert lane.pair_count() == wave.pair_count()
    m = O.Model(cfg, R11, contract=1)
    tgt = O.make_target("temp", 1.0)
    oc = O.OracleChain(m, 300, xs[299])
    for _ in range(6):
        oc.step(tgt, 0.0, True)
    _oracle_equal(oc, wave, 299)
    lane.close(); wave.close()
    # AUTO: small ensembles one chain per wavefront, large ones one per lane
    s = H.CovidSampler(make_cfg(40, R11, contract=1, **SMALL), R10)
    c40 = s.cfg
    s.set_chains(perturbed(c40, 10, 0.05)); assert s.mapping() == "wave"
    s.set_chains(H.perturb_ics(CC.initial_params(c40), 16385, 0.01)); assert s.mapping() == "lane"
    s.set_chains(H.perturb_ics(CC.initial_params(c40), 16384, 0.01)); assert s.mapping() == "wave"
    s.close()


def test_wave_mapping_under_a_tempering_ladder(R11, R10):
    cfg = make_cfg(40, R11, contract=1, **SMALL)
    temps = [1.0, 1.7, 3.0, 8.0]
    xs = perturbed(cfg, 72, 0.05)
    lane, wave = _pair(cfg, R10, xs)
    for s in (lane, wave):
        s.set_ladder(temps, swap_seed=7)
        s.run_tempered(7, 3)
    _same(lane, wave)
    assert np.array_equal(lane.rungs(), wave.rungs()) and np.array_equal(lane.swap_counts()[0], wave.swap_counts()[0])
    lane.close(); wave.close()


def test_wave_mapping_refuses_what_it_cannot_run(R11, R10):
    strict = H.CovidSampler(make_cfg(40, R11, contract=0, **SMALL), R10)      # strict arithmetic runs under both mappings
    strict.set_chains(perturbed(strict.cfg, 8, 0.05)); assert strict.mapping() == "wave"
    strict.close()
    fast = H.CovidSampler(make_cfg(40, R11, contract=1, **SMALL), R10, precision="f32_model")
    with pytest.raises(H.McmcHipError, match="fast mode"):
        fast.set_mapping("wave")
    fast.set_chains(perturbed(fast.cfg, 8, 0.05)); assert fast.mapping() == "lane"
    fast.close()
    long = H.CovidSampler(make_cfg(1600, R11, contract=1), R10)      # 365 days: 4 chains per workgroup; 1200: 1; 1600: none fits
    with pytest.raises(H.McmcHipError, match="too long"):
        long.set_mapping("wave")
    long.set_chains(H.perturb_ics(CC.initial_params(long.cfg), 8, 0.01)); assert long.mapping() == "lane"
    mid = H.CovidSampler(make_cfg(500, R11, contract=1), R10)        # two chains per workgroup: AUTO's limit halves
    x500 = H.perturb_ics(CC.initial_params(mid.cfg), 8193, 0.01)
    mid.set_chains(x500); assert mid.mapping() == "lane"
    mid.set_chains(x500[:8192]); assert mid.mapping() == "wave"
    mid.close()
    with pytest.raises(H.McmcHipError, match="unknown mapping"):
        H._check(H.lib().mcmc_hip_set_mapping(long._h, 9))
    long.close()
