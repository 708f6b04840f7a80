"""GPU tests of the pybind11 module `mcmc` (Python surface of the reference for the covid
path): the flow of notebooks/covid.ipynb cells 3-7 -- data, params, DiseaseSpread,
SmoothnessPrior, State, TempTarget, Chain.run, ChainManager.run_all -- gives, bit for bit,
what the CPU oracle gives for the same chain ids and inputs."""
import os
import sys

import numpy as np
import pytest

from conftest import ABI_CURVE_ORDER, ROOT

sys.path.insert(0, os.path.join(ROOT, "mcmc_amd", "pymod"))
mcmc = pytest.importorskip("mcmc", reason="build it with `make -C mcmc_amd/pymod`")

from mcmc_amd import covid_config as CC  # noqa: E402
from oracle import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu

N_TRAJ = 200_000
DISC = np.array([[1, 9, 12, 29], [1, .6, .5, .4], [1, 1, 1, 1]], dtype=float)


def _setup(T=1.0, cooling=None):
    """covid.ipynb cells 3-4 with the Italy series that is hard-coded there."""
    deaths = CC.ITALY_DEATHS
    sigma = CC.sigma_recipe(deaths)
    data = mcmc.DiseaseData(np.array([deaths, sigma]), DISC)
    data.initialBetaMild, data.initialBetaHigh, data.initialDelay = 0.8, 2.5, 23.0
    p = mcmc.DiseaseParams()
    dp = CC.ipynb_disease_params()
    for k, v in dp.items():
        setattr(p, k, v)
    ds = mcmc.DiseaseSpread(data, p, 60000000, 7500.0, 100.0, 40, 60, N_TRAJ)
    nB = len(deaths) - data.fixBehaviorInAdvance
    state = mcmc.State()
    state.add(ds)
    state.add(mcmc.SmoothnessPrior("behavior", 50.0, float(nB)))
    state.init()
    target = mcmc.CoolingTarget(*cooling) if cooling else mcmc.TempTarget(T)
    target.set_posterior(state)
    # the same model for the oracle (FMA contraction: the host side's default)
    R11 = np.zeros((O.NCURVES, 500))
    R11[ABI_CURVE_ORDER] = ds.response_table()
    kind = dict(target_kind="cooling", target_slope=cooling[0], target_Tinit=cooling[1]) if cooling else \
        dict(target_kind="temp", target_T=T)
    cfg = CC.make_config(deaths, sigma, disease_params=dp, initial_beta_mild=0.8, initial_beta_high=2.5,
                         initial_delay=23.0, contract=1, **kind)
    m = O.Model(cfg, R11, contract=1)
    tgt = O.make_target("cooling", 1.0, *cooling) if cooling else O.make_target("temp", T)
    return data, p, ds, state, target, cfg, m, tgt


def test_state_init_evaluates_on_the_gpu():
    data, p, ds, state, target, cfg, m, tgt = _setup()
    x0 = CC.initial_params(cfg)
    ll, curves = m.eval(x0, curves=True)
    assert state.loglike() == ll + m.prior(x0)
    allc = state.get_all()
    assert set(allc) == set(ds.get_names())
    assert np.array_equal(allc["dead"], curves[7]) and np.array_equal(allc["totalBehavior"], curves[8])
    assert allc["delay"] == [23.0]


@pytest.mark.parametrize("cooling", [None, (2.0, 5.0)])
def test_chain_run_equals_oracle_chain(cooling):
    data, p, ds, state, target, cfg, m, tgt = _setup(T=2.0, cooling=cooling)
    x0 = CC.initial_params(cfg)
    chain = mcmc.Chain(target, 0)                      # id 0 like test.cpp:36
    assert chain.computeMean is False                  # mcmc.h:990
    chain.computeMean = True
    chain.run(60, 4, 3, 5)
    oc = O.OracleChain(m, 0, x0)
    ll, samples, mean = oc.run(tgt, 60, 4, 3, 5, record=True, mean=True)
    assert np.array_equal(chain.get_loglikes(), ll)
    nB, nD = cfg["nB"], cfg["n_disc"]
    assert np.array_equal(chain.get_samples("behavior"), samples[:, :nB])
    assert np.array_equal(chain.get_samples("discontinuousVals"), samples[:, nB:nB + nD])
    assert np.array_equal(chain.get_samples("delay"), samples[:, -1:])
    assert np.array_equal(chain.get_mean("betaMild"), mean[nB + nD:nB + nD + 1])
    assert np.array_equal(chain.get_mean("behavior"), mean[:nB])
    assert np.array_equal(chain.get_weights(), np.ones(len(ll)))
    # derived coords of the saved samples are computed on request (compute_derived_late)
    dead = chain.get_samples("dead")
    want = np.array([m.eval(s, curves=True)[1][7] for s in samples])
    assert dead.shape == (len(ll), cfg["size"]) and np.array_equal(dead, want)
    # the target's state now is the chain's end state, step size included
    assert ds.getCoords("delay") == [oc.x[-1]] and ds.stepsizeCorrectionFac == oc.c.stepsize
    assert state.loglike() == oc.ll[0] + oc.ll[1]
    # reevaluate drops burn-in points (mcmc.h:1160-1161)
    chain.reevaluate(state, 2)
    assert np.array_equal(chain.get_loglikes(), ll[2:])
    assert np.array_equal(chain.get_samples("betaHigh"), samples[2:, -2:-1])


def test_a_second_run_continues_the_chain():
    """A MetropolisChain seeds `rnd` once, in its constructor (mcmc.h:994-999): run(a) then run(b) goes on
    drawing from the same stream, APPENDS to loglikes / samples (mcmc.h:1039-1061) and keeps adding to
    the mean it has (mcmc.h:1064-1065,1133-1134) -- the notebooks do call run repeatedly
    (covid2.ipynb, covid3.ipynb).  Against one oracle chain run twice."""
    data, p, ds, state, target, cfg, m, tgt = _setup(T=2.0)
    x0 = CC.initial_params(cfg)
    chain = mcmc.Chain(target, 7)
    chain.computeMean = True
    chain.run(12, 0, 2, 1)
    chain.run(9, 0, 0, 1)
    oc = O.OracleChain(m, 7, x0)
    ll1, s1, mean1 = oc.run(tgt, 12, 0, 2, 1, record=True, mean=True)
    ll2, s2, _ = oc.run(tgt, 9, 0, 0, 1, record=True, mean=False)
    assert len(ll1) == 14 and len(ll2) == 9
    assert np.array_equal(chain.get_loglikes(), np.concatenate([ll1, ll2]))
    both = np.concatenate([s1, s2])
    nB, nD = cfg["nB"], cfg["n_disc"]
    assert np.array_equal(chain.get_samples("delay"), both[:, -1:])
    assert np.array_equal(chain.get_samples("behavior"), both[:, :nB])
    # the second run adds its points i > nBurnin (thinning 1: every saved point but the first) to the
    # first run's scaled mean and scales again
    mu = mean1.copy()
    for pt in s2[1:]:
        mu = mu + pt
    mu = mu * (1.0 / 9)
    assert np.array_equal(chain.get_mean("behavior"), mu[:nB])
    assert np.array_equal(chain.get_mean("delay"), mu[-1:])
    assert ds.getCoords("delay") == [oc.x[-1]] and ds.stepsizeCorrectionFac == oc.c.stepsize
    assert state.loglike() == oc.ll[0] + oc.ll[1]


def test_fast_mode_through_the_module():
    """DiseaseSpread.fastMode (extension): the module's chains run the packed-FP32 convolution;
    log-likelihoods stay within the fast mode's tolerance of the double ones (tests/test_gpu_fast_mode.py)."""
    data, p, ds, state, target, cfg, m, tgt = _setup(T=1.0)
    ll64 = state.loglike()
    ds.fastMode = True
    state.init()
    ll32 = state.loglike()
    assert ll32 != ll64 and abs(ll32 - ll64) <= 2e-5 * abs(ll64) + 2e-2
    chain = mcmc.Chain(target, 3)
    chain.run(40, 0, 2, 5)
    lls = np.asarray(chain.get_loglikes())
    oc = O.OracleChain(m, 3, CC.initial_params(cfg))
    ref, _, _ = oc.run(tgt, 40, 0, 2, 5, record=False)
    assert len(lls) == len(ref) and np.all(np.isfinite(lls))
    assert abs(lls[0] - ref[0]) <= 2e-5 * abs(ref[0]) + 2e-2          # the start point, before the chains diverge


def test_tempering_ladders_through_the_module():
    """ChainManager.run_tempered (extension, BASELINE configs[3]) is the C ABI's ladder: the same chains
    through the ctypes view -- which tests/test_gpu_tempering.py holds to the oracle -- end bit-identically."""
    from mcmc_amd import hip as H
    data, p, ds, state, target, cfg, m, tgt = _setup(T=1.0)
    x0 = CC.initial_params(cfg)
    rng = np.random.default_rng(5)
    xs = x0 * (1 + 0.02 * (rng.random((8, x0.size)) - 0.5))
    temps = [1.0, 2.0, 4.0, 8.0]
    cm = mcmc.ChainManager(target, 2)
    cm.set_initial_conditions([list(x) for x in xs])
    cm.run_tempered(temps, 5, 4, 9)                      # 2 ladders x 4 rungs, 5 rounds of 4 steps
    R10 = np.ascontiguousarray(ds.response_table())
    s = H.CovidSampler(cfg, R10)
    s.set_chains(xs)
    s.set_ladder(temps, swap_seed=9, first_ladder=0)
    s.run_tempered(5, 4)
    assert np.array_equal(cm.all_final_params(), s.params())
    assert list(cm.tempered_rungs()) == list(s.rungs())
    acc, prop = cm.swap_counts()
    a2, p2 = s.swap_counts()
    assert list(acc) == list(a2) and list(prop) == list(p2) and sum(prop) > 0
    c = cm.get_chain(5)
    assert c.weight == 1 and ds is not None
    s.close()
    cm.run_tempered([1.0, 2.0, 3.0], 1, 1, 1)            # 8 chains are not a multiple of 3: refused, prints
    with pytest.raises(IndexError):
        cm.get_chain(0)


def test_text_dump_of_run_and_reevaluate(tmp_path, monkeypatch):
    """samples<id>.txt: "chainweight w" from run (mcmc.h:1018), "chainweight = w" from
    reevaluate (mcmc.h:1167); then per saved point "weight w loglike l" and one
    "name v v ..." line per coordinate in std::map order (mcmc.h:1050-1058)."""
    data, p, ds, state, target, cfg, m, tgt = _setup(T=1.0)
    x0 = CC.initial_params(cfg)
    monkeypatch.chdir(tmp_path)
    chain = mcmc.Chain(target, 7)
    chain.writeSamplesToDisk = True
    chain.run(20, 0, 2, 5)
    oc = O.OracleChain(m, 7, x0)
    ll, samples, _ = oc.run(tgt, 20, 0, 2, 5, record=True)

    def blocks(txt):
        head, *rest = txt.strip().split("\n\n")
        return head, [b.split("\n") for b in rest]

    head, bl = blocks(open(tmp_path / "samples7.txt").read())
    assert head == "chainweight 1"
    assert len(bl) == len(ll)
    names = sorted(ds.get_names())
    for b, l, x in zip(bl, ll, samples):
        w = b[0].split()
        assert w[0] == "weight" and float(w[1]) == 1 and w[2] == "loglike"
        assert float(w[3]) == pytest.approx(l, rel=1e-5)              # operator<< prints 6 digits
        assert [line.split()[0] for line in b[1:]] == names
        row = {line.split()[0]: np.array(line.split()[1:], dtype=float) for line in b[1:]}
        assert row["delay"][0] == pytest.approx(x[-1], rel=1e-5)
        assert len(row["dead"]) == cfg["size"] and len(row["behavior"]) == cfg["nB"]
        assert np.allclose(row["dead"], m.eval(x, curves=True)[1][7], rtol=1e-5, atol=1e-300)
    chain.reevaluate(state, 1)
    head, bl = blocks(open(tmp_path / "samples7.txt").read())
    assert head == "chainweight = 1" and len(bl) == len(ll) - 1


def test_chain_manager_runs_all_chains_in_one_batch(tmp_path, monkeypatch):
    data, p, ds, state, target, cfg, m, tgt = _setup(T=1.0)
    x0 = CC.initial_params(cfg)
    cm = mcmc.ChainManager(target, 5)
    cm.run_all(30, 2, 3, 4)                # bound to run_all_adjust(nSteps, nAdjust, thinning, nThread), test.cpp:124
    for i in (0, 3, 4):
        oc = O.OracleChain(m, i + 1, x0)   # chain i draws pcg32(0, i+1), mcmc.h:1527
        ll, samples, _ = oc.run(tgt, 30, 0, 2, 3, record=True)
        c = cm.get_chain(i)
        assert np.array_equal(c.get_loglikes(), ll)
        assert c.weight == 1
    assert cm.all_loglikes().shape == (5, len(ll))
    oc0 = O.OracleChain(m, 1, x0)
    oc0.run(tgt, 30, 0, 2, 3, record=False)
    assert cm.stepsizes[0] == oc0.c.stepsize          # stepsizes = chains[0]'s, mcmc.h:1550
    with pytest.raises(IndexError):
        cm.get_chain(7)
    # the saved points stay on the device and come over on request: point 1 of every chain
    pts = cm.saved_sample(1)
    oc3 = O.OracleChain(m, 4, x0)
    assert pts.shape == (5, cfg["dim"]) and np.array_equal(pts[3], oc3.run(tgt, 30, 0, 2, 3, record=True)[1][1])
    # recordInitialConditions = False: the log-likelihood trace only
    lean = mcmc.ChainManager(target, 5)
    lean.recordInitialConditions = False
    lean.run_all(30, 2, 3, 4)
    assert np.array_equal(lean.get_chain(3).get_loglikes(), cm.get_chain(3).get_loglikes())
    assert np.array_equal(lean.all_final_params(), cm.all_final_params())
    with pytest.raises(IndexError):
        lean.saved_sample(0)
    # samples are available after reevaluate_all (run_all records only the points, mcmc.h:1490)
    monkeypatch.chdir(tmp_path)
    cm.reevaluate_all(state, 1, True, True, 1)
    c = cm.get_chain(3)
    oc3 = O.OracleChain(m, 4, x0)
    _, samples3, _ = oc3.run(tgt, 30, 0, 2, 3, record=True)
    assert np.array_equal(c.get_samples("delay"), samples3[1:, -1:])
    txt = open(tmp_path / "samples4.txt").read()
    assert txt.startswith("chainweight = 1") and "loglike" in txt and "\ndead " in txt
    # run_all_adjust is bound to run_all(nSteps, thinning, stepsizefac, nThread), test.cpp:125
    ss = cm.stepsizes[0]
    cm.run_all_adjust(10, 2, 0.5, 1)
    oc = O.OracleChain(m, 2, x0, stepsize=ss * 0.5)
    ll2, _, _ = oc.run(tgt, 10, 0, 0, 2, record=False)
    assert np.array_equal(cm.get_chain(1).get_loglikes(), ll2)


def test_reservoir_and_generator_constructors():
    data, p, ds, state, target, cfg, m, tgt = _setup(T=1.0)
    # reservoir: DiseaseSpread has no sampleInitialConditions (mcmc.h:112-114), so all
    # candidates coincide; the bootstrap then finds one distinct state with weight nChain
    cm = mcmc.ChainManager(target, 6, 1)
    assert cm.n_chains() == 1 and cm.chain_weight(0) == 1
    # generator chain at T=10 seeds chains for the T=1 target (notebooks/test2.ipynb flow)
    hot = mcmc.TempTarget(10.0)
    hot.set_posterior(state)
    gen = mcmc.Chain(hot, 0)
    gen.run(200, 0, 2, 10)
    cm2 = mcmc.ChainManager(gen, target, 2)
    assert cm2.n_chains() == 2 and sum(cm2.chain_weight(i) for i in range(2)) >= 2
    cm2.run_all(10, 0, 5, 1)
    assert cm2.all_loglikes().shape == (2, 2) and np.all(np.isfinite(cm2.all_loglikes()))
    # the bootstrap picks follow the discrete CDF of exp(new - old) with pcg32(0,0) draws
    lls = gen.get_loglikes()
    pts = np.hstack([gen.get_samples(k) for k in ("behavior", "discontinuousVals", "betaMild", "betaHigh", "delay")])
    new = np.array([m.eval(x) + m.prior(x) for x in pts])
    w = np.exp((new - lls) - np.max(new - lls))
    cdf = np.cumsum(w)
    cdf[-1] = 1.000001 * w.sum()
    total = w.sum()
    r = O.Rng(0, 0)
    picks = []
    while len(picks) < 2:
        idx = int(np.searchsorted(cdf, r.next_double() * total, side="right"))
        if idx not in picks:
            picks.append(idx)
    first = mcmc.ChainManager(gen, target, 2)
    first.run_all(1, 0, 1, 1)              # one step, one saved point = the initial conditions
    assert np.array_equal(first.saved_sample(0), pts[picks])


def test_initial_spread_lets_a_manager_seed_distinct_covid_chains():
    """DiseaseSpread.initialSpread (extension, off by default, no counterpart in the reference: its
    DiseaseSpread keeps the base's empty sampleInitialConditions, mcmc.h:112-114, so a manager's chains
    all start at one point -- the test above).  With a > 0 draw i is x0 + (nextFloat() - .5f) * x0 * a per
    sampled coordinate in the coords' order (State::perturb's rule, mcmc.h:306-314) from the manager's
    pcg32(0, 0), then force_bounds; fixed coordinates take their draw and keep their value."""
    data, p, ds, state, target, cfg, m, tgt = _setup(T=1.0)
    x0 = CC.initial_params(cfg)
    nB, nD = cfg["nB"], cfg["n_disc"]
    a = 0.05
    ds.initialSpread = a
    cm = mcmc.ChainManager(target, 6)
    assert cm.n_chains() == 6
    cm.run_all(1, 0, 1, 1)                  # one step, one saved point = the initial conditions
    got = cm.saved_sample(0)
    r = O.Rng(0, 0)
    want = []
    for _ in range(6):
        f = np.array([np.float32(r.next_float()) - np.float32(0.5) for _ in range(cfg["dim"])], dtype=np.float64)
        x = x0 + f * x0 * a
        x = m.force_bounds(x)               # covid.h:629-646 (DISC's flags are >= 0: every plateau value is sampled)
        want.append(x)
    want = np.array(want)
    assert np.array_equal(got, want)
    assert len({tuple(row) for row in got}) == 6
    # every chain starts from its own point and runs as the oracle's chain from there
    oc = O.OracleChain(m, 3, want[2])       # chain 2 draws pcg32(0, 3), mcmc.h:1527
    assert cm.get_chain(2).get_loglikes()[0] == oc.ll[0] + oc.ll[1]
    # the reservoir constructor now finds distinct candidates to pick from (a hot target: at T = 1 the
    # posterior is so peaked that one or two of 40 candidates carry all of the weight)
    ds2 = _setup(T=1e7)
    ds2[2].initialSpread = a
    picked = mcmc.ChainManager(ds2[4], 40, 5)
    assert picked.n_chains() == 5 and sum(picked.chain_weight(i) for i in range(5)) >= 5
    picked.run_all(1, 0, 1, 1)
    assert len({tuple(row) for row in picked.saved_sample(0)}) == 5
    # off again: the reference's behaviour, one point
    ds2[2].initialSpread = 0.0
    assert mcmc.ChainManager(ds2[4], 6, 1).n_chains() == 1
