"""GPU tests of the toy targets through the pybind11 module `mcmc` (BASELINE configs[0]):
the CLI run of test.cpp:27-40 (MyState, chain 0, run(100000, 10, 500, 5)) and the flow of
notebooks/test2.ipynb (FourGaussians(6): hot generator chain, reservoir and generator
ChainManagers), bit for bit against the CPU oracle."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "mcmc_amd", "pymod"))
mcmc = pytest.importorskip("mcmc", reason="build it with `make -C mcmc_amd/pymod`")

from oracle import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu


def _fg_ll(x, difficulty=6.0):
    L = O.lib()
    return np.array([L.orc_fourgauss_eval(difficulty, np.ascontiguousarray(v, dtype=np.float64).ctypes.data_as(
        O.C.POINTER(O.C.c_double))) for v in np.atleast_2d(x)])


def test_cli_config_through_the_module():
    t = mcmc.Target()
    s = mcmc.MyState()                                 # evaluates MyLike1 at (1,1,1) on the GPU
    assert s.loglike() == -11.0 and s.get_all() == {"position": [1, 1, 1], "max": [1]}
    t.set_posterior(s)
    chain = mcmc.Chain(t, 0)
    chain.computeMean = True                           # default false, mcmc.h:990
    chain.run(100000, 10, 500, 5)
    ll, smp, sf, nacc, mean = O.mylike1_run(0, 100000, 10, 500, 5, with_mean=True)
    assert np.array_equal(chain.get_loglikes(), ll)
    assert np.array_equal(chain.get_samples("position"), smp)
    assert np.array_equal(chain.get_samples("max"), smp.max(axis=1, keepdims=True))
    assert np.array_equal(chain.get_mean("position"), mean[:3]) and np.array_equal(chain.get_mean("max"), mean[3:])
    assert np.array_equal(chain.get_weights(), np.ones(len(ll)))
    # posterior of the two unit Gaussians at (2,2,0), (-2,-2,0): mean 0, variance 1/2
    post = chain.get_samples("position")[200:]
    assert np.all(np.abs(post.mean(axis=0)) < 0.05) and np.all(np.abs(post.var(axis=0) - 0.5) < 0.05)
    # the target's state is the chain's end state
    assert s.get_all()["max"] == [max(s.get_all()["position"])]


def _four_gaussians(T):
    lh = mcmc.FourGaussians(6)
    s = mcmc.State()
    s.add(lh)
    t = mcmc.TempTarget(T)
    t.set_posterior(s)
    return lh, s, t


def test_four_gaussians_generator_chain_and_reevaluate():
    lh, s, t = _four_gaussians(10)                     # test2.ipynb cell 2
    assert s.loglike() == _fg_ll([0.0, 0.0])[0]
    c = mcmc.Chain(t, 0)
    c.computeMean = True
    c.run(5000, 500, 200, 10)                          # cell 3, smaller
    ll, smp, sf, nacc, mean = O.fourgauss_run(6.0, 10.0, 0, [0.0, 0.0], 5000, 500, 200, 10, with_mean=True)
    assert np.array_equal(c.get_loglikes(), ll) and np.array_equal(c.get_samples("pos"), smp)
    assert np.array_equal(c.get_mean("pos"), mean) and lh.stepsizeCorrectionFac == sf
    c.reevaluate(s, 100)                               # cell 4
    assert np.array_equal(c.get_loglikes(), ll[100:]) and np.array_equal(c.get_samples("pos"), smp[100:])


def test_toy_chain_continues_over_two_runs():
    """rnd is seeded once per chain (mcmc.h:994-999): without an adjust phase, under a constant beta and with
    thinning 1, run(300) followed by run(200) IS run(500) -- same stream, points appended."""
    lh1, s1, t1 = _four_gaussians(10)
    a = mcmc.Chain(t1, 4)
    a.run(300, 0, 0, 1)
    a.run(200, 0, 0, 1)
    lh2, s2, t2 = _four_gaussians(10)
    b = mcmc.Chain(t2, 4)
    b.run(500, 0, 0, 1)
    assert np.array_equal(a.get_samples("pos"), b.get_samples("pos")) and len(a.get_loglikes()) == 500
    assert np.array_equal(a.get_loglikes(), b.get_loglikes())
    assert s1.get_all() == s2.get_all()
    # and a fresh chain object with the same id starts the stream again
    lh3, s3, t3 = _four_gaussians(10)
    c = mcmc.Chain(t3, 4)
    c.run(300, 0, 0, 1)
    assert np.array_equal(c.get_samples("pos"), b.get_samples("pos")[:300])


def _bootstrap_picks(r, w_log, n):
    w = np.exp(w_log - np.max(w_log))
    cdf = np.cumsum(w)
    total = w.sum()
    cdf[-1] = 1.000001 * total
    picks, weights = [], {}
    while len(picks) < n:
        idx = int(np.searchsorted(cdf, r.next_double() * total, side="right"))
        if idx in weights:
            weights[idx] += 1
        else:
            weights[idx] = 1
            picks.append(idx)
    return picks, [weights[i] for i in picks]


def test_reservoir_manager_follows_the_reference_draws():
    """ChainManager(target, nReservoir, nChain), mcmc.h:1348-1374 (test2.ipynb cell 7)."""
    lh, s, t2 = _four_gaussians(1.0)
    n_res, n_chain = 3000, 40
    cm = mcmc.ChainManager(t2, n_res, n_chain)
    r = O.Rng(0, 0)
    cand = []
    while len(cand) < n_res:                           # candidates with a finite log probability, in draw order
        x = [-5 + 10 * r.next_double(), -5 + 10 * r.next_double()]      # test.h:116-123
        if np.isfinite(_fg_ll(x)[0]):
            cand.append(x)
    cand = np.array(cand)
    picks, weights = _bootstrap_picks(r, _fg_ll(cand), n_chain)
    assert cm.n_chains() == n_chain and [cm.chain_weight(i) for i in range(n_chain)] == weights
    cm.run_all_adjust(100, 10, 1.0, 1)                 # = run_all(nSteps, thinning, stepsizefac, nThread), test.cpp:125
    assert np.array_equal(cm.saved_sample(0), cand[picks])
    for i in (0, 7, n_chain - 1):
        ll, smp, sf, nacc = O.fourgauss_run(6.0, 1.0, i + 1, cand[picks[i]], 100, 0, 0, 10)
        ch = cm.get_chain(i)
        assert np.array_equal(ch.get_loglikes(), ll) and ch.weight == weights[i]
    cm.reevaluate_all(s, 0, True, False, 1)            # cell 8
    _, smp, _, _ = O.fourgauss_run(6.0, 1.0, 8, cand[picks[7]], 100, 0, 0, 10)
    assert np.array_equal(cm.get_chain(7).get_samples("pos"), smp)


def test_plain_manager_and_generator_manager():
    # ChainManager(target, nChain): nChain draws of sampleInitialConditions with pcg32(0,0), mcmc.h:1317-1347
    lh, s, t2 = _four_gaussians(1.0)
    cm = mcmc.ChainManager(t2, 30)
    r = O.Rng(0, 0)
    ics = []
    while len(ics) < 30:
        x = [-5 + 10 * r.next_double(), -5 + 10 * r.next_double()]
        if np.isfinite(_fg_ll(x)[0]):
            ics.append(x)
    cm.run_all(200, 10, 20, 1)                         # = run_all_adjust(nSteps, nAdjust, thinning, nThread)
    assert np.array_equal(cm.saved_sample(0), np.array(ics))
    ll, smp, sf, nacc = O.fourgauss_run(6.0, 1.0, 3, ics[2], 200, 0, 10, 20)
    assert np.array_equal(cm.get_chain(2).get_loglikes(), ll)
    ll0, _, sf0, _ = O.fourgauss_run(6.0, 1.0, 1, ics[0], 200, 0, 10, 20)
    assert cm.stepsizes == [sf0]                       # mcmc.h:1550
    # generator chain at T=10 seeds chains at T=1 (test2.ipynb cell 10)
    lh, s, t = _four_gaussians(10)
    gen = mcmc.Chain(t, 0)
    gen.run(4000, 0, 100, 10)
    t2 = mcmc.TempTarget(1.0)
    t2.set_posterior(s)
    cm2 = mcmc.ChainManager(gen, t2, 25)
    pts, old = gen.get_samples("pos"), gen.get_loglikes()
    picks, weights = _bootstrap_picks(O.Rng(0, 0), _fg_ll(pts) - old, 25)
    assert [cm2.chain_weight(i) for i in range(25)] == weights
    ss = cm2.stepsizes[0]
    assert ss == lh.stepsizeCorrectionFac and ss != 1.0
    cm2.run_all_adjust(100, 10, 1 / 5., 1)
    assert np.array_equal(cm2.saved_sample(0), pts[picks])
    ll, smp, _, _ = O.fourgauss_run(6.0, 1.0, 5, pts[picks[4]], 100, 0, 0, 10, stepsize=ss * (1 / 5.))
    assert np.array_equal(cm2.get_chain(4).get_loglikes(), ll)


def test_module_reproduces_the_reference_notebook_reservoir_flow(capfd):
    """notebooks/test2.ipynb cell 8 as saved by the reference's author (tests/golden/
    test2_ipynb_outputs.json + README): ChainManager(TempTarget(1), 100000, 1000) -- a reservoir
    of sampleInitialConditions draws, bootstrap -- then 100 steps; the printed samples of chain 0
    are its states one step after every 10th iteration."""
    import json
    from conftest import GOLDEN
    g = json.load(open(os.path.join(GOLDEN, "test2_ipynb_outputs.json")))
    lh, s, t2 = _four_gaussians(1.0)
    cm = mcmc.ChainManager(t2, 100000, 1000)
    cm.run_all_adjust(100, 1, 1.0, 1)                  # = run_all(nSteps, thinning, stepsizefac, nThread)
    capfd.readouterr()                                 # the reference's progress prints
    got = np.array([cm.saved_sample(10 * k + 1)[0] for k in range(1, 10)])
    assert np.allclose(got, g["cell8"]["chain0_pos"], rtol=0, atol=5.1e-9)


def test_rosenbrock_through_the_module():
    """mcmc.Rosenbrock (extension, BASELINE configs[0]) as the single likelihood of a State: Chain.run equals the
    oracle chain bit for bit."""
    lh = mcmc.Rosenbrock(100.0)
    s = mcmc.State()
    s.add(lh)
    t = mcmc.TempTarget(1.0)
    t.set_posterior(s)
    assert s.loglike() == -1.0 / 20
    c = mcmc.Chain(t, 2)
    c.computeMean = True
    c.run(4000, 100, 50, 7)
    ll, smp, sf, nacc, mean = O.rosenbrock_run(100.0, 1.0, 2, [0.0, 0.0], 4000, 100, 50, 7, with_mean=True)
    assert np.array_equal(c.get_loglikes(), ll) and np.array_equal(c.get_samples("pos"), smp)
    assert np.array_equal(c.get_mean("pos"), mean) and lh.stepsizeCorrectionFac == sf
