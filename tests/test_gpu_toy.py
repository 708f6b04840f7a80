"""GPU parity of the toy targets of test.h (BASELINE configs[0]): MyLike1 under the CLI run
of test.cpp:27-40 and FourGaussians under TempTarget, bit-exact against the oracle."""
import numpy as np
import pytest

from mcmc_amd import hip as H
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_mylike1_cli_config_bitexact():
    """test.cpp:36-38: MyState, chain id 0, run(100000, 10, 500, 5)."""
    r = H.toy_run("mylike1", [[1.0, 1.0, 1.0]], 100000, 10, 500, 5, first_chain_id=0)
    ll, s, sf, nacc, mean = O.mylike1_run(0, 100000, 10, 500, 5, with_mean=True)
    assert np.array_equal(r["loglikes"][0], ll) and np.array_equal(r["samples"][0][:, :3], s)
    assert np.array_equal(r["samples"][0][:, 3], s.max(axis=1))          # the coordinate "max", test.h:37
    assert r["stepsizes"][0] == sf and r["n_accept"][0] == nacc
    assert np.array_equal(r["mean"][0], mean)
    post = r["samples"][0][200:, :3]
    assert np.all(np.abs(post.mean(axis=0)) < 0.05) and np.all(np.abs(post.var(axis=0) - 0.5) < 0.05)


def test_mylike1_many_chains_and_ids():
    n = 300
    r = H.toy_run("mylike1", np.ones((n, 3)), 400, 0, 20, 4, first_chain_id=1)
    for c in (0, 63, 64, 299):
        ll, s, sf, nacc, mean = O.mylike1_run(c + 1, 400, 0, 20, 4, with_mean=True)
        assert np.array_equal(r["loglikes"][c], ll) and np.array_equal(r["samples"][c][:, :3], s)
        assert r["stepsizes"][c] == sf and r["n_accept"][c] == nacc and np.array_equal(r["mean"][c], mean)


@pytest.mark.parametrize("T", [1.0, 10.0])
def test_four_gaussians_tempered_bitexact(T):
    """notebooks/test2.ipynb: FourGaussians(6) under TempTarget(10) as the hot generator chain."""
    x0 = np.array([[0.0, 0.0], [1.0, 0.5], [-4.9, 4.9]])
    r = H.toy_run("fourgauss", x0, 3000, 5, 100, 10, difficulty=6.0, target="temp", T=T, chain_ids=[0, 5, 9])
    for i, cid in enumerate([0, 5, 9]):
        ll, s, sf, nacc, mean = O.fourgauss_run(6.0, T, cid, x0[i], 3000, 5, 100, 10, with_mean=True)
        assert np.array_equal(r["loglikes"][i], ll) and np.array_equal(r["samples"][i], s)
        assert r["stepsizes"][i] == sf and r["n_accept"][i] == nacc and np.array_equal(r["mean"][i], mean)
    assert np.all(np.abs(r["samples"]) <= 5.0)


def test_toy_eval_and_errors():
    rs = np.random.RandomState(3)
    x = rs.uniform(-5, 5, (100, 2))
    got = H.toy_eval("fourgauss", x, difficulty=6.0)
    L = O.lib()
    want = np.array([L.orc_fourgauss_eval(6.0, np.ascontiguousarray(v).ctypes.data_as(O.C.POINTER(O.C.c_double))) for v in x])
    assert np.array_equal(got, want)
    x3 = rs.uniform(-3, 3, (50, 3))
    got = H.toy_eval("mylike1", x3)
    want = np.array([L.orc_mylike1_eval(np.ascontiguousarray(v).ctypes.data_as(O.C.POINTER(O.C.c_double))) for v in x3])
    assert np.array_equal(got, want)
    with pytest.raises(H.McmcHipError):
        H.toy_run("mylike1", np.ones((2, 3)), 10, 0, 0, 0)       # thinning 0
    assert H.lib().mcmc_hip_toy_dim(7) < 0


def test_gpu_chain_reproduces_the_reference_notebook():
    """notebooks/test2.ipynb as saved by the reference's author (tests/golden/
    test2_ipynb_outputs.json + README): the GPU chain, 5.2 million steps on, prints the same
    samples.  Printed sample k of cell 5 is the state one step after saved point 101 + k."""
    import json, os
    from conftest import GOLDEN
    g = json.load(open(os.path.join(GOLDEN, "test2_ipynb_outputs.json")))
    tol = 5.1e-9
    burn = 505000
    r = H.toy_run("fourgauss", [[0.0, 0.0]], 5000000, burn, 10000, 1, difficulty=6.0, target="temp", T=10.0,
                  first_chain_id=0)
    st = r["samples"][0]                                # st[k] = state at iteration burn + k
    first = 500000 + 50 * 101 + 1 - burn
    last = 500000 + 50 * (90200 - 1) + 1 - burn
    head, tail = st[[first, first + 50, first + 100]], st[[last - 100, last - 50, last]]
    c5 = g["cell5"]
    assert np.allclose(head[:, 0], c5["x_head"], rtol=0, atol=tol) and np.allclose(head[:, 1], c5["y_head"], rtol=0, atol=tol)
    assert np.allclose(tail[:, 0], c5["x_tail"], rtol=0, atol=tol) and np.allclose(tail[:, 1], c5["y_tail"], rtol=0, atol=tol)


def test_rosenbrock_chains_equal_the_oracle_and_the_analytic_moments():
    """TOY_ROSENBROCK (extension, BASELINE configs[0]): bit-exact against the oracle's chains for three ids
    under TempTarget(2) with adjust phase, burn-in, thinning and mean; batched loglike bit-exact; and the
    moments of the target itself: x ~ N(1, 10), y | x ~ N(x^2, 10/difficulty), so E[x] = 1, Var[x] = 10,
    E[y] = 11 -- kept by 4096 chains x 40000 steps started from exact draws (Monte-Carlo tolerance)."""
    x0 = np.array([[0.0, 0.0], [1.5, -0.5], [-2.0, 3.0]])
    r = H.toy_run("rosenbrock", x0, 3000, 5, 100, 10, difficulty=100.0, target="temp", T=2.0, chain_ids=[0, 5, 9])
    for i, cid in enumerate([0, 5, 9]):
        ll, s, sf, nacc, mean = O.rosenbrock_run(100.0, 2.0, cid, x0[i], 3000, 5, 100, 10, with_mean=True)
        assert np.array_equal(r["loglikes"][i], ll) and np.array_equal(r["samples"][i], s)
        assert r["stepsizes"][i] == sf and r["n_accept"][i] == nacc and np.array_equal(r["mean"][i], mean)
    x = np.random.default_rng(3).normal(size=(500, 2)) * 3
    L = O.lib()
    want = np.array([L.orc_rosenbrock_eval(100.0, np.ascontiguousarray(v).ctypes.data_as(O.C.POINTER(O.C.c_double))) for v in x])
    assert np.array_equal(H.toy_eval("rosenbrock", x, difficulty=100.0), want)
    assert H.toy_eval("rosenbrock", [[1.0, 1.0]], difficulty=100.0)[0] == 0.0          # the mode
    n = 4096
    rng = np.random.default_rng(11)
    # a random walk explores the curved ridge slowly: start from exact draws of the target and check that
    # the sampler keeps its moments
    xs = np.stack([1 + np.sqrt(10.0) * rng.standard_normal(n), np.zeros(n)], axis=1)
    xs[:, 1] = xs[:, 0] ** 2 + np.sqrt(0.1) * rng.standard_normal(n)
    big = H.toy_run("rosenbrock", xs, 40000, 10000, 500, 100, difficulty=100.0, first_chain_id=1)
    pts = big["samples"].reshape(-1, 2)
    got = (pts[:, 0].mean(), pts[:, 0].var(), pts[:, 1].mean(), (pts[:, 1] - pts[:, 0] ** 2).var())
    assert abs(got[0] - 1.0) < 0.3 and abs(got[1] - 10.0) < 2.0, got
    assert abs(got[2] - 11.0) < 2.0, got
    assert abs(got[3] - 0.1) < 0.01, got                                 # the ridge's width
