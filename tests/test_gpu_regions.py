"""GPU parity tests of the multi-region model (EXTENSION, BASELINE.json configs[4]; definition in
include/mcmc_hip.h at mcmc_hip_set_regions and in oracle/mcmc_oracle.c): the HIP path through the C ABI
against the oracle's orc_multi_* chains, bit for bit, at R = 1, 3 and 50 regions; R = 1 must be the
single-region path exactly."""
import numpy as np
import pytest

from conftest import make_cfg, perturbed
from mcmc_amd import hip as H
from oracle import oracle as O

pytestmark = pytest.mark.gpu

SMALL = dict(max_delay=12, n_predict=5, fix_behavior_in_advance=5, initial_delay=7.0)


def region_cfgs(R11, R, n_data=40, contract=0, **kw):
    """R regions that differ in population, hospital capacity and (through the forward model at their own
    constants) observations"""
    out = []
    for r in range(R):
        out.append(make_cfg(n_data, R11, contract=contract, pop_size=int(6e7 / (1 + 0.37 * r)), cap0=7500.0 / (1 + r % 5),
                            cap_incr_rate=100.0 - r, **{**SMALL, **kw}))
    return out


def starts(cfgs, n, a=0.05):
    return np.concatenate([perturbed(c, n, a, seq=11 + r) for r, c in enumerate(cfgs)], axis=1)


def test_one_region_is_the_single_region_path(R11, R10):
    cfg = make_cfg(40, R11, contract=1, **SMALL)
    xs = perturbed(cfg, 70, 0.05)
    plain = H.CovidSampler(cfg, R10)
    plain.set_chains(xs, first_chain_id=1)
    multi = H.CovidSampler(cfg, R10)
    multi.set_regions([cfg])
    assert multi.dim == plain.dim
    multi.set_chains(xs, first_chain_id=1)
    for s in (plain, multi):
        s.run(40, 2, 3, 3, record_samples=True, compute_mean=True)
        s.step(5, 0.7, False)
    assert np.array_equal(plain.params(), multi.params()) and np.array_equal(plain.loglike(), multi.loglike())
    assert np.array_equal(plain.rng_state(), multi.rng_state()) and np.array_equal(plain.accept_counts(), multi.accept_counts())
    assert np.array_equal(plain.stepsizes(), multi.stepsizes()) and np.array_equal(plain.loglikes(), multi.loglikes())
    assert np.array_equal(plain.mean(), multi.mean()) and np.array_equal(plain.samples(66), multi.samples(66))
    assert np.array_equal(multi.region_loglike()[:, 0], plain.loglike())
    assert np.array_equal(plain.eval(xs[:5]), multi.eval(xs[:5]))
    assert plain.pair_count() == multi.pair_count()
    plain.close(); multi.close()


@pytest.mark.parametrize("R,contract", [(3, 0), (3, 1), (50, 1)])
def test_region_chains_equal_the_oracle(R11, R10, R, contract):
    cfgs = region_cfgs(R11, R, contract=contract)
    mm = O.MultiModel(cfgs, R11, contract=contract)
    n = 70
    xs = starts(cfgs, n)
    s = H.CovidSampler(cfgs[0], R10)
    s.set_regions(cfgs)
    assert s.dim == R * cfgs[0]["dim"] == mm.dim
    s.set_chains(xs, first_chain_id=1)
    tgt = O.make_target("temp", 1.0)
    # the start: every region evaluated, betas mirrored from region 0
    check = (0, 63, 64, 69) if R < 50 else (0, 64, 69)
    for c in check:
        oc = O.MultiChain(mm, c + 1, xs[c])
        assert np.array_equal(s.region_loglike()[c], oc.ll) and np.array_equal(s.params()[c], oc.x)
    # run(): adjust groups (the head stays for 20 steps, its own step size is corrected), then the scan
    n_samples, n_burn, n_adj, thin = (70, 3, 4, 2) if R < 50 else (130, 0, 2, 5)
    s.run(n_samples, n_burn, n_adj, thin, record_samples=True, compute_mean=True)
    P, LLr, SSr, RNG, ACC, LLS, MEAN = (s.params(), s.region_loglike(), s.region_stepsizes(), s.rng_state(),
                                        s.accept_counts(), s.loglikes(), s.mean())
    assert ACC.sum() > 0
    for c in check:
        oc = O.MultiChain(mm, c + 1, xs[c])
        ll, smp, mean = oc.run(tgt, n_samples, n_burn, n_adj, thin, record=True, mean=True)
        assert np.array_equal(P[c], oc.x), f"chain {c}: parameters differ"
        assert np.array_equal(LLr[c], oc.ll) and np.array_equal(SSr[c], oc.stepsizes)
        assert oc.rng_state == tuple(int(v) for v in RNG[c]) and oc.c.n_accept == ACC[c]
        assert np.array_equal(LLS[c], ll) and np.array_equal(MEAN[c], mean) and np.array_equal(s.samples(c), smp)
        # a few more given-subspace steps under another beta continue the same chain
    s.step(4, 0.5, False)
    s.step(3, 1.0, True)
    P, LLr, RNG = s.params(), s.region_loglike(), s.rng_state()
    t2 = O.make_target("temp", 2.0)
    for c in check[:2]:
        oc = O.MultiChain(mm, c + 1, xs[c])
        oc.run(tgt, n_samples, n_burn, n_adj, thin, record=False)
        for _ in range(4):
            oc.step(t2, 0.0, False)
        for _ in range(3):
            oc.step(tgt, 0.0, True)
        assert np.array_equal(P[c], oc.x) and np.array_equal(LLr[c], oc.ll) and oc.rng_state == tuple(int(v) for v in RNG[c])
    # the betas stay shared, and some chain's betas moved
    dimR, iB = cfgs[0]["dim"], cfgs[0]["nB"] + cfgs[0]["n_disc"]
    blocks = P.reshape(n, R, dimR)
    assert np.all(blocks[:, :, iB] == blocks[:, :1, iB]) and np.all(blocks[:, :, iB + 1] == blocks[:, :1, iB + 1])
    assert np.any(blocks[:, 0, iB] != xs.reshape(n, R, dimR)[:, 0, iB])
    # summed log-likelihoods, batched evaluation of external vectors
    assert np.array_equal(s.loglike(), LLr.sum(axis=1)) or np.allclose(s.loglike(), LLr.sum(axis=1), rtol=1e-15)
    ev = s.eval(xs[:3])
    for c in range(3):
        oc = O.MultiChain(mm, 1, xs[c])
        m = p = 0.0
        for q in range(R):
            m += oc.ll[q, 0]; p += oc.ll[q, 1]
        assert ev[c, 0] == m and ev[c, 1] == p
    s.close()


def test_regions_do_not_depend_on_the_batch_and_resume(R11, R10):
    """a chain's trajectory does not depend on which chains share its wave; set_state plants a multi-region
    chain (per-region log-likelihoods [n][R][2]) in another context"""
    R = 3
    cfgs = region_cfgs(R11, R, contract=1)
    xs = starts(cfgs, 130)
    a = H.CovidSampler(cfgs[0], R10); a.set_regions(cfgs); a.set_chains(xs, first_chain_id=1)
    a.run(20, 0, 1, 4)
    idx = np.array([0, 64, 129])
    b = H.CovidSampler(cfgs[0], R10); b.set_regions(cfgs); b.set_chains(xs[idx], chain_ids=idx + 1)
    b.run(20, 0, 1, 4)
    assert np.array_equal(a.params()[idx], b.params()) and np.array_equal(a.region_loglike()[idx], b.region_loglike())
    assert np.array_equal(a.rng_state()[idx], b.rng_state()) and np.array_equal(a.region_stepsizes()[idx], b.region_stepsizes())
    a.close(); b.close()


def test_regions_refuse_what_they_do_not_support(R11, R10):
    cfgs = region_cfgs(R11, 2, contract=1)
    s = H.CovidSampler(cfgs[0], R10)
    s.set_regions(cfgs)
    xs = starts(cfgs, 4)
    s.set_chains(xs)
    with pytest.raises(H.McmcHipError, match="single-region"):
        s.set_ladder([1.0, 2.0])
    with pytest.raises(H.McmcHipError, match="one region at a time"):
        s.derive(xs[:1])
    with pytest.raises(H.McmcHipError, match="double only"):
        s.set_precision("f32_model")
    with pytest.raises(H.McmcHipError):
        s.set_regions([])
    # set_data brings the single-region State back
    s2 = H.CovidSampler(cfgs[0], R10)
    s2.set_regions(cfgs)
    import ctypes as C
    d = np.ascontiguousarray(cfgs[0]["deaths"]); g = np.ascontiguousarray(cfgs[0]["sigma"])
    assert H.lib().mcmc_hip_set_data(s2._h, d.ctypes.data_as(C.POINTER(C.c_double)), g.ctypes.data_as(C.POINTER(C.c_double))) == 0
    assert H.lib().mcmc_hip_dim(s2._h) == cfgs[0]["dim"] and H.lib().mcmc_hip_n_regions(s2._h) == 1
    s.close(); s2.close()
