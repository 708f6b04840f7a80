"""CPU tests of the ORACLE: pinned against the published PCG32 vector, the committed
golden fixtures, libm/scipy, and an independent pure-numpy restatement of the model."""
import hashlib
import json
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN, make_cfg, perturbed
from mcmc_amd import covid_config as CC
from oracle import oracle as O


def test_pcg32_published_known_answer():
    kat = json.load(open(os.path.join(GOLDEN, "pcg32_kat.json")))
    r = O.Rng(42, 54)
    assert [r.next_uint() for _ in range(6)] == kat["published_pcg32_42_54"]


def test_pcg32_golden_streams():
    kat = json.load(open(os.path.join(GOLDEN, "pcg32_kat.json")))["oracle_streams"]
    for key, v in kat.items():
        if key == "default":
            r = O.Rng()
            assert [r.next_uint() for _ in range(16)] == v["uint"]
            continue
        st, sq = (int(t) for t in key.split(","))
        r = O.Rng(st, sq)
        assert [r.next_uint() for _ in range(16)] == v["uint"]
        r = O.Rng(st, sq)
        assert [float(r.next_float()).hex() for _ in range(16)] == v["float_hex"]
        r = O.Rng(st, sq)
        assert [r.next_double().hex() for _ in range(16)] == v["double_hex"]


def test_pcg32_float_double_are_exact_scalings_of_the_uint():
    a, b, c = O.Rng(0, 3), O.Rng(0, 3), O.Rng(0, 3)
    for _ in range(200):
        u = a.next_uint()
        assert b.next_float() == (u >> 9) * 2.0 ** -23       # (u>>9)|0x3f800000, minus 1
        assert c.next_double() == u * 2.0 ** -32              # (u<<20)|0x3ff0..., minus 1


def test_bound_reflects_into_range():
    L = O.lib()
    rs = np.random.RandomState(0)
    for _ in range(2000):
        lo = rs.uniform(-3, 3); hi = lo + rs.uniform(0.1, 5); v = rs.uniform(-20, 20)
        b = L.orc_bound(v, lo, hi)
        assert lo - 1e-12 <= b <= hi + 1e-12
        if lo <= v <= hi:
            assert abs(b - v) < 1e-12 * max(1, abs(v))
    assert abs(L.orc_bound(2.25, 0.0, 2.0) - 1.75) < 1e-15      # mirror at the upper bound
    assert abs(L.orc_bound(-0.25, 0.0, 2.0) - 0.25) < 1e-15     # mirror at the lower bound
    assert math.isnan(L.orc_bound(1.0, 1.0, 1.0))               # zero-width interval: 0/0 as in mcmc.h:50


def _ulps(a, b):
    ia = np.frombuffer(np.float64(a).tobytes(), dtype=np.int64)[0]
    ib = np.frombuffer(np.float64(b).tobytes(), dtype=np.int64)[0]
    return abs(int(ia) - int(ib))


def test_detmath_within_one_ulp_of_libm():
    L = O.lib()
    rs = np.random.RandomState(1)
    for x in np.concatenate([rs.uniform(-2, 40, 20000), rs.uniform(0, 1e5, 5000), [0.0, np.pi / 4, np.pi / 2, 1e-9]]):
        assert _ulps(L.orc_det_cos(x), math.cos(x)) <= 1
    for x in np.concatenate([rs.uniform(-745, 5, 20000), [0.0, -1e-10, 1e-10, -0.3, 0.3, -744.5, 709.0]]):
        assert _ulps(L.orc_det_exp(x), math.exp(x)) <= 1
    for x in np.concatenate([rs.uniform(1e-300, 10, 20000), 10.0 ** rs.uniform(-320, 300, 5000), 1 + rs.uniform(-1e-6, 1e-6, 2000),
                             [1.0, 0.5, 2.0, 5e-324, 1.7976931348623157e308]]):
        assert _ulps(L.orc_det_log(x), math.log(x)) <= 1
    assert L.orc_det_log(0.0) == -math.inf and math.isnan(L.orc_det_log(-1.0)) and L.orc_det_log(math.inf) == math.inf
    assert L.orc_det_exp(-1e4) == 0.0 and L.orc_det_exp(1e4) == math.inf
    assert math.isnan(L.orc_det_exp(math.nan)) and math.isnan(L.orc_det_cos(math.inf))


def test_erfinv_is_giles_single_precision_quality():
    from scipy.special import erfinv
    L = O.lib()
    for x in np.linspace(-0.999, 0.999, 401):
        assert abs(L.orc_erfinv(x) - erfinv(x)) <= 2e-6 * max(1.0, abs(erfinv(x)))


# ---------------------------------------------------------------------------------------
def numpy_eval(cfg, R11, x):
    """Independent restatement of DiseaseSpread::eval (covid.h:325-496) in plain Python,
    written from the reference text, scatter form, no FMA.  Slow: small cases only."""
    nd, md, npred = cfg["n_data"], cfg["max_delay"], cfg["n_predict"]
    size, nB, nDisc = md + npred + nd, cfg["nB"], cfg["n_disc"]
    beh, disc = x[:nB], x[nB:nB + nDisc]
    bm, bh, delay = x[nB + nDisc:]
    Rm = {n: R11[i] for i, n in enumerate(O.CURVE_NAMES)}
    out = np.zeros((9, size))
    ll = 0.0
    start = int(md - delay)
    frac = md - delay - start
    for shift in (1, 0):
        w = (1 - shift) * (1 - frac) + shift * frac
        buf = {k: np.zeros(size) for k in ("mi", "hi", "inc", "asy", "mild", "ser", "rec", "dead")}
        tb = np.ones(size)
        n = 1.0
        src = [("mi", "infectiousMild"), ("hi", "infectiousHigh"), ("inc", "incubating"), ("asy", "mild"),
               ("mild", "mild"), ("ser", "serious"), ("rec", "recovered"), ("dead", "dead")]
        for i in range(start + shift, size):
            for b, t in src:
                for d in range(min(500, size - i)):
                    buf[b][i + d] += n * Rm[t][d]
            cap = cfg["cap0"] + max(cfg["cap_incr_rate"] * (i - md), 0.0)
            e = cfg["prob_lethal_daily_untreated"] * (buf["ser"][i] * cfg["prob_icu_if_serious"] - cap)
            if e > 0:
                for d in range(min(500, size - i)):
                    buf["dead"][i + d] += e
                for b, t in (("ser", "seriousFromSerious"), ("rec", "recoveredFromSerious"), ("dead", "deadFromSerious")):
                    for d in range(min(500, size - i)):
                        buf[b][i + d] += -e * Rm[t][d]
            found = -1
            days = cfg["disc_days"]
            for m in range(nDisc - 1):
                if days[m] <= i - md < days[m + 1]:
                    found = m
                    break
            if found == -1 and i - md >= days[-1]:
                found = nDisc - 1
            pcf = disc[found] if found != -1 else 1.0
            sb = 1.0
            if i >= md:
                sb = pcf * (beh[i - md] if i - md < nB else beh[-1])
            tb[i] = pcf * sb
            nS = cfg["pop_size"] - buf["inc"][i] - buf["asy"][i] - buf["mild"][i] - buf["ser"][i] - buf["rec"][i] - buf["dead"][i]
            n = tb[i] * nS / cfg["pop_size"] * (bm * buf["mi"][i] + bh * buf["hi"][i])
        for c, k in enumerate(("mi", "hi", "inc", "asy", "mild", "ser", "rec", "dead")):
            out[c] = w * buf[k] if shift == 1 else out[c] + w * buf[k]
        out[8] = tb
        for i in range(nd):
            delta = out[7][i + md] - cfg["deaths"][i]
            ll += w * (-0.5 * delta * delta / (cfg["sigma"][i] * cfg["sigma"][i]))
    return ll, out


@pytest.mark.parametrize("force_overflow", [False, True])
def test_eval_matches_independent_numpy_restatement(R11, force_overflow):
    kw = dict(max_delay=12, n_predict=7, fix_behavior_in_advance=5, initial_delay=6.3,
              disc_days=(1, 4, 9), disc_vals=(1.0, 0.7, 0.5))
    if force_overflow:
        kw.update(cap0=3.0, cap_incr_rate=0.5, initial_beta_mild=1.5, initial_beta_high=4.0)
    cfg = make_cfg(40, R11, **kw)
    m = O.Model(cfg, R11)
    for x in perturbed(cfg, 3, 0.1):
        ll, curves = m.eval(x, curves=True)
        ll2, curves2 = numpy_eval(cfg, R11, x)
        assert ll == ll2
        assert np.array_equal(curves, curves2)
        if force_overflow:
            assert curves[7].max() > 0          # the capacity branch really ran
    if force_overflow:
        # make sure extraDeaths > 0 happened: dead exceeds the no-overflow run
        cfg0 = make_cfg(40, R11, **{**kw, "cap0": 1e12})
        x = CC.initial_params(cfg)
        assert O.Model(cfg, R11).eval(x, curves=True)[1][7][-1] != O.Model({**cfg0, "deaths": cfg["deaths"], "sigma": cfg["sigma"]}, R11).eval(x, curves=True)[1][7][-1]


def test_loglike_does_not_see_the_prediction_tail(R11):
    """The HIP sampling kernels stop at day maxDelay+nData; the oracle shows that this is
    bit-identical for loglike (covid.h:488-492 never reads later days)."""
    for nd in (60, 200):
        cfg = make_cfg(nd, R11)
        m = O.Model(cfg, R11)
        for x in perturbed(cfg, 4, 0.05):
            assert m.eval(x) == m.eval(x, sim_days=cfg["max_delay"] + nd)


def test_faithful_getday_path_equals_response_table_path():
    p = O.DiseaseParams(**CC.ipynb_disease_params())
    grid = O.traj_generate(p, 200_000)
    R = O.traj_response(grid)
    cfg = make_cfg(50, R)
    lean, faithful = O.Model(cfg, R), O.Model(cfg, R, grid=grid, faithful=1)
    for x in perturbed(cfg, 3, 0.05):
        assert lean.eval(x) == faithful.eval(x)
    # deltas really are differences of the cumulative grid (covid.h:221-235)
    assert R[0, 0] == grid[0, 0] and abs(R[6, 499] - grid[6, 99800]) < 1e-12


def test_contract_mode_is_a_rounding_level_change(R11):
    cfg = make_cfg(100, R11)
    strict, fused = O.Model(cfg, R11, contract=0), O.Model(cfg, R11, contract=1)
    x = perturbed(cfg, 1, 0.05)[0]
    (a, ca), (b, cb) = strict.eval(x, curves=True), fused.eval(x, curves=True)
    assert not np.array_equal(ca, cb)                          # single vs double rounding shows up
    assert np.allclose(ca, cb, rtol=1e-9, atol=1e-9) and abs(a - b) <= 1e-9 * abs(a)


def test_smoothness_prior(R11):
    cfg = make_cfg(60, R11)
    m = O.Model(cfg, R11)
    x = CC.initial_params(cfg)
    assert m.prior(x) == 0.0                                   # behavior == 1 has no curvature
    x[3] += 0.1
    n, L, Ls = cfg["nB"], cfg["smooth_L"], cfg["smooth_Ls"]
    dx = L / n
    want = -(0.1 ** 2 + 0.2 ** 2 + 0.1 ** 2) / dx ** 3 * Ls * Ls / L
    assert abs(m.prior(x) - want) < 1e-9 * abs(want)


def test_golden_traces_regression(R11):
    gold = json.load(open(os.path.join(GOLDEN, "covid200_traces.json")))
    for contract in (0, 1):
        g = gold[f"contract{contract}"]
        cfg = make_cfg(200, R11, contract=contract)
        assert [float(v).hex() for v in cfg["deaths"]] == g["deaths_hex"]
        m = O.Model(cfg, R11, contract=contract)
        xs = perturbed(cfg, 4, 0.05)
        tgt = O.make_target("temp", 1.0)
        for c in range(4):
            tr = g["chains"][c]
            ch = O.OracleChain(m, c + 1, xs[c])
            assert [ch.ll[0].hex(), ch.ll[1].hex()] == tr["ll0"]
            for k in range(40):
                assert ch.step(tgt, 0.0, True) == tr["accept"][k]
                assert float(ch.c.last_delta).hex() == tr["delta_hex"][k]
                assert [int(ch.rng_state[0]), int(ch.rng_state[1])] == tr["rng"][k]
            assert [ch.ll[0].hex(), ch.ll[1].hex()] == tr["ll_end"]
            assert hashlib.sha256(ch.x.tobytes()).hexdigest() == tr["x_end_sha"]


def test_run_equals_manual_step_sequence(R11):
    """MetropolisChain::run (mcmc.h:1005-1138): 20 steps per adjust iteration, the first
    with a random sub-space, step size corrected after each group, records before stepping."""
    cfg = make_cfg(40, R11, max_delay=12, n_predict=5, fix_behavior_in_advance=5, initial_delay=7.0)
    m = O.Model(cfg, R11)
    x0 = perturbed(cfg, 1, 0.05)[0]
    tgt = O.make_target("temp", 2.0)
    a = O.OracleChain(m, 3, x0)
    ll, samples, mean = a.run(tgt, nSamples=30, nBurnin=2, nAdjust=4, thinning=3, mean=True)
    assert ll.shape[0] == O.lib().orc_n_saved(30, 2, 4, 3) == len([i for i in range(34) if i >= 2 and i % 3 == 0])
    b = O.OracleChain(m, 3, x0)
    rec, mu = [], x0.copy()
    for i in range(34):
        if i == 4:
            b.c.n_accept = 0
        if i >= 2 and i % 3 == 0:
            rec.append(0.5 * (b.ll[0] + b.ll[1]))
        if i > 2:
            mu += b.x
        if i < 4:
            b.c.n_accept = 0
            b.step(tgt, i / 4, True)
            for _ in range(19):
                b.step(tgt, i / 4, False)
            r = b.c.n_accept / 20
            b.c.stepsize *= (1 + 0.726484 * r * r * r * r) / (0.82051 + 0.0427315 / (r + 0.0001))
        else:
            b.step(tgt, (i - 4) / 30, True)
    assert np.array_equal(a.x, b.x) and a.rng_state == b.rng_state
    assert a.c.stepsize == b.c.stepsize and a.c.n_accept == b.c.n_accept
    assert np.array_equal(ll, np.array(rec))
    assert np.allclose(mean, mu / (4 + 30 - 2), rtol=0, atol=1e-12)
    assert np.array_equal(samples[-1], samples[-1]) and samples.shape == (ll.shape[0], m.dim)


def test_run_chains_pool_matches_single_chain_runs(R11):
    cfg = make_cfg(40, R11, max_delay=12, n_predict=5, fix_behavior_in_advance=5, initial_delay=7.0)
    m = O.Model(cfg, R11)
    xs = perturbed(cfg, 6, 0.05)
    tgt = O.make_target("temp", 1.0)
    res = O.run_chains(m, tgt, xs, 1.0, 15, 0, 2, 5, n_threads=3, want_loglikes=True)
    for i in range(6):
        ch = O.OracleChain(m, i + 1, xs[i])                       # ids i+1, mcmc.h:1527
        ll, _, _ = ch.run(tgt, 15, 0, 2, 5, record=False)
        assert np.array_equal(ch.x, res["x"][i]) and np.array_equal(ll, res["loglikes"][i])
        assert ch.c.stepsize == res["stepsizes"][i] and ch.c.n_accept == res["n_accept"][i]


def test_toy_mylike1_cli_config_statistics():
    """test.cpp:27-40 runs MyLike1 with chain id 0; the target is the product of two unit
    Gaussians at (+-2,+-2,0): N(0, 1/2 I) in the position (test.h:14-67)."""
    ll, s, sf, nacc = O.mylike1_run(0, 100000, 10, 500, 5)
    assert ll.shape[0] == O.lib().orc_n_saved(100000, 10, 500, 5)
    post = s[200:]
    assert np.all(np.abs(post.mean(axis=0)) < 0.05)
    assert np.all(np.abs(post.var(axis=0) - 0.5) < 0.05)
    assert 0.1 < nacc / 100000 < 0.5 and 0 < sf < 1
    # deterministic given the seed
    ll2, s2, sf2, nacc2 = O.mylike1_run(0, 100000, 10, 500, 5)
    assert np.array_equal(s, s2) and sf == sf2 and nacc == nacc2


def test_toy_four_gaussians_tempered():
    ll, s, sf, nacc = O.fourgauss_run(6.0, 10.0, 0, np.zeros(2), 20000, 0, 200, 10)
    assert np.all(np.isfinite(ll)) and np.all(np.abs(s) <= 5.0) and nacc > 0


def test_reference_notebook_outputs():
    """The outputs the reference's author recorded in notebooks/test2.ipynb (tests/golden/
    test2_ipynb_outputs.json, README_test2_ipynb.md): 5.2 million MH steps of FourGaussians(6)
    under TempTarget(10), and chain 0 of a 100 000-candidate reservoir ChainManager."""
    import json
    g = json.load(open(os.path.join(GOLDEN, "test2_ipynb_outputs.json")))
    tol = 5.1e-9                                           # printed with 8 decimals
    # cell 5: printed sample k = state one step after saved point 101 + k (see the README)
    _, st, _, _ = O.fourgauss_run(6.0, 10.0, 0, [0.0, 0.0], 5000000, 0, 10000, 1)   # every iteration
    first = 500000 + 50 * 101 + 1
    last = 500000 + 50 * (90200 - 1) + 1
    head = st[[first, first + 50, first + 100]]
    tail = st[[last - 100, last - 50, last]]
    c5 = g["cell5"]
    assert np.allclose(head[:, 0], c5["x_head"], rtol=0, atol=tol) and np.allclose(head[:, 1], c5["y_head"], rtol=0, atol=tol)
    assert np.allclose(tail[:, 0], c5["x_tail"], rtol=0, atol=tol) and np.allclose(tail[:, 1], c5["y_tail"], rtol=0, atol=tol)
    # cell 8: reservoir of 100 000 sampleInitialConditions draws from pcg32(0,0) (test.h:116-123,
    # mcmc.h:1348-1374), bootstrap (mcmc.h:1599-1650), chain 0 has id 1 (mcmc.h:1487)
    L = O.lib()
    r = O.Rng(0, 0)
    cand, lls = [], []
    while len(cand) < 100000:
        x = np.array([-5 + 10 * r.next_double(), -5 + 10 * r.next_double()])
        l = L.orc_fourgauss_eval(6.0, x.ctypes.data_as(O.C.POINTER(O.C.c_double)))
        if np.isfinite(l):
            cand.append(x)
            lls.append(l)
    lls = np.array(lls)
    w = np.exp(lls - lls.max())
    cdf = np.cumsum(w)
    total = w.sum()
    cdf[-1] = 1.000001 * total
    pick0 = int(np.searchsorted(cdf, r.next_double() * total, side="right"))
    _, st, _, _ = O.fourgauss_run(6.0, 1.0, 1, cand[pick0], 100, 0, 0, 1)
    assert np.allclose(st[[10 * k + 1 for k in range(1, 10)]], g["cell8"]["chain0_pos"], rtol=0, atol=tol)


# ---------------------------------------------------------------------------------------
# An independent restatement of DiseaseSpread::step up to its eval call (covid.h:499-622),
# written from the reference text in plain Python: own pcg32 on Python integers, float32 gates
# through numpy.float32, C's round() spelled out, libm cos.  Checked draw for draw against
# orc_covid_propose (run with use_libm=1, since the reference calls std::cos).
class PyPcg32:
    MULT = 0x5851f42d4c957f2d
    M64 = (1 << 64) - 1

    def __init__(self, initstate, initseq):            # pcg32::seed
        self.state, self.inc = 0, ((initseq << 1) | 1) & self.M64
        self.next_uint()
        self.state = (self.state + initstate) & self.M64
        self.next_uint()

    def next_uint(self):
        old = self.state
        self.state = (old * self.MULT + self.inc) & self.M64
        xorshifted = (((old >> 18) ^ old) >> 27) & 0xffffffff
        rot = old >> 59
        return ((xorshifted >> rot) | (xorshifted << ((-rot) & 31))) & 0xffffffff

    def next_float(self):                               # [0,1) float: mantissa trick, then - 1.0f
        bits = np.array([(self.next_uint() >> 9) | 0x3f800000], dtype=np.uint32)
        return np.float32(bits.view(np.float32)[0] - np.float32(1.0))

    def next_double(self):
        bits = np.array([(self.next_uint() << 20) | 0x3ff0000000000000], dtype=np.uint64)
        return float(bits.view(np.float64)[0]) - 1.0


def c_round(v):
    """std::round: to nearest, halves away from zero (Python's round() goes to even)."""
    a = abs(v)
    f = math.floor(a)
    return math.copysign(f + 1.0 if a - f >= 0.5 else f, v)


def py_bound(val, lower, upper):                        # mcmc.h:49-53
    val = 0.5 * (val - lower) / (upper - lower)
    val = 2.0 * abs(val - c_round(val))
    return lower + val * (upper - lower)


def python_step_proposal(cfg, rnd, x, stepsize_fac):
    """covid.h:499-622 on the coords vector [behavior | discontinuousVals | betaMild betaHigh delay]."""
    nB, nDisc = cfg["nB"], cfg["n_disc"]
    new = [float(v) for v in x]                         # newstate = copy()
    i_bm, i_bh, i_delay = nB + nDisc, nB + nDisc + 1, nB + nDisc + 2
    fixed = [bool(f) for f in cfg["disc_fixed"]]
    sf = min(stepsize_fac, 1.0)
    big = [rnd.next_float() < np.float32(0.6) for _ in range(5)]           # 503-507
    if rnd.next_float() < np.float32(0.3):                                  # start time, 510-515
        new[i_delay] += (rnd.next_double() - 0.5) * 0.5 * 0.1 * sf
        if big[0]:
            new[i_delay] += (rnd.next_double() - 0.5) * sf
        new[i_delay] = py_bound(new[i_delay], 5.0, float(cfg["max_delay"]))
    if rnd.next_float() < np.float32(0.3):                                  # betas, 518-532
        if not big[1]:
            new[i_bm] += (rnd.next_double() - 0.5) * 0.1 * sf
        else:
            new[i_bm] += (rnd.next_double() - 0.5) * sf
        new[i_bm] = py_bound(new[i_bm], 0.0, 100.0)
        if not big[2]:
            new[i_bh] += (rnd.next_double() - 0.5) * 0.1 * sf
        else:
            new[i_bh] += (rnd.next_double() - 0.5) * sf
        new[i_bh] = py_bound(new[i_bh], 0.0, 100.0)
        if new[i_bh] < new[i_bm]:
            new[i_bh], new[i_bm] = new[i_bm], new[i_bh]
    if rnd.next_float() < np.float32(0.9):                                  # breakpoint values, 557-598
        n_not_fixed = np.float32(0)
        for i in range(nDisc):
            if fixed[i]:                                                    # (sic) counts the fixed ones
                n_not_fixed += np.float32(1)
        order = list(range(nDisc))
        if rnd.next_double() < 0.5:
            order.reverse()                                                 # start = n-1, stop = -1, incr = -1
        with np.errstate(divide="ignore"):
            thr = np.float32(2) / n_not_fixed                               # int 2 / float -> float; inf if zero
        for i in order:
            if fixed[i]:
                continue
            xx = rnd.next_float()
            if xx < thr:
                if not big[3]:
                    new[nB + i] += (rnd.next_double() - 0.5) * 0.1 * sf
                else:
                    new[nB + i] += (rnd.next_double() - 0.5) * sf
                lower, upper = 0.0, 1.0
                if i >= 1:
                    upper = new[nB + i - 1]
                if i < nDisc - 1:
                    lower = new[nB + i + 1]
                new[nB + i] = py_bound(new[nB + i], lower, upper)
    if rnd.next_float() < np.float32(0.3):                                  # behaviour function, 601-622
        T = float(cfg["n_data"])
        xx = rnd.next_double()
        omega = 2 * 3.1415 / T * 5 * xx * xx
        A = min(1.0, stepsize_fac) * 0.1 * (rnd.next_double() - 0.5)
        if big[4]:
            A += min(1.0, stepsize_fac) * (rnd.next_double() - 0.5)
        for i in range(nB):
            new[i] += A * (math.cos(omega * i) - 1)
            new[i] = py_bound(new[i], 0.0, 2.0)
    return np.array(new)


@pytest.mark.parametrize("variant", ["ipynb", "some_fixed", "all_fixed", "eight_breakpoints", "one_breakpoint"])
def test_propose_matches_independent_python_restatement(R11, variant):
    """orc_covid_propose against python_step_proposal, draw for draw: same proposal vector bit for
    bit and the same generator state after every one of 400 proposals per chain (proposals are
    chained: each starts from the previous proposal, as if every step were accepted), for step
    size factors below and above 1."""
    kw = dict(max_delay=12, n_predict=5, fix_behavior_in_advance=5, initial_delay=7.0)
    if variant == "some_fixed":
        kw.update(disc_fixed=[1, 0, 0, 1])
    elif variant == "all_fixed":
        kw.update(disc_fixed=[1, 1, 1, 1])
    elif variant == "eight_breakpoints":
        kw.update(disc_days=(1, 3, 5, 9, 12, 20, 25, 29), disc_vals=(1.0, 0.9, 0.8, 0.7, 0.6, 0.5, 0.45, 0.4),
                  disc_fixed=[0, 0, 1, 0, 0, 0, 0, 0])
    elif variant == "one_breakpoint":
        kw.update(disc_days=(4,), disc_vals=(0.7,))
    cfg = make_cfg(40, R11, **kw)
    m = O.Model(cfg, R11, use_libm=1)
    for chain_id, sfac in ((1, 1.0), (2, 0.37), (3, 2.5)):
        x = perturbed(cfg, 1, 0.05, seq=chain_id)[0]
        orng, prng = O.Rng(0, chain_id), PyPcg32(0, chain_id)
        n_changed = 0
        for it in range(400):
            want = python_step_proposal(cfg, prng, x, sfac)
            got = m.propose(orng, x, sfac)
            assert np.array_equal(got, want), (variant, chain_id, it, np.nonzero(got != want))
            assert orng.state == (prng.state, prng.inc), (variant, chain_id, it)
            n_changed += int(not np.array_equal(got, x))
            x = got
        assert n_changed > 200


def test_python_pcg32_reproduces_the_published_vector():
    """the restatement's own generator against the PCG32 known-answer vector pcg32(42, 54)"""
    r = PyPcg32(42, 54)
    assert [r.next_uint() for _ in range(6)] == [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e]


def test_faithful_copy_mode_runs_the_same_chain(R11):
    """faithful=1 (a fresh copy of the trajectory arrays per proposal, running sums re-formed from
    them: covid.h:290,501,376-386) only costs time: the chain is bit-identical to the lean one."""
    p = O.DiseaseParams(**CC.ipynb_disease_params())
    grid = O.traj_generate(p, 100_000)
    R = O.traj_response(grid)
    cfg = make_cfg(40, R, max_delay=12, n_predict=5, fix_behavior_in_advance=5, initial_delay=7.0)
    lean, faithful = O.Model(cfg, R), O.Model(cfg, R, grid=grid, faithful=1)
    tgt = O.make_target("temp", 1.0)
    x0 = perturbed(cfg, 1, 0.05)[0]
    a, b = O.OracleChain(lean, 1, x0), O.OracleChain(faithful, 1, x0)
    for _ in range(25):
        assert a.step(tgt) == b.step(tgt)
    assert np.array_equal(a.x, b.x) and a.ll == b.ll and a.rng_state == b.rng_state
    ra = O.run_chains(lean, tgt, perturbed(cfg, 3, 0.05), 1.0, 10, 0, 1, 1, n_threads=2)
    rb = O.run_chains(faithful, tgt, perturbed(cfg, 3, 0.05), 1.0, 10, 0, 1, 1, n_threads=2)
    assert np.array_equal(ra["x"], rb["x"]) and np.array_equal(ra["rng"], rb["rng"])


def test_toy_rosenbrock_oracle_moments():
    """the oracle's Rosenbrock chains (extension, BASELINE configs[0]) against the target's own moments:
    x ~ N(1, 10), y | x ~ N(x^2, 0.1).  A random walk explores the curved ridge slowly, so the chains start
    from exact draws of the target and the check is that the sampler keeps its moments (48 chains: loose
    Monte-Carlo tolerances); runs are deterministic per chain id."""
    rng = np.random.default_rng(7)
    pts = []
    for cid in range(48):
        x = 1 + np.sqrt(10.0) * rng.standard_normal()
        start = [x, x * x + np.sqrt(0.1) * rng.standard_normal()]
        ll, s, sf, nacc = O.rosenbrock_run(100.0, 1.0, cid, start, 30000, 0, 300, 50)
        assert np.all(np.isfinite(ll)) and nacc > 1000
        pts.append(s)
    a = O.rosenbrock_run(100.0, 1.0, 3, [1.0, 1.0], 2000, 0, 10, 5)
    b = O.rosenbrock_run(100.0, 1.0, 3, [1.0, 1.0], 2000, 0, 10, 5)
    assert np.array_equal(a[1], b[1])
    pts = np.concatenate(pts)
    assert abs(pts[:, 0].mean() - 1.0) < 1.2 and 5.0 < pts[:, 0].var() < 15.0
    assert abs((pts[:, 1] - pts[:, 0] ** 2).var() - 0.1) < 0.02          # the ridge's width: y - x^2 ~ N(0, 0.1)
    x = np.array([0.3, -0.7])
    assert O.lib().orc_rosenbrock_eval(100.0, x.ctypes.data_as(O.C.POINTER(O.C.c_double))) == \
        -((1 - 0.3) * (1 - 0.3) + 100.0 * ((-0.7 - 0.3 * 0.3) * (-0.7 - 0.3 * 0.3))) / 20


def test_multi_region_oracle_reduces_to_the_single_region_chain(R11):
    """orc_multi_* (extension, BASELINE configs[4]) with one region IS orc_chain_*; with three regions the
    betas stay shared, every region's log-likelihood is the stand-alone model's at its block, and the scan
    visits the regions in turn."""
    cfg = make_cfg(40, R11, max_delay=12, n_predict=5, fix_behavior_in_advance=5, initial_delay=7.0)
    x0 = perturbed(cfg, 1, 0.05)[0]
    tgt = O.make_target("temp", 1.0)
    a = O.MultiChain(O.MultiModel([cfg], R11), 3, x0)
    b = O.OracleChain(O.Model(cfg, R11), 3, x0)
    ra, rb = a.run(tgt, 30, 2, 3, 2, mean=True), b.run(tgt, 30, 2, 3, 2, mean=True)
    assert all(np.array_equal(u, v) for u, v in zip(ra, rb))
    assert a.rng_state == b.rng_state and np.array_equal(a.x, b.x) and a.stepsizes[0] == b.c.stepsize
    cfgs = [cfg, dict(cfg, pop_size=30000000, cap0=3000.0), dict(cfg, deaths=cfg["deaths"] * 1.3, sigma=cfg["sigma"] * 1.1)]
    mm = O.MultiModel(cfgs, R11)
    x3 = np.concatenate([perturbed(cfg, 1, 0.05, seq=k)[0] for k in range(3)])
    c = O.MultiChain(mm, 5, x3)
    dimR, iB = cfg["dim"], cfg["nB"] + cfg["n_disc"]
    heads = []
    for _ in range(90):
        c.step(tgt)
        heads.append(c.c.head)
        blocks = c.x.reshape(3, dimR)
        assert np.all(blocks[:, iB] == blocks[0, iB]) and np.all(blocks[:, iB + 1] == blocks[0, iB + 1])
        for q in range(3):
            assert c.ll[q, 0] == mm.models[q].eval(blocks[q]) and c.ll[q, 1] == mm.models[q].prior(blocks[q])
    assert heads[:7] == [0, 1, 2, 0, 1, 2, 0] and c.c.n_accept > 10
    assert c.x.reshape(3, dimR)[0, iB] != x3[iB]
