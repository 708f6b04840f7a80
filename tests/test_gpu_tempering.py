"""GPU parity of the tempering ladder with swap exchange (EXTENSION, BASELINE configs[3];
the reference has no replica exchange).  The oracle side is the CPU chain of
oracle/mcmc_oracle.c stepped under TempTarget(T_rung) plus the exchange rule of
include/mcmc_hip.h restated here; the GPU must reproduce chains, rungs and swap counts bit
for bit, and sharding whole ladders over two contexts must change nothing."""
import numpy as np
import pytest

from conftest import make_cfg, perturbed
from mcmc_amd import hip as H
from oracle import oracle as O

pytestmark = pytest.mark.gpu

SMALL = dict(max_delay=12, n_predict=5, fix_behavior_in_advance=5, initial_delay=7.0)


class OracleLadders:
    def __init__(self, m, xs, temps, swap_seed, first_chain_id=1, first_ladder=0):
        self.K = len(temps)
        self.n = len(xs)
        self.chains = [O.OracleChain(m, first_chain_id + c, xs[c]) for c in range(self.n)]
        self.targets = [O.make_target("temp", float(T)) for T in temps]
        self.beta = [1 / float(T) for T in temps]                     # TempTarget: bet = 1/T, mcmc.h:797-808
        self.rung = [c % self.K for c in range(self.n)]
        self.at = [[L * self.K + r for r in range(self.K)] for L in range(self.n // self.K)]
        self.rng = [O.Rng(swap_seed, first_ladder + L) for L in range(self.n // self.K)]
        self.accepted = [0] * (self.K - 1)

    def steps(self, n):
        for c, ch in enumerate(self.chains):
            for _ in range(n):
                ch.step(self.targets[self.rung[c]], 0.0, True)

    def swap(self, parity):
        exp = O.lib().orc_det_exp
        for L, at in enumerate(self.at):
            for r in range(parity, self.K - 1, 2):
                a, b = at[r], at[r + 1]
                la = 0.0 + self.chains[a].ll[0] + self.chains[a].ll[1]
                lb = 0.0 + self.chains[b].ll[0] + self.chains[b].ll[1]
                delta = (self.beta[r] - self.beta[r + 1]) * (lb - la)
                u = float(self.rng[L].next_float())
                if u < exp(delta):
                    at[r], at[r + 1] = b, a
                    self.rung[a], self.rung[b] = r + 1, r
                    self.accepted[r] += 1

    def run(self, rounds, steps):
        for k in range(rounds):
            self.steps(steps)
            self.swap(k & 1)


def _check(s, o):
    P, LL, RNG, ACC = s.params(), s.loglike(), s.rng_state(), s.accept_counts()
    for c, ch in enumerate(o.chains):
        assert np.array_equal(ch.x, P[c]), f"chain {c}"
        assert ch.ll == tuple(LL[c]) and ch.rng_state == tuple(int(v) for v in RNG[c]) and ch.c.n_accept == ACC[c]
    assert list(s.rungs()) == o.rung


def test_ladder_matches_the_oracle(R11, R10):
    cfg = make_cfg(40, R11, **SMALL)
    m = O.Model(cfg, R11)
    temps = [1.0, 1.7, 3.0, 8.0]
    xs = perturbed(cfg, 72, 0.05)                       # 18 ladders: one full wave and a ragged one
    s = H.CovidSampler(cfg, R10)
    s.set_chains(xs)
    s.set_ladder(temps, swap_seed=7)
    o = OracleLadders(m, xs, temps, 7)
    assert list(s.rungs()) == o.rung
    s.run_tempered(7, 3)
    o.run(7, 3)
    _check(s, o)
    acc, prop = s.swap_counts()
    assert list(acc) == o.accepted and list(prop) == [18 * 4, 18 * 3, 18 * 4]
    assert 0 < sum(o.accepted) < sum(prop)              # some exchanges happened, not all
    # every ladder still holds each rung exactly once
    assert np.array_equal(np.sort(s.rungs().reshape(18, 4), axis=1), np.tile(np.arange(4), (18, 1)))
    # single pieces of the same machinery: steps and one swap round by hand
    s.step(2, 1.0, True)
    s.swap(1)
    o.steps(2)
    o.swap(1)
    _check(s, o)
    s.close()


def test_whole_ladders_shard_without_exchange(R11, R10):
    """Two contexts with 5 + 4 ladders of 8 rungs = one context with 9 (chain ids and ladder ids
    continue across the shard boundary): no exchange step crosses a device."""
    cfg = make_cfg(40, R11, **SMALL)
    temps = [1.0, 1.3, 1.8, 2.5, 3.5, 5.0, 7.0, 10.0]
    xs = perturbed(cfg, 72, 0.05)
    one = H.CovidSampler(cfg, R10)
    one.set_chains(xs)
    one.set_ladder(temps, swap_seed=3)
    one.run_tempered(4, 2)
    a, b = H.CovidSampler(cfg, R10), H.CovidSampler(cfg, R10)
    a.set_chains(xs[:40], first_chain_id=1)
    a.set_ladder(temps, swap_seed=3, first_ladder=0)
    b.set_chains(xs[40:], first_chain_id=41)
    b.set_ladder(temps, swap_seed=3, first_ladder=5)
    a.run_tempered(4, 2)
    b.run_tempered(4, 2)
    assert np.array_equal(one.params(), np.vstack([a.params(), b.params()]))
    assert np.array_equal(one.rungs(), np.concatenate([a.rungs(), b.rungs()]))
    assert np.array_equal(one.rng_state(), np.vstack([a.rng_state(), b.rng_state()]))
    for s in (one, a, b):
        s.close()


def test_ladder_errors_and_removal(R11, R10):
    cfg = make_cfg(40, R11, **SMALL)
    xs = perturbed(cfg, 6, 0.05)
    s = H.CovidSampler(cfg, R10)
    with pytest.raises(H.McmcHipError):
        s.set_ladder([1.0, 2.0])                        # no chains yet
    s.set_chains(xs)
    with pytest.raises(H.McmcHipError):
        s.swap(0)                                       # no ladder yet
    with pytest.raises(H.McmcHipError):
        s.set_ladder([1.0, 2.0, 3.0, 4.0])              # 6 chains are not a multiple of 4
    with pytest.raises(H.McmcHipError):
        s.set_ladder([1.0, -2.0])
    s.set_ladder([1.0, 2.0, 4.0], swap_seed=1)
    with pytest.raises(H.McmcHipError):
        s.swap(2)
    # removing the ladder (n_temps = 0) gives plain chains again: steps equal a fresh sampler's
    H._check(H.lib().mcmc_hip_set_ladder(s._h, 0, None, 0, 0))
    s.step(3, 1.0, True)
    t = H.CovidSampler(cfg, R10)
    t.set_chains(xs)
    t.step(3, 1.0, True)
    assert np.array_equal(s.params(), t.params()) and np.array_equal(s.loglike(), t.loglike())
    s.close()
    t.close()
