"""The experimental cooperative step kernel (MCMC_HIP_COOP=1: two waves per tile, DESIGN.md
section 8) must give, bit for bit, what the default kernel gives -- which the other tests pin
to the oracle.  It is not the default: on MI355X it measured slower."""
import numpy as np
import pytest

from conftest import make_cfg, perturbed
from mcmc_amd import hip as H

pytestmark = pytest.mark.gpu

SMALL = dict(max_delay=12, n_predict=5, fix_behavior_in_advance=5, initial_delay=7.0)


def _run(cfg, R10, xs, coop, monkeypatch):
    monkeypatch.setenv("MCMC_HIP_COOP", "1" if coop else "0")      # read by mcmc_hip_create
    s = H.CovidSampler(cfg, R10)
    s.set_chains(xs)
    s.run(24, 2, 2, 3, record_samples=True, compute_mean=True)
    out = (s.params(), s.loglike(), s.rng_state(), s.accept_counts(), s.stepsizes(), s.loglikes(), s.mean(),
           s.sample_all(s.n_saved - 1))
    s.close()
    return out


@pytest.mark.parametrize("overflow", [False, True])
def test_cooperative_kernel_equals_the_default_kernel(R11, R10, monkeypatch, overflow):
    kw = dict(cap0=3.0, cap_incr_rate=0.5, initial_beta_mild=1.5, initial_beta_high=4.0) if overflow else {}
    cfg = make_cfg(60, R11, **kw, **SMALL)
    xs = perturbed(cfg, 130, 0.1)                       # two full tiles and a ragged one
    a = _run(cfg, R10, xs, False, monkeypatch)
    b = _run(cfg, R10, xs, True, monkeypatch)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
