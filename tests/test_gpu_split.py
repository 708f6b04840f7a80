"""The split step kernel (MCMC_HIP_SPLIT=1: the two passes of a model evaluation in two waves of
one SIMD, mh_step_split_kernel) must give, bit for bit, what the default kernel gives -- which
the other tests pin to the oracle.  It is not the default: on MI355X it measured slower (DESIGN.md
section 8)."""
import numpy as np
import pytest

from conftest import make_cfg, perturbed
from mcmc_amd import hip as H

pytestmark = pytest.mark.gpu

SMALL = dict(max_delay=12, n_predict=5, fix_behavior_in_advance=5, initial_delay=7.0)


def _run(cfg, R10, xs, split, monkeypatch):
    monkeypatch.setenv("MCMC_HIP_SPLIT", "1" if split else "0")     # read by mcmc_hip_create
    monkeypatch.setenv("MCMC_HIP_COOP", "0")
    s = H.CovidSampler(cfg, R10)
    s.set_chains(xs)
    s.run(24, 2, 2, 3, record_samples=True, compute_mean=True)
    out = (s.params(), s.loglike(), s.rng_state(), s.accept_counts(), s.stepsizes(), s.loglikes(), s.mean(),
           s.sample_all(s.n_saved - 1), np.array([s.pair_count()]))
    s.close()
    return out


@pytest.mark.parametrize("case", ["plain", "overflow", "contract", "days200"])
def test_split_kernel_equals_the_default_kernel(R11, R10, monkeypatch, case):
    if case == "overflow":
        cfg = make_cfg(60, R11, cap0=3.0, cap_incr_rate=0.5, initial_beta_mild=1.5, initial_beta_high=4.0, **SMALL)
    elif case == "contract":
        cfg = make_cfg(60, R11, contract=1, **SMALL)
    elif case == "days200":
        cfg = make_cfg(200, R11, contract=1)
    else:
        cfg = make_cfg(60, R11, **SMALL)
    xs = perturbed(cfg, 130, 0.1)                       # two full tiles and a ragged one
    a = _run(cfg, R10, xs, False, monkeypatch)
    b = _run(cfg, R10, xs, True, monkeypatch)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
