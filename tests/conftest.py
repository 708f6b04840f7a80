import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# oracle curve index of each of the 10 curves the C ABI takes (MCMC_HIP_NCURVES order)
ABI_CURVE_ORDER = [3, 4, 0, 2, 5, 7, 6, 8, 9, 10]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def R11():
    """Response table of the covid.ipynb DiseaseParams, 10M trajectories, made by
    tests/golden/make_golden.py with the oracle (44 KB)."""
    return np.load(os.path.join(GOLDEN, "R_ipynb_oracle.npy"))


@pytest.fixture(scope="session")
def R10(R11):
    return np.ascontiguousarray(R11[ABI_CURVE_ORDER])


def make_cfg(n_data, R11, contract=0, **kw):
    """Synthetic workload of SURVEY 8(d) built with the ORACLE's forward model (CPU)."""
    from mcmc_amd import covid_config as CC
    from oracle import oracle as O

    def fwd(cfg, x):
        return O.Model(cfg, R11).eval(x, curves=True)[1][7]

    kw.setdefault("initial_beta_mild", 0.8)
    kw.setdefault("initial_beta_high", 2.5)
    return CC.synthetic_config(n_data, fwd, contract=contract, **kw)


def perturbed(cfg, n, a=0.05, seq=0):
    """State::perturb-style initial conditions from the oracle's pcg32 (CPU only)."""
    from mcmc_amd import covid_config as CC
    from oracle import oracle as O
    x0 = CC.initial_params(cfg)
    r = O.Rng(0, seq)
    out = np.zeros((n, x0.size))
    for c in range(n):
        for k in range(x0.size):
            out[c, k] = x0[k] + (np.float32(r.next_float()) - np.float32(0.5)) * x0[k] * a
    return out
