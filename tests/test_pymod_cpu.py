"""CPU tests of the pybind11 module `mcmc` (the reference's Python surface for the covid
path): it imports, exposes the reference's names (test.cpp:46-141 + the covid classes of
notebooks/covid.ipynb), parses its inputs like the reference, and fails loudly without a GPU."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "mcmc_amd", "pymod"))
mcmc = pytest.importorskip("mcmc", reason="build it with `make -C mcmc_amd/pymod`")

from mcmc_amd import covid_config as CC, hip as H  # noqa: E402


def _data():
    deaths = CC.ITALY_DEATHS
    d = mcmc.DiseaseData(np.array([deaths, CC.sigma_recipe(deaths)]),
                         np.array([[1, 9, 12, 29], [1, .6, .5, .4], [1, 1, -1, 1]], dtype=float))
    return d


def test_surface_names_match_the_reference():
    for name in ("SubspaceState", "SmoothnessPrior", "State", "Target", "TempTarget", "CoolingTarget", "Chain",
                 "ChainManager", "DiseaseParams", "DiseaseData", "DiseaseSpread", "FourGaussians", "MyState"):
        assert hasattr(mcmc, name)
    for meth in ("run", "get_mean", "get_samples", "get_weights", "get_loglikes", "reevaluate"):
        assert hasattr(mcmc.Chain, meth)
    for meth in ("run_all", "run_all_adjust", "reevaluate_all", "get_chain"):
        assert hasattr(mcmc.ChainManager, meth)
    for meth in ("add", "init", "loglike", "force_bounds", "get_all"):
        assert hasattr(mcmc.State, meth)
    fg = mcmc.FourGaussians(6)                                       # test.h:73-77
    assert fg.get_names() == {"pos": 0} and fg.getCoords("pos") == [0, 0]
    assert mcmc.MyLike1().getCoords("position") == [1, 1, 1]         # test.h:18-22
    p = mcmc.DiseaseParams()
    for k, v in CC.DISEASE_PARAMS_DEFAULT.items():                  # covid.h:75-90
        assert getattr(p, k) == v
    p.timeSeriousToRecovered = 8.0                                   # the notebook's spelling
    assert p.timeSeriousToRec == 8.0


def test_disease_data_and_spread_layout():
    d = _data()
    assert (d.initialBetaMild, d.initialBetaHigh, d.initialDelay, d.fixBehaviorInAdvance) == (2, 7, 14, 14)  # covid.h:52-56
    d.initialDelay = 23
    p = mcmc.DiseaseParams()
    ds = mcmc.DiseaseSpread(d, p, 60000000, 7500.0, 100.0, 40, 60, 20000)
    names = ds.get_names()
    assert names["behavior"] == 0 and names["delay"] == 4 and names["totalBehavior"] == 14 and len(names) == 15
    n_data = len(CC.ITALY_DEATHS)
    assert len(ds.getCoords("behavior")) == n_data - 14
    assert len(ds.getCoords("dead")) == 40 + 60 + n_data             # covid.h:297
    assert ds.getCoords("delay") == [23.0] and ds.getCoords("discontinuousVals") == [1, .6, .5, .4]
    cap = ds.getCoords("capacity")
    assert cap[0] == 7500 and cap[41] == 7600                        # covid.h:314
    # the host precompute is the C ABI's (bit-identical to the oracle's, see test_host_cpu)
    want = H.make_response_tables(CC.DISEASE_PARAMS_DEFAULT, n_traj=20000)
    assert np.array_equal(ds.response_table(), want)
    # force_bounds (covid.h:629-646) runs on the host like the reference's
    st = mcmc.State()
    st.add(ds)
    ds.setCoords([[2.5] * (n_data - 14), [1.2, .6, .5, .4], [-1.0], [7.0], [50.0]] + [ds.getCoords(k) for k in
                 ("mildlyInfectious", "highlyInfectious", "incubating", "asymptomatic", "mild", "serious",
                  "recovered", "dead", "capacity", "totalBehavior")])
    st.force_bounds()
    assert ds.getCoords("behavior")[0] == 1.5 and ds.getCoords("betaMild") == [1.0]
    assert ds.getCoords("delay")[0] == pytest.approx(30.0)           # 50 mirrored at maxDelay 40
    assert ds.getCoords("discontinuousVals")[0] == pytest.approx(0.8)
    assert ds.getCoords("discontinuousVals")[2] == 0.5               # fixed entry untouched


def test_bad_shapes_raise():
    with pytest.raises(Exception):
        mcmc.DiseaseData(np.zeros((3, 5)), np.zeros((3, 2)))
    with pytest.raises(Exception):
        mcmc.DiseaseData(np.zeros((2, 5)), np.zeros((2, 2)))


@pytest.mark.skipif(H.lib().mcmc_hip_device_count() > 0, reason="needs a box WITHOUT a GPU")
def test_no_gpu_means_an_error_not_a_cpu_fallback(capfd):
    d = _data()
    p = mcmc.DiseaseParams()
    ds = mcmc.DiseaseSpread(d, p, 60000000, 7500.0, 100.0, 40, 60, 20000)
    st = mcmc.State()
    st.add(ds)
    st.add(mcmc.SmoothnessPrior("behavior", 50.0, float(len(CC.ITALY_DEATHS) - 14)))
    with pytest.raises(mcmc.McmcHipError, match="no HIP device"):
        st.init()
    t = mcmc.TempTarget(1.0)
    t.set_posterior(st)                       # swallows the failure like the reference (mcmc.h:711-725)
    assert "set_posterior failed" in capfd.readouterr().out
    # an unsupported composition is refused with the reference's message convention
    bad = mcmc.State()
    bad.add(mcmc.SmoothnessPrior("behavior", 50.0, 30.0))
    with pytest.raises(RuntimeError):
        bad.init()
    # the toy states evaluate on the device too: MyState() runs init() (test.h:300-308)
    with pytest.raises(mcmc.McmcHipError, match="no HIP device"):
        mcmc.MyState()
    fg = mcmc.State()
    fg.add(mcmc.FourGaussians(6))
    with pytest.raises(mcmc.McmcHipError, match="no HIP device"):
        fg.init()
