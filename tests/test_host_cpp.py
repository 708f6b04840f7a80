"""The host mirror mcmc_amd/host/mcmc_host.hpp used from plain C++ (tests/cpp/host_mirror_main.cpp
is the CLI run of the reference's test.cpp:27-40): it must compile warning-free on its own,
fail loudly without a GPU, and with one print what the oracle computes."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from mcmc_amd import hip as H

EXE = os.path.join(ROOT, "tests", "cpp", "host_mirror_main")


def _built():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.dirname(EXE)])
    return EXE


@pytest.mark.skipif(H.lib().mcmc_hip_device_count() > 0, reason="needs a box WITHOUT a GPU")
def test_cpp_program_fails_loudly_without_a_gpu():
    r = subprocess.run([_built(), "1000"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "no HIP device" in r.stdout


@pytest.mark.gpu
def test_cpp_program_runs_the_cli_config_on_the_gpu():
    from oracle import oracle as O
    assert os.path.exists(EXE), "build it beforehand: python -c 'import __graft_entry__ as g; g.build()'"
    r = subprocess.run([EXE, "100000"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"saved (\d+) x (\d+), mean (\S+) (\S+) (\S+), last loglike (\S+)", r.stdout)
    ll, smp, sf, nacc, mean = O.mylike1_run(0, 100000, 10, 500, 5, with_mean=True)
    assert (int(m.group(1)), int(m.group(2))) == smp.shape
    assert [float(m.group(k)) for k in (3, 4, 5)] == list(mean[:3]) and float(m.group(6)) == ll[-1]
