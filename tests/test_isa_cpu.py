"""Build-time guard on the generated code of the step kernel's two lag loops (no GPU needed:
hipcc cross-compiles gfx950).  What it protects is measured in DESIGN.md section 5: a full
vmcnt(0) wait or scratch traffic inside these loops costs 5-10 % of the whole benchmark and
comes and goes with unrelated source changes."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_lag_loops_keep_their_prefetch_in_flight(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_guard.py"), str(tmp_path)],
                         check=True, capture_output=True, text=True, timeout=600).stdout
    r = json.loads(out.strip().splitlines()[-1])
    plain, eterm = r["plain_loop"], r["eterm_loop"]
    # two chunks of 8 lag steps per iteration: 56 (88) multiply-adds per step plus <= 7 (12) others
    assert plain["ops"] == 896 and plain["instructions"] <= 1010 and plain["instances"] == 2     # both runs of a block
    assert eterm["ops"] == 1408 and eterm["instructions"] <= 1600
    assert plain["scratch"] == 0 and eterm["scratch"] == 0
    # the history window is refilled two chunks ahead: waits inside the loops must leave at
    # least one chunk's worth of loads (8, resp. 16 with the e-window) outstanding
    assert plain["min_vmcnt"] >= 8 and eterm["min_vmcnt"] >= 16
    # the fast mode's loops (mcmc_model_f32.hip.h): 32 (48) packed FP32 multiply-adds per lag step
    fplain, feterm = r["fast_plain_loop"], r["fast_eterm_loop"]
    assert fplain["ops"] == 512 and fplain["instructions"] <= 620
    assert feterm["ops"] == 768 and feterm["instructions"] <= 950
    assert fplain["scratch"] == 0 and feterm["scratch"] == 0
    assert fplain["sgpr_spill_moves"] == 0 and feterm["sgpr_spill_moves"] == 0
    assert fplain["min_vmcnt"] >= 8 and feterm["min_vmcnt"] >= 16
    # the wave mapping (mcmc_wave_kernel.hip.h): no scratch, no matrix instruction; its plain lag loop takes two groups of
    # 16 lags x 7 curves per trip (224 DPP multiply-adds + 24 LDS reads and their addresses) and waits only for the group
    # it is about to use (lgkmcnt(12): the other group's 12 reads stay in flight)
    wf, ws = r["wave_kernel_fused"], r["wave_kernel_strict"]
    assert wf["scratch"] == 0 and ws["scratch"] == 0 and wf["mfma"] == 0 and ws["mfma"] == 0
    assert wf["dpp_mov"] == 0 and ws["dpp_mov"] > 1000            # the strict form broadcasts with a move of its own
    assert wf["plain_loop"]["ops"] == 224 and wf["plain_loop"]["instructions"] <= 280
    assert wf["plain_loop"]["min_lgkmcnt"] >= 12
