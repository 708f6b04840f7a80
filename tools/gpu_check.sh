#!/bin/bash
# On the GPU box: parity suite, phase profile (stamps build), bench line without the CPU leg.
set -o pipefail
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | grep -E "passed|failed|rror|assert" | tail -5 || exit 1
MCMC_HIP_LIB=$PWD/mcmc_amd/libmcmc_hip_stamps.so timeout -k 10 200 python tools/phase_profile.py || exit 1
timeout -k 10 240 python bench.py --no-cpu-baseline | cut -c1-140
