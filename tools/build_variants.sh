#!/bin/bash
# Builds the product library and the -DMCMC_STAMPS variant used by tools/phase_profile.py.
set -e
cd "$(dirname "$0")/../mcmc_amd/csrc"
make 2>&1 | grep -E "error|warning" || true
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -I. -DMCMC_T=${T:-8} -DMCMC_STAMPS -mllvm -amdgpu-mfma-vgpr-form $EXTRA -Wno-unused-function -shared -o ../libmcmc_hip_stamps.so mcmc_hip.hip 2>&1 | grep -E "error|warning" || true
