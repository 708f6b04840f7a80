#!/bin/bash
# On the GPU box: bench.py (no CPU leg, no ESS leg) alternately with the libraries given, twice each.
#   tools/ab_bench.sh libA.so libB.so ...
for rep in 1 2; do
  for lib in "$@"; do
    MCMC_HIP_LIB=$PWD/mcmc_amd/$lib timeout -k 10 240 python bench.py --no-cpu-baseline --ess-chains 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value']/1e6,3), 'M steps/s', round(d['roofline']['kernel_ms_per_launch'],2), 'ms/launch')" || exit 1
  done
done
