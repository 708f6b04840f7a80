#!/bin/bash
# On the GPU box: bench.py (no CPU leg, no ESS legs, no side legs) with library A, then B (, then C ...), and
# the whole round a second time -- every GPU step under its own time limit, chained: the first failure ends
# the script.   tools/ab_bench.sh libA.so libB.so [libC.so ...]      (EXTRA_ARGS="--config c1" for other workloads)
one() { MCMC_HIP_LIB=$PWD/mcmc_amd/$1 timeout -k 10 240 python bench.py --no-cpu-baseline --ess-chains 0 --no-fast-leg --no-strict-leg --no-toy-ess $EXTRA_ARGS 2>/dev/null | python tools/bench_line.py "$1"; }
for round in 1 2; do
  for lib in "$@"; do one "$lib" || exit 1; done
done
