#!/bin/bash
# On the GPU box: bench.py (no CPU leg, no ESS leg) with library A, then B, then A, then B -- four
# GPU steps, each under its own time limit, chained: the first failure ends the script.
#   tools/ab_bench.sh libA.so libB.so
A=$1; B=$2
one() { MCMC_HIP_LIB=$PWD/mcmc_amd/$1 timeout -k 10 240 python bench.py --no-cpu-baseline --ess-chains 0 2>/dev/null | python tools/bench_line.py "$1"; }
one "$A" && one "$B" && one "$A" && one "$B"
