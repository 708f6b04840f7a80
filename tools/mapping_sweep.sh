#!/bin/bash
# On the GPU box: bench.py's stepping rate under both mappings over ensemble sizes (where does one chain per
# wavefront stop paying?).   tools/mapping_sweep.sh <n_data> <chains> [<chains> ...]
nd=$1; shift
for n in "$@"; do
  for mp in wave lane; do
    timeout -k 10 240 python bench.py --n-data $nd --chains-per-gpu $n --mapping $mp --no-cpu-baseline --ess-chains 0 --no-fast-leg --no-strict-leg --no-toy-ess 2>/dev/null \
      | python tools/bench_line.py "n_data=$nd chains=$n $mp" || exit 1
  done
done
