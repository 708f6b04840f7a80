#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats of bench.py and the
# separate PMC passes the MI355X guide prescribes (FETCH_SIZE and WRITE_SIZE cannot share
# a pass).  Output under gpurun_out/prof_<tag>/; copy the summaries to profiles/ afterwards.
#   tools/profile_gpu.sh <tag> [bench args...]
set -u
tag=${1:-r01}; shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
args="--no-cpu-baseline --ess-chains 0 --no-fast-leg --no-strict-leg --no-toy-ess $*"
# PMC_ARGS: what the counter passes append (default: one launch of 16 steps after 16 warm-up steps)
pmc_args=${PMC_ARGS:---steps 16 --warmup 16}
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$root/bench.py" $args > "$out/bench_stats.json" 2> "$out/stats.err" || { echo "stats pass failed"; tail -5 "$out/stats.err"; exit 1; }
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo "$pass" | cut -d' ' -f1)
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d "$out/pmc_$name" -- python3 "$root/bench.py" $args $pmc_args > "$out/bench_pmc_$name.json" 2> "$out/pmc_$name.err" || { echo "pmc pass $name failed"; tail -5 "$out/pmc_$name.err"; exit 1; }
  echo "pass $name done"
done
# matrix-core counters (fast mode): optional, a name this rocprofv3 does not know must not fail the run
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES --output-format csv -d "$out/pmc_MFMA" -- python3 "$root/bench.py" $args $pmc_args > "$out/bench_pmc_MFMA.json" 2> "$out/pmc_MFMA.err" && echo "pass MFMA done" || { echo "pass MFMA not available"; rm -rf "$out/pmc_MFMA"; }
find "$out" -name '*.csv' | head -40
