#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / L2 hit counters of bench.py's kernel for one kernel selector (kernel-trace only, separate passes)
# usage: tools/profile_fetch.sh <tag> [workload] [selector 43|44|45]
set -o pipefail
TAG=${1:-dev}; WL=${2:-C3}; VAR=${3:-0}
OUT=gpurun_out/fetch_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B2="python3 bench.py --workload $WL --steps 4 --warmup 2 --reps 1 --no-others --no-cpu-baseline --no-parity --variant $VAR"
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_fetch -- $B2 > $OUT/pmc_fetch.log 2>&1 || echo "pmc_fetch failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -- $B2 > $OUT/pmc_write.log 2>&1 || echo "pmc_write failed"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -f csv -d $OUT/pmc_l2 -- $B2 > $OUT/pmc_l2.log 2>&1 || echo "pmc_l2 failed"
python3 tools/prof_summary.py $OUT 2>&1 | grep -E "FETCH_SIZE|WRITE_SIZE|TCC_|fa3_fwd dispatches"
