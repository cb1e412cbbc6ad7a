#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then PMC passes (separate runs, kernel-trace only).
# usage: tools/profile.sh <tag> [workload]
set -o pipefail
TAG=${1:-dev}; WL=${2:-C3}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# same K/W as the default bench line, so that the average kernel duration here is the one bench.py reports
B="python3 bench.py --workload $WL --no-others --no-cpu-baseline --no-parity"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; }
B2="python3 bench.py --workload $WL --steps 4 --warmup 2 --reps 1 --no-others --no-probe --no-cpu-baseline --no-parity --variant ${VAR:-0}"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d $OUT/pmc_sq -- $B2 > $OUT/pmc_sq.log 2>&1 || { echo "pmc_sq failed"; tail -5 $OUT/pmc_sq.log; }
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -f csv -d $OUT/pmc_sq2 -- $B2 > $OUT/pmc_sq2.log 2>&1 || { echo "pmc_sq2 failed"; tail -5 $OUT/pmc_sq2.log; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_fetch -- $B2 > $OUT/pmc_fetch.log 2>&1 || { echo "pmc_fetch failed"; tail -5 $OUT/pmc_fetch.log; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -- $B2 > $OUT/pmc_write.log 2>&1 || { echo "pmc_write failed"; tail -5 $OUT/pmc_write.log; }
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -f csv -d $OUT/pmc_l2 -- $B2 > $OUT/pmc_l2.log 2>&1 || { echo "pmc_l2 failed"; tail -5 $OUT/pmc_l2.log; }
python3 tools/prof_summary.py $OUT > $OUT/summary.md 2>&1
cat $OUT/summary.md
