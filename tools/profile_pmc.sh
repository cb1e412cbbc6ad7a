#!/bin/bash
# Quick SQ-only PMC comparison of kernel variants: tools/profile_pmc.sh <tag> <workload> <variant>
set -o pipefail
TAG=$1; WL=$2; VAR=${3:-0}
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B2="python3 bench.py --workload $WL --steps 4 --warmup 2 --reps 1 --no-others --no-probe --no-cpu-baseline --no-parity --variant ${VAR:-0}"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d $OUT/pmc_sq -- $B2 > $OUT/pmc_sq.log 2>&1 || tail -3 $OUT/pmc_sq.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -f csv -d $OUT/pmc_sq2 -- $B2 > $OUT/pmc_sq2.log 2>&1 || tail -3 $OUT/pmc_sq2.log
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_SCA SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_fetch -- $B2 > $OUT/pmc_fetch.log 2>&1 || tail -3 $OUT/pmc_fetch.log
python3 tools/prof_summary.py $OUT
