#!/bin/bash
# rocprofv3 of the backward (tools/bwd_bench.py): kernel trace + stats, then one SQ PMC pass.  usage: tools/profile_bwd.sh <tag> [workloads]
set -o pipefail
TAG=${1:-dev}; WL=${2:-C3}
OUT=gpurun_out/profbwd_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -- python3 tools/bwd_bench.py $WL > $OUT/trace.log 2>&1 || tail -3 $OUT/trace.log
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d $OUT/pmc_sq -- python3 tools/bwd_bench.py $WL > $OUT/pmc_sq.log 2>&1 || tail -3 $OUT/pmc_sq.log
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
st = glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)
print("| kernel | calls | avg ns | % |\n|---|---|---|---|")
for r in csv.DictReader(open(st[0])):
    print(f"| {r['Name'][:70]} | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['Percentage']} |")
f = glob.glob(os.path.join(root, "pmc_sq", "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if "fa3_bwd" in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(f"- {k}: MFMA busy {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * m['SQ_WAVE_CYCLES']) * 100:.1f} % of wave cycles x4; wait_any {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'] * 100:.1f} %; "
          f"wait_inst {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES'] * 100:.1f} %; active {m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES'] * 100:.1f} %; LDS conflicts {m['SQ_LDS_BANK_CONFLICT']:.3g} of {m['SQ_LDS_IDX_ACTIVE']:.3g}")
PY
