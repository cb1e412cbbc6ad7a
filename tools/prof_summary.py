#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (tools/profile.sh) into a markdown summary for profiles/."""
import csv, glob, os, sys, collections
root = sys.argv[1]
def find(sub, pat):
    r = glob.glob(os.path.join(root, sub, "**", pat), recursive=True)
    return r[0] if r else None
print(f"# rocprofv3 summary: {root}\n")
st = find("trace", "*kernel_stats.csv")
if st:
    print("## kernel stats (--kernel-trace --stats)\n")
    print("| kernel | calls | total ns | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|---|")
    for r in csv.DictReader(open(st)):
        print(f"| {r['Name'][:90]} | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {r['Percentage']} |")
tr = find("trace", "*kernel_trace.csv")
if tr:
    rows = [r for r in csv.DictReader(open(tr)) if "fa3_fwd" in r["Kernel_Name"]]
    if rows:
        d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
        r0 = rows[0]
        print(f"\nfa3_fwd dispatches: {len(d)}, median {d[len(d)//2]} ns, min {d[0]} ns, max {d[-1]} ns; "
              f"grid {r0.get('Grid_Size_X')} wg {r0.get('Workgroup_Size_X')} VGPR {r0.get('VGPR_Count')} accum {r0.get('Accum_VGPR_Count')} "
              f"SGPR {r0.get('SGPR_Count')} LDS {r0.get('LDS_Block_Size')} scratch {r0.get('Scratch_Size')}")
print("\n## PMC (per fa3_fwd dispatch, mean over dispatches)\n")
for sub in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write", "pmc_l2"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        print(f"- {sub}: no output"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "fa3_fwd" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"- {k}: {sum(v)/len(v):.6g}  (n={len(v)})")
