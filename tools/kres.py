#!/usr/bin/env python3
"""Summarise `make asm` kernel-resource-usage remarks + MFMA/LDS instruction counts per kernel."""
import re, subprocess, sys, collections
out = subprocess.run(["make", "-C", "photonic_flash_attention_amd/csrc", "asm"], capture_output=True, text=True)
txt = out.stdout + out.stderr
cur = None; rows = collections.OrderedDict()
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark: .*?:\s+([A-Za-z \[\]/]+): (\S+) \[", line)
    if m and cur: rows[cur][m.group(1).strip()] = m.group(2)
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for k, v in rows.items():
    d = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    d = d.replace("void pfa::", "").replace("(pfa::FwdParams)", "")
    if flt and flt not in d: continue
    print(f"{d:70s} VGPR {v.get('VGPRs')} AGPR {v.get('AGPRs')} SGPR {v.get('TotalSGPRs')} scratch {v.get('ScratchSize [bytes/lane]')} occ {v.get('Occupancy [waves/SIMD]')}")
if "error" in txt: print(txt[-3000:])
