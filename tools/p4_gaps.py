#!/usr/bin/env python3
"""Static issue-budget table of the persistent forward's FULL body: per MFMA gap, the vector-issue cycles of its fillers (prices from
MI355X_MICROARCH.md 'vector-instruction ISSUE cost': MFMA 8 of its 32, v_exp 8, other VALU 4, s_* 4, LDS read ~2, LDS-DMA piece ~30).
A gap runs ~max(32, sum).   python3 tools/p4_gaps.py [parity] [--list]"""
import os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "photonic_flash_attention_amd", "csrc"))
import gen_fa3_fwd_p4 as G

def cost(op_line):
    return 8 if op_line.startswith("v_mfma") else G.Gen.price(op_line)

g = G.Gen("bf16", True)
p = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 0
g.body_full(p)
ins = [l.strip() for l in g.main if l.startswith("\t") and not l.strip().startswith(";")]
gaps, cur = [], None
pre = []
for x in ins:
    op = x.split()[0]
    if op.startswith("v_mfma"):
        cur = [x]; gaps.append(cur)
    elif cur is None: pre.append(x)
    else: cur.append(x)
tot = 0
print(f"parity {p}: {len(gaps)} MFMA gaps; {len(pre)} instructions before the first MFMA ({sum(cost(x) for x in pre)} cycles)")
for i, gp in enumerate(gaps):
    c = sum(cost(x) for x in gp)
    kinds = {}
    for x in gp[1:]:
        op = x.split()[0]
        k = "exp" if op.startswith("v_exp") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "dma" if op.startswith("buffer") else "wait" if op.startswith("s_waitcnt") else "salu"
        kinds[k] = kinds.get(k, 0) + 1
    tot += max(32, c)
    flag = " <-- over" if c > 32 else ""
    print(f"  gap {i:2d} ({'QK' if i < 32 else 'PV'}): issue {c:3d}  {kinds}{flag}")
    if "--list" in sys.argv:
        for x in gp: print("        " + x)
print(f"sum of max(32, issue) over the gaps: {tot} cycles per wave-tile ({tot / 64:.1f} per MFMA); instructions per tile: {len(ins)}")
