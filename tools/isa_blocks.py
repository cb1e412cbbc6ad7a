#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in the -save-temps .s (after `make asm`)."""
import re, sys, collections
name = sys.argv[1]
s = open('/tmp/pfa_asm/pfa_capi-hip-amdgcn-amd-amdhsa-gfx950.s').read()
i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
body = s[i:j]; open('/tmp/kernel.s', 'w').write(body)
blocks = []; cur = ['entry', collections.Counter(), 0]; blocks.append(cur)
for ln in body.splitlines():
    m = re.match(r'^(\.LBB\d+_\d+):', ln)
    if m: cur = [m.group(1), collections.Counter(), 0]; blocks.append(cur); continue
    t = ln.strip().split()
    if not t or t[0].startswith(';') or t[0].startswith('.') or t[0].endswith(':'): continue
    op = t[0]; cur[2] += 1
    k = 'mfma' if op.startswith('v_mfma') else 'ds' if op.startswith('ds_') else 'exp' if op.startswith('v_exp') else \
        'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'vmem' if op.startswith(('global_', 'buffer_')) else 'other'
    cur[1][k] += 1
for b in blocks:
    if b[2] >= int(sys.argv[2]) if len(sys.argv) > 2 else 8: print(b[0], b[2], dict(b[1]))
