#!/usr/bin/env python3
"""Ragged problems on the persistent assembly forward (fa3_fwd_p4_*_kl_*: Sq no multiple of 256 / Sk no multiple of 128) against the
8-wave HIP kernel (selector 44): outputs, LSE, and that nothing is written outside the output rows (guard rows around the buffers).
    timeout -k 10 300 python3 tools/p4_ragged_check.py [--time] [--d 64]"""
import argparse, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photonic_flash_attention_amd import ops, _capi

ap = argparse.ArgumentParser()
ap.add_argument("--time", action="store_true")
ap.add_argument("--d", type=int, default=128)
a = ap.parse_args()
D = a.d
dev = torch.device("cuda:0")
dt = torch.bfloat16
CASES = [(1, 8, 300, 300, False), (2, 4, 257, 193, False), (1, 8, 1000, 1000, True), (2, 3, 512, 1000, False), (3, 5, 700, 640, False),
         (1, 16, 2000, 2000, True), (2, 8, 1024, 1025, False), (1, 8, 129, 4000, False), (2, 8, 1500, 1500, True), (1, 4, 255, 255, False),
         (4, 8, 3000, 3000, False), (1, 8, 768, 767, False),
         # an odd number of Q blocks under the causal mask: the middle block is a unit of its own
         (1, 8, 768, 768, True), (2, 4, 256, 256, True), (1, 16, 700, 700, True), (3, 8, 1280, 1280, True), (1, 40, 2304, 2304, True), (9, 32, 200, 200, True)]
bad = 0
for (B, H, Sq, Sk, causal) in CASES:
    g = torch.Generator(device=dev).manual_seed(B * 1000 + H * 10 + Sq)
    G = 3                                                       # guard rows before and after every operand / result
    qb = torch.randn(B, Sq + 2 * G, H, D, device=dev, generator=g).to(dt)
    kb, vb = (torch.randn(B, Sk + 2 * G, H, D, device=dev, generator=g).to(dt) for _ in range(2))
    qb[:, :G] = float("nan"); qb[:, -G:] = float("nan"); kb[:, :G] = float("nan"); kb[:, -G:] = float("nan"); vb[:, :G] = float("nan"); vb[:, -G:] = float("nan")
    q, k, v = (t[:, G:-G].permute(0, 2, 1, 3) for t in (qb, kb, vb))
    res = {}
    for var in (44, 45):
        for o32 in (False, True):
            ob = torch.full((B, Sq + 2 * G, H, D), 7.0, device=dev, dtype=torch.float32 if o32 else dt)
            lb = torch.full((B, H, Sq + 2 * G), 7.0, device=dev, dtype=torch.float32)
            out = ob[:, G:-G].permute(0, 2, 1, 3)
            o, lse = ops.fa3_forward(q, k, v, causal=causal, out=out, out_dtype=torch.float32 if o32 else None, return_lse=True, _variant=var)
            torch.cuda.synchronize()
            guard_ok = bool((ob[:, :G] == 7.0).all()) and bool((ob[:, -G:] == 7.0).all())
            res[var, o32] = (o.float().clone(), lse.clone(), guard_ok)
    names = [_capi.describe(ops.build_args(q, k, v, torch.empty(B, Sq, H, D, device=dev, dtype=torch.float32 if o32 else dt).permute(0, 2, 1, 3),
                                           causal=causal, split_p=o32, variant=45)[0])[0] for o32 in (False, True)]
    ok = all(("_kl_" in n) == (Sq % 256 != 0 or Sk % 128 != 0) and "p4" in n for n in names)
    msg = []
    for o32 in (False, True):
        d_o = (res[45, o32][0] - res[44, o32][0]).abs().nan_to_num(1e9)
        d_l = (res[45, o32][1] - res[44, o32][1]).abs().nan_to_num(1e9)
        tol = 3e-5 if o32 else 2e-2
        ok = ok and float(d_o.max()) <= tol and float(d_l.max()) <= 1e-4 and res[45, o32][2]
        msg.append(f"{'o32' if o32 else 'o16'} max|dO| {float(d_o.max()):.2e} max|dLSE| {float(d_l.max()):.2e} guard {'ok' if res[45, o32][2] else 'WRITTEN'}")
    print(f"B{B} H{H} Sq{Sq} Sk{Sk} {'causal' if causal else 'full  '} {names[0]}: {' | '.join(msg)}  {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print("FAILED" if bad else "ALL OK", flush=True)
if a.time and not bad:
    for (B, H, Sq, Sk, causal) in [(16, 16, 2000, 2000, False), (4, 16, 4000, 4000, True), (16, 16, 1000, 1000, False), (8, 16, 3000, 3000, True)]:
        q = torch.randn(B, Sq, H, D, device=dev).to(dt).permute(0, 2, 1, 3)
        k, v = (torch.randn(B, Sk, H, D, device=dev).to(dt).permute(0, 2, 1, 3) for _ in range(2))
        out = torch.empty(B, Sq, H, D, device=dev, dtype=dt).permute(0, 2, 1, 3)
        fl = 4.0 * B * H * Sq * Sk * D / (2 if causal else 1)
        times = {43 if D == 128 else 44: [], 45: []}
        for var in times:
            for _ in range(20):
                ops.fa3_forward(q, k, v, causal=causal, out=out, _variant=var)
        torch.cuda.synchronize()
        for r in range(7):
            for var in times:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    ops.fa3_forward(q, k, v, causal=causal, out=out, _variant=var)
                e1.record(); torch.cuda.synchronize()
                times[var].append(e0.elapsed_time(e1) / 20)
        for var in times:
            med = statistics.median(times[var])
            print(f"B{B} H{H} Sq{Sq} Sk{Sk} {'causal' if causal else 'full'} var {var}: {med:.4f} ms {fl / med / 1e9:.1f} TF", flush=True)
sys.exit(1 if bad else 0)
