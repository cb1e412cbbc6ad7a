#!/usr/bin/env python3
"""A/B kernel variants interleaved in ONE process (cdna guide rule 24): median/min ms and TFLOP/s."""
import argparse, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photonic_flash_attention_amd import ops, _capi

CONFIGS = {"C3": (4, 16, 4096, 128, True), "C4": (4, 16, 4096, 128, False), "C5": (1, 32, 16384, 128, True),
           "C2": (4, 12, 1024, 64, False), "S2K": (16, 16, 2048, 128, False), "S8Kc": (2, 16, 8192, 128, True),
           "S1K": (16, 16, 1024, 128, False), "S1Kc": (16, 16, 1024, 128, True), "S512": (32, 16, 512, 128, False),
           "S512c": (32, 16, 512, 128, True), "S256": (64, 16, 256, 128, False), "S2Kc": (16, 16, 2048, 128, True),
           # few (batch, head) pairs: does the persistent kernel (one 256-row unit -- causal: a PAIR of blocks -- per CU) leave CUs idle?
           "b1h1S4K": (1, 1, 4096, 128, False), "b1h1S4Kc": (1, 1, 4096, 128, True), "b1h4S4Kc": (1, 4, 4096, 128, True), "b1h8S2K": (1, 8, 2048, 128, False),
           "b1h8S8Kc": (1, 8, 8192, 128, True), "b2h8S1Kc": (2, 8, 1024, 128, True), "b1h12S1Kd64": (1, 12, 1024, 64, False), "b1h16S16Kc": (1, 16, 16384, 128, True)}
ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="C4,C3")
ap.add_argument("--variants", default="0,1,2,3,4,5")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--check", action="store_true")
ap.add_argument("--data", default="randn", choices=["randn", "zeros", "const", "small"], help="operand data: DVFS probe (cdna guide: zero operands clock higher)")
a = ap.parse_args()
dev = torch.device("cuda:0")
variants = [int(x) for x in a.variants.split(",")]
for name in a.configs.split(","):
    B, H, S, D, causal = CONFIGS[name]
    q, k, v = (torch.randn(B, S, H, D, device=dev, dtype=torch.float32).to(torch.bfloat16) for _ in range(3))
    if a.data == "zeros": q, k, v = (torch.zeros_like(t) for t in (q, k, v))
    elif a.data == "const": q, k, v = (torch.full_like(t, 0.5) for t in (q, k, v))
    elif a.data == "small": q, k, v = ((t * 0.01).to(torch.bfloat16) for t in (q, k, v))
    out = torch.empty(B, S, H, D, device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3)
    qv, kv, vv = (t.permute(0, 2, 1, 3) for t in (q, k, v))
    fl = 4.0 * B * H * S * S * D / (2 if causal else 1)
    times = {v_: [] for v_ in variants}
    ref = None
    for v_ in variants:
        for _ in range(3):
            ops.fa3_forward(qv, kv, vv, causal=causal, out=out, _variant=v_)
        if a.check:
            torch.cuda.synchronize()
            if ref is None: ref = out.clone()
            else: print(f"  variant {v_} max|diff| vs variant {variants[0]}: {float((out.float()-ref.float()).abs().max()):.3e}")
    torch.cuda.synchronize()
    for r in range(a.rounds):
        for v_ in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                ops.fa3_forward(qv, kv, vv, causal=causal, out=out, _variant=v_)
            e1.record(); torch.cuda.synchronize()
            times[v_].append(e0.elapsed_time(e1) / a.iters)
    for v_ in variants:
        nm = _capi.describe(ops.build_args(qv, kv, vv, out, causal=causal, variant=v_)[0])[0]
        med, mn = statistics.median(times[v_]), min(times[v_])
        print(f"{name:5s} var {v_} {nm:44s} median {med:7.4f} ms {fl/med/1e9:7.1f} TF | min {mn:7.4f} ms {fl/mn/1e9:7.1f} TF", flush=True)
