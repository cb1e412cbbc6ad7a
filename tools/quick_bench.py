#!/usr/bin/env python3
"""Developer loop: time the kernel on the BASELINE configs (1 GPU), print TFLOP/s.
Not the judged bench (that is /bench.py)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photonic_flash_attention_amd import ops  # noqa: E402

CONFIGS = {
    "C2": (4, 12, 1024, 64, False),
    "C3": (4, 16, 4096, 128, True),
    "C4": (4, 16, 4096, 128, False),
    "C5": (1, 32, 16384, 128, True),
    "C3d64": (4, 32, 4096, 64, True),
}


def run(name, iters, split, dtype, out32):
    B, H, S, D, causal = CONFIGS[name]
    dev = torch.device("cuda:0")
    q, k, v = (torch.randn(B, S, H, D, device=dev, dtype=torch.float32).to(dtype) for _ in range(3))
    odt = torch.float32 if out32 else None
    for _ in range(3):
        ops.fa3_forward_bshd(q, k, v, causal=causal, out_dtype=odt, split_p=split)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.fa3_forward_bshd(q, k, v, causal=causal, out_dtype=odt, split_p=split)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    ms = min(ts)
    fl = 4.0 * B * H * S * S * D / (2 if causal else 1)
    print(f"{name:6s} {str(dtype)[6:]:9s} split={int(split)} o32={int(out32)}  {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s  "
          f"({fl / ms / 1e9 / 2500 * 100:.1f}% of 2.5 PF)", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="C2,C3,C4,C5")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    for n in a.configs.split(","):
        run(n, a.iters, False, torch.bfloat16, False)
    run("C3", a.iters, True, torch.bfloat16, True)
    run("C3", a.iters, False, torch.float16, False)
