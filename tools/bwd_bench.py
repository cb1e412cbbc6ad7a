#!/usr/bin/env python3
"""Time the backward (pfa_fa3_bwd: delta + dQ + dK/dV kernels).  FLOP convention: 2.5 x forward = 10*B*H*S*S*D
(five S x S x D products), halved for causal -- the kernels execute 7 products (dQ has its own recompute pass)."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photonic_flash_attention_amd import ops
CONFIGS = {"C3": (4, 16, 4096, 128, True), "C4": (4, 16, 4096, 128, False), "C2": (4, 12, 1024, 64, False),
           "C5": (1, 32, 16384, 128, True)}
dev = torch.device("cuda:0")
for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["C3", "C4"]):
    B, H, S, D, causal = CONFIGS[name]
    q, k, v, g = (torch.randn(B, S, H, D, device=dev).to(torch.bfloat16).permute(0, 2, 1, 3) for _ in range(4))
    out, lse = ops.fa3_forward(q, k, v, causal=causal, return_lse=True)
    for _ in range(5):
        ops.fa3_backward(q, k, v, out, g, lse, causal=causal)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.fa3_backward(q, k, v, out, g, lse, causal=causal)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ms = statistics.median(ts)
    fl = 10.0 * B * H * S * S * D / (2 if causal else 1)
    print(f"{name}: backward {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s (2.5x-forward convention; 7/5 of that executed)", flush=True)
