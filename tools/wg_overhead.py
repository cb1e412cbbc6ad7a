#!/usr/bin/env python3
"""Fixed cost per workgroup: time the forward at Sq = 256 (one Q block per head, 1024 heads = 4 workgroups per CU; env B / SQ / D
change the shape: B=4 SQ=4096 is C4's geometry, far from the HBM limit) against the number of key tiles; the intercept of the line is what a workgroup pays outside its tile loop (Q load, first DMA, epilogue)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photonic_flash_attention_amd import ops
dev = torch.device("cuda:0")
B, H, Sq, D = int(os.environ.get("B", "64")), 16, int(os.environ.get("SQ", "256")), int(os.environ.get("D", "128"))
ROUNDS = B * H * ((Sq + 255) // 256) / 256.0          # workgroups per CU
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0      # kernel selector 43 / 44 / 45
rows = []
for Sk in (64, 128, 256, 512, 1024, 2048, 4096):
    q = torch.randn(B, Sq, H, D, device=dev).to(torch.bfloat16).permute(0, 2, 1, 3)
    k, v = (torch.randn(B, Sk, H, D, device=dev).to(torch.bfloat16).permute(0, 2, 1, 3) for _ in range(2))
    out = torch.empty(B, Sq, H, D, device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3)
    for _ in range(200):
        ops.fa3_forward(q, k, v, out=out, _variant=variant)
    best = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.fa3_forward(q, k, v, out=out, _variant=variant)
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 50 * 1e3)
    us = sorted(best)[len(best) // 2]
    tiles = Sk // 64
    rows.append((tiles, us))
    print(f"Sk {Sk:5d} tiles {tiles:3d}: {us:8.1f} us  ({us / ROUNDS:6.2f} us per workgroup round, {4 * B * H * Sq * Sk * D / us / 1e6:7.1f} TF)")
(t0, u0), (t1, u1) = rows[-2], rows[-1]
slope = (u1 - u0) / (t1 - t0)
print(f"slope {slope / ROUNDS:.3f} us per tile per workgroup; intercept {(u1 - slope * t1) / ROUNDS:.2f} us per workgroup")
