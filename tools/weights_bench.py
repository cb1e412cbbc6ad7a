#!/usr/bin/env python3
"""Bandwidth of the need_weights path (pfa_fa3_weights): bytes of the [B,H,Sq,Sk] output / time."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photonic_flash_attention_amd import ops
dev = torch.device("cuda:0")
for (B, H, S, D, causal) in [(4, 12, 1024, 64, False), (2, 16, 2048, 128, False), (2, 16, 2048, 128, True), (1, 8, 4096, 128, False)]:
    q, k, v = (torch.randn(B, S, H, D, device=dev).to(torch.bfloat16).permute(0, 2, 1, 3) for _ in range(3))
    for _ in range(3):
        ops.fa3_forward(q, k, v, causal=causal, return_weights=True)
    torch.cuda.synchronize()
    ts, t0s = [], []
    for _ in range(5):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        ops.fa3_forward(q, k, v, causal=causal, return_lse=True)
        e1.record()
        ops.fa3_forward(q, k, v, causal=causal, return_weights=True)
        e2.record(); torch.cuda.synchronize()
        t0s.append(e0.elapsed_time(e1)); ts.append(e1.elapsed_time(e2))
    ms = statistics.median(ts) - statistics.median(t0s)
    by = B * H * S * S * 2
    print(f"B{B} H{H} S{S} D{D} causal={causal}: weights pass {ms:.3f} ms for {by / 1e6:.0f} MB = {by / ms / 1e9:.2f} TB/s "
          f"(W allocated uninitialised: the kernel writes masked elements as zeros)", flush=True)
