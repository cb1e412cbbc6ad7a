#!/usr/bin/env python3
"""Achievable HBM rate on this box: torch copy / read-only sum of a buffer the size of the short-sequence problems."""
import torch
dev = torch.device("cuda:0")
for mb in (64, 134, 268, 1024):
    a = torch.empty(mb * 1024 * 1024 // 2, dtype=torch.bfloat16, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(20):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"copy {mb:5d} MiB: {us:8.1f} us  {2 * a.numel() * 2 / us / 1e6:6.2f} TB/s (read + write)")
