#!/usr/bin/env python3
"""Backward at C3's shape: causal flag / unmasked against 4-D element masks (the triangle, a 1024-wide sliding window) and a [B,Sk] key mask."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photonic_flash_attention_amd import ops
dev = torch.device("cuda:0")
B, H, S, D = 4, 16, 4096, 128
q, k, v, g = (torch.randn(B, S, H, D, device=dev).to(torch.bfloat16).permute(0, 2, 1, 3) for _ in range(4))
i = torch.arange(S, device=dev)
tril = (i[None, :] <= i[:, None])[None, None]
sw = ((i[:, None] - i[None, :] >= 0) & (i[:, None] - i[None, :] < 1024))[None, None]
km = (i[None, :] < torch.tensor([S, S * 9 // 16, S * 13 // 16, S * 11 // 16], device=dev)[:, None])
for name, kw in (("causal flag", dict(causal=True)), ("no mask", {}), ("triangle as a 4-D mask", dict(mask=tril)), ("sliding window 1024 as a 4-D mask", dict(mask=sw)),
                 ("[B,Sk] key mask + causal flag", dict(causal=True, key_mask=km))):
    out, lse = ops.fa3_forward(q, k, v, return_lse=True, **kw)
    f = lambda: ops.fa3_backward(q, k, v, out, g, lse, **kw)
    for _ in range(3): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): f()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 5)
    print(f"{name}: backward {statistics.median(ts):.3f} ms", flush=True)
