#!/usr/bin/env python3
"""Cost of the mask paths of the forward: no mask vs seqlens_k vs [B,Sk] key mask vs full [B,1,Sq,Sk] element mask."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photonic_flash_attention_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=30):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, H, S, D) in [(4, 12, 1024, 64), (8, 16, 2048, 128), (16, 12, 512, 64)]:
    q, k, v = (torch.randn(B, S, H, D, device=dev).to(torch.bfloat16).permute(0, 2, 1, 3) for _ in range(3))
    out = torch.empty(B, S, H, D, device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3)
    lens = [S - (S // 8) * (b % 3) for b in range(B)]
    km = torch.zeros(B, S, dtype=torch.bool, device=dev)
    for b, n in enumerate(lens): km[b, :n] = True
    em = km[:, None, None, :].expand(B, 1, S, S).contiguous()
    fl = 4.0 * B * H * S * S * D
    for name, kw in (("no mask", {}), ("seqlens_k", dict(seqlens_k=lens)), ("key mask [B,Sk]", dict(key_mask=km)), ("element mask [B,1,Sq,Sk]", dict(mask=em))):
        us = t(lambda: ops.fa3_forward(q, k, v, out=out, **kw))
        print(f"B{B} H{H} S{S} D{D} {name:26s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF (dense count)", flush=True)
