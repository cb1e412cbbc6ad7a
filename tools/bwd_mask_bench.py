#!/usr/bin/env python3
"""Backward with masks: no mask vs seqlens_k vs [B,Sk] key mask vs [B,1,Sq,Sk] element mask (pfa_fa3_bwd reads a mask byte per score)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photonic_flash_attention_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, H, S, D) in [(4, 12, 1024, 64), (8, 16, 2048, 128)]:
    q, k, v, g = (torch.randn(B, S, H, D, device=dev).to(torch.bfloat16).permute(0, 2, 1, 3) for _ in range(4))
    lens = [S - (S // 8) * (b % 3) for b in range(B)]
    km = torch.zeros(B, S, dtype=torch.bool, device=dev)
    for b, n in enumerate(lens): km[b, :n] = True
    em = km[:, None, None, :].expand(B, 1, S, S).contiguous()
    for name, kw in (("no mask", {}), ("seqlens_k", dict(seqlens_k=lens)), ("key mask [B,Sk]", dict(key_mask=km)), ("element mask [B,1,Sq,Sk]", dict(mask=em))):
        out, lse = ops.fa3_forward(q, k, v, return_lse=True, **kw)
        us = t(lambda: ops.fa3_backward(q, k, v, out, g, lse, **kw))
        print(f"B{B} H{H} S{S} D{D} backward, {name:26s} {us:8.1f} us", flush=True)
