#!/usr/bin/env python3
"""Host-side cost of one forward call (ctypes marshalling + hipLaunchKernel), and CUDA-graph replay of the same call."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photonic_flash_attention_amd import ops
dev = torch.device("cuda:0")
B, H, S, D = 2, 4, 128, 64          # BASELINE config C1 shape
q, k, v = (torch.randn(B, S, H, D, device=dev).to(torch.bfloat16).permute(0, 2, 1, 3) for _ in range(3))
out = torch.empty(B, S, H, D, device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3)
for _ in range(50):
    ops.fa3_forward(q, k, v, out=out)
torch.cuda.synchronize()
N = 3000
t = time.perf_counter()
for _ in range(N):
    ops.fa3_forward(q, k, v, out=out)
t_enq = (time.perf_counter() - t) / N
torch.cuda.synchronize()
t_all = (time.perf_counter() - t) / N
print(f"eager: enqueue {t_enq * 1e6:.1f} us/call, end-to-end {t_all * 1e6:.1f} us/call")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ops.fa3_forward(q, k, v, out=out)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(10):
            ops.fa3_forward(q, k, v, out=out)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(N // 10):
    g.replay()
torch.cuda.synchronize()
print(f"graph (10 calls per replay): {(time.perf_counter() - t) / N * 1e6:.1f} us/call")
