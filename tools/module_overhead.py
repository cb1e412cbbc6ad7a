#!/usr/bin/env python3
"""Where a small module forward (BASELINE C1: B2 S128 E256 H4 fp32) spends its host time; eager vs HIP-graph replay."""
import cProfile, pstats, os, sys, time, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photonic_flash_attention_amd import FlashAttention3, PhotonicFlashAttention
dev = "cuda:0"
for name, (B, S, E, H, dt) in {"C1 fp32": (2, 128, 256, 4, torch.float32), "C2 bf16": (4, 1024, 768, 12, torch.bfloat16)}.items():
    m = FlashAttention3(E, H, dtype=dt).to(dev).eval()
    x = torch.randn(B, S, E, device=dev, dtype=dt)
    with torch.no_grad():
        for _ in range(20): m(x)
        torch.cuda.synchronize()
        N = 500
        t = time.perf_counter()
        for _ in range(N): m(x)
        t_enq = (time.perf_counter() - t) / N
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t) / N
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            m(x); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                y = m(x)[0]
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(N): g.replay()
        torch.cuda.synchronize()
        t_graph = (time.perf_counter() - t) / N
        print(f"{name}: eager enqueue {t_enq*1e6:.0f} us, end-to-end {t_all*1e6:.0f} us; graph replay {t_graph*1e6:.0f} us per forward")
        if name.startswith("C1"):
            pr = cProfile.Profile(); pr.enable()
            for _ in range(200): m(x)
            pr.disable(); torch.cuda.synchronize()
            out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(14)
            print("\n".join(out.getvalue().splitlines()[4:26]))
