#!/usr/bin/env python3
"""Bandwidth of the need_weights pass (pfa_fa3_weights alone, HIP events around `steps` launches on the forward's arguments): bytes of the
[B,H,Sq,Sk] output / time.  WDT=fp32: fp32 weights; WB_FROM=n: skip the first n shapes.  Beside it: tools/write_rate.py (the box's fill rate)."""
import ctypes as C, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photonic_flash_attention_amd import ops, _capi
dev = torch.device("cuda:0")
WDT = torch.float32 if os.environ.get("WDT") == "fp32" else torch.bfloat16
ES = 4 if WDT == torch.float32 else 2
lib = _capi.load()
for (B, H, S, D, causal) in [(4, 12, 1024, 64, False), (2, 16, 2048, 128, False), (2, 16, 2048, 128, True), (1, 8, 4096, 128, False),
                            (4, 16, 4096, 128, False), (4, 16, 4096, 128, True), (16, 12, 2048, 64, False)][int(os.environ.get("WB_FROM", "0")):]:
    q, k, v = (torch.randn(B, S, H, D, device=dev).to(torch.bfloat16).permute(0, 2, 1, 3) for _ in range(3))
    out = torch.empty((B, S, H, D), dtype=torch.bfloat16, device=dev).permute(0, 2, 1, 3)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=dev)
    args, keep = ops.build_args(q, k, v, out, causal=causal, lse=lse)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _capi.check_status(lib.pfa_fa3_fwd(C.byref(args), C.c_void_p(stream)))
    w = torch.empty((B, H, S, S), dtype=WDT, device=dev)
    launch = lambda: _capi.check_status(lib.pfa_fa3_weights(C.byref(args), C.c_void_p(w.data_ptr()), ops._DT[WDT], w.stride(0), w.stride(1),
                                                            w.stride(2), C.c_void_p(stream)))
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    ts, steps = [], 10
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            launch()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / steps)
    ms, by = statistics.median(ts), B * H * S * S * ES
    print(f"B{B} H{H} S{S} D{D} causal={causal}: weights pass {ms * 1e3:.1f} us for {by / 1e6:.0f} MB = {by / ms / 1e9:.2f} TB/s "
          f"(W allocated uninitialised: the kernel writes masked elements as zeros)", flush=True)
