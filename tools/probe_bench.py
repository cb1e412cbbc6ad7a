#!/usr/bin/env python3
"""The in-run MFMA ceiling probe of bench.py, stand-alone: pfa_probe_mfma (csrc/pfa_probe.hip) on random and on zero operands."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photonic_flash_attention_amd import _capi
lib = _capi.load()
dev = torch.device("cuda:0")
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
sink = torch.empty(n_cu * 256, device=dev, dtype=torch.float32)
for tag, rnd in (("random", torch.randn(32768, device=dev).to(torch.bfloat16)), ("zeros", torch.zeros(32768, device=dev, dtype=torch.bfloat16))):
    fl = C.c_double()
    st = torch.cuda.current_stream().cuda_stream
    for iters in (2000, 8000):
        for _ in range(3):
            lib.pfa_probe_mfma(rnd.data_ptr(), sink.data_ptr(), iters, 0, st, C.byref(fl))
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = lib.pfa_probe_mfma(rnd.data_ptr(), sink.data_ptr(), iters, 0, st, C.byref(fl))
            e1.record(); torch.cuda.synchronize()
            assert n > 0, n
            ts.append(e0.elapsed_time(e1))
        ms = sorted(ts)[2]
        print(f"probe {tag}: {n} workgroups, {iters} x 64 MFMAs per wave: {ms:.3f} ms -> {fl.value / ms / 1e9:.1f} TFLOP/s "
              f"({fl.value / ms / 1e9 / 2500:.3f} of 2.5 PF nominal)", flush=True)
