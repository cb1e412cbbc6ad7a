#!/usr/bin/env python3
"""Key-mask kernels of the persistent assembly forward (fa3_fwd_p4_*_km_*) against the 8-wave HIP kernel (selector 44) on the same
[B, Sk] byte masks: padding masks (a visible prefix per batch), random masks, a fully masked batch row; then timing against the
unmasked launch.  Run on the GPU box:  timeout -k 10 300 python3 tools/p4_kmask_check.py [--time] [--d 64]"""
import argparse, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photonic_flash_attention_amd import ops, _capi

ap = argparse.ArgumentParser()
ap.add_argument("--time", action="store_true")
ap.add_argument("--d", type=int, default=128)
a = ap.parse_args()
D = a.d
dev = torch.device("cuda:0")
dt = torch.bfloat16


def masks(B, Sk, kind, g):
    m = torch.ones(B, Sk, dtype=torch.uint8, device=dev)
    if kind == "pad":
        lens = torch.randint(1, Sk + 1, (B,), generator=g, device=dev)
        lens[0] = Sk
        m = (torch.arange(Sk, device=dev)[None, :] < lens[:, None]).to(torch.uint8)
    elif kind == "rand":
        m = (torch.rand(B, Sk, generator=g, device=dev) < 0.7).to(torch.uint8)
    elif kind == "row0":
        m = (torch.rand(B, Sk, generator=g, device=dev) < 0.9).to(torch.uint8)
        m[B - 1] = 0                                   # a batch with no visible key at all: output 0, LSE -inf
    return m


bad = 0
for (B, H, Sq, Sk, causal) in [(2, 8, 512, 512, True), (3, 4, 256, 384, False), (2, 8, 1024, 1024, False), (4, 8, 512, 512, True), (5, 3, 768, 640, False),
                               (2, 16, 2048, 2048, True),
                               # ragged shapes (no causal mask): per-item descriptor records, byte loads limited to the lanes whose keys exist
                               (2, 8, 300, 300, False), (3, 4, 257, 193, False), (2, 8, 1000, 1000, False), (1, 8, 129, 3001, False), (4, 4, 1024, 1025, False),
                               (2, 8, 1000, 1000, True), (3, 4, 769, 769, True), (1, 16, 2000, 2000, True), (2, 8, 200, 200, True)]:
    for kind in ("pad", "rand", "row0"):
        g = torch.Generator(device=dev).manual_seed(B * 1000 + H * 10 + Sq + len(kind))
        q = torch.randn(B, Sq, H, D, device=dev, generator=g).to(dt).permute(0, 2, 1, 3)
        k, v = (torch.randn(B, Sk, H, D, device=dev, generator=g).to(dt).permute(0, 2, 1, 3) for _ in range(2))
        km = masks(B, Sk, kind, g)
        res = {}
        for var in (44, 45):
            for o32 in (False, True):
                o, lse = ops.fa3_forward(q, k, v, causal=causal, key_mask=km, return_lse=True, _variant=var,
                                         out_dtype=torch.float32 if o32 else None)
                torch.cuda.synchronize()
                res[var, o32] = (o.float().clone(), lse.clone())
        names = [_capi.describe(ops.build_args(q, k, v, torch.empty_like(q, dtype=torch.float32 if o32 else dt), causal=causal, key_mask=km,
                                               split_p=o32, variant=45)[0])[0] for o32 in (False, True)]
        ok = all("_km_" in n for n in names)
        msg = []
        for o32 in (False, True):
            d_o = (res[45, o32][0] - res[44, o32][0]).abs().nan_to_num(1e9)
            l45, l44 = res[45, o32][1], res[44, o32][1]
            same_inf = bool((torch.isinf(l45) == torch.isinf(l44)).all())
            d_l = (l45 - l44).abs().nan_to_num(0.0)
            tol = 3e-5 if o32 else 2e-2
            ok = ok and float(d_o.max()) <= tol and float(d_l.max()) <= 1e-4 and same_inf and not bool(torch.isnan(res[45, o32][0]).any())
            msg.append(f"{'o32' if o32 else 'o16'} max|dO| {float(d_o.max()):.2e} max|dLSE| {float(d_l.max()):.2e}")
        print(f"B{B} H{H} Sq{Sq} Sk{Sk} {'causal' if causal else 'full  '} {kind:5s} {names[0]}: {' | '.join(msg)}  {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += 0 if ok else 1
print("FAILED" if bad else "ALL OK", flush=True)
if a.time and not bad:
    for (B, H, S, causal) in [(16, 16, 2048, False), (4, 16, 4096, True), (16, 16, 2048, True)]:
        q, k, v = (torch.randn(B, S, H, D, device=dev).to(dt).permute(0, 2, 1, 3) for _ in range(3))
        out = torch.empty(B, S, H, D, device=dev, dtype=dt).permute(0, 2, 1, 3)
        lens = torch.randint(S // 2, S + 1, (B,), device=dev)
        lens[0] = S
        pad = (torch.arange(S, device=dev)[None, :] < lens[:, None]).to(torch.uint8)
        ones = torch.ones(B, S, dtype=torch.uint8, device=dev)
        fl = 4.0 * B * H * S * S * D / (2 if causal else 1)
        cases = {"unmasked": dict(), "km all ones": dict(key_mask=ones), "km padding": dict(key_mask=pad), "km padding, 8-wave": dict(key_mask=pad, _variant=44)}
        times = {n: [] for n in cases}
        for n, kw in cases.items():
            for _ in range(20):
                ops.fa3_forward(q, k, v, causal=causal, out=out, **kw)
        torch.cuda.synchronize()
        for r in range(7):
            for n, kw in cases.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    ops.fa3_forward(q, k, v, causal=causal, out=out, **kw)
                e1.record(); torch.cuda.synchronize()
                times[n].append(e0.elapsed_time(e1) / 20)
        for n in cases:
            med = statistics.median(times[n])
            print(f"B{B} H{H} S{S} {'causal' if causal else 'full'} {n}: {med:.4f} ms {fl / med / 1e9:.1f} TF (dense-equivalent flops)", flush=True)
sys.exit(1 if bad else 0)
