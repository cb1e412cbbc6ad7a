#!/usr/bin/env python3
"""seqlens_k on the persistent assembly forward (no causal mask): the ragged kernels' length word with a per-batch length and an item's
tile count cut to it; with a key mask as well, the *_km_* kernels with the tile count cut.  Against the 8-wave HIP kernel (44).
    timeout -k 10 300 python3 tools/p4_seqlens_check.py [--time] [--d 64]"""
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
bad = 0
for (B, H, Sq, Sk, lens, with_mask) in [(4, 8, 512, 1024, [1024, 0, 1, 300], False), (3, 4, 256, 384, [384, 64, 383], False), (2, 8, 1000, 1000, [1000, 517], False),
                                        (5, 3, 768, 2048, [2048, 65, 128, 129, 1999], False), (4, 8, 512, 1024, [1024, 0, 700, 300], True),
                                        (2, 16, 2048, 2048, [1100, 2048], True), (8, 4, 300, 3000, [3000, 1, 2, 63, 64, 191, 192, 2999], False)]:
    g = torch.Generator(device=dev).manual_seed(B * 1000 + H * 10 + Sq)
    q = torch.randn(B, Sq, H, D, device=dev, generator=g).to(dt).permute(0, 2, 1, 3)
    k, v = (torch.randn(B, Sk, H, D, device=dev, generator=g).to(dt).permute(0, 2, 1, 3) for _ in range(2))
    km = (torch.rand(B, Sk, generator=g, device=dev) < 0.8) if with_mask else None
    res = {}
    for var in (44, 45):
        for o32 in (False, True):
            o, lse = ops.fa3_forward(q, k, v, seqlens_k=lens, key_mask=km, return_lse=True, _variant=var, out_dtype=torch.float32 if o32 else None)
            torch.cuda.synchronize()
            res[var, o32] = (o.float().clone(), lse.clone())
    name = _capi.describe(ops.build_args(q, k, v, torch.empty_like(q), seqlens_k=lens, key_mask=km, variant=45)[0])[0]
    ok = ("_km_" in name) if with_mask else ("_kl_" in name)
    msg = []
    for o32 in (False, True):
        d_o = (res[45, o32][0] - res[44, o32][0]).abs().nan_to_num(1e9)
        l45, l44 = res[45, o32][1], res[44, o32][1]
        same_inf = bool((torch.isinf(l45) == torch.isinf(l44)).all())
        d_l = (l45 - l44).abs().nan_to_num(0.0)
        ok = ok and float(d_o.max()) <= (3e-5 if o32 else 2e-2) and float(d_l.max()) <= 1e-4 and same_inf
        msg.append(f"{'o32' if o32 else 'o16'} max|dO| {float(d_o.max()):.2e} max|dLSE| {float(d_l.max()):.2e}")
    print(f"B{B} H{H} Sq{Sq} Sk{Sk} lens {lens} {'+mask' if with_mask else ''} {name}: {' | '.join(msg)}  {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print("FAILED" if bad else "ALL OK", flush=True)
if a.time and not bad:
    B, H, S = 16, 16, 2048
    q, k, v = (torch.randn(B, S, H, D, device=dev).to(dt).permute(0, 2, 1, 3) for _ in range(3))
    out = torch.empty(B, S, H, D, device=dev, dtype=dt).permute(0, 2, 1, 3)
    lens = torch.randint(S // 2, S + 1, (B,), device=dev)
    lens[0] = S
    pad = (torch.arange(S, device=dev)[None, :] < lens[:, None])
    li = lens.to(torch.int32)
    fl = 4.0 * B * H * S * S * D
    cases = {"unmasked": dict(), "seqlens": dict(seqlens_k=li), "seqlens, 8-wave": dict(seqlens_k=li, _variant=44), "key mask": dict(key_mask=pad),
             "key mask + seqlens": dict(key_mask=pad, seqlens_k=li)}
    times = {n: [] for n in cases}
    for n, kw in cases.items():
        for _ in range(20):
            ops.fa3_forward(q, k, v, out=out, **kw)
    torch.cuda.synchronize()
    for r in range(7):
        for n, kw in cases.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.fa3_forward(q, k, v, out=out, **kw)
            e1.record(); torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) / 20)
    for n in cases:
        med = statistics.median(times[n])
        print(f"B{B} H{H} S{S} lengths in [S/2, S] {n}: {med:.4f} ms {fl / med / 1e9:.1f} TF dense-equivalent", flush=True)
sys.exit(1 if bad else 0)
