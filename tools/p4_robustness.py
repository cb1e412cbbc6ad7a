#!/usr/bin/env python3
"""Persistent kernels in awkward settings: extreme shapes through the default dispatch, a side stream, HIP-graph capture and replay
(plain, key-masked with the derived seqlens_k, ragged), four host threads on four streams.  Checked against the 8-wave kernel / torch."""
import os, sys, threading, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photonic_flash_attention_amd import ops, _capi
dev = torch.device("cuda:0")


def mk(B, H, Sq, Sk, D, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    q = torch.randn(B, Sq, H, D, device=dev, generator=g).to(torch.bfloat16).permute(0, 2, 1, 3)
    k, v = (torch.randn(B, Sk, H, D, device=dev, generator=g).to(torch.bfloat16).permute(0, 2, 1, 3) for _ in range(2))
    return q, k, v


def check(tag, q, k, v, **kw):
    o0, l0 = ops.fa3_forward(q, k, v, return_lse=True, **kw)
    o1, l1 = ops.fa3_forward(q, k, v, return_lse=True, _variant=44, **kw)
    torch.cuda.synchronize()
    name = _capi.describe(ops.build_args(q, k, v, o0, **kw)[0])[0]
    d = float((o0.float() - o1.float()).abs().max()); dl = float((l0 - l1).abs().nan_to_num(0.0).max())
    print(f"{tag}: {name}  |dO| {d:.1e} |dLSE| {dl:.1e}", flush=True)
    assert "p4" in name and bool(torch.isfinite(o0.float()).all()) and d <= 2e-2 and dl <= 1e-4, tag


# 1. extreme shapes
check("S65536 causal one head", *mk(1, 1, 65536, 65536, 128), causal=True)
check("S32768 two heads", *mk(1, 2, 32768, 32768, 128))
check("2048 heads x 512 causal", *mk(64, 32, 512, 512, 128), causal=True)
check("S65535 causal D64", *mk(1, 1, 65535, 65535, 64), causal=True)
check("Sq 128 x Sk 100001", *mk(1, 8, 128, 100001, 128))
check("4096 heads x 256", *mk(128, 32, 256, 256, 128))

# 2. side stream + 3. graph capture / replay
q, k, v = mk(4, 8, 1024, 1024, 128, 3)
lens = torch.tensor([1024, 300, 77, 640], device=dev)
km = torch.arange(1024, device=dev)[None, :] < lens[:, None]
qr, kr, vr = mk(2, 8, 1000, 1000, 128, 4)
ref = {"plain": ops.fa3_forward(q, k, v, causal=True)[0].clone(), "km": ops.fa3_forward(q, k, v, key_mask=km)[0].clone(),
       "ragged": ops.fa3_forward(qr, kr, vr, causal=True)[0].clone()}
torch.cuda.synchronize()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    o = ops.fa3_forward(q, k, v, causal=True)[0]
s.synchronize()
assert torch.equal(o, ref["plain"]), "side stream"
outs = {n: torch.empty_like(t) for n, t in ref.items()}
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        ops.fa3_forward(q, k, v, causal=True, out=outs["plain"])
        ops.fa3_forward(q, k, v, key_mask=km, out=outs["km"])
        ops.fa3_forward(qr, kr, vr, causal=True, out=outs["ragged"])
for rep in range(3):
    for t in outs.values():
        t.fill_(float("nan"))
    g.replay()
    torch.cuda.synchronize()
    for n in ref:
        assert torch.equal(outs[n], ref[n]), ("graph replay", n, rep)
print("side stream, graph capture + 3 replays: bit-identical", flush=True)

# 4. threads
errs = []


def worker(i):
    try:
        st = torch.cuda.Stream()
        qq, kk, vv = mk(2, 8, 512 + 256 * i, 512 + 256 * i, 128, 10 + i)
        with torch.cuda.stream(st):
            a = ops.fa3_forward(qq, kk, vv, causal=True)[0]
            for _ in range(50):
                b = ops.fa3_forward(qq, kk, vv, causal=True)[0]
            st.synchronize()
            if not torch.equal(a, b):
                errs.append(i)
    except Exception as e:      # noqa: BLE001
        errs.append((i, repr(e)))


th = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
[t.start() for t in th]; [t.join() for t in th]
assert not errs, errs
print("4 threads x 4 streams x 50 launches: bit-identical per thread\nok")
