#!/usr/bin/env python3
"""One-off soak: random backward problems (D 64/128, causal, key lengths, element masks, grouped-query heads) vs autograd through a torch
fp32 reference on the GPU.  Error measure: max |err| / max |ref| per gradient (bf16 P/dS: <= 2e-2)."""
import os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photonic_flash_attention_amd import ops, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
worst = 0.0
for it in range(N):
    B, H = rnd.choice([(1, 1), (1, 2), (2, 2), (1, 4), (1, 6), (2, 8)])
    G = rnd.choice([x for x in (1, 1, 2, 3, 4) if H % x == 0])          # query heads per K/V head (ABI v7: summed inside the dK/dV kernel)
    D = rnd.choice([64, 128])
    Sq = rnd.choice([1, 17, 64, 65, 128, 200, 256, 257, 400, 512, 777, 1024])
    Sk = rnd.choice([1, 33, 64, 65, 127, 128, 129, 256, 300, 511, 512, 1000, 1024])
    causal = rnd.random() < 0.5
    dtype = rnd.choice(["bf16", "fp16"])
    mode = rnd.choice(["none", "none", "lens", "key", "mask"])
    q, k, v = (t.to("cuda:0").permute(0, 2, 1, 3) for t in synth.qkv(B, H, Sq, Sk, D, 70000 + it, dtype))
    k, v = k[:, ::G], v[:, ::G]                                         # H / G K/V heads (strided views: the kernels take any head stride)
    g = torch.from_numpy(synth.normal_f32((B, Sq, H, D), 80000 + it)).to("cuda:0", q.dtype).permute(0, 2, 1, 3)
    keep = torch.ones(B, H, Sq, Sk, dtype=torch.bool, device="cuda:0")
    kw = {}
    gen = torch.Generator().manual_seed(it)
    if mode == "lens":
        lens = [rnd.randint(1, Sk) for _ in range(B)]
        kw["seqlens_k"] = lens
        for b_, n_ in enumerate(lens): keep[b_, :, :, n_:] = False
    elif mode == "key":
        m = torch.rand(B, Sk, generator=gen) < 0.8
        m[:, 0] = True
        kw["key_mask"] = m.to("cuda:0")
        keep &= m.to("cuda:0")[:, None, None, :]
    elif mode == "mask":
        m = torch.rand(B, 1, Sq, Sk, generator=gen) < 0.7
        m[..., 0] = True
        kw["mask"] = m.to("cuda:0")
        keep &= m.to("cuda:0")
    if causal: keep &= torch.tril(torch.ones(Sq, Sk, dtype=torch.bool, device="cuda:0"))
    out, lse = ops.fa3_forward(q, k, v, causal=causal, return_lse=True, **kw)
    dq, dk, dv = ops.fa3_backward(q, k, v, out, g, lse, causal=causal, grad_dtype=torch.float32, **kw)
    qf, kf, vf = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    ke, ve = kf.repeat_interleave(G, dim=1), vf.repeat_interleave(G, dim=1)      # (autograd sums the group's gradients into kf / vf)
    s = (qf @ ke.transpose(-1, -2)) * D ** -0.5
    p = torch.nan_to_num(torch.softmax(s.masked_fill(~keep, float("-inf")), dim=-1), nan=0.0)
    (p @ ve).backward(g.float())
    tag = (it, B, H, G, Sq, Sk, D, causal, dtype, mode)
    for name, got, ref in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        assert bool(torch.isfinite(got).all()), (tag, name)
        scale = float(ref.abs().max())
        err = float((got - ref).abs().max())
        assert err <= 2e-2 * scale + 1e-4, (tag, name, err, scale)      # (floor: a gradient that is exactly 0 -- one key: dS = P (dP - delta) -- comes out as the rounding of two summation orders)
        if scale > 1e-3: worst = max(worst, err / scale)
print(f"{N} problems ok; worst relative-to-max gradient error {worst:.2e}")
