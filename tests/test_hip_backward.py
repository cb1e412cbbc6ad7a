"""GPU tests of the backward kernels (pfa_fa3_bwd) against autograd through the
CPU oracle (fp32; the reference itself gets its gradients from autograd through the same eager math,
tests/unit/test_flash_attention_3.py:137-160 of the reference only require that gradients exist).

Tolerance: P and dS enter the MFMAs as bf16/fp16 (relative 2^-9 / 2^-12 per element), so gradients are compared
with  |err| <= tol_rel * max|ref|  (tol_rel = 1.5e-2 bf16, 4e-3 fp16) and a cosine similarity >= 0.9995."""

from __future__ import annotations

import pytest
import torch

from photonic_flash_attention_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [
    # B, H, Sq, Sk, D, causal, seqlens
    (1, 2, 64, 64, 64, False, None),
    (2, 2, 128, 128, 128, True, None),
    (1, 3, 200, 333, 64, False, None),
    (2, 2, 384, 384, 128, True, None),
    (1, 2, 300, 300, 128, False, [257]),
    (1, 4, 1024, 1024, 128, True, None),
    (2, 1, 97, 513, 64, True, [513, 40]),
    # head dims without a kernel instantiation run zero-padded to 64 / 128 (ops._padded_head_dim)
    (1, 2, 130, 200, 32, True, None),
    (2, 2, 256, 256, 96, False, [256, 77]),
]


def _ref_grads(q, k, v, dout, causal, lens):
    from oracle import fa3_oracle as orc
    qf, kf, vf = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    out = orc.attention_bshd(qf, kf, vf, causal=causal, seqlens_k=lens)
    out.backward(dout.float())
    return out.detach(), qf.grad, kf.grad, vf.grad


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("case", CASES)
def test_backward_matches_autograd_of_oracle(case, dtype):
    from photonic_flash_attention_amd import ops
    B, H, Sq, Sk, D, causal, lens = case
    q, k, v = synth.qkv(B, H, Sq, Sk, D, 500 + Sq, dtype)
    dout = torch.from_numpy(synth.normal_f32((B, Sq, H, D), 900 + Sk)).to(q.dtype)
    _, rq, rk, rv = _ref_grads(q, k, v, dout, causal, lens)
    qd, kd, vd, gd = (t.to(DEV).permute(0, 2, 1, 3) for t in (q, k, v, dout))
    out, lse = ops.fa3_forward(qd, kd, vd, causal=causal, seqlens_k=lens, return_lse=True)
    dq, dk, dv = ops.fa3_backward(qd, kd, vd, out, gd, lse, causal=causal, seqlens_k=lens, grad_dtype=torch.float32)
    torch.cuda.synchronize()
    tol = 1.5e-2 if dtype == "bf16" else 4e-3
    for name, got, ref in (("dq", dq, rq), ("dk", dk, rk), ("dv", dv, rv)):
        got = got.permute(0, 2, 1, 3).cpu()
        assert bool(torch.isfinite(got).all()), name
        err = float((got - ref).abs().max())
        scale = float(ref.abs().max())
        cos = float(torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0))
        assert err <= tol * scale + 1e-6, f"{name} {case} {dtype}: err {err:.3e} vs max {scale:.3e}"
        assert cos >= 0.9995, f"{name} cosine {cos}"


def test_module_is_differentiable_like_the_reference():
    """tests/unit/test_flash_attention_3.py:137-160 of the reference: gradients exist, are non-zero, have the
    parameter's shape -- here through FlashAttention3 (projections by autograd, core by the HIP kernels)."""
    from photonic_flash_attention_amd import FlashAttention3
    from oracle import fa3_oracle as orc
    E, H = 256, 4
    m = FlashAttention3(E, H, dtype=torch.bfloat16).to(DEV).train()
    x = torch.from_numpy(synth.normal_f32((2, 192, E), 77)).to(DEV, torch.bfloat16).requires_grad_(True)
    out, w = m(x, is_causal=True)
    assert w is None and out.requires_grad
    out.float().square().mean().backward()
    assert x.grad is not None and x.grad.shape == x.shape and float(x.grad.float().abs().sum()) > 0
    for n_, p_ in m.named_parameters():
        assert p_.grad is not None and p_.grad.shape == p_.shape and bool(torch.isfinite(p_.grad.float()).all()), n_
    # numbers: same module in fp32 on the CPU with the oracle core under autograd
    sd = {k_: v_.detach().float().cpu().clone().requires_grad_(True) for k_, v_ in m.state_dict().items()}
    xc = x.detach().float().cpu().requires_grad_(True)
    ref = orc.module_forward(sd, H, xc, mask=orc.causal_mask(192, 192))
    ref.square().mean().backward()
    gx = x.grad.float().cpu()
    assert float((gx - xc.grad).abs().max()) <= 0.05 * float(xc.grad.abs().max()) + 1e-6
    gw = m.qkv_proj.weight.grad.float().cpu()
    cos = float(torch.nn.functional.cosine_similarity(gw.flatten(), sd["qkv_proj.weight"].grad.flatten(), dim=0))
    assert cos >= 0.995


def test_backward_is_bitwise_reproducible():
    from photonic_flash_attention_amd import ops
    q, k, v = (t.to(DEV).permute(0, 2, 1, 3) for t in synth.qkv(1, 4, 640, 640, 128, 6, "bf16"))
    g = torch.from_numpy(synth.normal_f32((1, 640, 4, 128), 7)).to(DEV, torch.bfloat16).permute(0, 2, 1, 3)
    out, lse = ops.fa3_forward(q, k, v, causal=True, return_lse=True)
    a = ops.fa3_backward(q, k, v, out, g, lse, causal=True)
    b = ops.fa3_backward(q, k, v, out, g, lse, causal=True)
    assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_backward_matches_reference_autograd_golden():
    """dq/dk/dv of the REAL reference (autograd through its eager core, tests/golden g8_*) vs the HIP backward."""
    from conftest import golden_inputs, golden_names, load_golden
    from photonic_flash_attention_amd import ops
    names = golden_names("grad")
    assert names
    for name in names:
        meta, arr = load_golden(name)
        q, k, v = golden_inputs(meta)
        dout = torch.from_numpy(synth.normal_f32((meta["B"], meta["Sq"], meta["H"], meta["D"]), meta["dout_seed"])).to(torch.bfloat16)
        qd, kd, vd, gd = (t.to(DEV).permute(0, 2, 1, 3) for t in (q, k, v, dout))
        kw = {}
        if meta.get("kv_valid") is not None:      # the reference was given a 4-D mask: feed the same one through `mask=`
            Sq, Sk = meta["Sq"], meta["Sk"]
            kw["mask"] = (torch.arange(Sk) < meta["kv_valid"]).view(1, 1, 1, Sk).expand(meta["B"], 1, Sq, Sk).to(DEV)
        out, lse = ops.fa3_forward(qd, kd, vd, causal=meta["causal"], return_lse=True, **kw)
        grads = ops.fa3_backward(qd, kd, vd, out, gd, lse, causal=meta["causal"], grad_dtype=torch.float32, **kw)
        for got, key in zip(grads, ("dq", "dk", "dv")):
            ref = torch.from_numpy(arr[key])
            got = got.permute(0, 2, 1, 3).cpu()
            err, scale = float((got - ref).abs().max()), float(ref.abs().max())
            print(f"{name} {key}: max-abs {err:.3e} (max |ref| {scale:.3e})")
            assert err <= 1.5e-2 * scale, (name, key, err, scale)


# ---- masks in the backward ------------------------------------------------------------------------------------------
def _torch_ref_masked(q, k, v, dout, keep, scale):
    """fp64 torch reference on [B,S,H,D] operands with a boolean keep-mask broadcastable to [B,H,Sq,Sk]; rows with no
    key kept give zero output and zero gradient (the kernel's documented convention for fully masked rows)."""
    qf, kf, vf = (t.double().permute(0, 2, 1, 3).clone().requires_grad_(True) for t in (q, k, v))
    s = (qf @ kf.transpose(-1, -2)) * scale
    s = s.masked_fill(~keep, float("-inf"))
    p = torch.nan_to_num(torch.softmax(s, dim=-1), nan=0.0)
    out = p @ vf
    out.backward(dout.double().permute(0, 2, 1, 3))
    return [g.permute(0, 2, 1, 3).float() for g in (qf.grad, kf.grad, vf.grad)]


MASK_CASES = [
    # B, H, Sq, Sk, D, causal, kind
    (2, 2, 128, 128, 64, False, "key"),
    (1, 2, 200, 333, 128, False, "key"),          # Sk % 4 != 0: the element-mask kernels
    (2, 2, 300, 640, 128, True, "key"),           # key-only masks on the unmasked kernels (Sk % 4 == 0), causal, ragged Sq
    (3, 1, 130, 1028, 64, False, "key"),
    (2, 3, 192, 192, 128, False, "b1qk"),
    (1, 2, 320, 320, 64, True, "bhqk"),
    (1, 2, 256, 256, 128, False, "dead_rows"),
]


@pytest.mark.parametrize("case", MASK_CASES)
def test_backward_with_masks(case):
    from photonic_flash_attention_amd import ops
    B, H, Sq, Sk, D, causal, kind = case
    q, k, v = synth.qkv(B, H, Sq, Sk, D, 1300 + Sq, "bf16")
    dout = torch.from_numpy(synth.normal_f32((B, Sq, H, D), 1400 + Sk)).to(q.dtype)
    g = torch.Generator().manual_seed(Sq + Sk)
    if kind == "key":
        m = (torch.rand(B, Sk, generator=g) < 0.8)
        m[:, 0] = True
        keep = m[:, None, None, :]
    elif kind == "b1qk":
        m = (torch.rand(B, 1, Sq, Sk, generator=g) < 0.7)
        m[..., 0] = True
        keep = m
    elif kind == "bhqk":
        m = (torch.rand(B, H, Sq, Sk, generator=g) < 0.7)
        m[..., 0] = True
        keep = m
    else:
        m = (torch.rand(B, 1, Sq, Sk, generator=g) < 0.7)
        m[..., 0] = True
        m[:, :, 5] = False                    # fully masked rows
        m[:, :, 100:133] = False
        keep = m
    full_keep = keep.expand(B, H, Sq, Sk).clone() if keep.shape[1] == H else keep.expand(B, 1, Sq, Sk).clone()
    if causal:
        full_keep = full_keep & torch.tril(torch.ones(Sq, Sk, dtype=torch.bool))
    ref = _torch_ref_masked(q, k, v, dout, full_keep, D ** -0.5)
    qd, kd, vd, gd = (t.to(DEV).permute(0, 2, 1, 3) for t in (q, k, v, dout))
    kw = dict(key_mask=m.to(DEV)) if kind == "key" else dict(mask=m.to(DEV))
    out, lse = ops.fa3_forward(qd, kd, vd, causal=causal, return_lse=True, **kw)
    grads = ops.fa3_backward(qd, kd, vd, out, gd, lse, causal=causal, grad_dtype=torch.float32, **kw)
    torch.cuda.synchronize()
    for name, got, r in zip(("dq", "dk", "dv"), grads, ref):
        got = got.permute(0, 2, 1, 3).cpu()
        assert bool(torch.isfinite(got).all()), name
        err, scale = float((got - r).abs().max()), float(r.abs().max())
        assert err <= 1.5e-2 * scale + 1e-6, f"{name} {case}: err {err:.3e} vs max {scale:.3e}"
    if kind == "dead_rows":
        dq = grads[0].permute(0, 2, 1, 3).cpu()
        assert float(dq[:, 5].abs().max()) == 0.0 and float(dq[:, 100:133].abs().max()) == 0.0


def test_masked_module_trains_and_fp32_module_is_differentiable():
    from photonic_flash_attention_amd import FlashAttention3
    from oracle import fa3_oracle as orc
    E, H, S = 128, 2, 96
    m = FlashAttention3(E, H, dtype=torch.bfloat16).to(DEV).train()
    x = torch.from_numpy(synth.normal_f32((2, S, E), 31)).to(DEV, torch.bfloat16).requires_grad_(True)
    am = torch.ones(2, S)
    am[0, 70:] = 0
    am[1, 33:50] = 0
    out, _ = m(x, attention_mask=am.to(DEV))
    out.float().square().mean().backward()
    sd = {k_: v_.detach().float().cpu().clone().requires_grad_(True) for k_, v_ in m.state_dict().items()}
    xc = x.detach().float().cpu().requires_grad_(True)
    ref = orc.module_forward(sd, H, xc, mask=am)
    ref.square().mean().backward()
    assert float((out.detach().float().cpu() - ref.detach()).abs().max()) <= 2e-2
    gx = x.grad.float().cpu()
    assert float((gx - xc.grad).abs().max()) <= 0.05 * float(xc.grad.abs().max()) + 1e-6
    # fp32 module: the exact fp32 kernels in both directions -- the forward under autograd is the forward under no_grad
    m32 = FlashAttention3(E, H).to(DEV).eval()
    x32 = torch.from_numpy(synth.normal_f32((2, S, E), 32)).to(DEV).requires_grad_(True)
    y = m32(x32, is_causal=True)[0]
    with torch.no_grad():
        y0 = m32(x32, is_causal=True)[0]
    assert y.dtype == torch.float32 and torch.equal(y.detach(), y0)
    y.square().mean().backward()
    assert x32.grad is not None and bool(torch.isfinite(x32.grad).all()) and float(x32.grad.abs().sum()) > 0
    for n_, p_ in m32.named_parameters():
        assert p_.grad is not None and bool(torch.isfinite(p_.grad).all()), n_


def test_backward_at_baseline_headline_shape():
    """C3 (B4 S4096 H16 D128 bf16 causal) backward at full size: two (batch, head) slices against a plain torch fp32
    reference of the same op evaluated on the GPU (the CPU oracle needs minutes here), all gradients finite."""
    from photonic_flash_attention_amd import ops
    B, H, S, D = 4, 16, 4096, 128
    q, k, v = (t.to(DEV).permute(0, 2, 1, 3) for t in synth.qkv(B, H, S, S, D, 2003, "bf16"))
    g = torch.from_numpy(synth.normal_f32((B, S, H, D), 2103)).to(DEV, torch.bfloat16).permute(0, 2, 1, 3)
    out, lse = ops.fa3_forward(q, k, v, causal=True, return_lse=True)
    dq, dk, dv = ops.fa3_backward(q, k, v, out, g, lse, causal=True, grad_dtype=torch.float32)
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(t).all()) for t in (dq, dk, dv))
    keep = torch.tril(torch.ones(S, S, dtype=torch.bool, device=DEV))
    for (b, h) in ((0, 0), (3, 15)):
        qf, kf, vf = (t[b, h].float().clone().requires_grad_(True) for t in (q, k, v))
        s = (qf @ kf.T) * D ** -0.5
        p_ = torch.softmax(s.masked_fill(~keep, float("-inf")), dim=-1)
        (p_ @ vf).backward(g[b, h].float())
        for name, got, ref in (("dq", dq[b, h], qf.grad), ("dk", dk[b, h], kf.grad), ("dv", dv[b, h], vf.grad)):
            err, scale = float((got - ref).abs().max()), float(ref.abs().max())
            assert err <= 1.5e-2 * scale, (name, b, h, err, scale)


# ---- fp32 kernels: exact gradients (fp32 modules) and attention dropout of the dense branch -------------------------------------------
@pytest.mark.parametrize("case", [(1, 2, 64, 64, 64, False, None), (2, 2, 130, 200, 128, True, None), (1, 3, 200, 333, 64, False, [257]),
                                  (2, 1, 97, 513, 64, True, [513, 40]), (1, 2, 256, 256, 40, False, None)])
def test_fp32_backward_matches_autograd_of_oracle(case):
    """fp32 operands: pfa_fa3_bwd runs the fp32 kernels -- the oracle's autograd gradients to fp32 rounding, not to 1e-2."""
    from photonic_flash_attention_amd import ops
    B, H, Sq, Sk, D, causal, lens = case
    q = torch.from_numpy(synth.normal_f32((B, Sq, H, D), 11 + Sq))
    k = torch.from_numpy(synth.normal_f32((B, Sk, H, D), 12 + Sk))
    v = torch.from_numpy(synth.normal_f32((B, Sk, H, D), 13 + D))
    dout = torch.from_numpy(synth.normal_f32((B, Sq, H, D), 14))
    _, rq, rk, rv = _ref_grads(q, k, v, dout, causal, lens)
    qd, kd, vd, gd = (t.to(DEV).permute(0, 2, 1, 3) for t in (q, k, v, dout))
    out, lse = ops.fa3_forward(qd, kd, vd, causal=causal, seqlens_k=lens, return_lse=True)
    dq, dk, dv = ops.fa3_backward(qd, kd, vd, out, gd, lse, causal=causal, seqlens_k=lens)
    torch.cuda.synchronize()
    for name, got, ref in (("dq", dq, rq), ("dk", dk, rk), ("dv", dv, rv)):
        got = got.permute(0, 2, 1, 3).cpu()
        assert got.dtype == torch.float32 and bool(torch.isfinite(got).all()), name
        assert float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), (name, case)


def test_dense_branch_dropout_statistics_and_gradients():
    """Attention dropout of the reference's dense branch (flash_attention_3.py:174-175), which cannot be matched stream for stream:
    (1) with the SAME keep-mask the kernels equal the oracle's dropout(softmax(s)) v and its autograd gradients to fp32 rounding
        -- so the forward applies and the backward replays exactly the mask it is given;
    (2) the mask the autograd Function draws keeps 1 - p of the weights, scaled by 1 / (1 - p): E[out] is the eval output;
    (3) module level: train != eval, eval deterministic (the reference's test_flash_attention_3.py:162-191), gradients finite."""
    from photonic_flash_attention_amd import FlashAttention3, ops
    B, H, S, D, p = 2, 3, 160, 64, 0.25
    q = torch.from_numpy(synth.normal_f32((B, S, H, D), 61))
    k = torch.from_numpy(synth.normal_f32((B, S, H, D), 62))
    v = torch.from_numpy(synth.normal_f32((B, S, H, D), 63))
    dout = torch.from_numpy(synth.normal_f32((B, S, H, D), 64))
    g = torch.Generator().manual_seed(5)
    keep = torch.rand((B, H, S, S), generator=g) >= p
    # (1) same mask on both sides
    qf, kf, vf = (t.clone().permute(0, 2, 1, 3).requires_grad_(True) for t in (q, k, v))          # [B,H,S,D]
    sc = (qf @ kf.transpose(-1, -2)) * D ** -0.5
    causal_keep = torch.tril(torch.ones(S, S, dtype=torch.bool))
    w = torch.softmax(sc.masked_fill(~causal_keep, float("-inf")), dim=-1)
    ref = (w * keep / (1 - p)) @ vf
    ref.backward(dout.permute(0, 2, 1, 3))
    qd, kd, vd, gd = (t.to(DEV).permute(0, 2, 1, 3) for t in (q, k, v, dout))
    kd_mask = keep.to(DEV).contiguous()
    out, lse = ops.fa3_forward(qd, kd, vd, causal=True, return_lse=True, drop_mask=kd_mask, drop_scale=1 / (1 - p))
    dq, dk, dv = ops.fa3_backward(qd, kd, vd, out, gd, lse, causal=True, drop_mask=kd_mask, drop_scale=1 / (1 - p))
    torch.cuda.synchronize()
    assert float((out.cpu() - ref.detach()).abs().max()) <= 2e-5
    for name, got, r in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        assert float((got.cpu() - r).abs().max()) <= 3e-5 * max(1.0, float(r.abs().max())), name
    # the LSE is the un-dropped normaliser
    assert float((lse.cpu() - torch.logsumexp(sc.detach().masked_fill(~causal_keep, float("-inf")), dim=-1)).abs().max()) <= 2e-5
    # (2) statistics of the Function's own mask: mean of many dropped outputs -> the eval output; a single one differs
    torch.manual_seed(0)
    ev = ops.fa3_forward(qd, kd, vd)[0]
    acc, acc2 = torch.zeros_like(ev, dtype=torch.float64), torch.zeros_like(ev, dtype=torch.float64)
    n = 200
    for _ in range(n):
        o_ = ops.fa3_attention_dropout(qd, kd, vd, p).double()
        acc += o_
        acc2 += o_ * o_
    one = ops.fa3_attention_dropout(qd, kd, vd, p)
    torch.cuda.synchronize()
    assert float((one - ev).abs().max()) > 1e-2
    mean = acc / n
    std = (acc2 / n - mean * mean).clamp(min=1e-12).sqrt()              # per element, over the draws
    z = (mean - ev.double()).abs() / (std / n ** 0.5)
    assert float(z.max()) <= 6.0 and float(z.mean()) <= 1.0, (float(z.max()), float(z.mean()))   # unbiased: |z| ~ half-normal, mean 0.8
    # (3) the module (bf16 parameters), dense branch: runs in training mode, differs from eval, eval is deterministic, trains
    m = FlashAttention3(128, 2, dropout=0.2, dtype=torch.bfloat16).to(DEV)
    x = torch.from_numpy(synth.normal_f32((2, 96, 128), 7)).to(DEV, torch.bfloat16).requires_grad_(True)
    m.train()
    y1, y2 = m(x)[0], m(x)[0]
    assert not torch.equal(y1, y2)                                  # two draws
    y1.float().square().mean().backward()
    assert x.grad is not None and bool(torch.isfinite(x.grad.float()).all()) and float(x.grad.float().abs().max()) > 0
    m.eval()
    with torch.no_grad():
        e1, e2 = m(x)[0], m(x)[0]
    assert torch.equal(e1, e2) and not torch.equal(e1, y1.detach())
