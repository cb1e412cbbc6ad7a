"""GPU parity tests: the HIP kernel, called through the C ABI, against
(1) outputs of the real reference (tests/golden) and (2) the CPU oracle on seeded inputs.

Tolerances (DESIGN.md "numerics"):
  * PARITY variant (fp32 store, P as bf16 hi+lo):   max-abs <= 1e-3   (north_star bar)
  * FAST variant  (16-bit store, P as ONE 16-bit operand): the error is two roundings and nothing else, and the bound says so
    per element:   |err| <= eps*|ref|  +  C_P * eps * max|v| * ||p_row||_2  (+ 2e-6)
    eps = 2^-9 (bf16) / 2^-11 (fp16) = half an ulp; first term = the rounding of the stored output (absent with an fp32 store);
    second term = the rounding of P before the PV product: one independent rounding of relative size <= eps per visible key,
    weighted by the softmax weights p_j, so its standard deviation is eps/sqrt(3) * sqrt(sum_j p_j^2 v_j^2) <= eps/sqrt(3) *
    max|v| * ||p||_2; C_P = 3 is the 5-sigma tail over ~1e7 elements.  ||p||_2 = 1/sqrt(n) for uniform weights over n keys
    (~ sqrt(e/n) on N(0,1) scores) and up to ~0.7 for a row dominated by one key: the first rows of a causal problem (1..10
    keys) sit at ~5e-3, rows with thousands of keys at ~3e-4.  ||p||_2 is computed here in fp32 from q and k.
"""

from __future__ import annotations

import numpy as np
import pytest
import torch

from conftest import golden_inputs, golden_names, load_golden

pytestmark = pytest.mark.gpu

PARITY_TOL = 1e-3


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _run(q, k, v, **kw):
    from photonic_flash_attention_amd import ops
    dev = _dev()
    o, lse = ops.fa3_forward_bshd(q.to(dev), k.to(dev), v.to(dev), **kw)
    torch.cuda.synchronize()
    return o.float().cpu(), (None if lse is None else lse.cpu())


C_P = 3.0


def _pnorm(q, k, causal, lens=None, rows=None, heads=None):
    """||p_row||_2 of the exact softmax, fp32 on the CPU: [B, Sq, H, 1] for [B,S,H,D] operands (K heads repeated for grouped-query
    problems); `rows` / `heads` = [(b, h), ...] restrict it to sampled rows of a few heads -> [len(heads), len(rows), 1]."""
    B, Sq, H, D = q.shape
    Sk, g = k.shape[1], H // k.shape[2]
    qf, kf = q.float(), k.float()
    if heads is not None:
        qf = torch.stack([qf[b, :, h] for b, h in heads])[:, None]            # [n, 1, Sq, D]
        kf = torch.stack([kf[b, :, h // g] for b, h in heads])[:, None]
        qf, kf = qf.permute(0, 2, 1, 3), kf.permute(0, 2, 1, 3)               # [n, Sq, 1, D]
        g, lens = 1, None
    if g > 1:
        kf = kf.repeat_interleave(g, dim=2)
    qi = torch.arange(Sq) if rows is None else rows
    s = torch.einsum("bqhd,bkhd->bhqk", qf[:, qi], kf) * D ** -0.5
    kj = torch.arange(Sk)
    if causal:
        s = s.masked_fill(kj[None, :] > qi[:, None], float("-inf"))
    if lens is not None:
        s = s.masked_fill(kj[None, None, None, :] >= torch.tensor(lens)[:, None, None, None], float("-inf"))
    p = torch.softmax(s, dim=-1).nan_to_num(0.0)
    pn = p.square().sum(-1).sqrt().permute(0, 2, 1)[..., None]                # [B, Sq', H, 1]
    return pn[:, :, 0] if heads is not None else pn


def _fast_bound(ref, pnorm, vmax, dtype="bf16", store16=True):
    eps = 2.0 ** (-9 if dtype == "bf16" else -11)
    return (ref.abs() * eps * 1.01 if store16 else 0.0) + C_P * eps * vmax * pnorm + 2e-6


def _fast_ok(out, ref, pnorm, vmax, dtype="bf16", store16=True, tag=""):
    err, bound = (out - ref).abs(), _fast_bound(ref, pnorm, vmax, dtype, store16)
    worst = float((err / bound).max())
    if tag:
        print(f"{tag}: max-abs {float(err.max()):.3e}, worst error/bound {worst:.2f}")
    return worst <= 1.0


def test_library_loaded_and_device_supported(hip):
    lib = hip.load()
    assert lib.pfa_abi_version() == hip.PFA_ABI_VERSION
    assert lib.pfa_device_supported(0) == 1


@pytest.mark.parametrize("name", golden_names("full"))
def test_golden_full(name):
    meta, arr = load_golden(name)
    if meta["D"] not in (64, 128):
        pytest.skip("head dim outside kernel domain")
    q, k, v = golden_inputs(meta)
    ref = torch.from_numpy(arr["out"])
    lens = None if meta["kv_valid"] is None else [meta["kv_valid"]] * meta["B"]
    out, _ = _run(q, k, v, causal=meta["causal"], seqlens_k=lens, out_dtype=torch.float32)
    err = float((out - ref).abs().max())
    print(f"{name}: parity variant max-abs {err:.3e}")
    assert err <= PARITY_TOL
    out16, _ = _run(q, k, v, causal=meta["causal"], seqlens_k=lens)
    pn = _pnorm(q, k, meta["causal"], lens)
    assert _fast_ok(out16, ref, pn, float(v.float().abs().max()), meta["dtype"], tag=f"{name}: fast variant")


@pytest.mark.parametrize("name", golden_names("sampled"))
def test_golden_sampled_baseline_shapes(name):
    """BASELINE.json configs C2..C5 at full size: sampled rows + whole-head sums of the reference."""
    meta, arr = load_golden(name)
    q, k, v = golden_inputs(meta)
    rows = torch.from_numpy(arr["rows"])
    out, _ = _run(q, k, v, causal=meta["causal"], out_dtype=torch.float32)
    out16, _ = _run(q, k, v, causal=meta["causal"])
    pns = _pnorm(q, k, meta["causal"], rows=rows, heads=[tuple(x) for x in meta["heads"]])
    for i, (b, h) in enumerate(meta["heads"]):
        ref = torch.from_numpy(arr["out"][i])
        err = float((out[b, rows, h] - ref).abs().max())
        print(f"{name} head {(b, h)}: parity max-abs {err:.3e}; fast {float((out16[b, rows, h] - ref).abs().max()):.3e}")
        assert err <= PARITY_TOL
        assert _fast_ok(out16[b, rows, h], ref, pns[i], float(v.float().abs().max()), meta["dtype"])
        hs = float(out[b, :, h].double().sum())
        assert abs(hs - arr["head_sum"][i]) <= 2e-5 * arr["head_abs_sum"][i] + 1e-2


CASES = [
    # B, H, Sq, Sk, D, causal, seqlens
    (1, 1, 1, 1, 64, False, None),
    (1, 2, 33, 33, 128, True, None),
    (2, 3, 255, 257, 64, False, None),
    (2, 2, 256, 256, 128, True, None),
    (1, 2, 300, 1000, 128, False, [777]),
    (3, 2, 513, 513, 64, True, [513, 100, 1]),
    (1, 4, 1024, 1024, 128, True, None),
    (2, 1, 700, 64, 128, False, [64, 63]),
    (1, 2, 64, 2048, 64, False, None),
]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("case", CASES)
def test_vs_oracle(case, dtype):
    from oracle import fa3_oracle as orc
    from photonic_flash_attention_amd import synth
    B, H, Sq, Sk, D, causal, lens = case
    q, k, v = synth.qkv(B, H, Sq, Sk, D, 31 + Sq + Sk, dtype)
    ref = orc.attention_bshd(q, k, v, causal=causal, seqlens_k=lens)
    out, lse = _run(q, k, v, causal=causal, seqlens_k=lens, out_dtype=torch.float32, return_lse=True)
    err = float((out - ref).abs().max())
    assert err <= PARITY_TOL, f"{case} {dtype}: {err}"
    ref_lse = orc.lse_bshd(q, k, causal=causal, seqlens_k=lens)
    assert float((lse - ref_lse).abs().max()) <= 2e-3
    out16, _ = _run(q, k, v, causal=causal, seqlens_k=lens)
    assert _fast_ok(out16, ref, _pnorm(q, k, causal, lens), float(v.float().abs().max()), dtype)


@pytest.mark.parametrize("name", ["g6_c3", "g6_c5"])
def test_single_p_error_is_p_rounding_and_store_rounding_only(name):
    """The benched schedule's 7e-3 on a causal problem, taken apart against the REFERENCE's own outputs (sampled rows of the
    C3 / C5 goldens): with an fp32 store and ONE 16-bit P operand the only rounding left is P's, and it obeys the per-row bound
    C_P * 2^-9 * max|v| * ||p_row||_2 on every row -- below 1e-3 wherever that bound is (rows with many comparable keys); the
    16-bit store adds its half ulp, 2^-9 |o|, and nothing else."""
    matches = [n for n in golden_names("sampled") if n.startswith(name)]
    assert matches, "sampled golden missing"
    meta, arr = load_golden(matches[0])
    q, k, v = golden_inputs(meta)
    rows = torch.from_numpy(arr["rows"])
    vmax = float(v.float().abs().max())
    o32, _ = _run(q, k, v, causal=meta["causal"], out_dtype=torch.float32, split_p=False)       # single P, fp32 store
    o16, _ = _run(q, k, v, causal=meta["causal"])                                                # the benched kernel
    pns = _pnorm(q, k, meta["causal"], rows=rows, heads=[tuple(x) for x in meta["heads"]])
    for i, (b, h) in enumerate(meta["heads"]):
        ref = torch.from_numpy(arr["out"][i])
        assert _fast_ok(o32[b, rows, h], ref, pns[i], vmax, store16=False, tag=f"{matches[0]} head {(b, h)} single P, fp32 store")
        assert _fast_ok(o16[b, rows, h], ref, pns[i], vmax, store16=True, tag=f"{matches[0]} head {(b, h)} benched kernel")
        late = (C_P * 2.0 ** -9 * vmax * pns[i][:, 0]) <= 1e-3      # rows whose P-rounding bound itself is below the north-star bar
        print(f"   rows with the bound <= 1e-3: {int(late.sum())} of {late.numel()} sampled (first at row {int(rows[late].min()) if bool(late.any()) else -1})")
        assert bool(late.any()) and float((o32[b, rows, h] - ref)[late].abs().max()) <= 1e-3
        # the store rounding alone: the 16-bit result is the fp32-store result rounded once
        assert float((o16[b, rows, h] - o32[b, rows, h]).abs().max()) <= float(ref.abs().max()) * 2.0 ** -8 + 1e-6


def test_strided_views_of_fused_qkv():
    """q,k,v as strided views of one [B,S,3E] buffer (flash_attention_3.py:88-99)."""
    from oracle import fa3_oracle as orc
    from photonic_flash_attention_amd import ops, synth
    B, S, H, D = 2, 384, 4, 64
    E = H * D
    qkv = torch.from_numpy(synth.normal_f32((B, S, 3 * E), 5)).to(torch.bfloat16)
    dev = _dev()
    g = qkv.to(dev)
    qv, kv, vv = (t.view(B, S, H, D).transpose(1, 2) for t in g.chunk(3, dim=-1))
    o, _ = ops.fa3_forward(qv, kv, vv, out_dtype=torch.float32)
    assert o.transpose(1, 2).is_contiguous()
    qc, kc, vc = (t.view(B, S, H, D) for t in qkv.chunk(3, dim=-1))
    ref = orc.attention_bshd(qc, kc, vc)
    assert float((o.transpose(1, 2).cpu() - ref).abs().max()) <= PARITY_TOL


def test_key_mask_2d():
    from oracle import fa3_oracle as orc
    from photonic_flash_attention_amd import synth
    B, H, S, D = 2, 2, 320, 64
    q, k, v = synth.qkv(B, H, S, S, D, 91, "bf16")
    g = torch.Generator().manual_seed(3)
    km = torch.rand(B, S, generator=g) > 0.3
    km[:, 0] = True
    mask4 = km.view(B, 1, 1, S).expand(B, 1, S, S)
    ref = orc.flash_attention_forward(q.float().permute(0, 2, 1, 3), k.float().permute(0, 2, 1, 3),
                                      v.float().permute(0, 2, 1, 3), mask4).permute(0, 2, 1, 3)
    out, _ = _run(q, k, v, key_mask=km, out_dtype=torch.float32)
    assert float((out - ref).abs().max()) <= PARITY_TOL


def test_fully_masked_rows_are_zero():
    from photonic_flash_attention_amd import synth
    q, k, v = synth.qkv(1, 1, 64, 64, 64, 9, "bf16")
    out, lse = _run(q, k, v, seqlens_k=[0], return_lse=True)
    assert float(out.abs().max()) == 0.0 and bool(torch.isinf(lse).all())


def test_online_softmax_rescale_branch_is_forced():
    """cdna guide rule 26: spike one key per row late in the sequence so the running max jumps
    at a chosen tile; an fp64 full-tensor reference checks the rescale path."""
    from photonic_flash_attention_amd import synth
    B, H, S, D = 1, 2, 1024, 128
    q, k, v = synth.qkv(B, H, S, S, D, 123, "bf16")
    k = k.clone()
    k[:, 700] = (q[:, 5] * 4).to(torch.bfloat16)    # row 5 (and friends) spike at key 700 (tile 10)
    k[:, 130] = (q[:, 900] * 3).to(torch.bfloat16)
    qd, kd, vd = (t.double().permute(0, 2, 1, 3) for t in (q, k, v))
    ref = torch.softmax(qd @ kd.transpose(-1, -2) * D ** -0.5, dim=-1) @ vd
    out, _ = _run(q, k, v, out_dtype=torch.float32)
    assert float((out - ref.permute(0, 2, 1, 3).float()).abs().max()) <= PARITY_TOL


def test_scaled_inputs_stay_finite():
    """tests/unit/test_flash_attention_3.py:249-262: x10 (bf16) / x5 (fp16) scaled inputs."""
    from photonic_flash_attention_amd import synth
    for dtype, sc in (("bf16", 10.0), ("fp16", 5.0)):
        q, k, v = synth.qkv(2, 4, 256, 256, 64, 17, dtype, scale=sc)
        out, _ = _run(q, k, v)
        assert bool(torch.isfinite(out).all())


def test_bad_arguments_raise_before_launch():
    from photonic_flash_attention_amd import ops
    dev = _dev()
    q = torch.zeros(1, 2, 16, 160, dtype=torch.bfloat16, device=dev)
    with pytest.raises(ValueError):
        ops.fa3_forward(q, q, q)                      # head dim 160 (> 128: no kernel; <= 128 runs zero-padded)
    q32 = torch.zeros(1, 2, 16, 64, dtype=torch.float32, device=dev)
    assert ops.fa3_forward(q32, q32, q32)[0].dtype == torch.float32      # fp32 inputs: the exact fp32 kernel ...
    with pytest.raises(ValueError):
        ops.fa3_forward(q32, q32, q32, split_p=True)  # ... which knows no split P / kernel selector / mixed dtypes
    with pytest.raises(ValueError):
        ops.fa3_forward(q32, q32.to(torch.bfloat16), q32)
    qc = torch.zeros(1, 2, 16, 64, dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        ops.fa3_forward(qc, qc, qc)                   # host tensors: no CPU path


def test_reentrant_from_threads_on_side_streams():
    """SURVEY.md section 5 (race row): the reference calls the electronic branch from a thread pool
    (hybrid_router.py:323,461-463) and its concurrency test uses 8 caller threads; the C ABI must be
    re-entrant and honour the caller's stream."""
    import threading
    from photonic_flash_attention_amd import ops, synth
    dev = _dev()
    q, k, v = (t.to(dev) for t in synth.qkv(2, 4, 512, 512, 64, 404, "bf16"))
    want, _ = ops.fa3_forward_bshd(q, k, v, causal=True)
    torch.cuda.synchronize()
    outs, errs = [None] * 8, []

    def work(i):
        try:
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                for _ in range(20):
                    o, _ = ops.fa3_forward_bshd(q, k, v, causal=True)
                s.synchronize()
            outs[i] = o
        except Exception as e:  # pragma: no cover
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs
    assert all(torch.equal(o, want) for o in outs)


def test_runs_to_run_determinism_and_variant_equivalence():
    """Bitwise reproducibility of every production kernel; the two HIP kernels (8 waves x 32 rows, 4 waves x 64 rows) share every
    formula and are bit-identical with 16-bit stores; the persistent assembly kernel sums its rows in another order and agrees
    to one rounding of the output.  Development variants do not exist in this library (PFA_ERR_FLAGS)."""
    from photonic_flash_attention_amd import ops, synth
    dev = _dev()
    q, k, v = (t.to(dev) for t in synth.qkv(1, 4, 1536, 1536, 128, 505, "bf16"))
    outs = {}
    for var in (0, 43, 44, 45):
        a, _ = ops.fa3_forward_bshd(q, k, v, causal=True, _variant=var)
        b, _ = ops.fa3_forward_bshd(q, k, v, causal=True, _variant=var)
        assert torch.equal(a, b), f"selector {var} is not reproducible"
        outs[var] = a.float()
    assert torch.equal(outs[43], outs[44])
    assert float((outs[45] - outs[44]).abs().max()) <= float(outs[44].abs().max()) * 2.0 ** -8
    with pytest.raises(ValueError):
        ops.fa3_forward_bshd(q, k, v, causal=True, _variant=9)


# ---- development variant 43: 4 waves x 64 rows, one wave per SIMD (csrc/fa3_fwd_w4_kernel.h) ----------------------------
W4_CASES = [
    # B, H, Sq, Sk, D, causal, seqlens
    (1, 2, 256, 256, 128, False, None),
    (2, 2, 512, 512, 128, True, None),
    (1, 3, 200, 333, 128, False, None),       # ragged, Sq != Sk
    (1, 2, 1000, 1000, 128, True, None),      # ragged causal
    (2, 2, 300, 300, 128, False, [300, 41]),  # key lengths, one shorter than a tile
    (1, 2, 640, 64, 128, False, None),        # a single key tile
    (1, 4, 2048, 2048, 128, True, None),
]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("case", W4_CASES)
def test_w4_variant_matches_production_and_oracle(case, dtype):
    """Same math, different schedule: the 4-wave kernel must agree with the production kernel to rounding noise and
    with the oracle within the single-P (16-bit P) tolerance of 1e-2 max-abs."""
    from photonic_flash_attention_amd import ops
    from oracle import fa3_oracle as orc
    from photonic_flash_attention_amd import _capi, synth
    B, H, Sq, Sk, D, causal, lens = case
    q, k, v = synth.qkv(B, H, Sq, Sk, D, 4300 + Sq, dtype)
    qd, kd, vd = (t.to("cuda:0").permute(0, 2, 1, 3) for t in (q, k, v))
    kw = dict(causal=causal, seqlens_k=lens, out_dtype=torch.float32, split_p=False, return_lse=True)
    o0, l0 = ops.fa3_forward(qd, kd, vd, **kw)
    o1, l1 = ops.fa3_forward(qd, kd, vd, _variant=43, **kw)
    torch.cuda.synchronize()
    args, _keep = ops.build_args(qd, kd, vd, o1, causal=causal, seqlens_k=lens, split_p=False, variant=43)
    assert "w4" in _capi.describe(args)[0]
    assert float((o0 - o1).abs().max()) <= 2e-5 and float((l0 - l1).abs().max()) <= 2e-5
    ref = orc.attention_bshd(q, k, v, causal=causal, seqlens_k=lens)
    assert float((o1.permute(0, 2, 1, 3).cpu() - ref).abs().max()) <= 1e-2
    o16 = ops.fa3_forward(qd, kd, vd, causal=causal, seqlens_k=lens, _variant=43)[0]
    assert float((o16.float() - o1).abs().max()) <= (2e-2 if dtype == "bf16" else 3e-3)      # 16-bit store rounding only


def test_w4_variant_random_shapes_against_the_8_wave_kernel():
    """40 random (Sq, Sk, causal, key lengths, dtype) problems at D = 128: the two forward kernels share every formula, so
    with fp32 stores they must agree to accumulation-order noise (rows with no visible key: zeros and lse = -inf in both)."""
    import random
    from photonic_flash_attention_amd import ops, synth
    rnd = random.Random(1234)
    for it in range(40):
        B, H = rnd.choice([(1, 1), (1, 3), (2, 2)])
        Sq = rnd.choice([1, 17, 64, 100, 255, 256, 257, 300, 511, 640, 1000])
        Sk = rnd.choice([1, 33, 64, 65, 127, 128, 200, 256, 320, 513, 777, 1024])
        causal = rnd.random() < 0.5
        dtype = rnd.choice(["bf16", "fp16"])
        lens = None
        if rnd.random() < 0.4:
            lens = [rnd.randint(0, Sk) for _ in range(B)]
        q, k, v = (t.to("cuda:0").permute(0, 2, 1, 3) for t in synth.qkv(B, H, Sq, Sk, 128, 9000 + it, dtype))
        kw = dict(causal=causal, seqlens_k=lens, out_dtype=torch.float32, split_p=False, return_lse=True)
        o0, l0 = ops.fa3_forward(q, k, v, _variant=44, **kw)
        o1, l1 = ops.fa3_forward(q, k, v, _variant=43, **kw)
        torch.cuda.synchronize()
        tag = (it, B, H, Sq, Sk, causal, lens, dtype)
        assert bool(torch.isfinite(o1).all()), tag
        assert float((o0 - o1).abs().max()) <= 3e-5, tag
        dead = torch.isinf(l0)
        assert torch.equal(dead, torch.isinf(l1)), tag
        assert float((l0 - l1)[~dead].abs().max() if bool((~dead).any()) else 0.0) <= 3e-5, tag


# ---- selector 45: the persistent 4-wave kernel in assembly (csrc/gen_fa3_fwd_p4.py) -----------------------------------------------
P4_CASES = [
    # B, H, Sq, Sk, causal, kv heads
    (1, 8, 512, 512, True, 8),          # one unit per XCD: heavy block, light block, cross-item prefetch
    (1, 8, 256, 256, False, 8),         # a single 4-tile item
    (2, 4, 1024, 1024, True, 4),
    (1, 3, 512, 512, True, 3),          # heads not a multiple of 8: linear workgroup -> unit map
    (2, 8, 768, 768, True, 8),          # an odd number of Q blocks under the causal mask: (2, 0) and the middle block alone
    (3, 5, 256, 256, True, 5),          # a single causal block
    (3, 5, 768, 640, False, 5),         # Sq != Sk
    (1, 16, 2048, 2048, True, 4),       # grouped-query heads
    (2, 16, 1024, 2304, False, 2),      # cross attention, long keys, GQA
    (9, 32, 256, 384, False, 32),       # more units than workgroups
    # head dim 64 (two keys per LDS row, half the fragments per tile): the last element is D
    (1, 8, 512, 512, True, 8, 64),
    (4, 12, 1024, 1024, False, 12, 64),  # BASELINE configs[1] (C2)
    (3, 5, 768, 640, False, 5, 64),
    (1, 16, 2048, 2048, True, 4, 64),
    (9, 32, 256, 384, False, 8, 64),
]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("case", P4_CASES)
def test_p4_kernel_against_the_oracle_and_the_8_wave_kernel(case, dtype):
    from oracle import fa3_oracle as orc
    from photonic_flash_attention_amd import _capi, ops, synth
    B, H, Sq, Sk, causal, Hkv = case[:6]
    D = case[6] if len(case) > 6 else 128
    q, k, v = synth.qkv(B, H, Sq, Sk, D, 4500 + Sq + H, dtype)
    k, v = k[:, :, :Hkv].contiguous(), v[:, :, :Hkv].contiguous()
    qd, kd, vd = (t.to("cuda:0").permute(0, 2, 1, 3) for t in (q, k, v))
    o45, l45 = ops.fa3_forward(qd, kd, vd, causal=causal, return_lse=True, _variant=45)
    o44, l44 = ops.fa3_forward(qd, kd, vd, causal=causal, return_lse=True, _variant=44)
    torch.cuda.synchronize()
    name = _capi.describe(ops.build_args(qd, kd, vd, o45, causal=causal, variant=45)[0])[0]
    assert name.startswith(f"fa3_fwd_p4_{dtype}_d{D}_"), name
    assert bool(torch.isfinite(o45.float()).all())
    assert float((o45.float() - o44.float()).abs().max()) <= float(o44.float().abs().max()) * 2.0 ** (-8 if dtype == "bf16" else -10)
    assert float((l45 - l44).abs().max()) <= 2e-5
    g = H // Hkv
    ref = orc.attention_bshd(q, k.repeat_interleave(g, dim=2), v.repeat_interleave(g, dim=2), causal=causal)
    assert _fast_ok(o45.permute(0, 2, 1, 3).float().cpu(), ref, _pnorm(q, k, causal), float(v.float().abs().max()), dtype,
                    tag=f"p4 {case} {dtype}")
    ref_lse = orc.lse_bshd(q, k.repeat_interleave(g, dim=2), causal=causal)
    assert float((l45.cpu() - ref_lse).abs().max()) <= 2e-3
    # the parity variant of the same schedule (fp32 store + split P): the north-star tolerance against the oracle
    p45, _ = ops.fa3_forward(qd, kd, vd, causal=causal, out_dtype=torch.float32, _variant=45)
    p44, _ = ops.fa3_forward(qd, kd, vd, causal=causal, out_dtype=torch.float32, _variant=44)
    torch.cuda.synchronize()
    assert "splitp_o32" in _capi.describe(ops.build_args(qd, kd, vd, p45, causal=causal, split_p=True, variant=45)[0])[0]
    err = float((p45.permute(0, 2, 1, 3).cpu() - ref).abs().max())
    assert err <= PARITY_TOL and float((p45 - p44).abs().max()) <= 3e-5, (case, dtype, err)


@pytest.mark.parametrize("case", [(2, 8, 512, 512, True, 128), (3, 4, 256, 384, False, 128), (2, 4, 1024, 768, False, 64), (2, 8, 1024, 1024, True, 64),
                                  (2, 8, 300, 300, False, 128), (4, 4, 1024, 1025, False, 64), (3, 4, 769, 769, True, 128)])     # the last three: ragged Sq / Sk under a key mask
@pytest.mark.parametrize("kind", ["padding", "random", "empty_row"])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_p4_key_mask_kernels_against_the_oracle(case, kind, dtype):
    """[B, Sk] key masks on the persistent schedule (fa3_fwd_p4_*_km_*: the waves read the mask bytes themselves): a visible prefix
    per batch, random bytes (every tile takes the bit-test path), and a batch without any visible key (output 0, LSE -inf)."""
    from oracle import fa3_oracle as orc
    from photonic_flash_attention_amd import _capi, ops, synth
    B, H, Sq, Sk, causal, D = case
    q, k, v = synth.qkv(B, H, Sq, Sk, D, 4700 + Sq + D, dtype)
    g = torch.Generator().manual_seed(17 + Sq)
    if kind == "padding":
        lens = torch.randint(1, Sk + 1, (B,), generator=g)
        lens[0] = Sk
        km = torch.arange(Sk)[None, :] < lens[:, None]
    else:
        km = torch.rand(B, Sk, generator=g) < 0.7
        if kind == "empty_row":
            km[B - 1] = False
    if causal:
        km[:B - 1 if kind == "empty_row" else B, 0] = True                  # (row 0 sees only key 0)
    qd, kd, vd = (t.to("cuda:0").permute(0, 2, 1, 3) for t in (q, k, v))
    kmd = km.to("cuda:0")
    o16, lse = ops.fa3_forward(qd, kd, vd, causal=causal, key_mask=kmd, return_lse=True)
    o32, _ = ops.fa3_forward(qd, kd, vd, causal=causal, key_mask=kmd, out_dtype=torch.float32)
    o44, l44 = ops.fa3_forward(qd, kd, vd, causal=causal, key_mask=kmd, out_dtype=torch.float32, return_lse=True, _variant=44)
    torch.cuda.synchronize()
    name = _capi.describe(ops.build_args(qd, kd, vd, o16, causal=causal, key_mask=kmd)[0])[0]
    small = D == 64 and B * H * ((Sq + 255) // 256) // (2 if causal else 1) > 256
    assert small or name.startswith(f"fa3_fwd_p4_{dtype}_d{D}_{'causal' if causal else 'full'}_km_o16"), name      # picked without a selector
    mask4 = km.view(B, 1, 1, Sk).expand(B, 1, Sq, Sk)
    if causal:
        mask4 = mask4 & orc.causal_mask(Sq, Sk)
    dead = ~mask4.any(dim=-1)                                                # [B,1,Sq]: rows without a visible key
    ref = orc.flash_attention_forward(q.float().permute(0, 2, 1, 3), k.float().permute(0, 2, 1, 3), v.float().permute(0, 2, 1, 3),
                                      mask4).permute(0, 2, 1, 3)
    ref = torch.where(dead.permute(0, 2, 1).unsqueeze(-1).expand_as(ref), torch.zeros_like(ref), ref)   # the kernels' convention for such rows
    err = float((o32.permute(0, 2, 1, 3).cpu() - ref).abs().max())
    assert err <= PARITY_TOL, (case, kind, err)
    assert float((o32 - o44).abs().max()) <= 3e-5
    assert float((o16.float() - o32).abs().max()) <= (2e-2 if dtype == "bf16" else 4e-3)
    deadh = dead.expand(B, H, Sq).to("cuda:0")
    assert bool((torch.isinf(lse) == deadh).all()) and float((lse - l44)[~deadh].abs().max()) <= 2e-5
    assert float(o16.float()[deadh].abs().max() if bool(deadh.any()) else 0.0) == 0.0
    # a mask whose rows are not contiguous stays on the HIP kernels
    a, _keep = ops.build_args(qd, kd, vd, o16, causal=causal, key_mask=kmd)
    a.key_mask_stride_b = Sk + 64
    assert "p4" not in _capi.describe(a)[0]


@pytest.mark.parametrize("case", [(1, 8, 300, 300, False, 128), (2, 4, 257, 193, False, 128), (1, 8, 1000, 1000, True, 128), (2, 3, 512, 1000, False, 128),
                                  (2, 8, 1024, 1025, False, 128), (1, 8, 129, 2000, False, 128), (2, 4, 1500, 1500, True, 64), (3, 5, 700, 640, False, 64),
                                  (1, 16, 700, 700, True, 128), (9, 32, 200, 200, True, 128), (3, 8, 1281, 1281, True, 64)])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_p4_ragged_kernels_against_the_oracle(case, dtype):
    """Sq no multiple of 256 / Sk no multiple of 128 on the persistent schedule (fa3_fwd_p4_*_kl_*): against the oracle, against the
    8-wave kernel, and nothing read or written outside the tensors (NaN guard rows around q / k / v, sentinel rows around out / lse)."""
    from oracle import fa3_oracle as orc
    from photonic_flash_attention_amd import _capi, ops, synth
    B, H, Sq, Sk, causal, D = case
    q, k, v = synth.qkv(B, H, Sq, Sk, D, 4800 + Sq + Sk, dtype)                      # [B,S,H,D]
    G = 2

    def guarded(t, fill):
        buf = torch.full((t.shape[0], t.shape[1] + 2 * G) + tuple(t.shape[2:]), fill, dtype=t.dtype, device="cuda:0")
        buf[:, G:-G] = t.to("cuda:0")
        return buf
    qb, kb, vb = (guarded(t, float("nan")) for t in (q, k, v))
    qd, kd, vd = (t[:, G:-G].permute(0, 2, 1, 3) for t in (qb, kb, vb))
    ob16 = torch.full((B, Sq + 2 * G, H, D), 7.0, device="cuda:0", dtype=torch.bfloat16 if dtype == "bf16" else torch.float16)
    ob32 = torch.full((B, Sq + 2 * G, H, D), 7.0, device="cuda:0", dtype=torch.float32)
    o16, lse = ops.fa3_forward(qd, kd, vd, causal=causal, out=ob16[:, G:-G].permute(0, 2, 1, 3), return_lse=True)
    o32, _ = ops.fa3_forward(qd, kd, vd, causal=causal, out=ob32[:, G:-G].permute(0, 2, 1, 3), out_dtype=torch.float32)
    o44, l44 = ops.fa3_forward(qd, kd, vd, causal=causal, out_dtype=torch.float32, return_lse=True, _variant=44)
    torch.cuda.synchronize()
    name = _capi.describe(ops.build_args(qd, kd, vd, o16, causal=causal)[0])[0]
    assert name == f"fa3_fwd_p4_{dtype}_d{D}_{'causal' if causal else 'full'}_kl_o16", name          # picked without a selector
    for ob in (ob16, ob32):
        assert bool((ob[:, :G] == 7.0).all()) and bool((ob[:, -G:] == 7.0).all())
    ref = orc.attention_bshd(q, k, v, causal=causal)
    err = float((o32.permute(0, 2, 1, 3).cpu() - ref).abs().max())
    assert err <= PARITY_TOL, (case, err)
    assert float((o32 - o44).abs().max()) <= 3e-5 and float((lse - l44).abs().max()) <= 2e-5
    assert _fast_ok(o16.permute(0, 2, 1, 3).float().cpu(), ref, _pnorm(q, k, causal), float(v.float().abs().max()), dtype, tag=f"p4 ragged {case} {dtype}")
    assert float((lse.cpu() - orc.lse_bshd(q, k, causal=causal)).abs().max()) <= 2e-3


@pytest.mark.parametrize("case", [(4, 8, 512, 1024, [1024, 0, 1, 300], False, 128), (5, 3, 768, 2048, [2048, 65, 128, 129, 1999], False, 128),
                                  (2, 8, 1000, 1000, [1000, 517], False, 128), (4, 8, 512, 1024, [1024, 0, 700, 300], True, 128),
                                  (8, 4, 300, 3000, [3000, 1, 2, 63, 64, 191, 192, 2999], False, 64), (3, 4, 256, 384, [384, 64, 383], True, 64)])
def test_p4_seqlens_on_the_persistent_kernel(case):
    """seqlens_k without the causal mask: the ragged kernels' length word with a per-batch length, every item's tile count cut to it
    (length 0: output 0, LSE -inf); with a key mask as well, the *_km_* kernels AND the length into the words made from the bytes."""
    from oracle import fa3_oracle as orc
    from photonic_flash_attention_amd import _capi, ops, synth
    B, H, Sq, Sk, lens, with_mask, D = case
    q, k, v = synth.qkv(B, H, Sq, Sk, D, 4900 + Sq + Sk, "bf16")
    qd, kd, vd = (t.to("cuda:0").permute(0, 2, 1, 3) for t in (q, k, v))
    km = (torch.rand(B, Sk, generator=torch.Generator().manual_seed(5)) < 0.8) if with_mask else None
    kmd = km.to("cuda:0") if with_mask else None
    o32, lse = ops.fa3_forward(qd, kd, vd, seqlens_k=lens, key_mask=kmd, out_dtype=torch.float32, return_lse=True)
    o16, _ = ops.fa3_forward(qd, kd, vd, seqlens_k=lens, key_mask=kmd)
    o44, l44 = ops.fa3_forward(qd, kd, vd, seqlens_k=lens, key_mask=kmd, out_dtype=torch.float32, return_lse=True, _variant=44)
    torch.cuda.synchronize()
    name = _capi.describe(ops.build_args(qd, kd, vd, o16, seqlens_k=lens, key_mask=kmd)[0])[0]
    assert name == f"fa3_fwd_p4_bf16_d{D}_full_{'km' if with_mask else 'kl'}_o16", name
    keep = torch.arange(Sk)[None, :] < torch.tensor(lens)[:, None]
    if with_mask:
        keep = keep & km
    dead = ~keep.any(dim=1)                                                     # [B]
    mask4 = keep.view(B, 1, 1, Sk).expand(B, 1, Sq, Sk)
    ref = orc.flash_attention_forward(q.float().permute(0, 2, 1, 3), k.float().permute(0, 2, 1, 3), v.float().permute(0, 2, 1, 3),
                                      mask4).permute(0, 2, 1, 3)
    ref[dead] = 0.0                                                             # the kernels' convention for rows without a visible key
    err = float((o32.permute(0, 2, 1, 3).cpu() - ref).abs().max())
    assert err <= PARITY_TOL, (case, err)
    assert float((o32 - o44).abs().max()) <= 3e-5
    assert float((o16.float() - o32).abs().max()) <= 2e-2
    inf = torch.isinf(lse)
    assert bool((inf == dead.to("cuda:0").view(B, 1, 1).expand(B, H, Sq)).all()) and float((lse - l44)[~inf].abs().max()) <= 2e-5


def test_key_mask_bound_is_handed_to_long_launches_as_seqlens():
    """ops derives seqlens_k = 1 + the last visible key from a [B, Sk] key mask (device-side, no sync) when the launch is long and the
    persistent kernel takes it: a padding mask's tail is then skipped, the result does not change."""
    from photonic_flash_attention_amd import ops, synth
    B, H, S, D = 8, 16, 2048, 128
    q, k, v = (t.to("cuda:0").permute(0, 2, 1, 3) for t in synth.qkv(B, H, S, S, D, 4990, "bf16"))
    lens = torch.tensor([2048, 1100, 64, 1, 777, 2047, 1024, 1500])
    km = (torch.arange(S)[None, :] < lens[:, None]).to("cuda:0")
    a, keep = ops.build_args(q, k, v, torch.empty_like(q), key_mask=km)
    assert a.seqlens_k and any(t.dtype == torch.int32 and t.tolist() == lens.tolist() for t in keep)
    a2, _ = ops.build_args(q[:1, :2], k[:1, :2], v[:1, :2], torch.empty_like(q[:1, :2]), key_mask=km[:1])      # a short launch: no extra ops
    assert not a2.seqlens_k
    o1, l1 = ops.fa3_forward(q, k, v, key_mask=km, return_lse=True)
    o2, l2 = ops.fa3_forward(q, k, v, key_mask=km, return_lse=True, _variant=44)
    torch.cuda.synchronize()
    assert float((o1.float() - o2.float()).abs().max()) <= 2e-2 and float((l1 - l2).abs().max()) <= 2e-5


def test_p4_kernel_reads_fused_qkv_views_and_is_the_default_for_long_aligned_problems():
    from oracle import fa3_oracle as orc
    from photonic_flash_attention_amd import _capi, ops, synth
    B, S, H, D = 2, 2048, 8, 128
    E = H * D
    qkv = torch.from_numpy(synth.normal_f32((B, S, 3 * E), 77)).to("cuda:0", torch.bfloat16)
    q, k, v = (t.view(B, S, H, D).transpose(1, 2) for t in qkv.chunk(3, dim=-1))          # strides (S*3E, D, 3E, 1)
    out = torch.full((B, S, H, D), float("nan"), device="cuda:0", dtype=torch.bfloat16).permute(0, 2, 1, 3)
    ops.fa3_forward(q, k, v, out=out)
    torch.cuda.synchronize()
    name = _capi.describe(ops.build_args(q, k, v, out, causal=False)[0])[0]
    assert name.startswith("fa3_fwd_p4_bf16_d128_full"), name                             # picked without a selector
    qc, kc, vc = (t.transpose(1, 2).float().cpu().contiguous() for t in (q, k, v))
    ref = orc.attention_bshd(qc, kc, vc)
    assert _fast_ok(out.permute(0, 2, 1, 3).float().cpu(), ref, _pnorm(qc, kc, False), float(v.float().abs().max()))
    # ragged lengths: the *_kl_* kernels (rows past the end kept out by the buffer descriptors, keys past Sk by a computed mask word)
    q2, k2, v2 = (t[:, :, :2000] for t in (q, k, v))
    assert _capi.describe(ops.build_args(q2, k2, v2, out[:, :, :2000], causal=False)[0])[0] == "fa3_fwd_p4_bf16_d128_full_kl_o16"
    q3, k3, v3 = (t[:, :, :700] for t in (q, k, v))            # an odd number of Q blocks under the causal mask: the middle block is a unit of its own
    assert _capi.describe(ops.build_args(q3, k3, v3, out[:, :, :700], causal=True)[0])[0] == "fa3_fwd_p4_bf16_d128_causal_kl_o16"
    # not eligible -> the HIP kernels: short key sequences, Sq != Sk under the causal mask, element masks, fp32 store with one P
    assert "p4" not in _capi.describe(ops.build_args(q2, k2[:, :, :192], v2[:, :, :192], out[:, :, :2000], causal=False)[0])[0]
    assert "p4" not in _capi.describe(ops.build_args(q3, k2, v2, out[:, :, :700], causal=True)[0])[0]
    o32 = torch.empty(B, S, H, D, device="cuda:0", dtype=torch.float32).permute(0, 2, 1, 3)
    assert _capi.describe(ops.build_args(q, k, v, o32, causal=False, split_p=True)[0])[0] == "fa3_fwd_p4_bf16_d128_full_splitp_o32"
    assert "p4" not in _capi.describe(ops.build_args(q, k, v, o32, causal=False, split_p=False)[0])[0]      # fp32 store, one P: HIP kernel
    assert _capi.describe(ops.build_args(q, k, v, out, causal=False, seqlens_k=[S, S - 1])[0])[0] == "fa3_fwd_p4_bf16_d128_full_kl_o16"
    assert _capi.describe(ops.build_args(q, k, v, out, causal=True, seqlens_k=[S, S - 1])[0])[0] == "fa3_fwd_p4_bf16_d128_causal_kl_o16"      # round 3
    km_ = torch.ones(B, S, dtype=torch.bool, device="cuda:0")
    assert _capi.describe(ops.build_args(q, k, v, out, causal=True, seqlens_k=[S, S - 1], key_mask=km_)[0])[0] == "fa3_fwd_p4_bf16_d128_causal_km_o16"
    # D = 64: since round 3 (fast loop + mid-phase barrier) every eligible problem, also with several units per CU
    for (b, h, s, want) in ((4, 12, 1024, True), (16, 16, 2048, True)):
        t = torch.empty(b, s, h, 64, device="cuda:0", dtype=torch.bfloat16).permute(0, 2, 1, 3)
        nm = _capi.describe(ops.build_args(t, t, t, torch.empty_like(t), causal=False)[0])[0]
        assert nm.startswith("fa3_fwd_p4_bf16_d64_full") == want, (b, h, s, nm)


def test_p4_kernel_random_eligible_shapes():
    """30 random eligible problems against the 8-wave kernel (same formulas, other summation order)."""
    import random
    from photonic_flash_attention_amd import ops, synth
    rnd = random.Random(4545)
    for it in range(30):
        causal = rnd.random() < 0.5
        B, H = rnd.choice([(1, 1), (1, 8), (2, 3), (1, 24), (4, 8), (1, 40)])
        if causal:
            Sq = Sk = 512 * rnd.randint(1, 4) - (rnd.randint(0, 255) if it % 3 == 0 else 0)
        else:
            Sq, Sk = 256 * rnd.randint(1, 6), 128 * rnd.randint(2, 12)
            if it % 3 == 0:
                Sq, Sk = max(128, Sq - rnd.randint(0, 255)), max(193, Sk - rnd.randint(0, 127))
        dtype = rnd.choice(["bf16", "fp16"])
        D = rnd.choice([128, 64])
        q, k, v = (t.to("cuda:0").permute(0, 2, 1, 3) for t in synth.qkv(B, H, Sq, Sk, D, 12000 + it, dtype))
        o0, l0 = ops.fa3_forward(q, k, v, causal=causal, return_lse=True, _variant=44)
        o1, l1 = ops.fa3_forward(q, k, v, causal=causal, return_lse=True, _variant=45)
        torch.cuda.synchronize()
        tag = (it, B, H, Sq, Sk, D, causal, dtype)
        assert bool(torch.isfinite(o1.float()).all()), tag
        assert float((o0.float() - o1.float()).abs().max()) <= float(o0.float().abs().max()) * 2.0 ** (-8 if dtype == "bf16" else -10), tag
        assert float((l0 - l1).abs().max()) <= 3e-5, tag


def test_speedup_over_the_eager_tiled_loop_on_this_gpu():
    """Orientation, not a parity test: the reference's two-level tile loop (`flash_attention_3.py:182-262`, here the oracle's
    restatement of it executed as eager torch ops on bf16 device tensors, which is what the reference does on a GPU) against
    one launch of the HIP kernel at the headline shape C3.  The reference has no causal skipping and materialises a
    [B,1,S,S] mask; its bf16 accumulators make it 1e-2 off, so only a loose agreement is checked."""
    import time
    from oracle import fa3_oracle as orc
    from photonic_flash_attention_amd import ops, synth
    B, H, S, D = 4, 16, 4096, 128
    q, k, v = (t.to("cuda:0").permute(0, 2, 1, 3) for t in synth.qkv(B, H, S, S, D, 2003, "bf16"))
    mask = torch.tril(torch.ones(S, S, dtype=torch.bool, device="cuda:0")).view(1, 1, S, S).expand(B, 1, S, S)
    qs = q * D ** -0.5
    ref = orc.tiled_attention(qs, k, v, mask, 512)                    # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ref = orc.tiled_attention(qs, k, v, mask, 512)
    torch.cuda.synchronize()
    t_eager = time.perf_counter() - t0
    out = ops.fa3_forward(q, k, v, causal=True)[0]
    for _ in range(20):
        ops.fa3_forward(q, k, v, causal=True, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        ops.fa3_forward(q, k, v, causal=True, out=out)
    torch.cuda.synchronize()
    t_kernel = (time.perf_counter() - t0) / 50
    print(f"C3: eager tiled loop {t_eager * 1e3:.1f} ms, HIP kernel {t_kernel * 1e3:.3f} ms -> {t_eager / t_kernel:.0f} x")
    assert float((out.float() - ref.float()).abs().max()) <= 6e-2
    assert t_eager / t_kernel >= 20


@pytest.mark.parametrize("case", [(2, 8, 2, 300, 300, 64, True), (1, 8, 4, 512, 2048, 128, False), (1, 4, 1, 4096, 4096, 128, True),
                                  # groups of 3 (the 4-wave kernel's multiply-high head index), and 32 heads x batches (its head-grouped order)
                                  (1, 6, 2, 512, 1024, 128, False), (4, 8, 2, 512, 1024, 128, True)])
def test_grouped_query_heads_read_in_place(case):
    """ABI v4 `kv_group`: K/V with H / g heads, query head h reading K/V head h // g, must equal the run on K/V expanded with
    repeat_interleave bit for bit (both forward kernels), also for the weights pass; the backward (ABI v7) sums dK / dV over each group."""
    from photonic_flash_attention_amd import ops, synth
    B, H, Hkv, Sq, Sk, D, causal = case
    q = synth.qkv(B, H, Sq, Sq, D, 611, "bf16")[0].to("cuda:0").permute(0, 2, 1, 3)
    _, k, v = (t.to("cuda:0").permute(0, 2, 1, 3) for t in synth.qkv(B, Hkv, Sk, Sk, D, 612, "bf16"))
    ke, ve = (t.repeat_interleave(H // Hkv, dim=1) for t in (k, v))
    for var in (44, 43) if D == 128 else (0,):
        o_g, l_g = ops.fa3_forward(q, k, v, causal=causal, return_lse=True, _variant=var)
        o_e, l_e = ops.fa3_forward(q, ke, ve, causal=causal, return_lse=True, _variant=var)
        torch.cuda.synchronize()
        assert torch.equal(o_g, o_e) and torch.equal(l_g, l_e), (case, var)
    if Sq * Sk <= 1 << 19:
        w_g = ops.fa3_forward(q, k, v, causal=causal, return_weights=True)[2]
        w_e = ops.fa3_forward(q, ke, ve, causal=causal, return_weights=True)[2]
        assert torch.equal(w_g, w_e)
    # backward (ABI v7): dQ bit for bit as on the expanded heads; dK / dV = the expanded run's gradients summed over each group (the kernel
    # sums in fp32 registers, the reference sum below adds rounded 16-bit gradients: a few bf16 ulps of the sum)
    g = H // Hkv
    dout = synth.qkv(B, H, Sq, Sq, D, 613, "bf16")[0].to("cuda:0").permute(0, 2, 1, 3)
    dq_g, dk_g, dv_g = ops.fa3_backward(q, k, v, o_g, dout, l_g, causal=causal, grad_dtype=torch.float32)
    dq_e, dk_e, dv_e = ops.fa3_backward(q, ke, ve, o_g, dout, l_g, causal=causal, grad_dtype=torch.float32)
    torch.cuda.synchronize()
    assert dk_g.shape == (B, Hkv, Sk, D) and dv_g.shape == (B, Hkv, Sk, D)
    assert torch.equal(dq_g, dq_e), case
    for got, exp in ((dk_g, dk_e), (dv_g, dv_e)):
        want = exp.reshape(B, Hkv, g, Sk, D).sum(2)
        assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max())), case     # fp32 sums in another order
    with pytest.raises(ValueError):
        ops.fa3_backward(q, k[:, :1].expand(-1, 5, -1, -1) if H % 5 else k[:, :1], v, o_g, dout, l_g, causal=causal)      # head counts that do not divide


@pytest.mark.parametrize("D", [8, 32, 40, 80, 96, 120])
@pytest.mark.parametrize("causal", [False, True])
def test_head_dims_without_a_kernel_run_zero_padded(D, causal):
    """The reference takes any head dim (E // H); kernels exist for 64 and 128, smaller ones run zero-padded to the next
    (same scores with the TRUE D^-0.5 scale, extra output columns dropped).  Tolerance 1e-3 on the parity variant."""
    from oracle import fa3_oracle as orc
    from photonic_flash_attention_amd import ops
    from photonic_flash_attention_amd import synth
    B, H, Sq, Sk = 2, 3, 150, 333
    q, k, v = synth.qkv(B, H, Sq, Sk, D, 4000 + D, "bf16")
    ref = orc.attention_bshd(q.float(), k.float(), v.float(), causal=causal)
    DEV = _dev()
    qd, kd, vd = (t.to(DEV).permute(0, 2, 1, 3) for t in (q, k, v))
    out, lse, w = ops.fa3_forward(qd, kd, vd, causal=causal, out_dtype=torch.float32, return_lse=True, return_weights=True,
                                  weights_dtype=torch.float32)
    assert out.shape == (B, H, Sq, D) and w.shape == (B, H, Sq, Sk)
    err = float((out.permute(0, 2, 1, 3).cpu() - ref).abs().max())
    assert err <= 1e-3, f"D={D} causal={causal}: {err:.3e}"
    assert float((w.sum(-1) - 1).abs().max()) <= 1e-3
    given = torch.empty(B, Sq, H, D, device=DEV, dtype=torch.bfloat16).permute(0, 2, 1, 3)
    o2 = ops.fa3_forward(qd, kd, vd, causal=causal, out=given)[0]
    assert o2.data_ptr() == given.data_ptr()
    assert float((o2.permute(0, 2, 1, 3).float().cpu() - ref).abs().max()) <= 2e-2     # bf16 store
    with pytest.raises(ValueError):
        ops.fa3_forward(*(torch.zeros(1, 1, 64, 256, device=DEV, dtype=torch.bfloat16) for _ in range(3)))


@pytest.mark.parametrize("kind", ["key", "b1qk", "bhqk", "11qk", "3d"])
@pytest.mark.parametrize("causal", [False, True])
def test_mask_words_and_mask_bytes_agree(kind, causal):
    """The C ABI reads a mask either as 64-bit words condensed into the caller's workspace (pfa_fa3_workspace_bytes) or, without
    workspace, a byte per score: both paths must give the same bits, for every broadcast shape of the reference's masks."""
    import ctypes as C
    from photonic_flash_attention_amd import _capi, ops, synth
    dev = _dev()
    B, H, Sq, Sk, D = 2, 3, 200, 330, 64
    q, k, v = (t.to(dev).permute(0, 2, 1, 3) for t in synth.qkv(B, H, Sq, Sk, D, 9100, "bf16"))
    g = torch.Generator().manual_seed(3)
    shape = {"key": (B, Sk), "b1qk": (B, 1, Sq, Sk), "bhqk": (B, H, Sq, Sk), "11qk": (1, 1, Sq, Sk), "3d": (B, Sq, Sk)}[kind]
    m = torch.rand(shape, generator=g) < 0.6
    m[..., :70] = True                    # tile 0 fully visible (the all-ones shortcut) ...
    m[..., 128:192] = False               # ... tile 2 fully masked (skipped) ...
    m = m.to(dev)                         # ... the others mixed (bit tests)
    outs = []
    for with_ws in (True, False):
        out = torch.empty(B, Sq, H, D, device=dev, dtype=torch.float32).permute(0, 2, 1, 3)
        lse = torch.empty(B, H, Sq, device=dev, dtype=torch.float32)
        kw = dict(key_mask=m) if kind == "key" else dict(mask=m)
        a, keep = ops.build_args(q, k, v, out, causal=causal, lse=lse, **kw)
        assert a.workspace_bytes == _capi.load().pfa_fa3_workspace_bytes(C.byref(a)) > 0
        if not with_ws:
            a.workspace, a.workspace_bytes = None, 0
        assert _capi.load().pfa_fa3_fwd(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
        torch.cuda.synchronize()
        outs.append((out.clone(), lse.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(torch.nan_to_num(outs[0][1], neginf=-1e30), torch.nan_to_num(outs[1][1], neginf=-1e30))


# ---- every (batch, head) of the headline shapes: sampled rows against the oracle's dense softmax ---------------------------------------
def _dense_rows(q, k, v, rows, causal, chunk=8):
    """oracle.standard_attention (flash_attention_3.py:152-180 restated) on the query rows `rows` of EVERY (b, h): -> [B, len(rows), H, D]
    fp32; the causal mask as the reference expresses it (a 0/1 mask over the keys of each row)."""
    from oracle import fa3_oracle as orc
    B, S, H, D = q.shape
    out = torch.empty(B, len(rows), H, D)
    kj = torch.arange(k.shape[1])
    mask = (kj[None, :] <= rows[:, None])[None, None] if causal else None          # [1, 1, R, Sk]
    for b in range(B):
        for h0 in range(0, H, chunk):
            hs = slice(h0, min(H, h0 + chunk))
            qq = q[b:b + 1, rows][:, :, hs].permute(0, 2, 1, 3).float()
            kk, vv = (t[b:b + 1, :, hs].permute(0, 2, 1, 3).float() for t in (k, v))
            o, _ = orc.standard_attention(qq * D ** -0.5, kk, vv, mask)
            out[b, :, hs] = o[0].permute(1, 0, 2)
    return out


def _edge_rows(S, seed, n_random=20):
    g = torch.Generator().manual_seed(seed)
    fixed = [0, 1, 63, 64, 255, 256, 257, S - 256, S - 255, S - 193, S - 192, S - 1]
    rnd = torch.randint(0, S, (n_random,), generator=g).tolist()
    return torch.tensor(sorted(set(x for x in fixed + rnd if 0 <= x < S)))


@pytest.mark.parametrize("shape", [("C3", 4, 16, 4096, True), ("C5", 1, 32, 16384, True), ("S2048x256", 16, 16, 2048, False)])
def test_every_head_of_the_headline_shapes_on_sampled_rows(shape):
    """The pipelined item seam, the (heavy, light) pairing and the per-XCD dealing of the persistent kernel put every (b, h) on its own
    CU, item position and seam: ~32 rows of EVERY head (block edges 255 / 256, first / last block, seam rows, random ones) against the
    oracle's dense softmax -- parity variant <= 1e-3, benched variant inside the per-row rounding bound."""
    from photonic_flash_attention_amd import _capi, ops, synth
    name, B, H, S, causal = shape
    rows = _edge_rows(S, S)
    dev = _dev()
    gen = torch.Generator(device=dev).manual_seed(9000 + S)                    # (device RNG: the counter-based generator needs minutes at this size)
    qd, kd, vd = (torch.randn(B, S, H, 128, device=dev, generator=gen).to(torch.bfloat16) for _ in range(3))
    q, k, v = (t.cpu() for t in (qd, kd, vd))
    o32, _ = ops.fa3_forward_bshd(qd, kd, vd, causal=causal, out_dtype=torch.float32)
    o16, _ = ops.fa3_forward_bshd(qd, kd, vd, causal=causal)
    torch.cuda.synchronize()
    kname = _capi.describe(ops.build_args(*(t.permute(0, 2, 1, 3) for t in (qd, kd, vd, o16)), causal=causal)[0])[0]
    assert kname.startswith("fa3_fwd_p4_"), kname
    ref = _dense_rows(q, k, v, rows, causal)
    e32 = (o32[:, rows].cpu() - ref).abs()
    print(f"{name} ({kname}): {B * H} heads x {len(rows)} rows: parity variant max-abs {float(e32.max()):.3e}")
    assert float(e32.max()) <= PARITY_TOL
    pn = torch.empty(B, len(rows), H, 1)
    kjs = torch.arange(S)
    for b in range(B):                                   # ||p_row||_2 per sampled row (fp32, CPU)
        s = torch.einsum("qhd,khd->hqk", q[b, rows].float(), k[b].float()) * 128 ** -0.5
        if causal:
            s = s.masked_fill(kjs[None, None, :] > rows[None, :, None], float("-inf"))
        pn[b] = torch.softmax(s, dim=-1).square().sum(-1).sqrt().permute(1, 0)[..., None]
    assert _fast_ok(o16[:, rows].float().cpu(), ref, pn, float(v.float().abs().max()), "bf16", tag=f"{name} benched kernel, every head")


def test_c4_at_its_full_batch_on_one_gpu():
    """BASELINE configs[3] is B = 32 over 8 GPUs; the sharded runs only ever see B = 4 per GPU.  Here the whole B = 32 problem runs on
    ONE GPU (2048 items on the persistent kernel, 8 per CU): determinism, the B = 4 shards bit for bit, sampled rows of every head."""
    from photonic_flash_attention_amd import ops, synth
    B, H, S, D = 32, 16, 4096, 128
    dev = _dev()
    gen = torch.Generator(device=dev).manual_seed(500)
    qd, kd, vd = (torch.randn(B, S, H, D, device=dev, generator=gen).to(torch.bfloat16) for _ in range(3))
    q, k, v = (t.cpu() for t in (qd, kd, vd))
    o1, _ = ops.fa3_forward_bshd(qd, kd, vd)
    o2, _ = ops.fa3_forward_bshd(qd, kd, vd)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2)
    for i in (0, 3, 7):                                   # a rank's shard computed alone gives the same bits
        sh, _ = ops.fa3_forward_bshd(qd[4 * i:4 * i + 4], kd[4 * i:4 * i + 4], vd[4 * i:4 * i + 4])
        assert torch.equal(sh, o1[4 * i:4 * i + 4])
    rows = _edge_rows(S, 77, n_random=4)
    ref = _dense_rows(q, k, v, rows, False, chunk=16)
    o32, _ = ops.fa3_forward_bshd(qd, kd, vd, out_dtype=torch.float32)
    err = float((o32[:, rows].cpu() - ref).abs().max())
    print(f"C4 at B = 32: parity variant max-abs on {len(rows)} rows of all {B * H} heads {err:.3e}; "
          f"benched variant {float((o1[:, rows].float().cpu() - ref).abs().max()):.3e}")
    assert err <= PARITY_TOL
    assert float((o1[:, rows].float().cpu() - ref).abs().max()) <= 2e-2


def test_hip_graph_capture_and_replay_of_the_persistent_kernel():
    """pfa_fa3_prepare / _capi.load() load the code object eagerly, so the FIRST forward of a p4-eligible shape may sit inside a stream
    capture; the replayed graph reproduces the eager result bit for bit, also on a side stream and with new data in the same buffers."""
    from photonic_flash_attention_amd import _capi, ops, synth
    dev = _dev()
    assert _capi.load().pfa_fa3_prepare(0) == 0
    q, k, v = (t.to(dev) for t in synth.qkv(2, 8, 1024, 1024, 128, 321, "bf16"))
    out = torch.empty_like(q)
    qv, kv, vv, ov = (t.permute(0, 2, 1, 3) for t in (q, k, v, out))
    assert _capi.describe(ops.build_args(qv, kv, vv, ov, causal=True)[0])[0].startswith("fa3_fwd_p4_")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            ops.fa3_forward(qv, kv, vv, causal=True, out=ov)
    eager, _ = ops.fa3_forward(qv, kv, vv, causal=True)
    torch.cuda.synchronize()
    for rep in range(3):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(ov, eager)
    q.copy_(synth.qkv(2, 8, 1024, 1024, 128, 322, "bf16")[0].to(dev))           # new data, same buffers
    g.replay()
    eager2, _ = ops.fa3_forward(qv, kv, vv, causal=True)
    torch.cuda.synchronize()
    assert torch.equal(ov, eager2) and not torch.equal(eager2, eager)


def test_reserved_cus_shrink_the_persistent_grid_and_keep_the_result():
    """pfa_fa3_args.reserve_cus (ABI v6): a caller that overlaps a kernel-based collective leaves it CUs; same bits on the smaller grid."""
    import ctypes as C
    from photonic_flash_attention_amd import _capi, ops, synth
    dev = _dev()
    q, k, v = (t.to(dev) for t in synth.qkv(2, 8, 2048, 2048, 128, 99, "bf16"))
    qv, kv, vv = (t.permute(0, 2, 1, 3) for t in (q, k, v))
    full, _ = ops.fa3_forward(qv, kv, vv, causal=True)
    out = torch.empty_like(q).permute(0, 2, 1, 3)
    a, keep = ops.build_args(qv, kv, vv, out, causal=True)
    n_full = _capi.describe(a)[1]
    a.reserve_cus = 32
    name, n_res = _capi.describe(a)
    assert name.startswith("fa3_fwd_p4_") and n_res == n_full - 32 and n_res % 8 == 0
    st = _capi.load().pfa_fa3_fwd(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0
    torch.cuda.synchronize()
    assert torch.equal(out, full)
    a.reserve_cus = -1
    assert _capi.load().pfa_fa3_check(C.byref(a)) == -3


def _sdma_worker(rank, world, port, q_out):
    import os
    import sys
    from conftest import REPO
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)       # control plane only; both ranks share the box's one GPU
    try:
        from photonic_flash_attention_amd import ops
        from photonic_flash_attention_amd.parallel import sharded
        dev = torch.device("cuda:0")
        gen = torch.Generator(device=dev).manual_seed(4)
        q, k, v = (torch.randn(4, 512, 8, 128, device=dev, generator=gen).to(torch.bfloat16) for _ in range(3))
        plan = sharded.shard_plan(4, 8, world)
        full_ref = ops.fa3_forward_bshd(q, k, v, causal=True)[0]
        ql, kl, vl = (sharded.local_slice(t, plan, rank).contiguous() for t in (q, k, v))
        out_l = ops.fa3_forward_bshd(ql, kl, vl, causal=True)[0].contiguous()
        torch.cuda.synchronize()
        full, ms = sharded.gather_outputs(out_l, plan, timed=True, algo="sdma")
        full2 = sharded.gather_outputs(out_l, plan, algo="sdma")
        ms_loop = sharded.overlapped_forward_gather(lambda: None, out_l, 4, algo="sdma") if os.environ.get("PFA_SDMA_LOOP") else -1.0
        q_out.put((rank, bool(torch.equal(full, full_ref)) and bool(torch.equal(full2, full_ref)), tuple(full.shape), ms, ms_loop))
    finally:
        dist.destroy_process_group()


def test_copy_engine_gather_between_two_processes():
    """parallel/sharded.py PeerGather ("sdma"): two rank processes (sharing this box's GPU; control plane over gloo) map each other's
    shard buffers through IPC handles and pull them with plain device copies -- the assembled tensor is the single-process result."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    procs = [ctx.Process(target=_sdma_worker, args=(r, 2, port, q_out)) for r in range(2)]
    [p.start() for p in procs]
    res = [q_out.get(timeout=300) for _ in range(2)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, ok, shape, ms, ms_loop in res:
        print(f"rank {rank}: copy-engine gather of {shape}: {ms:.3f} ms")
        assert ok and shape == (4, 512, 8, 128)


@pytest.mark.parametrize("case", [(128, False), (128, True), (64, False), (64, True)])
def test_fast_loop_fixup_on_spiked_keys(case):
    """The persistent kernel's fast loop takes a tile's exponentials against the running maximum without looking for a new one; a row
    sum past 2^14 sends the wave to its fix-up subroutine, which redoes the tile from the scaled scores it kept (gen_fa3_fwd_p4.py
    finish_fast / fixup).  Keys that outgrow the running maximum by 2^10 .. far beyond 2^128 (the exponential itself overflows), in
    even and odd tiles, both key blocks, a wave's last tile and the diagonal: fp64 full-tensor reference (cdna guide rule 26)."""
    from photonic_flash_attention_amd import _capi, ops, synth
    D, causal = case
    B, H, S = 1, 4, 1024
    q, k, v = synth.qkv(B, H, S, S, D, 777 + D, "bf16")
    k = k.clone()
    for key, row, f in ((70, 100, 0.9), (300, 400, 1.5), (352, 500, 25.0), (453, 600, 3.0), (520, 700, 0.8), (1023, 1023, 30.0), (960, 990, 2.0)):
        k[:, key] = (q[:, row].float() * f).to(torch.bfloat16)      # row `row` (and its friends) jump at key `key`
    dev = _dev()
    qd, kd, vd = (t.to(dev) for t in (q, k, v))
    out, lse = ops.fa3_forward_bshd(qd, kd, vd, causal=causal, return_lse=True)
    torch.cuda.synchronize()
    name = _capi.describe(ops.build_args(*(t.permute(0, 2, 1, 3) for t in (qd, kd, vd, out)), causal=causal)[0])[0]
    assert name.startswith("fa3_fwd_p4_") and name.endswith("_o16"), name
    s = torch.einsum("bqhd,bkhd->bhqk", q.double(), k.double()) * D ** -0.5
    if causal:
        s = s.masked_fill(torch.arange(S)[None, :] > torch.arange(S)[:, None], float("-inf"))
    ref_lse = torch.logsumexp(s, dim=-1)
    p = torch.softmax(s, dim=-1)
    ref = torch.einsum("bhqk,bkhd->bqhd", p, v.double()).float()
    el = (lse.cpu().double() - ref_lse).abs() / (1 + ref_lse.abs() / 64)
    print(f"{name}: LSE max err {float(el.max()):.3e} (row max of scores up to {float(s.max()):.0f})")
    assert float(el.max()) <= 1e-4
    pn = p.square().sum(-1).sqrt().permute(0, 2, 1)[..., None].float()
    assert _fast_ok(out.float().cpu(), ref, pn, float(v.float().abs().max()), "bf16", tag=name)


@pytest.mark.parametrize("case", [(6, 4, 1024, [1024, 0, 1, 300, 512, 769], 128), (4, 2, 2048, [2047, 1025, 256, 64], 128),
                                  (3, 4, 1000, [1000, 999, 130], 128), (4, 4, 1024, [700, 1, 257, 1024], 64), (2, 8, 1280, [513, 1279], 128)])
@pytest.mark.parametrize("with_mask", [False, True])
def test_p4_seqlens_under_the_causal_mask(case, with_mask):
    """Round 3: seqlens_k UNDER the causal mask (the padded decoder batch) on the persistent ragged kernels: a block that lies behind its
    batch's cut runs only the 256-key groups that hold visible keys, without a diagonal; the block that holds the cut keeps its diagonal and
    masks what lies past the length.  Cuts inside the first / a middle / the last group, on group and tile edges, lengths 0 and S, ragged S."""
    from oracle import fa3_oracle as orc
    from photonic_flash_attention_amd import _capi, ops, synth
    B, H, S, lens, D = case
    if with_mask and S % 256:
        pytest.skip("key-mask kernels take whole blocks")
    q, k, v = synth.qkv(B, H, S, S, D, 5200 + S, "bf16")
    qd, kd, vd = (t.to("cuda:0").permute(0, 2, 1, 3) for t in (q, k, v))
    # with_mask: the same lengths plus holes, as a [B, Sk] key mask AND seqlens_k (what ops derives from a padding mask): *_km_* kernels
    km = (torch.rand(B, S, generator=torch.Generator().manual_seed(6)) < 0.85) if with_mask else None
    if with_mask:
        km[:, 0] = True                                                        # (key 0 stays: every live row of a causal problem sees it)
    kw = dict(causal=True, seqlens_k=lens, key_mask=km.to("cuda:0") if with_mask else None)
    o32, lse = ops.fa3_forward(qd, kd, vd, out_dtype=torch.float32, return_lse=True, **kw)
    o16, _ = ops.fa3_forward(qd, kd, vd, **kw)
    o44, l44 = ops.fa3_forward(qd, kd, vd, out_dtype=torch.float32, return_lse=True, _variant=44, **kw)
    torch.cuda.synchronize()
    name = _capi.describe(ops.build_args(qd, kd, vd, o16, **kw)[0])[0]
    assert name == f"fa3_fwd_p4_bf16_d{D}_causal_{'km' if with_mask else 'kl'}_o16", name
    if with_mask:
        keep = (torch.arange(S)[None, :] < torch.tensor(lens)[:, None]) & km
        mask4 = keep.view(B, 1, 1, S) & (torch.arange(S)[None, :] <= torch.arange(S)[:, None]).view(1, 1, S, S)
        ref = orc.flash_attention_forward(q.float().permute(0, 2, 1, 3), k.float().permute(0, 2, 1, 3), v.float().permute(0, 2, 1, 3),
                                          mask4).permute(0, 2, 1, 3)
    else:
        ref = orc.attention_bshd(q, k, v, causal=True, seqlens_k=lens)
    dead = torch.tensor(lens) == 0
    ref[dead] = 0.0                                                             # the kernels' convention for rows without a visible key
    err = float((o32.permute(0, 2, 1, 3).cpu() - ref).abs().max())
    print(f"{case}: {name}: parity variant max-abs {err:.3e}")
    assert err <= PARITY_TOL, (case, err)
    assert float((o32 - o44).abs().max()) <= 3e-5
    assert float((o16.float() - o32).abs().max()) <= 2e-2
    inf = torch.isinf(lse)
    assert bool((inf == dead.to("cuda:0").view(B, 1, 1).expand(B, H, S)).all()) and float((lse - l44)[~inf].abs().max()) <= 2e-5


@pytest.mark.parametrize("case", [("S65536 causal, one head", 1, 1, 65536, 65536, 128, True), ("S32768, two heads", 1, 2, 32768, 32768, 128, False),
                                  ("2048 heads x 512 causal", 64, 32, 512, 512, 128, True), ("S65535 causal, D64", 1, 1, 65535, 65535, 64, True),
                                  ("Sq 128 x Sk 100001", 1, 8, 128, 100001, 128, False), ("4096 heads x 256", 128, 32, 256, 256, 128, False)])
def test_persistent_kernel_extreme_shapes(case):
    """The persistent kernels at the ends of their range (through the default dispatch): very long sequences, one head, thousands of
    heads, a ragged 65535, a key sequence of 100001 -- against the 8-wave HIP kernel (same formulas, other summation order)."""
    from photonic_flash_attention_amd import _capi, ops
    tag, B, H, Sq, Sk, D, causal = case
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(len(tag))
    q = torch.randn(B, Sq, H, D, device=dev, generator=g).to(torch.bfloat16).permute(0, 2, 1, 3)
    k, v = (torch.randn(B, Sk, H, D, device=dev, generator=g).to(torch.bfloat16).permute(0, 2, 1, 3) for _ in range(2))
    o0, l0 = ops.fa3_forward(q, k, v, causal=causal, return_lse=True)
    o1, l1 = ops.fa3_forward(q, k, v, causal=causal, return_lse=True, _variant=44)
    torch.cuda.synchronize()
    name = _capi.describe(ops.build_args(q, k, v, o0, causal=causal)[0])[0]
    d, dl = float((o0.float() - o1.float()).abs().max()), float((l0 - l1).abs().nan_to_num(0.0).max())
    print(f"{tag}: {name}  |dO| {d:.1e} |dLSE| {dl:.1e}")
    assert "p4" in name and bool(torch.isfinite(o0.float()).all()) and d <= 2e-2 and dl <= 1e-4


def test_persistent_kernel_graph_of_three_flavours_side_stream_and_threads():
    """One HIP graph holding a plain causal, a key-masked (with the seqlens_k hint derived on the device) and a ragged causal launch of
    the persistent kernels, captured on a side stream and replayed three times bit for bit; then four host threads on four streams."""
    import threading
    from photonic_flash_attention_amd import ops
    dev = _dev()

    def mk(B, H, S, seed):
        g = torch.Generator(device=dev).manual_seed(seed)
        return tuple(torch.randn(B, S, H, 128, device=dev, generator=g).to(torch.bfloat16).permute(0, 2, 1, 3) for _ in range(3))
    q, k, v = mk(4, 8, 1024, 3)
    lens = torch.tensor([1024, 300, 77, 640], device=dev)
    km = torch.arange(1024, device=dev)[None, :] < lens[:, None]
    qr, kr, vr = mk(2, 8, 1000, 4)
    ref = {"plain": ops.fa3_forward(q, k, v, causal=True)[0].clone(), "km": ops.fa3_forward(q, k, v, key_mask=km)[0].clone(),
           "ragged": ops.fa3_forward(qr, kr, vr, causal=True)[0].clone()}
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        o = ops.fa3_forward(q, k, v, causal=True)[0]
    s.synchronize()
    assert torch.equal(o, ref["plain"])
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
    errs = []

    def worker(i):
        try:
            st = torch.cuda.Stream()
            qq, kk, vv = mk(2, 8, 512 + 256 * i, 10 + i)
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
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs


@pytest.mark.parametrize("case", [(3, 4, 1024, False, 128), (3, 4, 1024, True, 128), (2, 8, 2048, True, 64), (2, 4, 512, False, 64)])
def test_p4_key_mask_fresh_rows_left_padding(case):
    """Key-mask kernels on the fast loop: a row that sees no key in its item's tile 0 (left padding, a hole at the start) is FRESH -- it
    keeps m = 0 and a negative limit, so the per-tile check fires for it and the fix-up subroutine gives it its first maximum at the
    first tile that shows it a key, whatever that maximum is (scores scaled up and down here).  Left padding of 1, 64, 65, 200 ... keys,
    holes in the middle, a batch with no key at all; against the oracle and the 8-wave kernel."""
    from photonic_flash_attention_amd import _capi, ops, synth
    B, H, S, causal, D = case
    q, k, v = synth.qkv(B, H, S, S, D, 6100 + S + D, "bf16")
    q = (q.float() * torch.tensor([1.0, 6.0, 0.05])[:B].view(B, 1, 1, 1)).to(torch.bfloat16)       # first maxima far above / below 0
    km = torch.ones(B, S, dtype=torch.bool)
    km[0, :200] = False                          # left padding past three tiles
    km[0, 400:600] = False                       # and a hole (all-masked tiles in the middle)
    km[1, :65] = False
    if B > 2:
        km[2, :] = False                         # a batch without any key
        km[2, 700:705] = True if causal else False
    qd, kd, vd = (t.to("cuda:0").permute(0, 2, 1, 3) for t in (q, k, v))
    kmd = km.to("cuda:0")
    o32, lse = ops.fa3_forward(qd, kd, vd, causal=causal, key_mask=kmd, out_dtype=torch.float32, return_lse=True, _variant=44)
    o16, l16 = ops.fa3_forward(qd, kd, vd, causal=causal, key_mask=kmd, return_lse=True)
    torch.cuda.synchronize()
    name = _capi.describe(ops.build_args(qd, kd, vd, o16, causal=causal, key_mask=kmd)[0])[0]
    assert name == f"fa3_fwd_p4_bf16_d{D}_{'causal' if causal else 'full'}_km_o16", name
    keep = km.view(B, 1, 1, S).expand(B, 1, S, S)
    if causal:
        keep = keep & (torch.arange(S)[None, :] <= torch.arange(S)[:, None]).view(1, 1, S, S)
    # (the reference's tiled branch poisons a row whose 512-key tile is fully masked, flash_attention_3.py:249 -- outside the parity domain;
    #  the yardstick here is the dense fp64 softmax, rows without any visible key 0 / -inf by the kernels' convention)
    sc = torch.einsum("bqhd,bkhd->bhqk", q.double(), k.double()) * D ** -0.5
    sc = sc.masked_fill(~keep, float("-inf"))
    dead = ~keep.any(dim=-1).expand(B, H, S)
    pr = torch.softmax(sc, dim=-1).nan_to_num(0.0)
    ref = torch.einsum("bhqk,bkhd->bhqd", pr, v.double()).float()
    err16 = (o16.float().cpu() - ref).abs()
    print(f"{name}: fast variant max-abs vs oracle {float(err16.max()):.3e}; vs the 8-wave parity kernel {float((o16.float() - o32).abs().max()):.3e}; "
          f"dead rows {int(dead.sum())}")
    assert float((o32.cpu() - ref).abs().max()) <= PARITY_TOL                  # (the reference kernel itself)
    assert float(err16.max()) <= 3e-2
    inf = torch.isinf(l16.cpu())
    assert bool((inf == dead).all()) and float((l16 - lse).cpu()[~inf].abs().max()) <= 1e-4


@pytest.mark.parametrize("case", [("window", 2, 3, 1536, 1536, 128), ("tril", 1, 2, 1024, 1024, 64), ("documents", 2, 2, 1280, 1280, 128),
                                  ("dead_block", 1, 2, 768, 1024, 128), ("left_padding_4d", 2, 2, 600, 900, 64), ("per_head", 1, 3, 512, 1024, 128)])
def test_element_masks_skip_the_tiles_they_hide(case):
    """Element masks on the HIP kernels: beside the 64-bit word per row and tile, the condensing pass leaves the first / last tile with
    any visible key per 256 mask rows (FwdParams::mrange); a Q block runs only that range.  Structured masks -- a sliding window, the
    triangle as a mask, block-diagonal documents, a Q block that sees nothing, left padding, a different band per head -- against the
    dense fp64 softmax (rows without a visible key: 0 / LSE -inf), parity variant <= 1e-3."""
    from photonic_flash_attention_amd import _capi, ops, synth
    kind, B, H, Sq, Sk, D = case
    q, k, v = synth.qkv(B, H, Sq, Sk, D, 7300 + Sq + D, "bf16")
    iq, ik = torch.arange(Sq)[:, None], torch.arange(Sk)[None, :]
    if kind == "window":
        keep = ((iq - ik >= 0) & (iq - ik < 200)).view(1, 1, Sq, Sk)
    elif kind == "tril":
        keep = (ik <= iq).view(1, 1, Sq, Sk)
    elif kind == "documents":                    # three documents of unequal length, per batch a different split
        keep = torch.zeros(B, 1, Sq, Sk, dtype=torch.bool)
        for b_, cuts in enumerate([(0, 300, 1000, 1280), (0, 64, 65, 1280)][:B]):
            for a0, a1 in zip(cuts[:-1], cuts[1:]):
                keep[b_, 0, a0:a1, a0:a1] = True
    elif kind == "dead_block":                   # rows 256..511 see nothing at all; the others a band in the middle of the keys
        keep = torch.zeros(1, 1, Sq, Sk, dtype=torch.bool)
        keep[..., :256, 300:700] = True
        keep[..., 512:, 650:1024] = True
    elif kind == "left_padding_4d":
        keep = torch.ones(B, 1, 1, Sk, dtype=torch.bool).expand(B, 1, Sq, Sk).clone()
        keep[0, ..., :333] = False
        keep[1, ..., :64] = False
    else:                                        # per head another band
        keep = torch.zeros(1, H, Sq, Sk, dtype=torch.bool)
        for h_ in range(H):
            keep[0, h_, :, 200 * h_:200 * h_ + 300] = True
    qd, kd, vd = (t.to("cuda:0").permute(0, 2, 1, 3) for t in (q, k, v))
    md = keep.to("cuda:0")
    o32, lse = ops.fa3_forward(qd, kd, vd, mask=md, out_dtype=torch.float32, return_lse=True)
    o16, l16 = ops.fa3_forward(qd, kd, vd, mask=md, return_lse=True)
    torch.cuda.synchronize()
    name = _capi.describe(ops.build_args(qd, kd, vd, o16, mask=md)[0])[0]
    assert "kmask" in name, name                                    # the HIP kernels (element masks never reach the assembly kernel)
    full = keep.expand(B, H, Sq, Sk)
    sc = torch.einsum("bqhd,bkhd->bhqk", q.double(), k.double()) * D ** -0.5
    sc = sc.masked_fill(~full, float("-inf"))
    dead = ~full.any(dim=-1)
    ref = torch.einsum("bhqk,bkhd->bhqd", torch.softmax(sc, dim=-1).nan_to_num(0.0), v.double()).float()
    ref_lse = torch.logsumexp(sc, dim=-1).float()
    e32, e16 = float((o32.cpu() - ref).abs().max()), float((o16.float().cpu() - ref).abs().max())
    print(f"{kind} {name}: parity variant {e32:.2e}, fast {e16:.2e}, rows without a key {int(dead.sum())}")
    assert e32 <= PARITY_TOL and e16 <= 3e-2
    for l_ in (lse.cpu(), l16.cpu()):
        assert bool((torch.isinf(l_) == dead).all())
        assert float((l_ - ref_lse)[~dead].abs().max()) <= 1e-4
    assert bool((o16.float().cpu()[dead] == 0).all())


@pytest.mark.parametrize("case", [("window", 1, 4, 2, 1024, 1024, 128), ("documents", 2, 2, 1, 1280, 1280, 64), ("dead_block", 1, 2, 1, 768, 1024, 128),
                                  ("per_head", 1, 4, 2, 512, 640, 64), ("ragged_band", 2, 3, 1, 333, 777, 128)])
def test_backward_element_masks_words_and_tile_ranges(case):
    """Backward under an element mask with `mask_workspace` (ops gives it): the dQ kernel reads a word per row and key tile, the dK/dV
    kernel the transposed word per key and row tile, both run only the tile range that holds visible entries (and with grouped-query
    heads the union over the group).  Structured masks against autograd through a torch fp32 reference; also against the byte paths
    (no workspace), which must agree bit for bit."""
    import ctypes as C
    from photonic_flash_attention_amd import _capi, ops, synth
    kind, B, H, G, Sq, Sk, D = case
    q, k, v = (t.to("cuda:0").permute(0, 2, 1, 3) for t in synth.qkv(B, H, Sq, Sk, D, 8100 + Sq + D, "bf16"))
    k, v = k[:, ::G], v[:, ::G]
    g = torch.from_numpy(synth.normal_f32((B, Sq, H, D), 8200 + Sq)).to("cuda:0", q.dtype).permute(0, 2, 1, 3)
    iq, ik = torch.arange(Sq, device="cuda:0")[:, None], torch.arange(Sk, device="cuda:0")[None, :]
    if kind == "window":
        keep = ((iq - ik >= 0) & (iq - ik < 150)).view(1, 1, Sq, Sk)
    elif kind == "documents":
        keep = torch.zeros(B, 1, Sq, Sk, dtype=torch.bool, device="cuda:0")
        for b_, cuts in enumerate([(0, 300, 1000, 1280), (0, 64, 65, 1280)][:B]):
            for a0, a1 in zip(cuts[:-1], cuts[1:]):
                keep[b_, 0, a0:a1, a0:a1] = True
    elif kind == "dead_block":
        keep = torch.zeros(1, 1, Sq, Sk, dtype=torch.bool, device="cuda:0")
        keep[..., :256, 300:700] = True
        keep[..., 512:, 650:1024] = True
    elif kind == "per_head":
        keep = torch.zeros(1, H, Sq, Sk, dtype=torch.bool, device="cuda:0")
        for h_ in range(H):
            keep[0, h_, :, 100 * h_:100 * h_ + 250] = True
    else:
        keep = ((ik >= iq // 2) & (ik < iq // 2 + 300)).view(1, 1, Sq, Sk).expand(B, 1, Sq, Sk).clone()
        keep[1, :, :100] = False
    out, lse = ops.fa3_forward(q, k, v, mask=keep, return_lse=True)
    dq, dk, dv = ops.fa3_backward(q, k, v, out, g, lse, mask=keep, grad_dtype=torch.float32)
    # the same call without the scratch: the byte paths
    orig = _capi.load().pfa_fa3_bwd_mask_workspace_bytes
    try:
        _capi.load().pfa_fa3_bwd_mask_workspace_bytes = lambda *_a: 0
        dq_b, dk_b, dv_b = ops.fa3_backward(q, k, v, out, g, lse, mask=keep, grad_dtype=torch.float32)
    finally:
        _capi.load().pfa_fa3_bwd_mask_workspace_bytes = orig
    torch.cuda.synchronize()
    assert torch.equal(dq, dq_b) and torch.equal(dk, dk_b) and torch.equal(dv, dv_b), case
    qf, kf, vf = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    ke, ve = kf.repeat_interleave(G, dim=1), vf.repeat_interleave(G, dim=1)
    s = (qf @ ke.transpose(-1, -2)) * D ** -0.5
    p = torch.nan_to_num(torch.softmax(s.masked_fill(~keep.expand(B, H, Sq, Sk), float("-inf")), dim=-1), nan=0.0)
    (p @ ve).backward(g.float())
    for name, got, ref in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        scale, err = float(ref.abs().max()), float((got - ref).abs().max())
        assert bool(torch.isfinite(got).all()) and err <= 2e-2 * scale + 1e-4, (case, name, err, scale)
