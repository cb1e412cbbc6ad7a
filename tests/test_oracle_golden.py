"""Pins the oracle: our CPU restatement vs. outputs of the REAL reference
(``tests/golden/*.npz``, made by ``oracle/make_golden.py`` in the build container)."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from conftest import golden_inputs, golden_names, load_golden
from oracle import fa3_oracle as orc
from photonic_flash_attention_amd import synth

ORACLE_TOL = 2e-6  # fp32 restatement vs fp32 reference (<= 1e-6 .. 1.8e-6 measured)


@pytest.mark.parametrize("name", golden_names("full"))
def test_oracle_matches_reference_full(name):
    meta, arr = load_golden(name)
    q, k, v = golden_inputs(meta)
    lens = None if meta["kv_valid"] is None else [meta["kv_valid"]] * meta["B"]
    out = orc.attention_bshd(q, k, v, causal=meta["causal"], seqlens_k=lens)
    err = float((out - torch.from_numpy(arr["out"])).abs().max())
    assert err <= ORACLE_TOL, f"{name}: {err}"


@pytest.mark.parametrize("name", [n for n in golden_names("sampled") if "c5" not in n])
def test_oracle_matches_reference_sampled(name):
    meta, arr = load_golden(name)
    q, k, v = golden_inputs(meta)
    rows = torch.from_numpy(arr["rows"])
    for i, (b, h) in enumerate(meta["heads"]):
        out = orc.attention_bshd(q[b:b + 1, :, h:h + 1], k[b:b + 1, :, h:h + 1], v[b:b + 1, :, h:h + 1],
                                 causal=meta["causal"])[0, :, 0]
        err = float((out[rows] - torch.from_numpy(arr["out"][i])).abs().max())
        assert err <= ORACLE_TOL, f"{name} head {(b, h)}: {err}"
        assert abs(float(out.double().sum()) - arr["head_sum"][i]) <= 1e-6 * arr["head_abs_sum"][i] + 1e-3


@pytest.mark.parametrize("name", golden_names("full"))
def test_c_restatement_matches_reference_full(name):
    """oracle/fa3_oracle.c (plain C, no BLAS) against the same reference outputs."""
    import os
    import subprocess
    from conftest import REPO
    from oracle import c_oracle
    if not os.path.exists(os.path.join(REPO, "oracle", "liboracle_fa3.so")):
        subprocess.run(["make", "-C", os.path.join(REPO, "oracle")], check=True)
    meta, arr = load_golden(name)
    q, k, v = golden_inputs(meta)
    lens = None if meta["kv_valid"] is None else [meta["kv_valid"]] * meta["B"]
    out = c_oracle.attention_bshd(q, k, v, causal=meta["causal"], seqlens_k=lens)
    err = float((out - torch.from_numpy(arr["out"])).abs().max())
    assert err <= 5e-6, f"{name}: {err}"


def test_oracle_module_plumbing():
    meta, arr = load_golden("g1_c1_module")
    E, H, seed = meta["E"], meta["H"], meta["seed"]
    sd = {
        "qkv_proj.weight": torch.from_numpy(synth.normal_f32((3 * E, E), seed + 10)) * E ** -0.5,
        "qkv_proj.bias": torch.from_numpy(synth.normal_f32((3 * E,), seed + 11)) * 0.1,
        "out_proj.weight": torch.from_numpy(synth.normal_f32((E, E), seed + 12)) * E ** -0.5,
        "out_proj.bias": torch.from_numpy(synth.normal_f32((E,), seed + 13)) * 0.1,
    }
    x = torch.from_numpy(synth.normal_f32((meta["B"], meta["S"], E), seed))
    y = orc.module_forward(sd, H, x)
    assert float((y - torch.from_numpy(arr["out"])).abs().max()) <= 5e-6


def test_tile_size_rule():
    # observed values of the reference's binary search
    for s, want in [(16, 32), (128, 128), (512, 512), (513, 512), (1024, 512), (4096, 512)]:
        assert orc.optimal_tile_size(s, s, 64) == want
    assert orc.optimal_tile_size(640, 330, 64) == 330


def test_dense_and_tiled_agree_with_sdpa():
    q, k, v = synth.qkv(1, 2, 700, 700, 64, 77, "bf16")
    ref = torch.nn.functional.scaled_dot_product_attention(
        q.float().permute(0, 2, 1, 3), k.float().permute(0, 2, 1, 3), v.float().permute(0, 2, 1, 3),
        is_causal=True).permute(0, 2, 1, 3)
    out = orc.attention_bshd(q, k, v, causal=True)
    assert float((out - ref).abs().max()) <= 5e-6


def test_generator_moments_and_exactness():
    x = synth.normal_f32((1 << 18,), 5)
    assert abs(float(x.mean())) < 0.01 and abs(float(x.std()) - 1.0) < 0.01
    # values are multiples of 2**-16: exact in fp32, unique bf16 rounding
    assert np.all(x * 65536 == np.round(x * 65536))
    b = synth.round_to_bf16_bits(x)
    t = torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(b, t)


@pytest.mark.parametrize("name", golden_names("grad"))
def test_oracle_autograd_matches_reference_autograd(name):
    """The backward oracle = autograd through our restatement; pinned by gradients of the REAL reference."""
    meta, arr = load_golden(name)
    q, k, v = golden_inputs(meta)
    dout = torch.from_numpy(synth.normal_f32((meta["B"], meta["Sq"], meta["H"], meta["D"]), meta["dout_seed"])).to(torch.bfloat16)
    qf, kf, vf = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    lens = None if meta.get("kv_valid") is None else [meta["kv_valid"]] * meta["B"]
    out = orc.attention_bshd(qf, kf, vf, causal=meta["causal"], seqlens_k=lens)
    (out * dout.float()).sum().backward()
    assert float((out.detach() - torch.from_numpy(arr["out"])).abs().max()) <= ORACLE_TOL
    for g, key in ((qf.grad, "dq"), (kf.grad, "dk"), (vf.grad, "dv")):
        ref = torch.from_numpy(arr[key])
        assert float((g - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), key
