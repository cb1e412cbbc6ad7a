#!/usr/bin/env python3
"""ORACLE tooling (test infrastructure): generate ``tests/golden/*.npz`` from the REAL reference.

Run in the build container only (it needs ``/root/reference``; the GPU box has none):

    PHOTONIC_LOG_LEVEL=CRITICAL PYTHONDONTWRITEBYTECODE=1 python3 oracle/make_golden.py

What is stored is data only: shapes/seeds (the inputs are re-generated bit-exactly by
``photonic_flash_attention_amd.synth`` and pinned by a checksum), and the outputs of

    photonic_flash_attention.core.flash_attention_3.FlashAttention3._flash_attention_forward
    photonic_flash_attention.core.flash_attention_3.FlashAttention3.forward            (G1 only)
    torch.autograd through _flash_attention_forward (G8: dq, dk, dv for a fixed dout)

run in fp32 on the CPU on the bf16-rounded inputs.  No reference
source text is copied anywhere.
"""

from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF_SRC = "/root/reference/src"
if not os.path.isdir(REF_SRC):
    sys.exit("make_golden.py needs /root/reference (build container only)")
os.environ.setdefault("PHOTONIC_LOG_LEVEL", "CRITICAL")
sys.path.insert(0, REF_SRC)

from photonic_flash_attention.core.flash_attention_3 import FlashAttention3 as RefFA3  # noqa: E402

from photonic_flash_attention_amd import synth  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def ref_core(q, k, v, mask):
    """q,k,v: [B,S,H,D] (bf16-rounded) -> reference output [B,Sq,H,D] fp32."""
    D = q.shape[-1]
    H = q.shape[2]
    m = RefFA3(H * D, H).eval()
    with torch.no_grad():
        o, _ = m._flash_attention_forward(
            q.float().permute(0, 2, 1, 3), k.float().permute(0, 2, 1, 3), v.float().permute(0, 2, 1, 3),
            mask, False)
    return o.permute(0, 2, 1, 3).contiguous()


def input_checksum(q, k, v):
    return [synth.checksum(t.view(torch.int16).numpy().view(np.uint16)) for t in (q, k, v)]


def tril(Sq, Sk):
    return torch.tril(torch.ones(Sq, Sk, dtype=torch.bool)).view(1, 1, Sq, Sk)


ONLY = [a for a in sys.argv[1:] if not a.startswith("-")]      # optional: regenerate only fixtures whose name starts so


def wanted(name):
    return not ONLY or any(name.startswith(o) for o in ONLY)


def save(name, meta, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB  {meta.get('note', '')}")


def full_case(name, B, H, Sq, Sk, D, seed, causal=False, kv_valid=None, note=""):
    if not wanted(name):
        return
    q, k, v = synth.qkv(B, H, Sq, Sk, D, seed, "bf16")
    mask = None
    if causal:
        mask = tril(Sq, Sk).expand(B, 1, Sq, Sk)
    if kv_valid is not None:
        kp = (torch.arange(Sk) < kv_valid).view(1, 1, 1, Sk).expand(B, 1, Sq, Sk)
        mask = kp if mask is None else (mask & kp)
    t = time.time()
    o = ref_core(q, k, v, mask)
    meta = dict(kind="full", B=B, H=H, Sq=Sq, Sk=Sk, D=D, seed=seed, causal=causal, kv_valid=kv_valid,
                dtype="bf16", layout="BSHD", in_checksum=input_checksum(q, k, v),
                ref_seconds=round(time.time() - t, 3), note=note)
    save(name, meta, out=o.numpy())


def sampled_case(name, B, H, S, D, seed, causal, heads, rows, note=""):
    """Large shapes: run the reference on single (b,h) problems (heads are independent,
    flash_attention_3.py:162/231 batch over them) and keep sampled rows + whole-head sums."""
    if not wanted(name):
        return
    q, k, v = synth.qkv(B, H, S, S, D, seed, "bf16")
    mask = tril(S, S) if causal else None
    outs, sums, asums = [], [], []
    t = time.time()
    for (b, h) in heads:
        o = ref_core(q[b:b + 1, :, h:h + 1], k[b:b + 1, :, h:h + 1], v[b:b + 1, :, h:h + 1], mask)[0, :, 0]
        outs.append(o[rows].numpy())
        sums.append(float(o.double().sum()))
        asums.append(float(o.double().abs().sum()))
    meta = dict(kind="sampled", B=B, H=H, Sq=S, Sk=S, D=D, seed=seed, causal=causal, kv_valid=None,
                dtype="bf16", layout="BSHD", in_checksum=input_checksum(q, k, v),
                heads=[list(x) for x in heads], ref_seconds=round(time.time() - t, 3), note=note)
    save(name, meta, out=np.stack(outs), rows=np.asarray(rows, dtype=np.int64),
         head_sum=np.asarray(sums), head_abs_sum=np.asarray(asums))


def grad_case(name, B, H, Sq, Sk, D, seed, causal, note="", kv_valid=None):
    """Gradients of the REAL reference's core by autograd (the only backward the reference has):
    loss = sum(out * dout) with a fixed dout, so dq/dk/dv are the vector-Jacobian products the HIP backward returns."""
    if not wanted(name):
        return
    q, k, v = synth.qkv(B, H, Sq, Sk, D, seed, "bf16")
    dout = torch.from_numpy(synth.normal_f32((B, Sq, H, D), seed + 5)).to(torch.bfloat16)
    m = RefFA3(H * D, H).eval()
    qf, kf, vf = (t.float().permute(0, 2, 1, 3).clone().requires_grad_(True) for t in (q, k, v))
    mask = tril(Sq, Sk).expand(B, 1, Sq, Sk) if causal else None
    if kv_valid is not None:
        kp = (torch.arange(Sk) < kv_valid).view(1, 1, 1, Sk).expand(B, 1, Sq, Sk)
        mask = kp if mask is None else (mask & kp)
    o, _ = m._flash_attention_forward(qf, kf, vf, mask, False)
    (o * dout.float().permute(0, 2, 1, 3)).sum().backward()
    meta = dict(kind="grad", B=B, H=H, Sq=Sq, Sk=Sk, D=D, seed=seed, causal=causal, kv_valid=kv_valid, dtype="bf16",
                layout="BSHD", in_checksum=input_checksum(q, k, v), dout_seed=seed + 5, note=note)
    save(name, meta, out=o.detach().permute(0, 2, 1, 3).contiguous().numpy(),
         dq=qf.grad.permute(0, 2, 1, 3).contiguous().numpy(), dk=kf.grad.permute(0, 2, 1, 3).contiguous().numpy(),
         dv=vf.grad.permute(0, 2, 1, 3).contiguous().numpy())


def module_case(name, B, S, E, H, seed):
    """G1b: whole-module plumbing (fused QKV chunk order, head split/merge, out_proj)."""
    if not wanted(name):
        return
    m = RefFA3(E, H).eval()
    sd = {
        "qkv_proj.weight": torch.from_numpy(synth.normal_f32((3 * E, E), seed + 10)) * E ** -0.5,
        "qkv_proj.bias": torch.from_numpy(synth.normal_f32((3 * E,), seed + 11)) * 0.1,
        "out_proj.weight": torch.from_numpy(synth.normal_f32((E, E), seed + 12)) * E ** -0.5,
        "out_proj.bias": torch.from_numpy(synth.normal_f32((E,), seed + 13)) * 0.1,
    }
    m.load_state_dict(sd)
    x = torch.from_numpy(synth.normal_f32((B, S, E), seed))
    with torch.no_grad():
        y, w = m(x)
    assert w is None
    meta = dict(kind="module", B=B, S=S, E=E, H=H, seed=seed, dtype="fp32",
                note="FlashAttention3.forward fp32 self-attention; weights from synth seeds +10..+13")
    save(name, meta, out=y.numpy())


def main():
    torch.manual_seed(0)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    # G1: BASELINE config 1 (dense branch) + module plumbing
    full_case("g1_c1_core", 2, 4, 128, 128, 64, 1001, note="C1 B2 S128 H4 D64, dense branch")
    module_case("g1_c1_module", 2, 128, 256, 4, 1101)
    # G2: crosses the 512 tile edge (tiled branch), D=64
    full_case("g2_s640_d64", 1, 2, 640, 640, 64, 1002, note="tiled branch, no mask")
    full_case("g2_s640_d64_causal", 1, 2, 640, 640, 64, 1003, causal=True, note="tiled branch, causal")
    # G3: D=128 causal
    full_case("g3_s1024_d128_causal", 1, 2, 1024, 1024, 128, 1004, causal=True)
    full_case("g3_s1024_d128", 1, 2, 1024, 1024, 128, 1005)
    # G4: key-padding tail (last 64 keys masked) as a 4-D mask
    full_case("g4_s640_d64_kvtail", 2, 2, 640, 640, 64, 1006, kv_valid=576, note="keys >= 576 masked")
    full_case("g4_s640_d128_kvtail_causal", 1, 2, 640, 640, 128, 1007, causal=True, kv_valid=601,
              note="keys >= 601 masked + causal")
    # G5: cross attention Sq != Sk (tests/unit/test_flash_attention_3.py:117-135 shape class)
    full_case("g5_cross_640x330_d64", 1, 2, 640, 330, 64, 1008, note="Sq=640 Sk=330")
    full_case("g5_cross_200x777_d128", 1, 3, 200, 777, 128, 1009, note="Sq=200 Sk=777, ragged tails")
    # dense-branch small / ragged shapes from the reference's fixture classes (conftest.py:31-38)
    full_case("g7_s97_d64", 2, 3, 97, 97, 64, 1010, note="ragged S, dense branch")
    full_case("g7_s512_d64_causal", 1, 4, 512, 512, 64, 1011, causal=True, note="S = tile edge, dense branch")
    # G8: gradients (autograd through the reference core), dense and tiled branch
    grad_case("g8_grad_s128_d128", 2, 2, 128, 128, 128, 3001, False, note="dense branch, autograd")
    grad_case("g8_grad_s640_d64_causal", 1, 2, 640, 640, 64, 3002, True, note="tiled branch, causal, autograd")
    grad_case("g8_grad_cross_200x333_d128", 1, 2, 200, 333, 128, 3003, False, note="Sq != Sk")
    grad_case("g8_grad_s320_d64_kvtail_causal", 1, 2, 320, 320, 64, 3004, True, note="keys >= 290 masked (4-D mask) + causal",
              kv_valid=290)
    # G6: BASELINE-shaped problems, sampled
    rows_1k = sorted(set(list(range(0, 8)) + list(range(500, 520)) + list(range(1016, 1024))))
    sampled_case("g6_c2", 4, 12, 1024, 64, 2002, False, [(0, 0), (1, 5), (3, 11)], rows_1k,
                 note="C2 B4 S1024 H12 D64")
    rows_4k = sorted(set(list(range(0, 16)) + list(range(250, 262)) + list(range(2040, 2056)) + list(range(4080, 4096))))
    sampled_case("g6_c3", 4, 16, 4096, 128, 2003, True, [(0, 0), (2, 7), (3, 15)], rows_4k,
                 note="C3 B4 S4096 H16 D128 causal")
    sampled_case("g6_c4", 4, 16, 4096, 128, 2004, False, [(0, 3), (3, 12)], rows_4k,
                 note="C4 per-GPU shard B4 S4096 H16 D128 non-causal")
    rows_16k = sorted(set(list(range(0, 8)) + list(range(8190, 8200)) + list(range(16376, 16384))))
    sampled_case("g6_c5", 1, 32, 16384, 128, 2005, True, [(0, 0), (0, 31)], rows_16k,
                 note="C5 B1 S16384 H32 D128 causal")


if __name__ == "__main__":
    main()
