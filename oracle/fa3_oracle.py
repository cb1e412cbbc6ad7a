"""ORACLE (test infrastructure, never shipped): CPU fp32 restatement of the reference's
electronic attention core.

This file restates, in our own words, the arithmetic of
``/root/reference/src/photonic_flash_attention/core/flash_attention_3.py``:

* ``optimal_tile_size``      <- ``_compute_optimal_tile_size``  (:264-293)
* ``standard_attention``     <- ``_standard_attention``         (:152-180)
* ``tiled_attention``        <- ``_tiled_attention``            (:182-262)
* ``flash_attention_forward``<- ``_flash_attention_forward``    (:120-150)
* ``module_forward``         <- ``FlashAttention3.forward``     (:49-118), plumbing only

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  The product package must never route through it.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the real reference in the build
container, runs it on the generator's inputs and stores its outputs under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement against those
vectors (<= 2e-6 max-abs) on every run, with no access to ``/root/reference``.

The reference's arithmetic *is* PyTorch's (matmul / exp / max / softmax on the CPU), so
the restatement uses the same primitives, in fp32, on bf16-rounded inputs
(run in bf16 the reference keeps bf16 accumulators and is 7e-3..1.4e-2
away from fp32 truth, so fp32 is the only meaningful target for a <= 1e-3 gate).

Parity domain: no mask, causal (4-D lower-triangular mask), key-padding
tails.  Outside it the reference has quirks we do not reproduce (fully masked rows give
NaN in the dense branch, zeros or poisoned tiles in the tiled one).
"""

from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

DEFAULT_MEMORY_BYTES = 8e9  # flash_attention_3.py:142 (no CUDA device visible)
MAX_MEMORY_USAGE = 0.8  # config.py:18


def optimal_tile_size(seq_len_q: int, seq_len_k: int, head_dim: int,
                      available_memory: float = DEFAULT_MEMORY_BYTES,
                      max_memory_usage: float = MAX_MEMORY_USAGE) -> int:
    """Largest tile in [32, min(Sq, Sk, 512)] whose fp32 working set fits the budget
    (flash_attention_3.py:264-293).  With >= 8 GB the bound never binds."""
    def need(t: int) -> float:
        return (t * head_dim + t * seq_len_k + t) * 4

    lo, hi = 32, min(seq_len_q, seq_len_k, 512)
    budget = available_memory * max_memory_usage
    while lo < hi:
        mid = (lo + hi + 1) // 2
        if need(mid) <= budget:
            lo = mid
        else:
            hi = mid - 1
    return max(lo, 32)


def causal_mask(seq_len_q: int, seq_len_k: int) -> torch.Tensor:
    """The reference has no ``is_causal``; causality is a 4-D mask, 0 = masked
    (flash_attention_3.py:168,236).  Top-left aligned: key j visible to row i iff j <= i."""
    return torch.tril(torch.ones(seq_len_q, seq_len_k, dtype=torch.bool)).view(1, 1, seq_len_q, seq_len_k)


def key_padding_mask_4d(seqlens_k, seq_len_k: int) -> torch.Tensor:
    """[B] valid lengths -> 4-D mask [B,1,1,Sk] (what a 2-D [B,Sk] mask becomes at :166-167)."""
    lens = torch.as_tensor(seqlens_k).view(-1, 1)
    return (torch.arange(seq_len_k).view(1, -1) < lens).view(-1, 1, 1, seq_len_k)


def standard_attention(q, k, v, mask: Optional[torch.Tensor] = None, need_weights: bool = False):
    """Dense branch (S <= tile), flash_attention_3.py:152-180.  ``q`` is already scaled."""
    scores = torch.matmul(q, k.transpose(-2, -1))
    if mask is not None:
        if mask.dim() == 2:
            mask = mask[:, None, None, :]
        scores = scores.masked_fill(mask == 0, float("-inf"))
    w = torch.softmax(scores, dim=-1)
    out = torch.matmul(w, v)
    return out, (w if need_weights else None)


def tiled_attention(q, k, v, mask: Optional[torch.Tensor] = None, tile: int = 512):
    """Two-level tiled online softmax, flash_attention_3.py:182-262.  ``q`` is already scaled.

    Follows the reference step for step: the running output is kept *normalised* after
    every K/V tile (``O = (e_old * O + P V) / l_new``, :250) rather than once at the end.
    The host-sync guard at :249 (``if l_new.sum() > 0``) is always true inside the parity
    domain and is not restated."""
    B, H, Sq, D = q.shape
    Sk = k.shape[2]
    out = torch.zeros_like(q)
    for i in range(0, Sq, tile):
        ie = min(i + tile, Sq)
        qi = q[:, :, i:ie]
        m = torch.full((B, H, ie - i), float("-inf"), dtype=q.dtype, device=q.device)
        l = torch.zeros((B, H, ie - i), dtype=q.dtype, device=q.device)
        o = torch.zeros((B, H, ie - i, D), dtype=q.dtype, device=q.device)
        for j in range(0, Sk, tile):
            je = min(j + tile, Sk)
            s = torch.matmul(qi, k[:, :, j:je].transpose(-2, -1))
            if mask is not None:
                s = s.masked_fill(mask[:, :, i:ie, j:je] == 0, float("-inf"))
            m_new = torch.maximum(m.unsqueeze(-1), s.max(dim=-1, keepdim=True).values)
            p = torch.exp(s - m_new)
            e_old = torch.exp(m.unsqueeze(-1) - m_new) * l.unsqueeze(-1)
            l_new = e_old.sum(dim=-1) + p.sum(dim=-1)
            o = (e_old * o + torch.matmul(p, v[:, :, j:je])) / l_new.unsqueeze(-1)
            m = m_new.squeeze(-1)
            l = l_new
        out[:, :, i:ie] = o
    return out


def flash_attention_forward(q, k, v, mask: Optional[torch.Tensor] = None,
                            scaling: Optional[float] = None,
                            available_memory: float = DEFAULT_MEMORY_BYTES) -> torch.Tensor:
    """Core seam, flash_attention_3.py:120-150.  q,k,v: ``[B,H,S,D]`` fp32 (any strides).

    ``mask``: None, or a 4-D tensor broadcastable to ``[B,H,Sq,Sk]`` with 0 = masked
    (2-D ``[B,Sk]`` masks are accepted only on the dense branch, as in the reference)."""
    q = q.float()
    k = k.float()
    v = v.float()
    D = q.shape[-1]
    if scaling is None:
        scaling = D ** -0.5
    q = q * scaling  # :138
    Sq, Sk = q.shape[2], k.shape[2]
    tile = optimal_tile_size(Sq, Sk, D, available_memory)
    if Sq <= tile and Sk <= tile:
        return standard_attention(q, k, v, mask)[0]
    if mask is not None and mask.dim() != 4:
        raise IndexError("tiled branch needs a 4-D mask (flash_attention_3.py:235)")
    if mask is not None and (mask.shape[2] != Sq or mask.shape[3] != Sk):
        # the reference slices dims 2/3 directly (:235) and needs them materialised
        mask = mask.expand(mask.shape[0], mask.shape[1], Sq, Sk)
    return tiled_attention(q, k, v, mask, tile)


def attention_bshd(q, k, v, causal: bool = False, seqlens_k=None,
                   scaling: Optional[float] = None) -> torch.Tensor:
    """Convenience for the parity tests: operands in the kernel's ``[B,S,H,D]`` layout
    (any float dtype, up-cast to fp32), mask given the way the C ABI takes it.
    Returns fp32 ``[B,Sq,H,D]``."""
    qh, kh, vh = (t.float().permute(0, 2, 1, 3) for t in (q, k, v))
    Sq, Sk = qh.shape[2], kh.shape[2]
    mask = None
    if causal:
        mask = causal_mask(Sq, Sk)
    if seqlens_k is not None:
        kp = key_padding_mask_4d(seqlens_k, Sk)
        mask = kp if mask is None else (mask & kp)
    if mask is not None:
        mask = mask.expand(qh.shape[0], 1, Sq, Sk)
    out = flash_attention_forward(qh, kh, vh, mask, scaling)
    return out.permute(0, 2, 1, 3).contiguous()


def lse_bshd(q, k, causal: bool = False, seqlens_k=None, scaling: Optional[float] = None) -> torch.Tensor:
    """Natural-log row log-sum-exp of the scaled scores, fp64 accumulate -> fp32 ``[B,H,Sq]``.
    Not a reference quantity (the reference never exposes it); used to check the kernel's
    optional LSE output."""
    qh = q.double().permute(0, 2, 1, 3)
    kh = k.double().permute(0, 2, 1, 3)
    if scaling is None:
        scaling = q.shape[-1] ** -0.5
    s = torch.matmul(qh, kh.transpose(-2, -1)) * scaling
    Sq, Sk = s.shape[-2:]
    if causal:
        s = s.masked_fill(~causal_mask(Sq, Sk), float("-inf"))
    if seqlens_k is not None:
        s = s.masked_fill(~key_padding_mask_4d(seqlens_k, Sk), float("-inf"))
    return torch.logsumexp(s, dim=-1).float()


def module_forward(state_dict, num_heads: int, query, key=None, value=None, mask=None) -> torch.Tensor:
    """Plumbing of ``FlashAttention3.forward`` (:85-110) on explicit weights: fused QKV
    projection (chunk order q,k,v), head split, core, head merge, output projection."""
    W, b = state_dict["qkv_proj.weight"].float(), state_dict.get("qkv_proj.bias")
    Wo, bo = state_dict["out_proj.weight"].float(), state_dict.get("out_proj.bias")
    B, S, E = query.shape
    D = E // num_heads
    lin = torch.nn.functional.linear
    if key is None and value is None:
        q, k, v = lin(query.float(), W, None if b is None else b.float()).chunk(3, dim=-1)
    else:
        key = query if key is None else key
        value = query if value is None else value
        bq = None if b is None else b.float()
        q = lin(query.float(), W, bq)[..., :E]
        k = lin(key.float(), W, bq)[..., E:2 * E]
        v = lin(value.float(), W, bq)[..., 2 * E:]
    q = q.reshape(B, S, num_heads, D).transpose(1, 2)
    k = k.reshape(B, -1, num_heads, D).transpose(1, 2)
    v = v.reshape(B, -1, num_heads, D).transpose(1, 2)
    o = flash_attention_forward(q, k, v, mask, D ** -0.5)
    o = o.transpose(1, 2).reshape(B, S, E)
    return lin(o, Wo, None if bo is None else bo.float())


def attention_flops(B: int, H: int, Sq: int, Sk: int, D: int, causal: bool) -> float:
    """4*B*H*Sq*Sk*D (QK^T + PV, fma = 2), halved for causal (SURVEY.md §8 notation)."""
    f = 4.0 * B * H * Sq * Sk * D
    return f / 2 if causal else f
