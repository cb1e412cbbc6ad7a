"""Hugging Face models on the MI355X attention kernels, without touching their modules.

The reference converts HF models by swapping attention layers and copying weights by name
(``integration/pytorch/convert.py:389-450``: BERT ``query/key/value``, GPT-2 ``c_attn/c_proj``, T5 ``q/k/v/o``), and
loads them by name from the hub (``:545``).  Current ``transformers`` (>= 4.48; 5.x here) route every model's attention
through a pluggable function table instead (``transformers.AttentionInterface``): a model whose
``config._attn_implementation`` names a registered function calls it with ``query/key/value`` already projected and
split into heads.  Registering the HIP kernel there converts BERT, GPT-2, Llama-style models ... alike, keeps their
weights where they are, and is autograd-aware through ``ops.fa3_attention``.

Only model OBJECTS are handled (``convert_hf_model(model)``); nothing is ever fetched from the network.
"""

from __future__ import annotations

from typing import Optional

import torch

from ... import ops

IMPLEMENTATION_NAME = "pfa_hip"


def _keep_mask(attention_mask: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """HF hands sdpa-style masks: bool (True = attend) or float additive (0 = attend, large negative = masked)."""
    if attention_mask is None:
        return None
    if attention_mask.dtype == torch.bool:
        return attention_mask
    return attention_mask > -1.0          # additive masks are 0 or <= -1e4 / finfo.min / -inf


def pfa_attention_mask(batch_size: int, q_length: int, kv_length: int, q_offset: int = 0, kv_offset: int = 0, mask_function=None,
                       attention_mask: Optional[torch.Tensor] = None, **kwargs):
    """``AttentionMaskInterface`` function.  For the two plain patterns -- causal or bidirectional over a prefill (every query has its own
    key position: ``q_length == kv_length``, no cache offset) -- the kernels need no ``[B,1,Sq,Sk]`` tensor at all: the causal triangle is a
    flag and the padding is the caller's 2-D ``[B, Sk]`` mask, which is handed on as it is (``None`` when there is no padding mask).  That
    is what sends a padded decoder batch to the persistent kernels (key-mask flavour, tile counts cut to each batch's keys) instead of the
    element-mask path.  Everything else (cached decoding, sliding windows, packed sequences, user mask functions) gets sdpa's 4-D mask."""
    from transformers import masking_utils as mu
    plain = mask_function in (getattr(mu, "causal_mask_function", None), getattr(mu, "bidirectional_mask_function", None), None)
    if plain and q_length == kv_length and q_length > 1 and not q_offset and not kv_offset:
        if attention_mask is None:
            return None
        am = attention_mask[:, -kv_length:]
        return am if am.shape[1] == kv_length else mu.sdpa_mask(batch_size=batch_size, q_length=q_length, kv_length=kv_length, q_offset=q_offset,
                                                                  kv_offset=kv_offset, mask_function=mask_function, attention_mask=attention_mask, **kwargs)
    kw = dict(batch_size=batch_size, q_length=q_length, kv_length=kv_length, q_offset=q_offset, kv_offset=kv_offset, attention_mask=attention_mask)
    if mask_function is not None:
        kw["mask_function"] = mask_function
    return mu.sdpa_mask(**kw, **kwargs)


def pfa_attention_forward(module, query, key, value, attention_mask, dropout: float = 0.0,
                          scaling: Optional[float] = None, is_causal: Optional[bool] = None, **kwargs):
    """``AttentionInterface`` function: ``query/key/value`` are ``[B, H(kv), S, D]``; returns ``([B, Sq, H, D], None)``.

    ``attention_mask``: ``None`` (the causal flag decides), the 2-D ``[B, Sk]`` padding mask of ``pfa_attention_mask`` (plus the causal
    flag), or an sdpa-style 4-D mask (then the flag is off, as in ``transformers.integrations.sdpa_attention.sdpa_attention_forward``)."""
    if kwargs.get("output_attentions", False):
        raise NotImplementedError(f"'{IMPLEMENTATION_NAME}' attention does not return attention weights; use 'eager'")
    if dropout and getattr(module, "training", False):
        raise NotImplementedError("attention dropout in training mode is not implemented on the HIP path")
    # (grouped-query attention: forward and backward read the shared K/V heads in place -- pfa_fa3_args.kv_group, and since ABI v7
    #  pfa_fa3_bwd_args.kv_group: the dK/dV kernel sums over each group in registers; nothing is expanded)
    if query.shape[-1] > 128:
        raise NotImplementedError(f"head_dim {query.shape[-1]} has no kernel (<= 128)")
    q_len, k_len = query.shape[2], key.shape[2]
    want_causal = is_causal if is_causal is not None else getattr(module, "is_causal", True)
    key_mask = keep = None
    if attention_mask is not None and attention_mask.dim() == 2:
        # the compact form: padding as a [B, Sk] key mask, the triangle as the flag (prefill only: q_len == k_len, see pfa_attention_mask)
        key_mask = attention_mask if attention_mask.dtype == torch.bool else attention_mask != 0
        causal = bool(want_causal and q_len > 1 and q_len == k_len)
        if want_causal and q_len > 1 and q_len != k_len:
            raise NotImplementedError("a 2-D mask with q_len != k_len under the causal mask: pfa_attention_mask never produces this")
    else:
        causal = bool(q_len > 1 and attention_mask is None and want_causal)
        keep = _keep_mask(attention_mask)
    in_dtype = query.dtype
    cd = in_dtype if in_dtype in (torch.bfloat16, torch.float16) else torch.bfloat16
    out = ops.fa3_attention(query.to(cd), key.to(cd), value.to(cd), causal=causal, mask=keep, key_mask=key_mask, softmax_scale=scaling,
                            out_dtype=in_dtype)
    return out.transpose(1, 2), None       # [B,H,S,D] view of a [B,S,H,D] buffer -> contiguous [B,S,H,D]


def register_hf_attention(name: str = IMPLEMENTATION_NAME) -> str:
    """Register the kernel as a ``transformers`` attention implementation (idempotent).  Masks are built like sdpa's."""
    from transformers import AttentionInterface, AttentionMaskInterface
    AttentionInterface.register(name, pfa_attention_forward)
    AttentionMaskInterface.register(name, pfa_attention_mask)
    return name


def convert_hf_model(model, name: str = IMPLEMENTATION_NAME):
    """Switch a ``transformers.PreTrainedModel`` OBJECT to the HIP attention path in place and return it."""
    register_hf_attention(name)
    if hasattr(model, "set_attn_implementation"):
        model.set_attn_implementation(name)
    else:                                   # older transformers
        model.config._attn_implementation = name
    return model
