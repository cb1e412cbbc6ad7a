"""``convert_to_photonic``: swap a model's ``torch.nn.MultiheadAttention`` layers for the MI355X attention block.

Counterpart of the reference's ``integration/pytorch/convert.py`` (``ModelConverter`` :205-520,
``convert_to_photonic`` :527-557) for what can work offline and is on this build's path:

* handled: ``torch.nn.MultiheadAttention`` with a packed ``in_proj_weight`` (the PyTorch rule of the reference,
  :441-452: ``in_proj_weight -> qkv_proj.weight``, ``in_proj_bias -> qkv_proj.bias``, ``out_proj`` copied), found
  anywhere in the module tree and replaced in place, e.g. inside ``nn.TransformerEncoderLayer``;
* handled: Hugging Face ``transformers`` model OBJECTS (BERT, GPT-2, Llama-style ...) by registering the kernel as an
  attention implementation (``hf.py``) -- the counterpart of the reference's per-architecture weight copies (:389-450);
* not handled (reported in ``skipped_layers``): ``kdim/vdim`` != ``embed_dim``, ``add_bias_kv``, ``add_zero_attn``;
* a model *name* (string) would mean ``AutoModel.from_pretrained`` = a network fetch (reference :545): refused.

The replacement keeps **PyTorch's** call convention (``key_padding_mask`` / boolean ``attn_mask``: True = masked;
float ``attn_mask``: additive), translating it to the reference convention (0 = masked) the kernel wrappers use,
so the surrounding ``nn.TransformerEncoderLayer`` code runs unchanged (its fused "fast path" is bypassed because the
layer is no longer an ``nn.MultiheadAttention``).
"""

from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from .modules import PhotonicMultiHeadAttention


@dataclass
class ConversionReport:
    """Subset of the reference's report (convert.py:60-75) that is meaningful here."""
    original_model_name: str
    converted_layers: List[str] = field(default_factory=list)
    skipped_layers: List[str] = field(default_factory=list)
    conversion_errors: List[str] = field(default_factory=list)
    compatibility_warnings: List[str] = field(default_factory=list)


class TorchMHAReplacement(PhotonicMultiHeadAttention):
    """``PhotonicMultiHeadAttention`` that speaks ``nn.MultiheadAttention``'s mask dialect."""

    # attributes nn.TransformerEncoderLayer inspects before choosing its fused path
    _qkv_same_embed_dim = True
    in_proj_weight = None
    in_proj_bias = None

    @staticmethod
    def _to_keep_mask(mask: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """PyTorch mask (bool: True = drop; float: additive) -> keep-mask (non-zero = attend).

        ``nn.MultiheadAttention`` ADDS a float mask to the scores.  The kernel takes binary masks only, so a float mask is
        accepted when it is binary in effect -- every entry 0 (attend) or at most -1e4 / finfo.min / -inf (drop), the same
        threshold as ``hf._keep_mask`` -- and refused otherwise: finite biases (ALiBi, relative position terms) would be
        dropped silently."""
        if mask is None:
            return None
        if mask.dtype == torch.bool:
            return ~mask
        keep = mask > -1.0
        soft = (keep & (mask != 0)) | (~keep & (mask > -1e4))
        if bool(soft.any()):
            raise ValueError("float attn_mask / key_padding_mask with finite biases is not supported by the converted layer: "
                             "entries must be 0 (attend) or <= -1e4 / -inf (masked); additive biases such as ALiBi or "
                             "relative-position terms cannot be expressed as a binary mask")
        return keep

    def forward(self, query, key, value, key_padding_mask=None, need_weights=True, attn_mask=None,
                average_attn_weights=True, is_causal=False):
        keep_kp = self._to_keep_mask(key_padding_mask)
        keep_am = None
        if attn_mask is not None and not is_causal:
            keep_am = self._to_keep_mask(attn_mask)
            if keep_am.dim() == 2:                        # [L, S] -> broadcast over batch and heads
                keep_am = keep_am[None, None]
            elif keep_am.dim() == 3:                      # [N*H, L, S]
                n = (query.shape[0] if self.batch_first else query.shape[1])
                keep_am = keep_am.view(n, self.num_heads, *keep_am.shape[1:])
        if keep_am is not None and keep_kp is not None:
            keep_am = keep_am & keep_kp[:, None, None, :]
            keep_kp = None
        # parent: masks use 0 = masked; a lone key-padding mask takes the cheap [B,Sk] path
        return super().forward(query, key, value, key_padding_mask=keep_kp, need_weights=need_weights,
                               attn_mask=keep_am, average_attn_weights=average_attn_weights, is_causal=is_causal)


def _replacement_for(mha: nn.MultiheadAttention, dtype: Optional[torch.dtype]) -> TorchMHAReplacement:
    w = mha.in_proj_weight
    new = TorchMHAReplacement(mha.embed_dim, mha.num_heads, dropout=mha.dropout, bias=mha.in_proj_bias is not None,
                              batch_first=mha.batch_first, device=w.device, dtype=dtype or w.dtype)
    core = new.gpu_attention
    with torch.no_grad():
        core.qkv_proj.weight.copy_(mha.in_proj_weight)
        core.out_proj.weight.copy_(mha.out_proj.weight)
        if mha.in_proj_bias is not None:
            core.qkv_proj.bias.copy_(mha.in_proj_bias)
            core.out_proj.bias.copy_(mha.out_proj.bias)
    new.train(mha.training)
    return new


def convert_to_photonic(model, dtype: Optional[torch.dtype] = None, inplace: bool = False,
                        **_unused) -> Tuple[nn.Module, ConversionReport]:
    """Return ``(converted_model, report)``.  ``model`` must be an ``nn.Module`` (no hub/network loading)."""
    if isinstance(model, str):
        raise ValueError("convert_to_photonic needs an nn.Module: loading a model by name would fetch it from the "
                         "network (reference convert.py:545), which this build never does")
    report = ConversionReport(original_model_name=type(model).__name__)
    if not inplace:
        model = copy.deepcopy(model)
    if hasattr(getattr(model, "config", None), "_attn_implementation"):
        # a transformers model OBJECT: its attention goes through the pluggable function table (integration/pytorch/hf.py);
        # weights stay where they are, so none of the reference's per-architecture copy rules (:389-450) is needed
        from .hf import convert_hf_model, IMPLEMENTATION_NAME
        convert_hf_model(model)
        report.converted_layers.append(f"<all attention layers via attn_implementation={IMPLEMENTATION_NAME!r}>")
        return model, report
    if isinstance(model, nn.MultiheadAttention):
        targets = [("", None, model)]
    else:
        targets = [(f"{pname}.{cname}".lstrip("."), parent, child)
                   for pname, parent in model.named_modules()
                   for cname, child in parent.named_children() if isinstance(child, nn.MultiheadAttention)]
    for path, parent, mha in targets:
        why = None
        if not getattr(mha, "_qkv_same_embed_dim", False) or mha.in_proj_weight is None:
            why = "kdim/vdim differ from embed_dim"
        elif mha.bias_k is not None or mha.bias_v is not None or mha.add_zero_attn:
            why = "add_bias_kv / add_zero_attn"
        elif mha.head_dim > 128:
            why = f"head_dim {mha.head_dim} has no kernel (<= 128)"
        if why:
            report.skipped_layers.append(path)
            report.compatibility_warnings.append(f"{path or '<root>'}: {why}")
            continue
        try:
            new = _replacement_for(mha, dtype)
            if parent is None:
                model = new
            else:
                setattr(parent, path.rsplit(".", 1)[-1], new)
            report.converted_layers.append(path)
        except Exception as exc:  # noqa: BLE001  (the reference also records and continues, :261-266)
            report.conversion_errors.append(f"{path}: {exc}")
            report.skipped_layers.append(path)
    return model, report
