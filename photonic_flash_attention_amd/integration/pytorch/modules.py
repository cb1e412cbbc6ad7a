"""Public drop-in modules on top of the MI355X electronic branch.

API mirror of the reference's ``integration/pytorch/modules.py`` (``PhotonicFlashAttention`` :12-233,
``PhotonicMultiHeadAttention`` :235-336): constructor arguments, ``forward`` signatures, return conventions
(a tensor, or ``(tensor, weights)`` when ``need_weights``), the public knobs (``set_photonic_threshold``,
``enable_photonic``, ``reset_performance_history``, ``get_performance_stats``) and the ``last_*`` attributes
are kept; the implementation is not.  There is exactly one compute branch here -- the HIP kernel behind
``FlashAttention3`` -- so device selection (reference :118-143) collapses to the constant ``"gpu"``, which is
also what the reference does on a machine without photonic hardware (:120-121).
"""

from __future__ import annotations

from collections import deque
from typing import Deque, Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from ...config import get_config
from ...core.flash_attention_3 import FlashAttention3
from ...photonic.hardware.detection import is_photonic_available

_HISTORY_LEN = 100   # reference keeps the last 100 calls (:185-187)


class _CallLog:
    """Bounded per-call record (device, latency, energy) behind ``get_performance_stats``."""

    def __init__(self, maxlen: int = _HISTORY_LEN):
        self._rows: Deque[Dict[str, float]] = deque(maxlen=maxlen)

    def add(self, device: str, latency_ms: float, energy_mj: float) -> None:
        self._rows.append({"device": device, "latency_ms": latency_ms, "energy_mj": energy_mj, "timestamp": 0})

    def clear(self) -> None:
        self._rows.clear()

    def __len__(self) -> int:
        return len(self._rows)

    def summary(self) -> Dict[str, float]:
        n = len(self._rows)
        if n == 0:
            return {}
        gpu = [r for r in self._rows if r["device"] == "gpu"]
        out = {"total_calls": n, "photonic_calls": n - len(gpu), "gpu_calls": len(gpu),
               "photonic_usage_ratio": (n - len(gpu)) / n}
        if gpu:
            out["avg_gpu_latency_ms"] = sum(r["latency_ms"] for r in gpu) / len(gpu)
            out["avg_gpu_energy_mj"] = sum(r["energy_mj"] for r in gpu) / len(gpu)
        return out


class PhotonicFlashAttention(nn.Module):
    """``[B, S, E] -> [B, S, E]`` attention block; state-dict compatible with the reference
    (``gpu_attention.{qkv_proj,out_proj}.{weight,bias}``)."""

    def __init__(self, embed_dim: int, num_heads: int, dropout: float = 0.0, bias: bool = True,
                 photonic_threshold: Optional[int] = None, device: Union[str, torch.device] = "auto",
                 dtype: Optional[torch.dtype] = None):
        super().__init__()
        if embed_dim % num_heads:
            raise AssertionError("embed_dim must be divisible by num_heads")
        cfg = get_config()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        self.photonic_threshold = photonic_threshold or cfg.photonic_threshold
        self.auto_device_selection = device == "auto" and cfg.auto_device_selection
        self.gpu_attention = FlashAttention3(embed_dim=embed_dim, num_heads=num_heads, dropout=dropout, bias=bias,
                                             device=None if device == "auto" else device, dtype=dtype)
        self.photonic_attention = None                      # out of scope (north_star): never built
        self.photonic_available = is_photonic_available()   # False on an MI355X box
        self.last_device_used, self.last_latency_ms, self.last_energy_mj = "gpu", 0.0, 0.0
        self._log = _CallLog()

    # the reference exposes the raw list; keep a read-only view for callers that peek at it
    @property
    def _performance_history(self):
        return list(self._log._rows)

    def _should_use_photonic(self, batch_size: int, seq_len: int) -> bool:
        return False

    def forward(self, query: torch.Tensor, key: Optional[torch.Tensor] = None, value: Optional[torch.Tensor] = None,
                attention_mask: Optional[torch.Tensor] = None, need_weights: bool = False,
                is_causal: bool = False) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        out, weights = self.gpu_attention(query, key, value, attention_mask, need_weights, is_causal=is_causal)
        core = self.gpu_attention.get_performance_stats()
        self.last_device_used = "gpu"
        self.last_latency_ms = core.get("latency_ms", 0.0)
        self.last_energy_mj = core.get("energy_mj", 0.0)
        self._log.add("gpu", self.last_latency_ms, self.last_energy_mj)
        return (out, weights) if need_weights else out

    def get_performance_stats(self) -> dict:
        stats = {"last_device_used": self.last_device_used, "last_latency_ms": self.last_latency_ms,
                 "last_energy_mj": self.last_energy_mj, "photonic_available": self.photonic_available,
                 "photonic_threshold": self.photonic_threshold}
        stats.update(self._log.summary())
        return stats

    def set_photonic_threshold(self, threshold: int) -> None:
        self.photonic_threshold = threshold

    def enable_photonic(self, enabled: bool = True) -> None:
        if enabled and not self.photonic_available:
            print("Warning: Photonic hardware not available")
        self.auto_device_selection = enabled

    def reset_performance_history(self) -> None:
        self._log.clear()


class PhotonicMultiHeadAttention(PhotonicFlashAttention):
    """``torch.nn.MultiheadAttention``-shaped facade: ``(L, N, E)`` unless ``batch_first``, ``key_padding_mask``
    / ``attn_mask`` arguments, weights averaged over heads by default.  Mask convention is the reference's:
    0 = masked, and a key-padding mask is *added* to ``attn_mask`` when both are given (:310-315)."""

    _UNSUPPORTED = "{} not yet supported"

    def __init__(self, embed_dim: int, num_heads: int, dropout: float = 0.0, bias: bool = True,
                 add_bias_kv: bool = False, add_zero_attn: bool = False, kdim: Optional[int] = None,
                 vdim: Optional[int] = None, batch_first: bool = False, photonic_threshold: Optional[int] = None,
                 device: Union[str, torch.device] = "auto", dtype: Optional[torch.dtype] = None):
        if add_bias_kv or add_zero_attn:
            raise NotImplementedError(self._UNSUPPORTED.format("add_bias_kv and add_zero_attn"))
        if kdim is not None or vdim is not None:
            raise NotImplementedError(self._UNSUPPORTED.format("Different key/value dimensions"))
        super().__init__(embed_dim, num_heads, dropout=dropout, bias=bias, photonic_threshold=photonic_threshold,
                         device=device, dtype=dtype)
        self.batch_first = batch_first

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                key_padding_mask: Optional[torch.Tensor] = None, need_weights: bool = True,
                attn_mask: Optional[torch.Tensor] = None, average_attn_weights: bool = True,
                is_causal: bool = False) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        k_is_q, v_is_q = key is query, value is query
        if not self.batch_first:                       # (L, N, E) -> (N, L, E); keep tensor identity for self-attention
            query = query.transpose(0, 1)
            key = query if k_is_q else key.transpose(0, 1)
            value = query if v_is_q else value.transpose(0, 1)
        mask = attn_mask
        if key_padding_mask is not None:
            mask = key_padding_mask if mask is None else mask + key_padding_mask.unsqueeze(1)
        res = super().forward(query, key, value, mask, need_weights, is_causal=is_causal)
        out, weights = res if need_weights else (res, None)
        if weights is not None and average_attn_weights:
            weights = weights.mean(dim=1)
        return (out if self.batch_first else out.transpose(0, 1)), weights
