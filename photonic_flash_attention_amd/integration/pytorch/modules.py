"""Public drop-in modules on top of the MI355X electronic branch.

Mirrors ``integration/pytorch/modules.py`` of the reference: ``PhotonicFlashAttention``
(:12-233) and ``PhotonicMultiHeadAttention`` (:235-336) keep their constructors, ``forward``
signatures, return conventions (tensor, or ``(tensor, weights)`` when ``need_weights``),
public knobs and ``last_*`` attributes.  Routing: with no photonic device the reference only
ever calls ``self.gpu_attention`` (:104-111, :118-121); here that is the only branch that
exists, so ``_should_use_photonic`` is constant False whatever the threshold says.
"""

from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from ...config import get_config
from ...core.flash_attention_3 import FlashAttention3
from ...photonic.hardware.detection import is_photonic_available


class PhotonicFlashAttention(nn.Module):
    def __init__(
        self,
        embed_dim: int,
        num_heads: int,
        dropout: float = 0.0,
        bias: bool = True,
        photonic_threshold: Optional[int] = None,
        device: Union[str, torch.device] = "auto",
        dtype: Optional[torch.dtype] = None,
    ):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.dropout = dropout
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == embed_dim, "embed_dim must be divisible by num_heads"

        config = get_config()
        self.photonic_threshold = photonic_threshold or config.photonic_threshold
        self.auto_device_selection = device == "auto" and config.auto_device_selection

        self.gpu_attention = FlashAttention3(
            embed_dim=embed_dim, num_heads=num_heads, dropout=dropout, bias=bias,
            device=device if device != "auto" else None, dtype=dtype)

        self.photonic_attention = None          # photonic branch: out of scope, never built
        self.photonic_available = is_photonic_available()

        self.last_device_used = "gpu"
        self.last_latency_ms = 0.0
        self.last_energy_mj = 0.0
        self._performance_history = []

    def forward(
        self,
        query: torch.Tensor,
        key: Optional[torch.Tensor] = None,
        value: Optional[torch.Tensor] = None,
        attention_mask: Optional[torch.Tensor] = None,
        need_weights: bool = False,
        is_causal: bool = False,
    ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        batch_size, seq_len, _ = query.shape
        use_photonic = self._should_use_photonic(batch_size, seq_len)
        assert not use_photonic
        output, weights = self._forward_gpu(query, key, value, attention_mask, need_weights, is_causal)
        self.last_device_used = "gpu"
        self._update_performance_stats()
        return (output, weights) if need_weights else output

    def _should_use_photonic(self, batch_size: int, seq_len: int) -> bool:
        """Reference: modules.py:118-143.  The first test there (no photonic device -> False)
        is the only reachable one on an MI355X box."""
        return False

    def _forward_gpu(self, query, key, value, attention_mask, need_weights, is_causal=False):
        return self.gpu_attention(query, key, value, attention_mask, need_weights, is_causal=is_causal)

    def _update_performance_stats(self) -> None:
        stats = self.gpu_attention.get_performance_stats()
        self.last_latency_ms = stats.get("latency_ms", 0.0)
        self.last_energy_mj = stats.get("energy_mj", 0.0)
        self._performance_history.append({
            "device": self.last_device_used,
            "latency_ms": self.last_latency_ms,
            "energy_mj": self.last_energy_mj,
            "timestamp": 0,
        })
        if len(self._performance_history) > 100:
            self._performance_history = self._performance_history[-100:]

    def get_performance_stats(self) -> dict:
        stats = {
            "last_device_used": self.last_device_used,
            "last_latency_ms": self.last_latency_ms,
            "last_energy_mj": self.last_energy_mj,
            "photonic_available": self.photonic_available,
            "photonic_threshold": self.photonic_threshold,
        }
        if self._performance_history:
            gpu_calls = [h for h in self._performance_history if h["device"] == "gpu"]
            stats.update({
                "total_calls": len(self._performance_history),
                "photonic_calls": 0,
                "gpu_calls": len(gpu_calls),
                "photonic_usage_ratio": 0.0,
            })
            if gpu_calls:
                stats["avg_gpu_latency_ms"] = sum(h["latency_ms"] for h in gpu_calls) / len(gpu_calls)
                stats["avg_gpu_energy_mj"] = sum(h["energy_mj"] for h in gpu_calls) / len(gpu_calls)
        return stats

    def set_photonic_threshold(self, threshold: int) -> None:
        self.photonic_threshold = threshold

    def enable_photonic(self, enabled: bool = True) -> None:
        if enabled and not self.photonic_available:
            print("Warning: Photonic hardware not available")
        self.auto_device_selection = enabled

    def reset_performance_history(self) -> None:
        self._performance_history.clear()


class PhotonicMultiHeadAttention(PhotonicFlashAttention):
    """``torch.nn.MultiheadAttention``-shaped facade (reference: modules.py:235-336)."""

    def __init__(
        self,
        embed_dim: int,
        num_heads: int,
        dropout: float = 0.0,
        bias: bool = True,
        add_bias_kv: bool = False,
        add_zero_attn: bool = False,
        kdim: Optional[int] = None,
        vdim: Optional[int] = None,
        batch_first: bool = False,
        photonic_threshold: Optional[int] = None,
        device: Union[str, torch.device] = "auto",
        dtype: Optional[torch.dtype] = None,
    ):
        if add_bias_kv or add_zero_attn:
            raise NotImplementedError("add_bias_kv and add_zero_attn not yet supported")
        if kdim is not None or vdim is not None:
            raise NotImplementedError("Different key/value dimensions not yet supported")
        super().__init__(embed_dim=embed_dim, num_heads=num_heads, dropout=dropout, bias=bias,
                         photonic_threshold=photonic_threshold, device=device, dtype=dtype)
        self.batch_first = batch_first

    def forward(
        self,
        query: torch.Tensor,
        key: torch.Tensor,
        value: torch.Tensor,
        key_padding_mask: Optional[torch.Tensor] = None,
        need_weights: bool = True,
        attn_mask: Optional[torch.Tensor] = None,
        average_attn_weights: bool = True,
        is_causal: bool = False,
    ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        same_qk, same_qv = key is query, value is query
        if not self.batch_first:
            query = query.transpose(0, 1)
            key = query if same_qk else key.transpose(0, 1)
            value = query if same_qv else value.transpose(0, 1)

        # mask merge as the reference does it (:310-315): masks ADD, and 0 still means "masked"
        attention_mask = attn_mask
        if key_padding_mask is not None:
            if attention_mask is not None:
                attention_mask = attention_mask + key_padding_mask.unsqueeze(1)
            else:
                attention_mask = key_padding_mask  # [B,Sk] key mask; the reference unsqueezes to [B,1,Sk] (:315)

        result = super().forward(query, key, value, attention_mask, need_weights, is_causal=is_causal)
        if need_weights:
            output, weights = result
            if weights is not None and average_attn_weights:
                weights = weights.mean(dim=1)
        else:
            output, weights = result, None
        if not self.batch_first:
            output = output.transpose(0, 1)
        return output, weights
