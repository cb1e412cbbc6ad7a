"""photonic_flash_attention_amd -- MI355X-native electronic branch of Photonic-Flash-Attention.

Drop-in for the reference package's hot path only (BASELINE.json north_star): the
Flash-Attention forward behind ``PhotonicFlashAttention`` / ``HybridFlashAttention``'s
"gpu" branch, as a hand-written gfx950 HIP kernel behind a C ABI (``include/pfa_hip.h``).
Import surface mirrors ``photonic_flash_attention/__init__.py:10-72``.
"""

from .config import GlobalConfig, get_config
from .core.flash_attention_3 import FlashAttention3
from .core.hybrid_router import AdaptiveRouter, HybridFlashAttention
from .integration.pytorch.modules import PhotonicFlashAttention, PhotonicMultiHeadAttention
from .ops import fa3_forward, fa3_forward_bshd, is_available

__version__ = "0.1.0"

__all__ = [
    "PhotonicFlashAttention", "PhotonicMultiHeadAttention", "FlashAttention3", "HybridFlashAttention",
    "AdaptiveRouter", "fa3_forward", "fa3_forward_bshd", "is_available", "get_config", "GlobalConfig",
    "get_version", "get_device_info", "set_global_config",
]


def get_version() -> str:
    return __version__


def get_device_info() -> dict:
    """Reference: __init__.py:44-66 (``cuda_*`` keys kept; on ROCm they describe the HIP devices)."""
    import torch
    info = {"photonic_available": False, "version": __version__,
            "cuda_available": torch.cuda.is_available(),
            "cuda_device_count": torch.cuda.device_count() if torch.cuda.is_available() else 0}
    if torch.cuda.is_available():
        info["cuda_version"] = torch.version.hip
        info["gpu_names"] = [torch.cuda.get_device_name(i) for i in range(torch.cuda.device_count())]
        info["pfa_hip_kernel"] = is_available()
    return info


def set_global_config(**kwargs) -> None:
    GlobalConfig.update(**kwargs)
