"""photonic_flash_attention_amd -- the electronic ("gpu") attention branch of Photonic-Flash-Attention, rebuilt
for AMD Instinct MI355X.

Scope (BASELINE.json north_star): ONE hot path -- the Flash-Attention forward behind ``PhotonicFlashAttention`` and
``HybridFlashAttention`` -- as a hand-written gfx950 HIP kernel behind a C ABI (``include/pfa_hip.h``), with the
reference package's module surface on top (import names as in ``photonic_flash_attention/__init__.py:10-72``).
"""

import torch as _torch

from .config import GlobalConfig, get_config
from .core.flash_attention_3 import FlashAttention3
from .core.hybrid_router import AdaptiveRouter, HybridFlashAttention
from .integration.pytorch.convert import convert_to_photonic
from .integration.pytorch.modules import PhotonicFlashAttention, PhotonicMultiHeadAttention
from .ops import fa3_forward, fa3_forward_bshd, is_available

__version__ = "0.1.0"
__all__ = ["PhotonicFlashAttention", "PhotonicMultiHeadAttention", "FlashAttention3", "HybridFlashAttention",
           "AdaptiveRouter", "convert_to_photonic", "fa3_forward", "fa3_forward_bshd", "is_available", "get_config", "GlobalConfig",
           "get_version", "get_device_info", "set_global_config"]


def get_version() -> str:
    return __version__


def set_global_config(**kwargs) -> None:
    """Same entry point as the reference (``__init__.py:69-72``)."""
    GlobalConfig.update(**kwargs)


def get_device_info() -> dict:
    """Keys of the reference's ``get_device_info`` (:44-66); the ``cuda_*`` entries describe the HIP devices."""
    n = _torch.cuda.device_count() if _torch.cuda.is_available() else 0
    info = {"version": __version__, "photonic_available": False, "cuda_available": n > 0, "cuda_device_count": n}
    if n:
        info.update(cuda_version=_torch.version.hip, pfa_hip_kernel=is_available(),
                    gpu_names=[_torch.cuda.get_device_name(i) for i in range(n)])
    return info
