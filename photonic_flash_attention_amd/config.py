"""Process-wide settings read by the hot path and its wrappers.

Keeps the names, defaults and environment overrides of the reference's ``GlobalConfig`` (config.py:8-101) so
``set_global_config(photonic_threshold=...)``, ``get_config().max_memory_usage`` and the ``PHOTONIC_THRESHOLD`` /
``AUTO_DEVICE_SELECTION`` / ``ENABLE_PROFILING`` / ``LOG_LEVEL`` / ``PHOTONIC_WAVELENGTHS`` / ``MAX_OPTICAL_POWER``
environment variables behave the same.  Only three fields matter on this path: ``photonic_threshold`` and
``auto_device_selection`` (read by the wrappers, modules.py:43-44 of the reference) and ``enable_profiling``
(turns the per-call device sync + timing of ``FlashAttention3.forward`` back on).  ``max_memory_usage`` fed the
reference's tile-size search (flash_attention_3.py:284); the HIP kernel has fixed tiles and ignores it.  The
photonic-hardware fields are carried so existing ``set_global_config`` calls do not fail.
"""

from __future__ import annotations

import os
from typing import Any, Callable, Dict, Optional, Tuple

_TRUE = {"true", "1", "yes", "on"}

# field -> default
_FIELDS: Dict[str, Any] = {
    "device_priority": ("photonic", "cuda"),
    "photonic_threshold": 512,
    "auto_device_selection": True,
    "max_memory_usage": 0.8,
    "memory_pool_enabled": True,
    "enable_profiling": False,
    "benchmark_mode": False,
    "cache_kernel_selections": True,
    "photonic_wavelengths": 80,
    "modulator_resolution": 6,
    "detector_noise_floor": 1e-12,
    "max_optical_power": 10e-3,
    "temperature_monitoring": True,
    "thermal_shutdown_temp": 85.0,
    "log_level": "INFO",
    "log_device_switches": True,
    "log_performance_metrics": False,
}

# environment variable -> (field, parser)
_ENV: Dict[str, Tuple[str, Callable[[str], Any]]] = {
    "PHOTONIC_THRESHOLD": ("photonic_threshold", int),
    "PHOTONIC_WAVELENGTHS": ("photonic_wavelengths", int),
    "MAX_OPTICAL_POWER": ("max_optical_power", float),
    "LOG_LEVEL": ("log_level", str),
    "ENABLE_PROFILING": ("enable_profiling", lambda s: s.lower() in _TRUE),
    "AUTO_DEVICE_SELECTION": ("auto_device_selection", lambda s: s.lower() in _TRUE),
}


class GlobalConfig:
    """Singleton settings object (``GlobalConfig.get_instance()`` / ``get_config()``)."""

    _instance: Optional["GlobalConfig"] = None

    def __init__(self, **overrides: Any):
        for name, default in _FIELDS.items():
            setattr(self, name, list(default) if isinstance(default, tuple) else default)
        for name, value in overrides.items():
            self._set(name, value)

    def _set(self, name: str, value: Any) -> None:
        if name not in _FIELDS:
            raise ValueError(f"Unknown config key: {name}")
        setattr(self, name, value)

    def _load_from_env(self) -> None:
        for var, (name, parse) in _ENV.items():
            raw = os.getenv(var)
            if raw is None:
                continue
            try:
                setattr(self, name, parse(raw))
            except (ValueError, TypeError) as exc:
                print(f"Warning: Invalid value for {var}: {raw}. Error: {exc}")

    @classmethod
    def get_instance(cls) -> "GlobalConfig":
        if cls._instance is None:
            cls._instance = cls()
            cls._instance._load_from_env()
        return cls._instance

    @classmethod
    def update(cls, **kwargs: Any) -> None:
        inst = cls.get_instance()
        for name, value in kwargs.items():
            inst._set(name, value)

    @classmethod
    def reset(cls) -> None:
        cls._instance = None

    def to_dict(self) -> Dict[str, Any]:
        return {name: getattr(self, name) for name in _FIELDS}

    def __repr__(self) -> str:
        return "GlobalConfig(" + ", ".join(f"{k}={v}" for k, v in self.to_dict().items()) + ")"


def get_config() -> GlobalConfig:
    return GlobalConfig.get_instance()
