"""Global configuration read by the hot path.

Mirrors the fields of the reference's ``GlobalConfig`` that the electronic branch and its
wrappers read (config.py:8-101): ``photonic_threshold`` / ``auto_device_selection`` feed the
routers (modules.py:43-44, hybrid_router.py:160-171), ``max_memory_usage`` fed the reference's
tile-size search (flash_attention_3.py:284; the HIP kernel has fixed tiles and needs no budget,
the field is kept so ``set_global_config(max_memory_usage=...)`` keeps working).
Photonic-hardware fields are out of scope but accepted and stored so
existing ``set_global_config`` calls do not fail.
"""

from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Any, Dict, Optional


def _str_to_bool(value: str) -> bool:
    return value.lower() in ("true", "1", "yes", "on")


@dataclass
class GlobalConfig:
    device_priority: list = field(default_factory=lambda: ["photonic", "cuda"])
    photonic_threshold: int = 512
    auto_device_selection: bool = True
    max_memory_usage: float = 0.8
    memory_pool_enabled: bool = True
    enable_profiling: bool = False
    benchmark_mode: bool = False
    cache_kernel_selections: bool = True
    photonic_wavelengths: int = 80
    modulator_resolution: int = 6
    detector_noise_floor: float = 1e-12
    max_optical_power: float = 10e-3
    temperature_monitoring: bool = True
    thermal_shutdown_temp: float = 85.0
    log_level: str = "INFO"
    log_device_switches: bool = True
    log_performance_metrics: bool = False

    _instance = None  # class attribute (not a dataclass field)

    _ENV = {
        "PHOTONIC_THRESHOLD": ("photonic_threshold", int),
        "PHOTONIC_WAVELENGTHS": ("photonic_wavelengths", int),
        "MAX_OPTICAL_POWER": ("max_optical_power", float),
        "LOG_LEVEL": ("log_level", str),
        "ENABLE_PROFILING": ("enable_profiling", _str_to_bool),
        "AUTO_DEVICE_SELECTION": ("auto_device_selection", _str_to_bool),
    }

    @classmethod
    def get_instance(cls) -> "GlobalConfig":
        if cls._instance is None:
            inst = cls()
            inst._load_from_env()
            cls._instance = inst
        return cls._instance

    @classmethod
    def update(cls, **kwargs) -> None:
        inst = cls.get_instance()
        for key, value in kwargs.items():
            if key.startswith("_") or not hasattr(inst, key):
                raise ValueError(f"Unknown config key: {key}")
            setattr(inst, key, value)

    @classmethod
    def reset(cls) -> None:
        cls._instance = None

    def _load_from_env(self) -> None:
        for env, (attr, conv) in self._ENV.items():
            raw = os.getenv(env)
            if raw is None:
                continue
            try:
                setattr(self, attr, conv(raw))
            except (ValueError, TypeError) as exc:
                print(f"Warning: Invalid value for {env}: {raw}. Error: {exc}")

    def to_dict(self) -> Dict[str, Any]:
        return {k: v for k, v in self.__dict__.items() if not k.startswith("_")}


def get_config() -> GlobalConfig:
    return GlobalConfig.get_instance()
