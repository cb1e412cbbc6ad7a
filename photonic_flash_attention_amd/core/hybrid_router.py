"""``HybridFlashAttention`` with the router pinned to the electronic ("gpu") branch.

Mirror of the reference's ``core/hybrid_router.py``: ``AdaptiveRouter`` (:56-259) keeps its
public surface (``select_device``, ``update_performance``, ``get_stats``) and
``HybridFlashAttention`` (:262-669) its constructor, ``forward`` -> raw ``(out, weights|None)``
tuple (:430), stats, ``enable_auto_scaling`` and ``reset_stats``.

Differences, all forced by BASELINE.json's north_star ("the router's threshold simply always
selects the GPU backend"):

* ``select_device`` always answers ``'gpu'``.  The reference's heuristic sends every BASELINE
  GPU shape to ``'photonic'`` (``S >= 512`` or ``B*S*S > 1e6``, :165-171) and later explores at
  random (:149-150); with separately initialised weights per branch (:297-315) that makes the
  reference's output non-deterministic.  The learned latency model is still updated (numpy,
  same SGD step as :221-242) so ``get_stats`` stays meaningful.
* no warm-up A/B (:403-404,:543-597): there is one branch.
* the overload path queues onto the thread pool exactly like :456-467; the batch-splitting
  variant (:471-541) needed a photonic branch and is gone.  The C ABI is re-entrant and takes
  the caller's current stream, so concurrent callers are safe (race row).
"""

from __future__ import annotations

import threading
import time
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from concurrent.futures import TimeoutError as FutureTimeout
from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from ..config import get_config
from ..utils.exceptions import PhotonicTimeoutError
from ..utils.logging import get_logger
from .flash_attention_3 import FlashAttention3


@dataclass
class PerformanceMetrics:
    """One measurement fed to the router (field names as in the reference, hybrid_router.py:20-29)."""
    latency_ms: float = 0.0
    throughput_tokens_per_sec: float = 0.0
    energy_mj: float = 0.0
    memory_mb: float = 0.0
    temperature_c: float = 0.0
    accuracy_score: float = 1.0
    timestamp: float = field(default_factory=time.time)


_DTYPE_FEATURE = {torch.float32: 1.0, torch.float16: 0.5}


@dataclass
class WorkloadCharacteristics:
    """Shape/dtype summary of a request; ``to_features`` is the 7-vector of the reference (:42-52)."""
    batch_size: int
    seq_length: int
    embed_dim: int
    num_heads: int
    is_training: bool = False
    has_mask: bool = False
    dtype: torch.dtype = torch.float32

    def to_features(self) -> np.ndarray:
        return np.asarray([self.batch_size, self.seq_length, self.embed_dim, self.num_heads, self.is_training,
                           self.has_mask, _DTYPE_FEATURE.get(self.dtype, 0.25)], dtype=np.float64)


class AdaptiveRouter:
    """Device selector pinned to ``'gpu'``; keeps the reference's bookkeeping."""

    DEVICES = ("gpu",)

    def __init__(self, history_size: int = 1000, learning_rate: float = 0.01,
                 exploration_rate: float = 0.1, min_samples_for_prediction: int = 50):
        self.history_size = history_size
        self.learning_rate = learning_rate
        self.exploration_rate = exploration_rate
        self.min_samples_for_prediction = min_samples_for_prediction
        self.logger = get_logger(self.__class__.__name__)
        self.gpu_history: deque = deque(maxlen=history_size)
        self.photonic_history: deque = deque(maxlen=history_size)   # stays empty
        self.gpu_weights = np.zeros(7)
        self._lock = threading.RLock()
        self._prediction_cache: Dict[str, Tuple[str, float]] = {}
        self._cache_hits = 0
        self._cache_misses = 0

    def select_device(self, workload: WorkloadCharacteristics) -> str:
        with self._lock:
            key = self._get_cache_key(workload)
            if key in self._prediction_cache:
                self._cache_hits += 1
            else:
                self._cache_misses += 1
                self._prediction_cache[key] = ("gpu", 1.0)
                if len(self._prediction_cache) > 1000:
                    del self._prediction_cache[next(iter(self._prediction_cache))]
            return "gpu"

    @staticmethod
    def _get_cache_key(w: WorkloadCharacteristics) -> str:
        return f"{w.batch_size}_{w.seq_length // 32 * 32}_{w.embed_dim}_{w.num_heads}_{w.is_training}_{w.has_mask}"

    def predicted_latency_ms(self, workload: WorkloadCharacteristics) -> float:
        return float(np.dot(self.gpu_weights, workload.to_features()))

    def update_performance(self, device: str, workload: WorkloadCharacteristics,
                           metrics: PerformanceMetrics) -> None:
        with self._lock:
            if device != "gpu":
                return
            self.gpu_history.append((workload.to_features(), metrics.latency_ms))
            if len(self.gpu_history) % 10 == 0 and len(self.gpu_history) >= 10:
                self._update_single_model(self.gpu_history, self.gpu_weights)

    def _update_single_model(self, history: deque, weights: np.ndarray) -> None:
        X = np.array([s[0] for s in history])
        y = np.array([s[1] for s in history])
        Xn = (X - X.mean(axis=0)) / (X.std(axis=0) + 1e-8)
        err = Xn @ weights - y
        weights -= self.learning_rate * (Xn.T @ err) / len(history)

    def get_stats(self) -> Dict[str, Any]:
        with self._lock:
            lookups = self._cache_hits + self._cache_misses
            n = len(self.gpu_history)
            return dict(gpu_samples=n, photonic_samples=0, total_samples=n, cache_size=len(self._prediction_cache),
                        cache_hit_rate=(self._cache_hits / lookups) if lookups else 0.0,
                        exploration_rate=self.exploration_rate, min_samples_for_ml=self.min_samples_for_prediction,
                        using_ml_prediction=False)


class HybridFlashAttention(nn.Module):
    def __init__(
        self,
        embed_dim: int,
        num_heads: int,
        dropout: float = 0.0,
        bias: bool = True,
        device: Union[str, torch.device] = "auto",
        dtype: Optional[torch.dtype] = None,
        enable_scaling: bool = True,
        max_concurrent_requests: int = 4,
    ):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.dropout = dropout
        self.enable_scaling = enable_scaling
        self.max_concurrent_requests = max_concurrent_requests
        self.logger = get_logger(self.__class__.__name__)
        self.config = get_config()

        self.gpu_attention = FlashAttention3(
            embed_dim=embed_dim, num_heads=num_heads, dropout=dropout, bias=bias,
            device=device if device != "auto" else None, dtype=dtype)
        self.photonic_attention = None   # out of scope (north_star): never constructed

        self.router = AdaptiveRouter()
        self.executor = ThreadPoolExecutor(max_workers=max_concurrent_requests) if enable_scaling else None
        self.active_requests = 0
        self._scaling_lock = threading.Lock()
        self.warmup_complete = True      # nothing to A/B
        self.total_requests = 0
        self.concurrent_requests = 0
        self.peak_concurrent = 0

    def forward(self, query, key=None, value=None, attention_mask=None, need_weights=False,
                is_causal: bool = False):
        with self._scaling_lock:
            self.total_requests += 1
            self.active_requests += 1
            self.concurrent_requests = self.active_requests
            self.peak_concurrent = max(self.peak_concurrent, self.concurrent_requests)
            overloaded = self.enable_scaling and self.active_requests > self.max_concurrent_requests
        try:
            if overloaded and self.executor is not None:
                return self._scaled_forward(query, key, value, attention_mask, need_weights, is_causal)
            return self._standard_forward(query, key, value, attention_mask, need_weights, is_causal)
        finally:
            with self._scaling_lock:
                self.active_requests -= 1

    def _standard_forward(self, query, key, value, attention_mask, need_weights, is_causal=False):
        workload = WorkloadCharacteristics(
            batch_size=query.shape[0], seq_length=query.shape[1], embed_dim=query.shape[2],
            num_heads=self.num_heads, is_training=self.training, has_mask=attention_mask is not None,
            dtype=query.dtype)
        device_used = self.router.select_device(workload)      # always 'gpu'
        t0 = time.perf_counter()
        try:
            result = self.gpu_attention(query, key, value, attention_mask, need_weights, is_causal=is_causal)
        except Exception as exc:   # the reference logs and re-raises GPU-branch failures (:432-438)
            self.logger.error(f"Attention computation failed on {device_used}: {exc}")
            raise
        dt = max(time.perf_counter() - t0, 1e-9)   # enqueue time: the C ABI is asynchronous
        self.router.update_performance(device_used, workload, PerformanceMetrics(
            latency_ms=dt * 1000,
            throughput_tokens_per_sec=workload.batch_size * workload.seq_length / dt,
            energy_mj=self._estimate_energy(device_used, workload),
            memory_mb=self._estimate_memory(workload)))
        return result

    def _scaled_forward(self, query, key, value, attention_mask, need_weights, is_causal=False):
        stream = torch.cuda.current_stream(query.device) if query.is_cuda else None

        def task():
            if stream is None:
                return self._standard_forward(query, key, value, attention_mask, need_weights, is_causal)
            with torch.cuda.device(query.device), torch.cuda.stream(stream):
                return self._standard_forward(query, key, value, attention_mask, need_weights, is_causal)

        future = self.executor.submit(task)
        try:
            return future.result(timeout=30.0)
        except FutureTimeout:
            raise PhotonicTimeoutError("Request timeout in scaled processing", 30.0, "scaled_forward")

    @staticmethod
    def _estimate_energy(device: str, workload: WorkloadCharacteristics) -> float:
        """Same constant model as the reference (:599-611): 300 W / 50 TOPS."""
        ops_ = workload.batch_size * workload.seq_length * workload.seq_length * workload.embed_dim
        return ops_ * (300 / 50e12 * 1000)

    @staticmethod
    def _estimate_memory(workload: WorkloadCharacteristics) -> float:
        elements = workload.batch_size * workload.seq_length * workload.embed_dim
        return elements * (4 if workload.dtype == torch.float32 else 2) / (1024 * 1024)

    def get_performance_stats(self) -> Dict[str, Any]:
        """Request counters + router counters + the core's stats (keys of the reference, :619-637)."""
        return dict(total_requests=self.total_requests, concurrent_requests=self.concurrent_requests,
                    peak_concurrent=self.peak_concurrent, warmup_complete=self.warmup_complete,
                    scaling_enabled=self.enable_scaling, max_concurrent=self.max_concurrent_requests,
                    **self.router.get_stats(), gpu_stats=self.gpu_attention.get_performance_stats())

    def enable_auto_scaling(self, enabled: bool = True, max_concurrent: Optional[int] = None) -> None:
        self.enable_scaling = enabled
        if max_concurrent is not None:
            self.max_concurrent_requests = max_concurrent
        if enabled and self.executor is None:
            self.executor = ThreadPoolExecutor(max_workers=self.max_concurrent_requests)
        elif not enabled and self.executor is not None:
            self.executor.shutdown(wait=False)
            self.executor = None

    def reset_stats(self) -> None:
        self.total_requests = 0
        self.concurrent_requests = 0
        self.peak_concurrent = 0
        self.router = AdaptiveRouter()

    def __del__(self):
        ex = getattr(self, "executor", None)
        if ex is not None:
            ex.shutdown(wait=False)
