"""Exception names the wrappers raise (reference: utils/exceptions.py:4,31,109)."""

from __future__ import annotations


class PhotonicFlashAttentionError(Exception):
    """Base class of the package's errors."""


class PhotonicComputationError(PhotonicFlashAttentionError):
    """A computation on the selected backend failed (reference: utils/exceptions.py:31-46)."""

    def __init__(self, message: str, operation: str = None, input_shapes: tuple = None):
        super().__init__(message)
        self.operation = operation
        self.input_shapes = input_shapes


class PhotonicConfigurationError(PhotonicFlashAttentionError):
    """Invalid configuration."""


class PhotonicTimeoutError(PhotonicFlashAttentionError):
    """A queued request did not finish in time (reference: utils/exceptions.py:109-120)."""

    def __init__(self, message: str, timeout_seconds: float, operation: str = None):
        super().__init__(message)
        self.timeout_seconds = timeout_seconds
        self.operation = operation
