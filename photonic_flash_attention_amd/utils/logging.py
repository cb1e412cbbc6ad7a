"""Logger factory: names live under ``photonic_flash_attention.`` like the reference's
(utils/logging.py:195-215) so existing log filters keep matching.  No handler is installed
at import (the reference configures the root logger as an import side effect, :249-259)."""

from __future__ import annotations

import logging
import os


def get_logger(name: str) -> logging.Logger:
    if not name.startswith("photonic_flash_attention"):
        name = f"photonic_flash_attention.{name}"
    logger = logging.getLogger(name)
    level = os.getenv("PHOTONIC_LOG_LEVEL")
    if level:
        logger.setLevel(getattr(logging, level.upper(), logging.INFO))
    return logger
