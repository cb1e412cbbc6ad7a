"""Shared fixtures.  ``-m "not gpu"`` runs in the build container (no GPU);
``-m gpu`` runs on an MI355X box where ``/root/reference`` does not exist."""

from __future__ import annotations

import glob
import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name: str):
    """-> (meta dict, {array name: ndarray})"""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    meta = json.loads(bytes(z["meta"]).decode())
    return meta, {k: z[k] for k in z.files if k != "meta"}


def golden_names(kind=None):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        n = os.path.basename(p)[:-4]
        if kind is None or load_golden(n)[0]["kind"] == kind:
            out.append(n)
    return out


def golden_inputs(meta):
    """Re-generate the fixture's inputs and verify them against the stored checksums."""
    import torch
    from photonic_flash_attention_amd import synth

    q, k, v = synth.qkv(meta["B"], meta["H"], meta["Sq"], meta["Sk"], meta["D"], meta["seed"], meta["dtype"])
    got = [synth.checksum(t.view(torch.int16).numpy().view(np.uint16)) for t in (q, k, v)]
    assert got == meta["in_checksum"], "synthetic generator is not reproducing the fixture inputs"
    return q, k, v


@pytest.fixture(scope="session")
def hip():
    """The C-ABI binding; GPU tests fail loudly if the native library is missing."""
    from photonic_flash_attention_amd import _capi
    _capi.load()
    return _capi
