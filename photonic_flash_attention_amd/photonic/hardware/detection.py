"""Photonic hardware probe -- OUT OF SCOPE stub.

The reference probes ``lspci`` / ``/dev/luminous`` / ``PHOTONIC_SIMULATION`` for photonic
accelerators (photonic/hardware/detection.py:24-234).  BASELINE.json's north_star leaves the
photonic/simulation path untouched and pins routing to the GPU backend, so this build ships
no photonic branch: the probe answers False and the wrappers' photonic slot stays ``None``.
"""


def is_photonic_available() -> bool:
    return False


def detect_photonic_hardware() -> bool:
    return False
