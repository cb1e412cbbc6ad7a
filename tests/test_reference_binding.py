"""The reference-side binding of INTEGRATION.md section B, run against the REAL reference (build container only: the reference
cannot travel to the GPU box, so this test is skipped wherever /root/reference is absent).

The ctypes stub a maintainer pastes into core/flash_attention_3.py is taken verbatim from INTEGRATION.md, patched into the
IMPORTED reference class, and driven through every entry of the reference that ends in the seam (flash_attention_3.py:97-102,
modules.py:154, photonic_attention.py:385-415).  `pfa_fa3_check` stands in for the launch (no GPU here): what is verified is
that the argument block the stub builds from the views the reference really produces -- strides (S*3E, D, 3E, 1) -- passes
the library's validation unchanged."""

from __future__ import annotations

import ctypes as C
import os
import sys

import pytest
import torch

from conftest import REPO

REF_SRC = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="the reference lives in the build container only")


@pytest.fixture(scope="module")
def binding():
    from photonic_flash_attention_amd import _capi
    lib = _capi.load()
    src = open(os.path.join(REPO, "INTEGRATION.md")).read()
    code = src[src.index("class PfaFa3Args(C.Structure):"):src.index("```", src.index("def fa3_forward("))]
    code = code.replace('_lib = C.CDLL("libpfa_hip.so")', "_lib = _FAKE_LIB")
    seen = []

    class FakeLib:                        # the stub's `_lib`: same entry points, the launch replaced by the validation
        pfa_status_string = lib.pfa_status_string

        class pfa_fa3_fwd:                # (has .argtypes assigned by the stub)
            argtypes = None

            def __new__(cls, a, stream):
                args = C.cast(a, C.POINTER(_capi.PfaFa3Args)).contents
                seen.append({f: getattr(args, f) for f in ("B", "H", "Sq", "Sk", "D", "q_stride_b", "q_stride_h", "q_stride_s",
                                                           "k_stride_s", "v_stride_s", "o_stride_s", "o_stride_h", "dtype_in")})
                return lib.pfa_fa3_check(C.byref(args))
    ns = {"_FAKE_LIB": FakeLib}
    exec("import ctypes as C, torch\n" + code, ns)
    return ns["fa3_forward"], seen


@pytest.fixture(scope="module")
def reference():
    os.environ.setdefault("PHOTONIC_LOG_LEVEL", "CRITICAL")
    sys.path.insert(0, REF_SRC)
    try:
        import photonic_flash_attention as ref
        from photonic_flash_attention.core import flash_attention_3 as ref_fa3
        yield ref, ref_fa3
    finally:
        sys.path.remove(REF_SRC)


def test_stub_in_the_imported_reference(binding, reference, monkeypatch):
    fa3_forward, seen = binding
    ref, ref_fa3 = reference

    def patched(self, q, k, v, attention_mask=None, need_weights=False):      # INTEGRATION.md, "replacement for lines 134-150"
        if need_weights or (attention_mask is not None and attention_mask.dim() != 2):
            raise NotImplementedError("this minimal stub: 2-D key masks only, no attention weights")
        return fa3_forward(q, k, v, self.scaling, key_mask=attention_mask), None

    monkeypatch.setattr(ref_fa3.FlashAttention3, "_flash_attention_forward", patched)
    B, S, E, H = 2, 128, 256, 4
    D = E // H
    x = torch.randn(B, S, E).to(torch.bfloat16)

    m = ref_fa3.FlashAttention3(E, H, dtype=torch.bfloat16).eval()
    with torch.no_grad():
        y, w = m(x)                                                           # flash_attention_3.py:85-118
    assert y.shape == (B, S, E) and w is None and len(seen) == 1
    a = seen[-1]
    assert (a["B"], a["H"], a["Sq"], a["Sk"], a["D"]) == (B, H, S, S, D) and a["dtype_in"] == 0
    # the strided views of the fused projection (:97-99), read in place: (S*3E, D, 3E) in elements
    assert (a["q_stride_b"], a["q_stride_h"], a["q_stride_s"]) == (S * 3 * E, D, 3 * E)
    assert a["k_stride_s"] == 3 * E and a["v_stride_s"] == 3 * E
    assert (a["o_stride_h"], a["o_stride_s"]) == (D, E)                       # [B,S,H,D] output: the transpose at :107 is a view

    km = torch.ones(B, S)
    km[:, 100:] = 0
    with torch.no_grad():
        m(x, attention_mask=km)                                               # the 2-D key mask of :166-167
        xkv = torch.randn(B, 77 * 8, E).to(torch.bfloat16)
        m(x, xkv, xkv)                                                        # cross attention: three separate projections (:92-94)
    assert len(seen) == 3 and seen[-1]["Sk"] == 77 * 8 and seen[-1]["k_stride_s"] == 3 * E

    pfa = ref.PhotonicFlashAttention(E, H, dtype=torch.bfloat16).eval()       # modules.py:77-116 -> gpu_attention
    if hasattr(pfa, "photonic_attention"):
        pfa.photonic_attention = None                                         # the router's threshold always selects the GPU branch
    pfa.photonic_available = False
    with torch.no_grad():
        out = pfa(x)
    assert torch.is_tensor(out) and out.shape == (B, S, E) and len(seen) == 4

    from photonic_flash_attention.core.photonic_attention import PhotonicAttention
    pa = PhotonicAttention(E, H, dtype=torch.bfloat16).eval()                 # photonic_attention.py:385-415
    with torch.no_grad():
        y2, _ = pa._fallback_forward(x, None, None, None, False)
    assert y2.shape == (B, S, E) and len(seen) == 5
    assert (seen[-1]["q_stride_b"], seen[-1]["q_stride_h"], seen[-1]["q_stride_s"]) == (S * 3 * E, D, 3 * E)


def test_stub_reports_library_errors(binding):
    fa3_forward, _ = binding
    q = torch.zeros(1, 2, 64, 48, dtype=torch.bfloat16)                       # head dim 48: PFA_ERR_HEAD_DIM through the stub
    with pytest.raises(RuntimeError, match="head dim"):
        fa3_forward(q, q, q, 48 ** -0.5)
