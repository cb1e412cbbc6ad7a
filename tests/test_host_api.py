"""CPU-side tests (no GPU): C-ABI surface, argument validation, and the reference-shaped
host modules (constructor contract, routing pinned to 'gpu', plumbing around the core seam)."""

from __future__ import annotations

import ctypes as C
import os
import re
import subprocess
import threading

import pytest
import torch

from conftest import REPO, load_golden
from photonic_flash_attention_amd import _capi, synth

HEADER = os.path.join(REPO, "include", "pfa_hip.h")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_capi.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(REPO, "photonic_flash_attention_amd", "csrc")], check=True)
    return _capi.load()


def test_library_exports_every_declared_symbol(lib):
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(pfa_[a-z0-9_]+)\s*\(", text))
    assert declared == set(_capi.EXPORTS), declared ^ set(_capi.EXPORTS)
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert lib.pfa_abi_version() == _capi.PFA_ABI_VERSION


def test_ctypes_struct_matches_c_layout(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pfa_hip.h"\nint main(){printf("%zu %zu %zu %zu",'
                   'sizeof(pfa_fa3_args),offsetof(pfa_fa3_args,q_stride_b),offsetof(pfa_fa3_args,B),'
                   'offsetof(pfa_fa3_args,workspace));return 0;}')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    A = _capi.PfaFa3Args
    assert got == [C.sizeof(A), A.q_stride_b.offset, A.B.offset, A.workspace.offset]


def _args(**over):
    base = dict(q=0x1000, k=0x2000, v=0x3000, o=0x4000, B=1, H=2, Sq=64, Sk=64, D=64,
                q_stride_b=8192, q_stride_h=64, q_stride_s=128, k_stride_b=8192, k_stride_h=64, k_stride_s=128,
                v_stride_b=8192, v_stride_h=64, v_stride_s=128, o_stride_b=8192, o_stride_h=64, o_stride_s=128,
                dtype_in=0, dtype_out=0, softmax_scale=0.125)
    base.update(over)
    return _capi.make_args(**base)


def test_argument_validation_without_a_gpu(lib):
    assert lib.pfa_fa3_check(C.byref(_args())) == 0
    cases = [
        (dict(D=80), -4), (dict(D=256), -4), (dict(dtype_in=2), -5), (dict(dtype_out=1), -5),
        (dict(q=0), -1), (dict(o=0), -1), (dict(B=0), -3), (dict(Sk=0), -3), (dict(softmax_scale=0.0), -3),
        (dict(softmax_scale=float("nan")), -3), (dict(q_stride_s=129), -6), (dict(o_stride_s=6), -6),
        (dict(k=0x2008), -7), (dict(flags=0x80), -10),
    ]
    for over, want in cases:
        assert lib.pfa_fa3_check(C.byref(_args(**over))) == want, over
    bad = _args()
    bad.size = 8
    assert lib.pfa_fa3_check(C.byref(bad)) == -2
    assert lib.pfa_fa3_check(None) == -1
    assert lib.pfa_fa3_workspace_bytes(C.byref(_args())) == 0
    for st in range(0, -11, -1):
        assert _capi.status_string(st) != "unknown pfa_status"
    assert _capi.status_string(-99) == "unknown pfa_status"


def test_production_library_refuses_development_variants(lib):
    """flags bits 8..15: 0 or one of the three production kernel selectors; the schedule experiments and the timing-only
    ablations (which compute wrong answers on purpose) exist only in a `make DEV=1` library."""
    for sel in (0, 43, 44, 45):
        assert lib.pfa_fa3_check(C.byref(_args(flags=sel << 8))) == 0, sel
    for var in (1, 9, 0x14, 19, 28, 30, 41, 42, 46, 49, 255):          # 0x14 << 8 = 0x1400
        assert lib.pfa_fa3_check(C.byref(_args(flags=var << 8))) == -10, var
    assert lib.pfa_fa3_check(C.byref(_args(flags=0x1400))) == -10


def test_no_environment_variable_selects_a_kernel(monkeypatch):
    """ops.fa3_forward must not read a kernel variant from the environment (only the explicit `_variant=` of tools / tests)."""
    import inspect
    from photonic_flash_attention_amd import ops
    src = inspect.getsource(ops)
    assert "os.environ" not in src and "PFA_VARIANT" not in src


def test_describe_picks_kernel_variant(lib):
    name, nwg = _capi.describe(_args(D=128, causal=1, Sq=4096, Sk=4096, B=4, H=16, dtype_out=2, flags=1))
    assert "d128" in name and "o32" in name and nwg == 16 * 64


# ---------------------------------------------------------------- reference-shaped modules
def test_constructor_contract():
    from photonic_flash_attention_amd import (FlashAttention3, HybridFlashAttention, PhotonicFlashAttention,
                                              PhotonicMultiHeadAttention)
    m = FlashAttention3(512, 8)
    # tests/unit/test_flash_attention_3.py:25-39 of the reference
    assert (m.embed_dim, m.num_heads, m.head_dim) == (512, 8, 64) and m.scaling == 64 ** -0.5
    assert m.qkv_proj.weight.shape == (1536, 512) and m.out_proj.weight.shape == (512, 512)
    assert sorted(m.state_dict()) == ["out_proj.bias", "out_proj.weight", "qkv_proj.bias", "qkv_proj.weight"]
    with pytest.raises(AssertionError):
        FlashAttention3(100, 8)
    assert set(m.get_performance_stats()) == {"latency_ms", "memory_mb", "device", "implementation"}
    p = PhotonicFlashAttention(256, 4)
    assert sorted(p.state_dict())[0].startswith("gpu_attention.") and p.photonic_attention is None
    assert p.last_device_used == "gpu" and p.photonic_threshold == 512
    p.set_photonic_threshold(64)
    assert not p._should_use_photonic(8, 100000)
    with pytest.raises(NotImplementedError):
        PhotonicMultiHeadAttention(256, 4, add_bias_kv=True)
    with pytest.raises(NotImplementedError):
        PhotonicMultiHeadAttention(256, 4, kdim=64)
    h = HybridFlashAttention(256, 4, enable_scaling=False)
    assert h.photonic_attention is None and h.executor is None
    assert "gpu_stats" in h.get_performance_stats()


def test_router_is_pinned_to_gpu_for_every_baseline_shape():
    from photonic_flash_attention_amd.core.hybrid_router import (AdaptiveRouter, PerformanceMetrics,
                                                                 WorkloadCharacteristics)
    r = AdaptiveRouter()
    for B, S, E, H in [(2, 128, 256, 4), (4, 1024, 768, 12), (4, 4096, 2048, 16), (1, 16384, 4096, 32)]:
        w = WorkloadCharacteristics(B, S, E, H, dtype=torch.bfloat16)
        for _ in range(120):   # far past the reference's 50-sample learned/exploring regime
            assert r.select_device(w) == "gpu"
            r.update_performance("gpu", w, PerformanceMetrics(latency_ms=1.0))
    st = r.get_stats()
    assert st["photonic_samples"] == 0 and st["gpu_samples"] == 480 and st["cache_hit_rate"] > 0.9


def test_no_cpu_or_eager_path_in_the_product():
    """Host tensors must fail loudly: the product has no CPU implementation of the core."""
    from photonic_flash_attention_amd import FlashAttention3, ops
    m = FlashAttention3(128, 2).eval()
    with torch.no_grad(), pytest.raises(RuntimeError, match="MI355X"):
        m(torch.zeros(1, 8, 128))
    z = torch.zeros(1, 2, 8, 64, dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        ops.fa3_forward(z, z, z)
    import photonic_flash_attention_amd as pkg
    src_dir = os.path.dirname(pkg.__file__)
    for root, _, files in os.walk(src_dir):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(root, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f


def _oracle_core(self, q, k, v, attention_mask=None, need_weights=False, is_causal=False):
    """Test double for the core seam (tests may use the oracle as the checker)."""
    from oracle import fa3_oracle as orc
    mask = orc.causal_mask(q.shape[2], k.shape[2]) if is_causal else attention_mask
    return orc.flash_attention_forward(q, k, v, mask, self.scaling), None


def test_module_plumbing_matches_reference_module(monkeypatch):
    """Projection plumbing around the seam (chunk order, head split/merge, out_proj) against the
    real reference's FlashAttention3.forward output (golden g1_c1_module), core replaced by the oracle."""
    from photonic_flash_attention_amd import FlashAttention3, PhotonicFlashAttention
    meta, arr = load_golden("g1_c1_module")
    E, H, seed = meta["E"], meta["H"], meta["seed"]
    sd = {
        "qkv_proj.weight": torch.from_numpy(synth.normal_f32((3 * E, E), seed + 10)) * E ** -0.5,
        "qkv_proj.bias": torch.from_numpy(synth.normal_f32((3 * E,), seed + 11)) * 0.1,
        "out_proj.weight": torch.from_numpy(synth.normal_f32((E, E), seed + 12)) * E ** -0.5,
        "out_proj.bias": torch.from_numpy(synth.normal_f32((E,), seed + 13)) * 0.1,
    }
    monkeypatch.setattr(FlashAttention3, "_flash_attention_forward", _oracle_core)
    x = torch.from_numpy(synth.normal_f32((meta["B"], meta["S"], E), seed))
    ref = torch.from_numpy(arr["out"])
    m = FlashAttention3(E, H).eval()
    m.load_state_dict(sd)
    with torch.no_grad():
        y, w = m(x)
        y2, _ = m(x, x.clone(), x.clone())       # cross-attention branch computes the same numbers
    assert w is None and float((y - ref).abs().max()) <= 5e-6 and float((y2 - ref).abs().max()) <= 5e-6
    p = PhotonicFlashAttention(E, H).eval()
    p.gpu_attention.load_state_dict(sd)
    with torch.no_grad():
        z = p(x)
    assert isinstance(z, torch.Tensor) and float((z - ref).abs().max()) <= 5e-6
    assert p.get_performance_stats()["gpu_calls"] == 1


def test_mha_facade_layouts(monkeypatch):
    from photonic_flash_attention_amd import FlashAttention3, PhotonicMultiHeadAttention
    monkeypatch.setattr(FlashAttention3, "_flash_attention_forward", _oracle_core)
    mb = PhotonicMultiHeadAttention(128, 2, batch_first=True).eval()
    ms = PhotonicMultiHeadAttention(128, 2, batch_first=False).eval()
    ms.load_state_dict(mb.state_dict())
    x = torch.from_numpy(synth.normal_f32((2, 40, 128), 3))
    with torch.no_grad():
        yb, wb = mb(x, x, x, need_weights=False)
        xs = x.transpose(0, 1)
        ys, _ = ms(xs, xs, xs, need_weights=False)
    assert wb is None and yb.shape == (2, 40, 128) and ys.shape == (40, 2, 128)
    assert float((ys.transpose(0, 1) - yb).abs().max()) <= 1e-6


def test_hybrid_overload_path_and_concurrency(monkeypatch):
    """tests/performance/test_benchmarks.py:298-376 of the reference: 8 caller threads."""
    from photonic_flash_attention_amd import FlashAttention3, HybridFlashAttention
    monkeypatch.setattr(FlashAttention3, "_flash_attention_forward", _oracle_core)
    h = HybridFlashAttention(128, 2, max_concurrent_requests=2).eval()
    x = torch.from_numpy(synth.normal_f32((2, 64, 128), 5))
    with torch.no_grad():
        want = h(x)[0]
    outs, errs = [None] * 8, []

    def work(i):
        try:
            with torch.no_grad():
                outs[i] = h(x)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs
    for o in outs:
        assert isinstance(o, tuple) and o[1] is None and torch.equal(o[0], want)
    st = h.get_performance_stats()
    assert st["total_requests"] == 9 and st["peak_concurrent"] >= 1 and st["photonic_samples"] == 0


def test_global_config_env_and_update(monkeypatch):
    from photonic_flash_attention_amd import set_global_config
    from photonic_flash_attention_amd.config import GlobalConfig, get_config
    GlobalConfig.reset()
    monkeypatch.setenv("PHOTONIC_THRESHOLD", "2048")
    monkeypatch.setenv("AUTO_DEVICE_SELECTION", "off")
    c = get_config()
    assert c.photonic_threshold == 2048 and c.auto_device_selection is False and c.max_memory_usage == 0.8
    set_global_config(max_memory_usage=0.5)
    assert get_config().max_memory_usage == 0.5
    with pytest.raises(ValueError):
        set_global_config(no_such_key=1)
    GlobalConfig.reset()


def test_convert_to_photonic_replaces_torch_mha(monkeypatch):
    """model conversion for local nn.Module objects (reference convert.py:441-452
    weight rule).  Core replaced by the oracle double; checks weight mapping + PyTorch mask dialect translation
    against the ORIGINAL torch layer."""
    import torch.nn as nn
    from photonic_flash_attention_amd import FlashAttention3
    from photonic_flash_attention_amd.integration.pytorch.convert import TorchMHAReplacement, convert_to_photonic
    monkeypatch.setattr(FlashAttention3, "_flash_attention_forward", _oracle_core)
    torch.manual_seed(0)
    layer = nn.TransformerEncoderLayer(d_model=256, nhead=4, dim_feedforward=512, dropout=0.0, batch_first=True).eval()
    conv, rep = convert_to_photonic(layer)
    assert rep.converted_layers == ["self_attn"] and not rep.skipped_layers and not rep.conversion_errors
    assert isinstance(conv.self_attn, TorchMHAReplacement) and isinstance(layer.self_attn, nn.MultiheadAttention)
    x = torch.from_numpy(synth.normal_f32((2, 50, 256), 8))
    pad = torch.zeros(2, 50, dtype=torch.bool)
    pad[0, 40:] = True                                               # PyTorch convention: True = ignore
    causal = nn.Transformer.generate_square_subsequent_mask(50)
    with torch.no_grad():
        for kw in (dict(), dict(src_key_padding_mask=pad), dict(src_mask=causal, is_causal=True),
                   dict(src_mask=causal), dict(src_mask=causal, src_key_padding_mask=pad)):
            want = layer(x, **kw)
            got = conv(x, **kw)
            keep = ~pad if "src_key_padding_mask" in kw else torch.ones(2, 50, dtype=torch.bool)
            assert float((got - want)[keep].abs().max()) <= 2e-5, kw
    # float masks: binary in effect (0 / -inf / finfo.min / -1e4) are translated, finite biases are refused -- never dropped silently
    big = torch.zeros(50, 50)
    big[:, 30:] = torch.finfo(torch.float32).min
    soft = torch.zeros(50, 50)
    soft[:, 30:] = -1e4
    with torch.no_grad():
        a = conv(x, src_mask=big)
        b = conv(x, src_mask=soft)
        assert float((a - layer(x, src_mask=big)).abs().max()) <= 2e-5 and torch.equal(a, b)
        alibi = -0.5 * torch.arange(50.0)[None, :].expand(50, 50).contiguous()
        with pytest.raises(ValueError, match="finite biases"):
            conv(x, src_mask=alibi)
    keep = TorchMHAReplacement._to_keep_mask(torch.tensor([[0.0, -1e4, float("-inf"), torch.finfo(torch.float32).min]]))
    from photonic_flash_attention_amd.integration.pytorch import hf
    assert keep.tolist() == [[True, False, False, False]] == hf._keep_mask(torch.tensor([[0.0, -1e4, float("-inf"), -3e38]])).tolist()
    # skipped layers are reported, strings are refused
    odd = nn.MultiheadAttention(768, 4)                              # head_dim 192 > 128: no kernel
    _, rep2 = convert_to_photonic(nn.Sequential(odd))
    assert rep2.skipped_layers == ["0"] and "head_dim" in rep2.compatibility_warnings[0]
    with pytest.raises(ValueError):
        convert_to_photonic("bert-base-uncased")


def test_integration_md_binding_matches_the_abi():
    """The ctypes struct INTEGRATION.md tells a maintainer to paste must be the library's struct (same fields, same size)."""
    import ctypes as C
    import os
    from photonic_flash_attention_amd import _capi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "INTEGRATION.md")).read()
    code = src[src.index("class PfaFa3Args(C.Structure):"):src.index("_lib = C.CDLL")]
    ns = {}
    exec("import ctypes as C\n" + code, ns)
    doc = ns["PfaFa3Args"]
    assert [f[0] for f in doc._fields_] == [f[0] for f in _capi.PfaFa3Args._fields_]
    assert C.sizeof(doc) == C.sizeof(_capi.PfaFa3Args)


def test_hf_attention_registration_and_mask_dialect():
    """transformers models take the kernel through AttentionInterface; on a CPU tensor the call fails loudly (no fallback)."""
    transformers = pytest.importorskip("transformers")
    import torch
    from photonic_flash_attention_amd.integration.pytorch import hf
    from photonic_flash_attention_amd.integration.pytorch.convert import convert_to_photonic
    name = hf.register_hf_attention()
    assert name in transformers.AttentionInterface().valid_keys()
    assert hf._keep_mask(None) is None
    b = torch.tensor([[True, False]])
    assert hf._keep_mask(b) is b
    add = torch.tensor([[0.0, -1e4, float("-inf"), torch.finfo(torch.float32).min]])
    assert hf._keep_mask(add).tolist() == [[True, False, False, False]]
    cfg = transformers.BertConfig(hidden_size=256, num_attention_heads=4, num_hidden_layers=1, intermediate_size=256, vocab_size=50)
    model = transformers.BertModel(cfg).eval()
    conv, report = convert_to_photonic(model)
    assert conv is not model and conv.config._attn_implementation == hf.IMPLEMENTATION_NAME
    assert model.config._attn_implementation != hf.IMPLEMENTATION_NAME and report.converted_layers
    with torch.no_grad(), pytest.raises((ValueError, RuntimeError)):
        conv(input_ids=torch.tensor([[1, 2, 3, 4]]))


def test_hf_mask_interface_hands_on_the_compact_form():
    """pfa_attention_mask: a plain causal / bidirectional prefill keeps the caller's 2-D padding mask (the triangle is a flag of the
    kernels); cached decoding, offsets and user mask functions get sdpa's 4-D mask."""
    pytest.importorskip("transformers")
    import torch
    from transformers import masking_utils as mu
    from photonic_flash_attention_amd.integration.pytorch import hf
    am = torch.ones(2, 16, dtype=torch.bool)
    am[1, 10:] = False
    kw = dict(batch_size=2, q_length=16, kv_length=16)
    for fn in (mu.causal_mask_function, mu.bidirectional_mask_function):
        got = hf.pfa_attention_mask(mask_function=fn, attention_mask=am, **kw)
        assert got.dim() == 2 and torch.equal(got, am)
        assert hf.pfa_attention_mask(mask_function=fn, attention_mask=None, **kw) is None
    dec = hf.pfa_attention_mask(batch_size=2, q_length=1, kv_length=16, mask_function=mu.causal_mask_function, attention_mask=am, q_offset=15)
    assert dec is None or dec.dim() == 4                                       # a decode step: sdpa's form (here nothing, or [B,1,1,Sk])
    win = mu.and_masks(mu.causal_mask_function, mu.sliding_window_causal_mask_function(4)) if hasattr(mu, "and_masks") else None
    if win is not None:
        m4 = hf.pfa_attention_mask(mask_function=win, attention_mask=am, **kw)
        assert m4 is not None and m4.dim() == 4 and m4.shape[-2:] == (16, 16)


def test_generator_takes_no_knobs_in_the_product_build_and_is_deterministic(tmp_path):
    """csrc/gen_fa3_fwd_p4.py: a P4_* knob left in the environment stops the product build (they are honoured only with P4_DEV=1,
    which tools/p4_variants.py sets); two runs give byte-identical assembly, 48 kernels."""
    import subprocess
    import sys
    gen = os.path.join(REPO, "photonic_flash_attention_amd", "csrc", "gen_fa3_fwd_p4.py")
    env = {k: v for k, v in os.environ.items() if not k.startswith("P4_")}
    a = subprocess.run([sys.executable, gen], env=env, capture_output=True, text=True, timeout=300)
    b = subprocess.run([sys.executable, gen], env=env, capture_output=True, text=True, timeout=300)
    assert a.returncode == 0 and a.stdout == b.stdout and a.stdout.count(".amdhsa_kernel ") == 48
    bad = subprocess.run([sys.executable, gen], env=dict(env, P4_STAMP="1"), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "P4_DEV" in bad.stderr and not bad.stdout


def test_bench_reads_the_traffic_figure_from_the_newest_profile():
    """bench.py roofline.traffic: parsed from profiles/r*_<workload>_rocprof_summary.md (FETCH_SIZE x 2 + WRITE_SIZE, KiB), not pasted."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("pfa_bench", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    b, src = bench.pmc_traffic("C3")
    assert b is not None and "r03_C3_rocprof_summary" in src and 2.5e8 < b < 5e8
    assert bench.pmc_traffic("nope") == (None, "no profiles/r*_nope_rocprof_summary.md")
