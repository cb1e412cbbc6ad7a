"""Host side of the hot path: tensor -> C-ABI argument marshalling for ``pfa_fa3_fwd``.

``fa3_forward`` has the call shape of the reference seam
``FlashAttention3._flash_attention_forward(q, k, v, attention_mask, need_weights)``
(core/flash_attention_3.py:120-150): operands are ``[B, H, S, D]`` tensors (usually strided
views of the fused QKV projection, :97-99), the result is ``[B, H, Sq, D]``.  The kernel
reads the views in place and writes a ``[B, Sq, H, D]`` buffer, returned as the transposed
view, so the reference's ``.transpose(1, 2).contiguous()`` at :107 costs nothing.

No fallback lives here: unsupported arguments raise ``ValueError`` (pre-launch
``pfa_status``) or ``PfaError``.
"""

from __future__ import annotations

import ctypes as C
import threading
from typing import Optional, Tuple

import torch

from . import _capi

_DT = {torch.bfloat16: _capi.PFA_DTYPE_BF16, torch.float16: _capi.PFA_DTYPE_FP16, torch.float32: _capi.PFA_DTYPE_FP32}

SUPPORTED_HEAD_DIMS = (64, 128)      # kernel instantiations; other head dims <= 128 run zero-padded to the next one


def _padded_head_dim(D: int) -> int:
    """Kernel head dim for a problem of head dim ``D`` (``D`` itself when a kernel exists for it)."""
    if D in SUPPORTED_HEAD_DIMS:
        return D
    if 1 <= D < SUPPORTED_HEAD_DIMS[-1]:
        return next(d for d in SUPPORTED_HEAD_DIMS if d > D)
    raise ValueError(f"head_dim {D} has no kernel (<= {SUPPORTED_HEAD_DIMS[-1]} supported)")


def _pad_d(t: torch.Tensor, Dp: int) -> torch.Tensor:
    """``[B,H,S,D]`` -> zero-padded ``[B,H,S,Dp]`` view of a fresh ``[B,S,H,Dp]`` buffer (zero q/k columns add nothing to the
    scores, zero v columns give zero output columns)."""
    B, H, S, D = t.shape
    buf = torch.zeros((B, S, H, Dp), dtype=t.dtype, device=t.device)
    buf[..., :D] = t.permute(0, 2, 1, 3)
    return buf.permute(0, 2, 1, 3)


def is_available(device: Optional[torch.device] = None) -> bool:
    """True when the native library is present and ``device`` is a gfx950 GPU."""
    if not torch.cuda.is_available():
        return False
    try:
        lib = _capi.load()
    except OSError:
        return False
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    return lib.pfa_device_supported(int(idx)) == 1


_SEQLENS_CACHE: "dict[tuple, torch.Tensor]" = {}
_SEQLENS_LOCK = threading.Lock()      # the wrappers above this module are entered from several threads (hybrid_router's pool)


def _seqlens_tensor(seqlens_k, device) -> torch.Tensor:
    """``seqlens_k`` as a contiguous int32 device tensor.  Python lists are uploaded once per distinct value (a pageable
    host-to-device copy costs ~15 us per call, more than the C2 kernel); tensors are used as they are."""
    if isinstance(seqlens_k, torch.Tensor):
        return seqlens_k.to(device=device, dtype=torch.int32).contiguous()
    key = (tuple(int(x) for x in seqlens_k), str(device))
    with _SEQLENS_LOCK:
        t = _SEQLENS_CACHE.get(key)
        if t is None:
            if len(_SEQLENS_CACHE) >= 64:
                _SEQLENS_CACHE.clear()
            t = _SEQLENS_CACHE[key] = torch.tensor(key[0], dtype=torch.int32, device=device)
    return t


def _drop_mask_ptr(drop_mask, B, H, Sq, Sk, like) -> int:
    """keep-mask of the dense branch's attention dropout: contiguous u8 / bool [B,H,Sq,Sk] on the operands' device, fp32 kernels only"""
    if like.dtype != torch.float32:
        raise ValueError("attention dropout runs on the fp32 kernels: hand over fp32 operands")
    if drop_mask.shape != (B, H, Sq, Sk) or drop_mask.dtype not in (torch.uint8, torch.bool) or not drop_mask.is_contiguous() \
            or drop_mask.device != like.device:
        raise ValueError("drop_mask must be a contiguous u8 / bool [B, H, Sq, Sk] tensor on the operands' device")
    return drop_mask.data_ptr()


def _bhsd_strides(t: torch.Tensor):
    sb, sh, ss, sd = t.stride()
    if sd != 1 and t.shape[3] != 1:
        raise ValueError("last (head_dim) stride must be 1")
    return sb, sh, ss


def _as_mask4(mask: torch.Tensor, B: int, H: int, Sq: int, Sk: int, device) -> torch.Tensor:
    """Normalise a reference-style mask (0 = masked) to a u8 tensor broadcastable as [B,H,Sq,Sk].
    2-D [B,Sk] -> [B,1,1,Sk] (flash_attention_3.py:166-167); 3-D [B,Sq|1,Sk] -> [B,1,Sq|1,Sk]; 4-D as is."""
    if mask.dim() == 2:
        mask = mask[:, None, None, :]
    elif mask.dim() == 3:
        mask = mask[:, None, :, :]
    elif mask.dim() != 4:
        raise ValueError(f"attention mask must be 2-D, 3-D or 4-D, got {mask.dim()}-D")
    for got, want, name in zip(mask.shape, (B, H, Sq, Sk), "BHQK"):
        if got not in (1, want):
            raise ValueError(f"mask dim {name} is {got}, expected 1 or {want}")
    if mask.shape[3] != Sk:
        mask = mask.expand(-1, -1, -1, Sk)
    dev = device if isinstance(device, torch.device) else torch.device(device)
    if mask.dtype == torch.bool and mask.device == dev:
        m = mask.view(torch.uint8)            # bools are 0 / 1 bytes already: no conversion pass over the mask
    elif mask.dtype == torch.uint8 and mask.device == dev:
        m = mask                              # the kernels test for non-zero
    else:
        m = (mask != 0).to(device=device, dtype=torch.uint8)
    if m.stride(3) not in (0, 1) or (m.shape[3] > 1 and m.stride(3) == 0):
        m = m.contiguous()
    return m


_ARANGE1 = {}


def _mask_bound(km: torch.Tensor) -> torch.Tensor:
    """int32 [B]: 1 + the index of the last non-zero byte of every row of a [B, Sk] uint8 mask (0 for an empty row)."""
    key = (km.shape[1], str(km.device))
    with _SEQLENS_LOCK:
        idx = _ARANGE1.get(key)
        if idx is None:
            if len(_ARANGE1) > 64:
                _ARANGE1.clear()
            idx = _ARANGE1[key] = torch.arange(1, km.shape[1] + 1, dtype=torch.int32, device=km.device)
    return (idx * (km != 0)).amax(dim=1).to(torch.int32)


def build_args(q, k, v, out, *, causal=False, seqlens_k=None, key_mask=None, softmax_scale=None,
               lse=None, split_p=False, variant=0, mask=None, drop_mask=None, drop_scale=1.0):
    """Fill a ``pfa_fa3_args`` from ``[B,H,S,D]``-shaped (arbitrarily strided) tensors."""
    B, H, Sq, D = q.shape
    Sk, Hkv = k.shape[2], k.shape[1]
    # grouped-query attention: k, v may carry H / g heads (query head h reads K/V head h // g); nothing is expanded
    if Hkv < 1 or H % Hkv or k.shape != (B, Hkv, Sk, D) or v.shape != (B, Hkv, Sk, D):
        raise ValueError(f"shape mismatch: q {tuple(q.shape)} k {tuple(k.shape)} v {tuple(v.shape)}")
    if out.shape != (B, H, Sq, D):
        raise ValueError("output shape mismatch")
    if q.dtype not in (torch.bfloat16, torch.float16, torch.float32) or k.dtype != q.dtype or v.dtype != q.dtype:
        raise ValueError("q, k, v must share dtype bf16, fp16 or fp32 (fp32: the exact, slow kernel)")
    if out.dtype not in (q.dtype, torch.float32):
        raise ValueError("output dtype must be the input dtype or fp32")
    if q.dtype == torch.float32 and (split_p or variant):
        raise ValueError("fp32 operands run the exact fp32 kernel: no split P, no kernel selector")
    if not (q.is_cuda and k.is_cuda and v.is_cuda and out.is_cuda):
        raise ValueError("pfa_fa3_fwd needs device tensors (there is no CPU path)")
    qs, ks, vs, os_ = (_bhsd_strides(t) for t in (q, k, v, out))
    a = _capi.make_args(
        flags=(_capi.PFA_FLAG_SPLIT_P if split_p else 0) | ((int(variant) & 0xFF) << 8),
        q=q.data_ptr(), k=k.data_ptr(), v=v.data_ptr(), o=out.data_ptr(),
        q_stride_b=qs[0], q_stride_h=qs[1], q_stride_s=qs[2],
        k_stride_b=ks[0], k_stride_h=ks[1], k_stride_s=ks[2],
        v_stride_b=vs[0], v_stride_h=vs[1], v_stride_s=vs[2],
        o_stride_b=os_[0], o_stride_h=os_[1], o_stride_s=os_[2],
        B=B, H=H, Sq=Sq, Sk=Sk, D=D,
        dtype_in=_DT[q.dtype], dtype_out=_DT[out.dtype], causal=1 if causal else 0,
        softmax_scale=float(D ** -0.5 if softmax_scale is None else softmax_scale),
        device_id=q.device.index if q.device.index is not None else torch.cuda.current_device(),
        kv_group=H // Hkv,
    )
    keep = []
    if seqlens_k is not None:
        sl = _seqlens_tensor(seqlens_k, q.device)
        if sl.numel() != B:
            raise ValueError("seqlens_k must have B entries")
        a.seqlens_k = sl.data_ptr()
        keep.append(sl)
    if key_mask is not None:
        km = key_mask
        if km.shape != (B, Sk):
            raise ValueError("key_mask must be [B, Sk]")
        if km.dtype == torch.bool and km.device == q.device:
            km = km.contiguous().view(torch.uint8)          # bools are 0 / 1 bytes already: no conversion kernel
        else:
            km = (km != 0).to(device=q.device, dtype=torch.uint8).contiguous()
        a.key_mask = km.data_ptr()
        a.key_mask_stride_b = km.stride(0)
        keep.append(km)
        # A padding mask's tail is dead weight the kernel cannot see coming (it reads the bytes two tiles ahead): hand it, as seqlens_k,
        # the position behind each row's LAST visible key -- two small device ops, no sync, nothing changes in the result (those keys are
        # masked anyway) -- and the persistent kernel cuts every item's tile count to it.  Only where that kernel takes the problem and
        # the launch is long enough to carry the two extra ops (>= ~0.15 ms of attention).
        if (seqlens_k is None and (not causal or Sq == Sk) and q.dtype != torch.float32 and D in (64, 128) and Sq >= 128 and Sk >= 193
                and 4.0 * B * H * Sq * Sk * D >= 1.5e11):       # (under the causal mask too since round 3: the padded decoder batch)
            sl = _mask_bound(km)
            a.seqlens_k = sl.data_ptr()
            keep.append(sl)
    if mask is not None:
        if key_mask is not None:
            raise ValueError("pass either key_mask or mask")
        m4 = _as_mask4(mask, B, H, Sq, Sk, q.device)
        a.mask = m4.data_ptr()
        st = [0 if m4.shape[i] == 1 else m4.stride(i) for i in range(4)]
        a.mask_stride_b, a.mask_stride_h, a.mask_stride_q, a.mask_stride_k = st[0], st[1], st[2], (st[3] or 1)
        keep.append(m4)
    if key_mask is not None or mask is not None:
        # scratch for the mask condensed to one 64-bit word per row and 64-key tile (pfa_fa3_workspace_bytes; optional for the
        # C ABI, always given here): a padding mask then costs about what seqlens_k costs, an element mask 1/3 of the byte path
        ws_bytes = int(_capi.load().pfa_fa3_workspace_bytes(C.byref(a)))
        if ws_bytes:
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=q.device)
            a.workspace, a.workspace_bytes = ws.data_ptr(), ws_bytes
            keep.append(ws)
    if lse is not None:
        if lse.shape != (B, H, Sq) or lse.dtype != torch.float32 or not lse.is_contiguous():
            raise ValueError("lse must be contiguous fp32 [B, H, Sq]")
        a.lse = lse.data_ptr()
    if drop_mask is not None:
        a.drop_mask, a.drop_scale = _drop_mask_ptr(drop_mask, B, H, Sq, Sk, q), float(drop_scale)
        keep.append(drop_mask)
    return a, keep


def fa3_forward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, causal: bool = False,
                seqlens_k=None, key_mask: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
                softmax_scale: Optional[float] = None, out_dtype: Optional[torch.dtype] = None,
                return_lse: bool = False, return_weights: bool = False, weights_dtype: Optional[torch.dtype] = None,
                split_p: Optional[bool] = None, drop_mask: Optional[torch.Tensor] = None, drop_scale: float = 1.0,
                out: Optional[torch.Tensor] = None, _variant: Optional[int] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """softmax(scale * q k^T + mask) v on the MI355X kernel.

    q: ``[B,H,Sq,D]``, k/v: ``[B,H,Sk,D]`` (any batch/head/seq strides that are multiples of 8
    elements; head_dim contiguous).  Returns ``(out [B,H,Sq,D] view of a [B,Sq,H,D] buffer, lse or None)``;
    with ``return_weights=True`` a third element, the softmax matrix ``[B,H,Sq,Sk]`` (second kernel pass,
    the reference's ``need_weights``).  ``mask``: any 2-/3-/4-D reference-style mask (0 = masked);
    ``key_mask``: the cheaper ``[B,Sk]`` special case.

    ``out_dtype=torch.float32`` selects the parity variant: fp32 store and, unless
    ``split_p=False`` is forced, P carried as bf16 hi+lo so the result is within 1e-3 of the
    fp32 reference ("numerics").
    """
    B, H, Sq, D = q.shape
    Dp = _padded_head_dim(D)
    if Dp != D:
        # no kernel instantiation for this head dim: run the next larger one on zero-padded operands (same scores, the
        # extra output columns are zero and dropped).  The scale stays the TRUE head dim's.
        res = fa3_forward(_pad_d(q, Dp), _pad_d(k, Dp), _pad_d(v, Dp), causal=causal, seqlens_k=seqlens_k,
                          key_mask=key_mask, mask=mask,
                          softmax_scale=float(D ** -0.5 if softmax_scale is None else softmax_scale), out_dtype=out_dtype,
                          return_lse=return_lse, return_weights=return_weights, weights_dtype=weights_dtype,
                          split_p=split_p, drop_mask=drop_mask, drop_scale=drop_scale, _variant=_variant)
        o = res[0][..., :D]
        if out is not None:
            out.copy_(o)
            o = out
        return (o,) + tuple(res[1:])
    if _variant is None:   # tools / tests only: kernel selector (include/pfa_hip.h PFA_FLAG_VARIANT_MASK); never read from the environment
        _variant = 0
    odt = q.dtype if out_dtype is None else out_dtype
    if q.dtype == torch.float32:
        # exact fp32 kernel (csrc/fa3_fwd_f32_kernel.h): every product and sum in fp32, ~two orders of magnitude slower than the
        # MFMA path; the softmax matrix, if wanted, comes from a 16-bit pass of its own (the weights kernel is MFMA only)
        if return_weights:
            raise ValueError("return_weights with fp32 operands: call again with 16-bit operands for the weights")
        if split_p or _variant:
            raise ValueError("fp32 operands run the exact fp32 kernel: no split P, no kernel selector")
        split_p = False
    elif split_p is None:
        split_p = odt == torch.float32
    if out is None:
        out = torch.empty((B, Sq, H, D), dtype=odt, device=q.device).permute(0, 2, 1, 3)
    lse = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device) if (return_lse or return_weights) else None
    args, keep = build_args(q, k, v, out, causal=causal, seqlens_k=seqlens_k, key_mask=key_mask, mask=mask,
                            softmax_scale=softmax_scale, lse=lse, split_p=split_p, variant=_variant,
                            drop_mask=drop_mask, drop_scale=drop_scale)
    stream = torch.cuda.current_stream(q.device).cuda_stream
    st = _capi.load().pfa_fa3_fwd(C.byref(args), C.c_void_p(stream))
    if st in (-3, -4, -5, -6, -7, -10):
        raise ValueError(f"pfa_fa3_fwd: {_capi.status_string(st)}")
    _capi.check_status(st)
    weights = None
    if return_weights:
        Sk = k.shape[2]
        wdt = q.dtype if weights_dtype is None else weights_dtype
        weights = torch.empty((B, H, Sq, Sk), dtype=wdt, device=q.device)     # the kernel writes every element, masked ones as zeros
        stw = _capi.load().pfa_fa3_weights(C.byref(args), C.c_void_p(weights.data_ptr()), _DT[wdt],
                                           weights.stride(0), weights.stride(1), weights.stride(2), C.c_void_p(stream))
        if stw in (-3, -4, -5, -6, -7, -10):
            raise ValueError(f"pfa_fa3_weights: {_capi.status_string(stw)}")
        _capi.check_status(stw)
    for t in keep:   # tensors made here must outlive the enqueued kernels
        t.record_stream(torch.cuda.current_stream(q.device))
    if return_weights:
        return out, (lse if return_lse else None), weights
    return out, lse


def fa3_forward_bshd(q, k, v, **kw):
    """Same, for operands laid out ``[B,S,H,D]``; returns ``[B,Sq,H,D]`` (+ lse)."""
    res = fa3_forward(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), **kw)
    return (res[0].permute(0, 2, 1, 3),) + tuple(res[1:])


def fa3_backward(q, k, v, out, dout, lse, *, causal: bool = False, seqlens_k=None, key_mask=None, mask=None,
                 softmax_scale: Optional[float] = None, grad_dtype: Optional[torch.dtype] = None,
                 drop_mask: Optional[torch.Tensor] = None, drop_scale: float = 1.0):
    """dQ, dK, dV of ``fa3_forward`` (``pfa_fa3_bwd``).  All operands ``[B,H,S,D]``-shaped (any strides, head dim
    contiguous), ``lse`` the forward's ``[B,H,Sq]`` fp32 LSE.  Returns gradients as ``[B,H,S,D]`` views of
    ``[B,S,H,D]`` buffers, in ``grad_dtype`` (input dtype by default, or fp32).  ``key_mask`` / ``mask``: the masks
    the forward was called with (same conventions as ``fa3_forward``).  Grouped-query heads: ``k`` / ``v`` may hold ``H / g`` heads as in
    the forward; ``dk`` / ``dv`` then have ``H / g`` heads too (summed over each group inside the dK/dV kernel, ABI v7)."""
    B, H, Sq, D = q.shape
    Sk = k.shape[2]
    Hkv = k.shape[1]
    if v.shape[1] != Hkv or Hkv <= 0 or H % Hkv:
        raise ValueError(f"k / v carry {k.shape[1]} / {v.shape[1]} heads: both must hold H / g heads for an integer g (H = {H})")
    if Hkv != H and q.dtype == torch.float32:
        # the exact-fp32 kernels take one K/V head per query head: expand, and sum dK / dV over each group here
        g = H // Hkv
        dq, dk, dv = fa3_backward(q, k.repeat_interleave(g, dim=1), v.repeat_interleave(g, dim=1), out, dout, lse, causal=causal,
                                  seqlens_k=seqlens_k, key_mask=key_mask, mask=mask, softmax_scale=softmax_scale, grad_dtype=grad_dtype,
                                  drop_mask=drop_mask, drop_scale=drop_scale)
        return dq, dk.reshape(B, Hkv, g, Sk, D).sum(2), dv.reshape(B, Hkv, g, Sk, D).sum(2)
    Dp = _padded_head_dim(D)
    if Dp != D:   # as in fa3_forward: zero-padded head dim; the gradients' extra columns are exactly zero and dropped
        grads = fa3_backward(_pad_d(q, Dp), _pad_d(k, Dp), _pad_d(v, Dp), _pad_d(out, Dp), _pad_d(dout, Dp), lse,
                             causal=causal, seqlens_k=seqlens_k, key_mask=key_mask, mask=mask,
                             softmax_scale=float(D ** -0.5 if softmax_scale is None else softmax_scale),
                             grad_dtype=grad_dtype, drop_mask=drop_mask, drop_scale=drop_scale)
        return tuple(g[..., :D] for g in grads)
    gdt = q.dtype if grad_dtype is None else grad_dtype
    if dout.stride(3) != 1:
        dout = dout.contiguous()
    dq = torch.empty((B, Sq, H, D), dtype=gdt, device=q.device).permute(0, 2, 1, 3)
    dk = torch.empty((B, Sk, Hkv, D), dtype=gdt, device=q.device).permute(0, 2, 1, 3)      # grouped-query heads: dK / dV summed over
    dv = torch.empty((B, Sk, Hkv, D), dtype=gdt, device=q.device).permute(0, 2, 1, 3)      # each group inside the kernel (ABI v7)
    delta = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)
    a = _capi.PfaFa3BwdArgs()
    a.size = C.sizeof(_capi.PfaFa3BwdArgs)
    for name, t in (("q", q), ("k", k), ("v", v), ("o", out), ("dout", dout), ("dq", dq), ("dk", dk), ("dv", dv)):
        setattr(a, name, t.data_ptr())
        sb, sh, ss = _bhsd_strides(t)
        pre = "do" if name == "dout" else name
        setattr(a, f"{pre}_stride_b", sb); setattr(a, f"{pre}_stride_h", sh); setattr(a, f"{pre}_stride_s", ss)
    if lse.shape != (B, H, Sq) or lse.dtype != torch.float32 or not lse.is_contiguous():
        raise ValueError("lse must be contiguous fp32 [B, H, Sq]")
    a.lse, a.delta = lse.data_ptr(), delta.data_ptr()
    keep = [delta]
    if seqlens_k is not None:
        sl = _seqlens_tensor(seqlens_k, q.device)
        a.seqlens_k = sl.data_ptr()
        keep.append(sl)
    if key_mask is not None:
        if mask is not None:
            raise ValueError("pass either key_mask or mask")
        if key_mask.shape != (B, Sk):
            raise ValueError("key_mask must be [B, Sk]")
        mask = key_mask
    if mask is not None:
        m4 = _as_mask4(mask, B, H, Sq, Sk, q.device)
        a.mask = m4.data_ptr()
        st = [0 if m4.shape[i] == 1 else m4.stride(i) for i in range(4)]
        a.mask_stride_b, a.mask_stride_h, a.mask_stride_q, a.mask_stride_k = st[0], st[1], st[2], (st[3] or 1)
        keep.append(m4)
    if drop_mask is not None:
        a.drop_mask, a.drop_scale = _drop_mask_ptr(drop_mask, B, H, Sq, Sk, q), float(drop_scale)
        keep.append(drop_mask)
    a.B, a.H, a.Sq, a.Sk, a.D = B, H, Sq, Sk, D
    a.kv_group = H // Hkv
    a.dtype, a.dtype_grad, a.causal = _DT[q.dtype], _DT[gdt], 1 if causal else 0
    a.softmax_scale = float(D ** -0.5 if softmax_scale is None else softmax_scale)
    a.device_id = q.device.index if q.device.index is not None else torch.cuda.current_device()
    stream = torch.cuda.current_stream(q.device)
    if a.mask:   # element masks: scratch for the condensed words / tile ranges (0 bytes for key-only masks; optional for the library)
        mws = int(_capi.load().pfa_fa3_bwd_mask_workspace_bytes(C.byref(a)))
        if mws:
            ws = torch.empty(mws, dtype=torch.uint8, device=q.device)
            a.mask_workspace, a.mask_workspace_bytes = ws.data_ptr(), mws
            keep.append(ws)
    st = _capi.load().pfa_fa3_bwd(C.byref(a), C.c_void_p(stream.cuda_stream))
    if st in (-3, -4, -5, -6, -7, -10):
        raise ValueError(f"pfa_fa3_bwd: {_capi.status_string(st)}")
    _capi.check_status(st)
    for t in keep:
        t.record_stream(stream)
    return dq, dk, dv


class _FA3Function(torch.autograd.Function):
    """Differentiable ``fa3_forward`` (causal / seqlens_k / key / element masks): saves q, k, v, o and the LSE.
    ``out_dtype`` fp32 (fp32 modules) runs the parity forward (fp32 store) and keeps a 16-bit copy of O for the
    backward's delta = rowsum(dO o O)."""

    @staticmethod
    def forward(ctx, q, k, v, causal, seqlens_k, softmax_scale, key_mask, mask, out_dtype, want_weights, weights_dtype):
        res = fa3_forward(q, k, v, causal=causal, seqlens_k=seqlens_k, key_mask=key_mask, mask=mask,
                          softmax_scale=softmax_scale, return_lse=True, out_dtype=out_dtype,
                          return_weights=want_weights, weights_dtype=weights_dtype)
        out, lse = res[0], res[1]
        o16 = out if out.dtype == q.dtype else out.to(q.dtype)     # (fp32 operands: the fp32 kernels in both directions, nothing is narrowed)
        ctx.save_for_backward(q, k, v, o16, lse)
        ctx.causal, ctx.seqlens_k, ctx.softmax_scale = causal, seqlens_k, softmax_scale
        ctx.key_mask, ctx.mask = key_mask, mask          # masks carry no gradient
        if want_weights:
            # the softmax matrix from the second pass on the saved LSE: returned for inspection (nn.MultiheadAttention's
            # default need_weights=True), detached -- the gradient flows through the output only
            ctx.mark_non_differentiable(res[2])
            return out, res[2]
        return out

    @staticmethod
    def backward(ctx, dout, *_dweights):
        q, k, v, out, lse = ctx.saved_tensors
        dq, dk, dv = fa3_backward(q, k, v, out, dout.to(q.dtype), lse, causal=ctx.causal, seqlens_k=ctx.seqlens_k,
                                  key_mask=ctx.key_mask, mask=ctx.mask, softmax_scale=ctx.softmax_scale)
        return dq, dk, dv, None, None, None, None, None, None, None, None


class _FA3DropoutFunction(torch.autograd.Function):
    """Attention with dropout on the softmax weights (the reference's dense branch, flash_attention_3.py:174-175), forward and
    backward on the fp32 kernels.  The keep-mask is drawn on the device with torch's generator (``torch.manual_seed`` governs it,
    as it governs ``nn.Dropout`` in the reference) and replayed by the backward."""

    @staticmethod
    def forward(ctx, q, k, v, p_drop, causal, softmax_scale, key_mask, mask):
        B, H, Sq, _ = q.shape
        Sk = k.shape[2]
        keep = torch.rand((B, H, Sq, Sk), device=q.device) >= p_drop
        scale = 1.0 / (1.0 - p_drop)
        q32, k32, v32 = q.float(), k.float(), v.float()
        out, lse = fa3_forward(q32, k32, v32, causal=causal, key_mask=key_mask, mask=mask, softmax_scale=softmax_scale,
                               return_lse=True, drop_mask=keep, drop_scale=scale)
        ctx.save_for_backward(q32, k32, v32, out, lse, keep)
        ctx.meta = (causal, softmax_scale, key_mask, mask, scale, q.dtype)
        return out.to(q.dtype)

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse, keep = ctx.saved_tensors
        causal, softmax_scale, key_mask, mask, scale, dt = ctx.meta
        dq, dk, dv = fa3_backward(q, k, v, out, dout.float(), lse, causal=causal, key_mask=key_mask, mask=mask,
                                  softmax_scale=softmax_scale, drop_mask=keep, drop_scale=scale)
        return dq.to(dt), dk.to(dt), dv.to(dt), None, None, None, None, None


def fa3_attention_dropout(q, k, v, p_drop: float, *, causal: bool = False, key_mask=None, mask=None,
                          softmax_scale: Optional[float] = None):
    """``dropout(softmax(scale q k^T + mask), p_drop) v`` on ``[B,H,S,D]`` operands of any float dtype, differentiable; fp32 kernels
    (meant for the short sequences of the reference's dense branch)."""
    if not 0.0 <= p_drop < 1.0:
        raise ValueError("dropout probability must be in [0, 1)")
    return _FA3DropoutFunction.apply(q, k, v, float(p_drop), causal, softmax_scale, key_mask, mask)


def fa3_attention(q, k, v, *, causal: bool = False, seqlens_k=None, key_mask=None, mask=None,
                  softmax_scale: Optional[float] = None, out_dtype: Optional[torch.dtype] = None,
                  return_weights: bool = False, weights_dtype: Optional[torch.dtype] = None):
    """Autograd-aware attention on ``[B,H,S,D]`` bf16/fp16 operands: forward + backward on the HIP kernels.
    ``return_weights=True`` -> ``(out, weights)``; the weights are detached (no gradient flows through them)."""
    return _FA3Function.apply(q, k, v, causal, seqlens_k, softmax_scale, key_mask, mask, out_dtype, bool(return_weights), weights_dtype)
