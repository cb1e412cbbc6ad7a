#!/usr/bin/env python3
"""Cycle buckets of the persistent assembly forward.  Needs the diagnostic build:
    rm photonic_flash_attention_amd/csrc/build/fa3_fwd_p4.s; P4_STAMP=1 make -C photonic_flash_attention_amd/csrc
(rebuild without P4_STAMP afterwards; a stamped kernel is slower and is never benchmarked)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photonic_flash_attention_amd import ops, _capi
CFG = {"C3": (4, 16, 4096, True), "C4": (4, 16, 4096, False), "C5": (1, 32, 16384, True), "S2K": (16, 16, 2048, False)}
dev = torch.device("cuda:0")
for name in (sys.argv[1:] or ["C3", "C4"]):
    B, H, S, causal = CFG[name]
    q, k, v = (torch.randn(B, S, H, 128, device=dev).to(torch.bfloat16).permute(0, 2, 1, 3) for _ in range(3))
    out = torch.empty(B, S, H, 128, device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3)
    dbg = torch.zeros(256 * 4 * 16, dtype=torch.int32, device=dev)
    args, keep = ops.build_args(q, k, v, out, causal=causal, variant=45)
    args.workspace = dbg.data_ptr(); args.workspace_bytes = dbg.numel() * 4
    for _ in range(200):
        st = _capi.load().pfa_fa3_fwd(C.byref(args), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert st == 0, st
    torch.cuda.synchronize()
    d = (dbg.view(256, 4, 16).long() & 0xffffffff).double().cpu()
    acc = d[..., 1:]                       # bucket k at index k
    nfull, nitems = acc[..., 7].sum(), acc[..., 8].sum()
    tot = acc[..., 10]
    if nfull == 0:                         # P4_STAMP=3: kernel totals only (the production control flow: lean loop, pipelined seam)
        clk = (tot / (acc[..., 11] * 10e-9) / 1e9)
        print(f"{name}: kernel cycles per wave: mean {tot.mean():.0f} (min {tot.min():.0f} max {tot.max():.0f}); in-kernel clock {clk.mean():.3f} GHz; "
              f"wall per wave {acc[..., 11].mean() * 10 / 1e3:.1f} us (max {acc[..., 11].max() * 10 / 1e3:.1f})")
        wall = acc[..., 11] * 10 / 1e3                       # us, [workgroup][wave]
        nx = 8
        per = [wall[x::nx].mean().item() for x in range(nx)]
        print("   wall per wave by workgroup % 8 (= XCD): " + " ".join(f"{v:.1f}" for v in per) + " us;  by wave: " +
              " ".join(f"{wall[:, w].mean().item():.1f}" for w in range(4)))
        continue
    print(f"{name}: FULL iterations {int(nfull)}, items {int(nitems)} (per workgroup-wave)")
    print(f"  per FULL iteration: QK^T phase {acc[..., 0].sum() / nfull:7.0f}   PV phase {acc[..., 1].sum() / nfull:7.0f} cycles")
    print(f"  wait + barrier + bookkeeping (all iterations) {acc[..., 2].sum() / nfull:7.0f} per FULL iteration")
    print(f"  per item: switch + prologue {acc[..., 3].sum() / nitems:7.0f}   epilogue {acc[..., 4].sum() / nitems:7.0f}   "
          f"LAST bodies {acc[..., 5].sum() / nitems:7.0f}   SKIP bodies {acc[..., 6].sum() / nitems:7.0f}")
    print(f"  per FULL iteration: vmcnt wait {acc[..., 9].sum() / nfull:6.0f}   barrier {acc[..., 12].sum() / nfull:6.0f}  (fine stamps only)")
    share = [acc[..., i].sum() / tot.sum() * 100 for i in range(7)]
    print("  share of wave time (percent): QK %.1f  PV %.1f  sync %.1f  prologue %.1f  epilogue %.1f  last %.1f  skip %.1f  (sum %.1f)" % (*share, sum(share)))
    clk = (tot / (acc[..., 11] * 10e-9) / 1e9)
    print(f"  kernel cycles per wave: mean {tot.mean():.0f} (min {tot.min():.0f} max {tot.max():.0f}); in-kernel clock {clk.mean():.3f} GHz; "
          f"wall per wave {acc[..., 11].mean() * 10 / 1e3:.1f} us (max {acc[..., 11].max() * 10 / 1e3:.1f})")
    for w in range(4):
        print(f"    wave {w}: QK {acc[:, w, 0].sum() / acc[:, w, 7].sum():6.0f}  PV {acc[:, w, 1].sum() / acc[:, w, 7].sum():6.0f}  sync/iter {acc[:, w, 2].sum() / acc[:, w, 7].sum():6.0f}")
