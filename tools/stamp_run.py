#!/usr/bin/env python3
"""Run the diagnostic VAR_STAMP kernel (variant 16) and print per-phase cycle shares."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photonic_flash_attention_amd import ops, _capi
B, H, S, D, causal = (4, 16, 4096, 128, len(sys.argv) > 1 and sys.argv[1] == "causal")
dev = torch.device("cuda:0")
q, k, v = (torch.randn(B, H, S, D, device=dev).to(torch.bfloat16) for _ in range(3))
out = torch.empty(B, S, H, D, device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3)
nwg = B * H * (S // 256)
dbg = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device=dev)
args, keep = ops.build_args(q, k, v, out, causal=causal, variant=16)
args.workspace = dbg.data_ptr(); args.workspace_bytes = dbg.numel() * 8
for _ in range(int(os.environ.get("STAMP_ITERS", "600"))):   # long enough to be in the steady DVFS state
    st = _capi.load().pfa_fa3_fwd(C.byref(args), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0, st
torch.cuda.synchronize()
d = dbg.view(nwg, 8, 8).double().cpu()
tiles = d[..., 4].sum()
names = ["dma issue", "compute (QK+softmax+PV)", "vmcnt wait", "lgkm+barrier", "tiles", "  of which QK^T segment"]
tot = d[..., :4].sum()
print(f"{'causal' if causal else 'full'}: {int(tiles)} wave-tiles, {tot / tiles:.0f} cycles per wave-tile (stamps add ~40 cyc each)")
for i in (0, 1, 5, 2, 3):
    print(f"  {names[i]:28s} {d[..., i].sum() / tiles:8.0f} cyc/tile  {100 * d[..., i].sum() / tot:5.1f} %")
clk = (d[..., 6] / d[..., 7].clamp(min=1) * 100.0)
print(f"  in-kernel shader clock (s_memtime / s_memrealtime x 100 MHz): median {clk.median():.0f} MHz, "
      f"min {clk.min():.0f}, max {clk.max():.0f}")
w = d[..., :4].sum(-1) / d[..., 4].clamp(min=1)
print("  per-wave cycles/tile by wave id:", [f"{w[:, i].mean():.0f}" for i in range(8)])
for i in (1, 3):
    x = d[..., i] / d[..., 4].clamp(min=1)
    print(f"  {names[i]} by wave id:", [f"{x[:, j].mean():.0f}" for j in range(8)])
