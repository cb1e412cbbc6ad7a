#!/usr/bin/env python3
"""Phase stamps of the 4-wave x 64-row forward: cycles per wave-tile.  Needs the diagnostic build of the library
(`make -C photonic_flash_attention_amd/csrc clean && make -C photonic_flash_attention_amd/csrc W4FLAGS=-DPFA_W4_STAMP`);
rebuild without the flag afterwards -- a stamped kernel is ~10 % slower and must never be benchmarked."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photonic_flash_attention_amd import ops, _capi
B, H, S, D, causal = (4, 16, 4096, 128, len(sys.argv) > 1 and sys.argv[1] == "causal")
dev = torch.device("cuda:0")
q, k, v = (torch.randn(B, H, S, D, device=dev).to(torch.bfloat16) for _ in range(3))
if os.environ.get("ZEROS"): q, k, v = (torch.zeros_like(t) for t in (q, k, v))
out = torch.empty(B, S, H, D, device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3)
nwg = B * H * (S // 256)
dbg = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
args, keep = ops.build_args(q, k, v, out, causal=causal, variant=43)
args.workspace = dbg.data_ptr(); args.workspace_bytes = dbg.numel() * 8
for _ in range(300):
    st = _capi.load().pfa_fa3_fwd(C.byref(args), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0, st
torch.cuda.synchronize()
d = dbg.view(nwg, 4, 8).double().cpu()
tiles = d[..., 4].sum()
for i, n in enumerate(["phase A (QK + finish softmax + DMA issue)", "phase B (PV + start softmax)", "vmcnt + barrier", "rescale check"]):
    print(f"  {n:45s} {d[..., i].sum() / tiles:8.0f} cyc/tile")
print(f"  total {d[..., :4].sum() / tiles:.0f} cycles per wave-tile over {int(tiles)} wave-tiles")
w = d[..., 5:8].mean(dim=(0, 1))
print(f"  per workgroup-wave: prologue {w[0]:.0f}, loop {w[1]:.0f}, epilogue (to last store done) {w[2]:.0f} cycles")
