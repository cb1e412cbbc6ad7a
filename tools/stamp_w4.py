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
dbg = torch.zeros(nwg * 4 * 24, dtype=torch.int64, device=dev)
args, keep = ops.build_args(q, k, v, out, causal=causal, variant=43)
args.workspace = dbg.data_ptr(); args.workspace_bytes = dbg.numel() * 8
for _ in range(300):
    st = _capi.load().pfa_fa3_fwd(C.byref(args), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0, st
torch.cuda.synchronize()
d = dbg.view(nwg, 4, 24).double().cpu()
tiles = d[..., 4].sum()
for i, n in enumerate(["phase A (QK + finish softmax + DMA issue)", "phase B (PV + start softmax)", "vmcnt + barrier", "rescale check"]):
    print(f"  {n:45s} {d[..., i].sum() / tiles:8.0f} cyc/tile")
print(f"  total {d[..., :4].sum() / tiles:.0f} cycles per wave-tile over {int(tiles)} wave-tiles")
w = d[..., 5:8].mean(dim=(0, 1))
print(f"  per workgroup-wave: prologue {w[0]:.0f}, loop {w[1]:.0f}, epilogue (to last store done) {w[2]:.0f} cycles; "
      f"outside the loop: {100 * (w[0] + w[2]) / w.sum():.1f} % of the workgroup's residency")
x = d[..., 8:13].mean(dim=(0, 1))
print(f"  prologue, cumulative from kernel entry: setup done {d[..., 16].mean():.0f}, loads issued {x[0]:.0f}, Q + K0 arrived {x[1]:.0f}, barrier {x[2]:.0f}, "
      f"QK^T(0) + softmax start {x[3]:.0f}, V0 / K1 published {w[0]:.0f}")
print(f"  epilogue: normalise + LDS staging + stores issued {x[4]:.0f}, stores acknowledged {w[2]:.0f}")
# ---- per-CU timelines from the absolute 100 MHz stamps: gaps between consecutive workgroups on a CU, and the tail
raw = dbg.view(nwg, 4, 24)[:, 0, :].cpu()
start, end, hw = raw[:, 13], raw[:, 14], raw[:, 15]
cu_key = ((hw >> 32) & 0xF) * 4096 + ((hw >> 13) & 7) * 256 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 15)    # xcc, se, sh, cu
t0k, t1k = int(start.min()), int(end.max())
gaps, busy, last_end, first_start = [], [], [], []
for key in cu_key.unique().tolist():
    sel = (cu_key == key).nonzero().flatten()
    order = sel[start[sel].argsort()]
    st, en = start[order], end[order]
    gaps += (st[1:] - en[:-1]).tolist()
    busy.append(int((en - st).sum())); last_end.append(int(en.max())); first_start.append(int(st.min()))
g = torch.tensor(gaps, dtype=torch.float64) * 10.0
le, fs = torch.tensor(last_end, dtype=torch.float64), torch.tensor(first_start, dtype=torch.float64)
span = (t1k - t0k) * 10.0
print(f"  CUs seen {len(busy)}; kernel span {span / 1e3:.1f} us; workgroup residency per CU {torch.tensor(busy, dtype=torch.float64).mean() * 10 / 1e3:.1f} us "
      f"({100 * torch.tensor(busy, dtype=torch.float64).mean() * 10 / span:.1f} % of the span)")
print(f"  gap between consecutive workgroups on a CU: median {g.median():.0f} ns, mean {g.mean():.0f} ns, max {g.max():.0f} ns")
print(f"  first start after kernel start: mean {(fs.mean() - t0k) * 10:.0f} ns; idle tail before kernel end: mean {(t1k - le.mean()) * 10:.0f} ns, max {(t1k - le.min()) * 10:.0f} ns")
