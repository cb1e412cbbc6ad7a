#!/usr/bin/env python3
"""debug: where a variant's LSE / output differ from the base build under spiked keys"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import p4_ab as A
dev = torch.device("cuda:0"); torch.zeros(1, device=dev)
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
B, H, S, D, causal = 1, 8, int(sys.argv[2]) if len(sys.argv) > 2 else 1024, 128, (sys.argv[3] == "c") if len(sys.argv) > 3 else False
spike, key = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0, int(sys.argv[4]) if len(sys.argv) > 4 else 300
g = torch.Generator(device=dev).manual_seed(1)
q, k, v = (torch.randn(B, S, H, D, device=dev, generator=g).to(torch.bfloat16).permute(0, 2, 1, 3) for _ in range(3))
k.permute(0, 2, 1, 3)[:, key] *= spike
st = torch.cuda.current_stream().cuda_stream
res = {}
for name in ("base", "fm"):
    vr = A.Variant(name)
    out = torch.full((B, S, H, D), float("nan"), device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3)
    lse = torch.full((B, H, S), float("nan"), device=dev)
    buf, grid = A.kernargs(q, k, v, out, lse, causal, n_cu)
    A.launch(vr.fn(A.kname("bf16", D, causal)), buf, grid, st)
    torch.cuda.synchronize()
    res[name] = (out.float().cpu(), lse.cpu())
dl = (res["fm"][1] - res["base"][1])
do = (res["fm"][0] - res["base"][0]).abs().amax(-1)
for h in range(min(H, 2)):
    bad = (dl[0, h].abs() > 1e-3).nonzero().flatten().tolist()
    print(f"head {h}: rows with LSE diff: {len(bad)}; first {bad[:8]} last {bad[-4:]}")
    for r in bad[:6] + bad[-3:]:
        print(f"   row {r}: LSE base {res['base'][1][0, h, r]:.4f} fm {res['fm'][1][0, h, r]:.4f}  d {dl[0, h, r]:.4f}  max|dO| {do[0, h, r]:.3e}")
    # reference LSE
    s = (q[0, h].float() @ k[0, h].float().T) * D ** -0.5
    if causal:
        s = s.masked_fill(torch.arange(S, device=dev)[None, :] > torch.arange(S, device=dev)[:, None], float("-inf"))
    ref = torch.logsumexp(s, -1).cpu()
    print(f"   vs exact: base max err {float((res['base'][1][0, h] - ref).abs().max()):.3e}  fm {float((res['fm'][1][0, h] - ref).abs().max()):.3e}")
    mxs = s.max(-1).values.cpu()
    for r in bad[:3]:
        print(f"   row {r}: exact row max {mxs[r]:.2f} (log2 units {mxs[r] * 1.4427:.2f}), exact LSE {ref[r]:.4f}")
print("---- anatomy of the first bad rows (log2 units, relative to the tile-0 maximum m0)")
h = 0
s = (q[0, h].float() @ k[0, h].float().T) * D ** -0.5 * 1.4426950408889634        # log2 units
t = key // 64
bad = (dl[0, h].abs() > 1e-3).nonzero().flatten().tolist()
for r in bad[:10]:
    row = s[r].cpu()
    m0 = float(row[:64].max())
    l_old = float(torch.exp2(row[:64 * t] - m0).sum())
    x = row[64 * t:64 * t + 64] - m0
    mx = float(x.max())
    tile_sum = float(torch.exp2(x - mx).sum())
    l_rest = float(torch.exp2(row[64 * t + 64:] - m0 - mx).sum())
    l_true = l_old * 2 ** (-mx) + tile_sum + l_rest
    err = float(dl[0, h, r]) / 0.6931
    print(f"row {r}: mx {mx:.2f}  l_old {l_old:.2f}  tile_sum(new units) {tile_sum:.3f} l_rest {l_rest:.3f}  l_true {l_true:.3f}; LSE err {err:.3f} log2 -> l_fm/l_true = {2 ** err:.3f}; "
          f"hyp alpha=1: {(l_old + tile_sum + l_rest) / l_true:.3f}  hyp tile twice: {(l_true + tile_sum) / l_true:.3f}; lane of spike h={(((key % 64) % 32) // 4) % 2}")
