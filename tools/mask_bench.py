import sys, os, statistics, torch
sys.path.insert(0, "/root/repo")
from photonic_flash_attention_amd import ops, _capi
dev = torch.device("cuda:0")
B, H, S, D = 4, 16, 4096, 128
q, k, v = (torch.randn(B, S, H, D, device=dev).to(torch.bfloat16).permute(0, 2, 1, 3) for _ in range(3))
out = torch.empty(B, S, H, D, device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3)
tril = torch.ones(S, S, device=dev, dtype=torch.bool).tril()[None, None]
win = (torch.arange(S, device=dev)[:, None] - torch.arange(S, device=dev)[None, :])
sw = ((win >= 0) & (win < 1024))[None, None]
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / n)
    return statistics.median(ts)
fl = 2 * B * H * S * S * D
for name, kw in (("causal flag", dict(causal=True)), ("tril as a 4-D element mask", dict(mask=tril)), ("sliding window 1024 as a 4-D mask", dict(mask=sw)), ("no mask", {})):
    ms = t(lambda: ops.fa3_forward(q, k, v, out=out, **kw))
    nm = _capi.describe(ops.build_args(q, k, v, out, **kw)[0])[0]
    print(f"{name}: {ms*1e3:.1f} us  dense-equivalent {2*fl/ms/1e9:.0f} TF (causal-equivalent {fl/ms/1e9:.0f})  {nm}", flush=True)
