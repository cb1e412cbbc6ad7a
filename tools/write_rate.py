#!/usr/bin/env python3
"""The box's fill / copy / elementwise rates on a 2 GiB tensor: the write-bandwidth ceiling the need_weights pass is read against."""
import torch, statistics
x = torch.empty(2147483648 // 2, dtype=torch.bfloat16, device="cuda")
y = torch.empty_like(x)
def t(f, n=10):
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return statistics.median(ts)
for _ in range(3): x.zero_(); y.copy_(x)
ms = t(lambda: x.zero_()); print(f"zero_ 2147 MB: {ms:.3f} ms = {2147.48 / ms / 1e3:.2f} TB/s written")
ms = t(lambda: x.fill_(1.5)); print(f"fill_ 2147 MB: {ms:.3f} ms = {2147.48 / ms / 1e3:.2f} TB/s written")
ms = t(lambda: y.copy_(x)); print(f"copy 2147 MB: {ms:.3f} ms = {2 * 2147.48 / ms / 1e3:.2f} TB/s read+written")
ms = t(lambda: torch.exp2(x, out=y)); print(f"exp2 out=: {ms:.3f} ms = {2 * 2147.48 / ms / 1e3:.2f} TB/s read+written")
# column slabs of a [R, 4096] bf16 matrix: W-like row segments of 128 .. 1024 bytes at an 8 KiB pitch
R = 262144
w = x[: R * 4096].view(R, 4096)
for cols in (64, 128, 256, 512, 1024, 4096):
    for _ in range(2): w[:, :cols].zero_()
    ms = t(lambda: w[:, :cols].zero_())
    print(f"slab of {cols * 2:5d}-byte row segments: {R * cols * 2 / 1e6:.0f} MB in {ms:.3f} ms = {R * cols * 2 / ms / 1e9:.2f} TB/s")
