#!/usr/bin/env python3
"""Same-process A/B of persistent-forward code objects (gen_fa3_fwd_p4.py builds), WITHOUT going through libpfa_hip.so: each
variant's .hsaco is loaded here with hipModuleLoadData and launched with hipModuleLaunchKernel on the kernarg block pfa_p4.hip
would build.  The product library therefore never carries an experimental, stamped or ablated code object.

    python3 tools/p4_variants.py base= fastmax=P4_FASTMAX=1 ...          (build container: writes csrc/build/variants/<name>.hsaco)
    python3 tools/p4_ab.py --variants base,fastmax --shapes C3,C4 [--check] [--stamp]         (GPU box)

--check : every variant's output / LSE against the first variant's (bitwise share, max-abs) and against the 8-wave HIP kernel (selector 44)
--stamp : the variants are P4_STAMP builds: dump their cycle buckets instead of timing them
Timing: rounds of `--steps` launches per variant, interleaved, HIP events; median and min over `--rounds` (cdna guide rule 24).
"""
import argparse
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "photonic_flash_attention_amd", "csrc")
sys.path.insert(0, CSRC)
for _k in [k for k in os.environ if k.startswith("P4_")]:
    del os.environ[_k]
import gen_fa3_fwd_p4 as G  # noqa: E402  (kernarg layout only)

SHAPES = {"C3": (4, 16, 4096, 128, True), "C4": (4, 16, 4096, 128, False), "C5": (1, 32, 16384, 128, True),
          "S2K": (16, 16, 2048, 128, False), "S1Kc": (16, 16, 1024, 128, True), "C2": (4, 12, 1024, 64, False),
          "D64S2K": (16, 16, 2048, 64, False), "D64S4Kc": (4, 16, 4096, 64, True), "S8Kc": (2, 16, 8192, 128, True),
          "small": (1, 8, 512, 128, True), "small2": (2, 8, 768, 128, False)}

hip = C.CDLL("libamdhip64.so")


def chk(e, what):
    if e != 0:
        raise RuntimeError(f"{what}: hipError {e}")


class Variant:
    def __init__(self, name):
        self.name = name
        path = os.path.join(CSRC, "build", "fa3_fwd_p4.hsaco") if name == "prod" else os.path.join(CSRC, "build", "variants", name + ".hsaco")
        self.blob = open(path, "rb").read()
        self.mod = C.c_void_p()
        chk(hip.hipModuleLoadData(C.byref(self.mod), self.blob), f"hipModuleLoadData({path})")
        self.fns = {}

    def fn(self, kname):
        if kname not in self.fns:
            f = C.c_void_p()
            chk(hip.hipModuleGetFunction(C.byref(f), self.mod, kname.encode()), f"{self.name}: {kname}")
            self.fns[kname] = f
        return self.fns[kname]


def magic(d):
    return 0 if d <= 1 else ((1 << 32) // d + 1) & 0xffffffff


def kernargs(q, k, v, o, lse, causal, n_cu, dbg=None, reserve=0, extra=None):
    """q, k, v, o: [B,H,S,D] views; -> (bytes-like ctypes buffer, grid)"""
    B, H, Sq, D = q.shape
    Sk = k.shape[2]
    buf = (C.c_uint32 * G.KARG_MEM_DWORDS)()
    KA = G.KA

    def put(name, val):
        buf[KA[name]] = val & 0xffffffff

    def put64(name, val):
        buf[KA[name]] = val & 0xffffffff
        buf[KA[name] + 1] = (val >> 32) & 0xffffffff
    osz = o.element_size()
    put64("q", q.data_ptr()); put64("k", k.data_ptr()); put64("v", v.data_ptr()); put64("o", o.data_ptr())
    put64("lse", lse.data_ptr() if lse is not None else 0)
    for nm, t, sz in (("q", q, 2), ("k", k, 2), ("v", v, 2), ("o", o, osz)):
        put(nm + "_sb", t.stride(0) * sz); put(nm + "_sh", t.stride(1) * sz); put(nm + "_ss", t.stride(2) * sz)
    NB = (Sq + 255) // 256
    NU = (NB + 1) // 2 if causal else NB
    put("H", H); put("Sq", Sq); put("Sk", Sk); put("NB", NB); put("NU", NU)
    put("magic_NU", magic(NU)); put("magic_H", magic(H)); put("kv_group", 1); put("magic_G", 0)
    import struct
    sl2 = float(D) ** -0.5 * 1.4426950408889634
    put("scale_log2", struct.unpack("I", struct.pack("f", sl2))[0])
    put("thr", struct.unpack("I", struct.pack("f", struct.unpack("f", struct.pack("f", 8.0))[0] / struct.unpack("f", struct.pack("f", sl2))[0]))[0])
    BH = B * H
    n = max(8, n_cu - reserve)
    grid = (n // 8) * 8 if BH % 8 == 0 else n
    xm = 1 if BH % 8 == 0 else 0
    put("xcd_mode", xm); put("hx", BH // 8 if xm else BH); put("SL", grid // 8 if xm else grid)
    put("nt_full", ((Sk + 127) // 128) * 2)
    put64("dbg", dbg.data_ptr() if dbg is not None else 0)
    for kk, vv in (extra or {}).items():
        put(kk, vv)
    return buf, grid


def launch(fn, buf, grid, stream):
    size = C.c_size_t(C.sizeof(buf))
    cfg = (C.c_void_p * 5)(1, C.cast(buf, C.c_void_p), 2, C.cast(C.pointer(size), C.c_void_p), 3)
    chk(hip.hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, C.c_void_p(stream), None, cfg), "hipModuleLaunchKernel")


def kname(dt, D, causal, flav="", parity=False):
    return f"fa3_fwd_p4_{dt}_d{D}_{'causal' if causal else 'full'}{flav}_{'splitp_o32' if parity else 'o16'}"


def stamp_report(name, dbg, grid):
    d = (dbg.view(-1, 4, 16)[:grid].long() & 0xffffffff).double().cpu()
    acc = d[..., 1:]
    nfull, nitems = acc[..., 7].sum(), acc[..., 8].sum()
    tot = acc[..., 10]
    clk = (tot / (acc[..., 11] * 10e-9) / 1e9)
    wall = acc[..., 11] * 10 / 1e3
    print(f"  {name}: kernel cycles per wave mean {tot.mean():.0f} (min {tot.min():.0f} max {tot.max():.0f}); in-kernel clock {clk.mean():.3f} GHz; "
          f"wall per wave {wall.mean():.1f} us (max {wall.max():.1f}, max/mean {wall.max() / wall.mean():.3f})")
    print("     wall by workgroup % 8: " + " ".join(f"{wall[x::8].mean().item():.1f}" for x in range(8)))
    xcc = (d[:, 0, 14].long() & 15)                     # HW_REG_XCC_ID as read by wave 0 of every workgroup
    hw = d[:, 0, 15].long()
    tab = [[int(((xcc == x) & (torch.arange(grid) % 8 == m)).sum()) for x in range(8)] for m in range(8)]
    print("     workgroup % 8 -> XCC id histogram (rows: wg % 8, columns: XCC 0..7): " + " | ".join(" ".join(str(c) for c in row) for row in tab))
    cu = ((hw >> 8) & 15) + 16 * ((hw >> 12) & 1) + 32 * ((hw >> 13) & 7)      # CU_ID, SH_ID, SE_ID of HW_ID
    pairs = {}
    for w in range(grid):
        pairs.setdefault((int(xcc[w]), int(cu[w])), []).append(w)
    dup = {k: v for k, v in pairs.items() if len(v) > 1}
    print(f"     distinct (XCC, SE/SH/CU) places: {len(pairs)} for {grid} workgroups; places holding more than one workgroup: {len(dup)}"
          + (f" e.g. {list(dup.items())[:3]}" if dup else ""))
    by_xcc = [wall[(xcc == x).nonzero().flatten()].mean().item() if int((xcc == x).sum()) else float('nan') for x in range(8)]
    print("     wall by XCC id: " + " ".join(f"{v:.1f}" for v in by_xcc))
    if nfull > 0:
        print(f"     per FULL iteration: QK^T {acc[..., 0].sum() / nfull:.0f}  PV {acc[..., 1].sum() / nfull:.0f}  sync+bookkeeping {acc[..., 2].sum() / nfull:.0f}"
              f"  (vmcnt {acc[..., 9].sum() / nfull:.0f}, barrier {acc[..., 12].sum() / nfull:.0f});  per item: switch+prologue {acc[..., 3].sum() / nitems:.0f}"
              f"  epilogue {acc[..., 4].sum() / nitems:.0f}  LAST {acc[..., 5].sum() / nitems:.0f}  SKIP {acc[..., 6].sum() / nitems:.0f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="prod")
    ap.add_argument("--shapes", default="C3,C4")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--stamp", action="store_true")
    ap.add_argument("--parity", action="store_true", help="the fp32-store + split-P kernels")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--layout", default="bshd", help="memory layout of q / k / v / out: bshd (what a fused projection yields) or bhsd (head-major)")
    ap.add_argument("--spike", type=float, default=0.0, help="--check: scale some key rows by this factor (forces online-softmax rescales)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    torch.zeros(1, device=dev)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    variants = [Variant(n) for n in a.variants.split(",")]
    stream = torch.cuda.current_stream().cuda_stream
    bad = 0
    for sname in a.shapes.split(","):
        B, H, S, D, causal = SHAPES[sname]
        g = torch.Generator(device=dev).manual_seed(1234)
        if a.layout == "bhsd":
            q, k, v = (torch.randn(B, H, S, D, device=dev, dtype=torch.float32, generator=g).to(dt) for _ in range(3))
        else:
            q, k, v = (torch.randn(B, S, H, D, device=dev, dtype=torch.float32, generator=g).to(dt).permute(0, 2, 1, 3) for _ in range(3))
        if a.spike:
            kk = k.permute(0, 2, 1, 3)
            kk[:, S // 3::S // 7] *= a.spike               # a few keys far out: row maxima jump mid-item
        odt = torch.float32 if a.parity else dt
        kn = kname(a.dtype, D, causal, parity=a.parity)
        fl = 4.0 * B * H * S * S * D / (2 if causal else 1)
        outs = []
        print(f"== {sname}: B{B} H{H} S{S} D{D} {'causal' if causal else 'full'}  {kn}", flush=True)
        for vr in variants:
            out = torch.full((B, H, S, D), float("nan"), device=dev, dtype=odt) if a.layout == "bhsd" else \
                torch.full((B, S, H, D), float("nan"), device=dev, dtype=odt).permute(0, 2, 1, 3)
            lse = torch.full((B, H, S), float("nan"), device=dev, dtype=torch.float32)
            dbg = torch.zeros(n_cu * 4 * 16, dtype=torch.int32, device=dev) if a.stamp else None
            buf, grid = kernargs(q, k, v, out, lse, causal, n_cu, dbg=dbg)
            fn = vr.fn(kn)
            launch(fn, buf, grid, stream)
            torch.cuda.synchronize()
            outs.append((out, lse, buf, grid, fn, dbg))
        if a.check:
            from photonic_flash_attention_amd import ops
            ref, rlse = ops.fa3_forward(q, k, v, causal=causal, return_lse=True, out_dtype=odt, _variant=44)
            torch.cuda.synchronize()
            o0, l0 = outs[0][0], outs[0][1]
            for vr, (o, l, *_r) in zip(variants, outs):
                nan = int(torch.isnan(o.float()).sum()) + int(torch.isnan(l).sum())
                same = float((o == o0).float().mean())
                d0 = float((o.float() - o0.float()).abs().nan_to_num(1e9).max())
                dr = float((o.float() - ref.float()).abs().nan_to_num(1e9).max())
                dl = float((l - rlse).abs().nan_to_num(1e9).max())
                scale = max(1.0, float(ref.float().abs().max()) / 4)          # (spiked inputs: the outputs and their half ulps grow with them)
                ok = nan == 0 and dr <= (3e-5 if a.parity else 2e-2) * scale and dl <= 1e-4 * scale
                bad += 0 if ok else 1
                print(f"  {vr.name:14s} nan {nan}  bitwise-equal to {variants[0].name}: {same * 100:.4f} %  max|d| vs {variants[0].name} {d0:.3e}  "
                      f"vs 8-wave kernel {dr:.3e}  LSE {dl:.3e}  {'ok' if ok else 'MISMATCH'}", flush=True)
        if a.stamp:
            for vr, (o, l, buf, grid, fn, dbg) in zip(variants, outs):
                for _ in range(200):
                    launch(fn, buf, grid, stream)
                torch.cuda.synchronize()
                stamp_report(vr.name, dbg, grid)
            continue
        for vr, (o, l, buf, grid, fn, dbg) in zip(variants, outs):
            for _ in range(30):
                launch(fn, buf, grid, stream)
        torch.cuda.synchronize()
        times = {vr.name: [] for vr in variants}
        for r in range(a.rounds):
            for vr, (o, l, buf, grid, fn, dbg) in zip(variants, outs):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.steps):
                    launch(fn, buf, grid, stream)
                e1.record()
                torch.cuda.synchronize()
                times[vr.name].append(e0.elapsed_time(e1) / a.steps)
        base = statistics.median(times[variants[0].name])
        for vr in variants:
            med, mn = statistics.median(times[vr.name]), min(times[vr.name])
            print(f"  {vr.name:14s} median {med * 1e3:8.1f} us  {fl / med / 1e9:7.1f} TF   min {mn * 1e3:8.1f} us  {fl / mn / 1e9:7.1f} TF   "
                  f"vs {variants[0].name}: {(base / med - 1) * 100:+.2f} %", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
