#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json: attention TFLOP/s + ms/fwd at
B=4 S=4096 H=16 D=128, 1/2/4/8 GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE forward of the MI355X kernel (through the C ABI) over one batch of synthetic
Q/K/V already resident in HBM: BASELINE config C3 (B=4, S=4096, H=16, D=128, bf16, causal) per
GPU.  With N GPUs every rank owns its own B=4 shard of a global batch B=4N (batch x head sharding,
SURVEY.md section 8(e)); the path has no data-path collective, so scaling is "weak".  The north
star's single gather of the outputs runs ONCE after the timed region and is reported separately
(``gather``: RCCL's all_gather_into_tensor and the direct W-1-link exchange), together with a
secondary loop that overlaps one gather per forward on a side stream (``end_to_end``).

What is timed, so that a short driver run (``--steps 20 --warmup 5``) still reports steady state:
  1. ``prewarm``: a fixed, disclosed clock-settle phase (>= 300 untimed launches and >= 0.25 s; a freshly started process
     runs its first ~60 launches 10-25 % slow while the clocks ramp), then the ``--warmup`` launches;
  2. ``--reps`` (default 7) repetitions of EXACTLY ``--steps`` launches, each bracketed by barrier + synchronize on both
     sides, max over ranks per repetition; ``value`` is the MEDIAN repetition, all of them are listed (``reps_ms_per_step``);
  3. in the same run: the tolerance-meeting variant (fp32 store + split P, ``parity_variant``) and the other BASELINE
     shapes (``other_workloads``), each with its own kernel name.
``python bench.py --gpus N`` without a launcher starts its N ranks itself (fresh child processes, before the parent touches a GPU).

FLOP convention: 4*B*H*Sq*Sk*D, halved for causal.
Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

MFMA_PEAK_TFLOPS = 2500.0   # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md, chip-level table)

# HBM-side bytes per launch from the PMC passes of tools/profile.sh (separate rocprofv3 --pmc runs; FETCH_SIZE x 2
# per the gfx950 wide-load correction of MI355X_MICROARCH.md section HBM, + WRITE_SIZE).  Not measurable from inside
# this process, so the committed profile is quoted, keyed by workload; null for anything not profiled.
PMC_TRAFFIC_BYTES = {
    "C3": (369.5e6, "profiles/r02_C3_rocprof_summary.md: FETCH_SIZE 151999 KB x 2 + WRITE_SIZE 65536 KB (1.38 x the algorithmic 268.4 MB)"),
}

WORKLOADS = {
    # name: (B per GPU, H, S, D, causal, BASELINE.json config it stands for)
    "C3": (4, 16, 4096, 128, True, "configs[2], the headline shape"),
    "C4": (4, 16, 4096, 128, False, "configs[3], one GPU's shard of B=32 over 8 GPUs"),
    "C2": (4, 12, 1024, 64, False, "configs[1]"),
    "C5": (1, 32, 16384, 128, True, "configs[4]"),
}


def flops(B, H, S, D, causal):
    f = 4.0 * B * H * S * S * D
    return f / 2 if causal else f


def self_launch(args) -> int:
    """``python bench.py --gpus N`` outside a launcher: start the N ranks as fresh child processes (one per GPU, the parent has
    not touched a GPU), rendezvous on 127.0.0.1, pass rank 0's JSON line through."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PFA_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = max(rc, p.wait())
    for ln in out.splitlines():          # rank 0's JSON line only (the gloo rehearsal backend chats on stdout)
        if ln.startswith("{"):
            print(ln, flush=True)
    return rc


def cpu_baseline(B, H, S, D, causal):
    """The oracle (our CPU restatement of the reference's fp32 path, kind "port") timed on this
    host's cores on a bounded sample of the same workload: one batch element (all H heads)."""
    import torch
    from oracle import fa3_oracle as orc
    from photonic_flash_attention_amd import synth

    # a 1-GPU job owns a 16-core share of the box; the affinity mask shows every host core, and 256 oversubscribed threads made
    # this leg 50x slower: use the share (or the mask, if smaller) and say so
    share = int(os.environ.get("PFA_CPU_THREADS", "16"))
    cores = min(len(os.sched_getaffinity(0)), share)
    torch.set_num_threads(cores)
    q, k, v = synth.qkv(1, H, S, S, D, 1234, "bf16")
    ts = []
    for i in range(3):
        t0 = time.perf_counter()
        orc.attention_bshd(q, k, v, causal=causal)
        ts.append(time.perf_counter() - t0)
        if sum(ts) > 45:
            break
    t = statistics.median(ts[1:] or ts)
    return {
        "value": round(flops(1, H, S, D, causal) / t / 1e12, 5),
        "unit": "TFLOP/s",
        "cores": cores,
        "kind": "port",
        "sample": f"1 of the {B} batch elements (B=1,H={H},S={S},D={D},causal={causal}), fp32 on bf16-rounded "
                  f"inputs, torch {torch.get_num_threads()} threads = the {share}-core share of a 1-GPU job "
                  f"({len(os.sched_getaffinity(0))} in the affinity mask), median of {max(1, len(ts) - 1)} after 1 warm-up, "
                  f"{t * 1e3:.0f} ms; the reference computes masked causal tiles in full "
                  f"({flops(1, H, S, D, False) / t / 1e12:.4f} TFLOP/s against the un-halved count)",
    }


def parity_check(q, k, v, causal, heads, variant):
    """Benched kernel (bf16 store) and parity variant (fp32 store, split P) vs the oracle on a few heads."""
    import torch
    from oracle import fa3_oracle as orc
    from photonic_flash_attention_amd import ops
    o16, _ = ops.fa3_forward_bshd(q, k, v, causal=causal, _variant=variant)
    o32, _ = ops.fa3_forward_bshd(q, k, v, causal=causal, out_dtype=torch.float32)
    torch.cuda.synchronize()
    e16 = e32 = 0.0
    for (b, h) in heads:
        sl = (slice(b, b + 1), slice(None), slice(h, h + 1))
        ref = orc.attention_bshd(q[sl].cpu(), k[sl].cpu(), v[sl].cpu(), causal=causal)
        e16 = max(e16, float((o16[sl].float().cpu() - ref).abs().max()))
        e32 = max(e32, float((o32[sl].cpu() - ref).abs().max()))
    return e16, e32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--reps", type=int, default=7, help="repetitions of exactly --steps launches; the median is reported")
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--variant", type=int, default=0, help="A/B only: kernel selector 43 / 44 / 45 of pfa_capi.hip (0 = production choice)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip parity_variant / other_workloads (profiling runs)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    import torch
    import torch.distributed as dist
    backend = os.environ.get("PFA_DIST_BACKEND", "nccl")   # "gloo" only to rehearse the N>1 control flow on one GPU / on CPU boxes
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (there is no CPU path to time)"
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from photonic_flash_attention_amd import _capi, ops
    from photonic_flash_attention_amd.parallel import sharded

    _capi.load()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def make(B, H, S, D, seed, out_dtype=torch.bfloat16):
        gen = torch.Generator(device=dev).manual_seed(seed)
        q, k, v = (torch.randn(B, S, H, D, device=dev, dtype=torch.float32, generator=gen).to(torch.bfloat16) for _ in range(3))
        out = torch.empty(B, S, H, D, device=dev, dtype=out_dtype)
        return q, k, v, out

    def timed(fn, steps, reps):
        """reps x (barrier, exactly `steps` launches, barrier): per-repetition wall ms/step (max over ranks) and HIP-event ms/step"""
        walls, kerns = [], []
        for _ in range(reps):
            barrier()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            ev0.record()
            for _ in range(steps):
                fn()
            ev1.record()
            barrier()
            walls.append((time.perf_counter() - t0) / steps * 1e3)
            kerns.append(ev0.elapsed_time(ev1) / steps)
        w = torch.tensor(walls, device=red_dev, dtype=torch.float64)
        kk = torch.tensor(kerns, device=red_dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(w, op=dist.ReduceOp.MAX)
            dist.all_reduce(kk, op=dist.ReduceOp.MAX)
        return w.tolist(), kk.tolist()

    B, H, S, D, causal, what = WORKLOADS[args.workload]
    q, k, v, out = make(B, H, S, D, 1234 + rank)
    outv = out.permute(0, 2, 1, 3)
    qv, kv, vv = (t.permute(0, 2, 1, 3) for t in (q, k, v))

    def step():
        ops.fa3_forward(qv, kv, vv, causal=causal, out=outv, _variant=args.variant)

    # ---- 1. disclosed clock-settle phase, then the caller's warm-up ------------------------------------------------------------
    t0 = time.perf_counter()
    n_pre = 0
    while n_pre < 300 or time.perf_counter() - t0 < 0.25:
        for _ in range(50):
            step()
        torch.cuda.synchronize()
        n_pre += 50
    prewarm = {"launches": n_pre, "ms": round((time.perf_counter() - t0) * 1e3, 1),
               "why": "clock-settle phase of a fresh process, untimed, fixed rule: >= 300 launches and >= 0.25 s"}
    for _ in range(args.warmup):
        step()
    # ---- 2. timed region: --reps x exactly --steps launches ------------------------------------------------------------------------
    walls, kerns = timed(step, args.steps, args.reps)
    ms_per_step = statistics.median(walls)
    kern_ms = statistics.median(kerns)
    f_rank = flops(B, H, S, D, causal)
    value = f_rank * world / (ms_per_step * 1e-3) / 1e12

    # ---- the single gather at the end (outside the timed steps) + overlapped end-to-end loop --------------
    gather = end_to_end = None
    if world > 1:
        try:   # secondary measurements must never cost the headline line
            src = out if backend == "nccl" else out.cpu()
            gather = {"bytes_per_rank": out.numel() * out.element_size(), "backend": backend}
            for algo in sharded.GATHER_ALGOS:
                gathered, g_ms = sharded.gather_outputs(src, timed=True, algo=algo)       # first call: connection set-up included
                gathered, g_ms = sharded.gather_outputs(src, timed=True, algo=algo)
                assert gathered.shape[0] == B * world
                t = torch.tensor([g_ms], device=red_dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                gather[algo + "_ms"] = round(float(t), 4)
            best = min(sharded.GATHER_ALGOS, key=lambda a_: gather[a_ + "_ms"])
            e2e_step = step if backend == "nccl" else (lambda: (step(), torch.cuda.synchronize()))
            e2e_ms = sharded.overlapped_forward_gather(e2e_step, src, min(args.steps, 50), algo=best)
            t = torch.tensor([e2e_ms], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            end_to_end = {"ms_per_step": round(float(t), 4),
                          "value": round(f_rank * world / (float(t) * 1e-3) / 1e12, 2), "unit": "TFLOP/s", "gather": best,
                          "note": "one gather per forward on a side stream, overlapped with the next forward"
                                  if backend == "nccl" else "gloo rehearsal: host copies, forward and gather alternate"}
        except Exception as exc:   # noqa: BLE001
            gather = {"error": f"{type(exc).__name__}: {exc}"[:300]}

    # ---- 3. the tolerance-meeting variant and the other BASELINE shapes, same process --------------------------------------------
    parity_variant = others = None
    if world == 1 and not args.no_others:
        out32 = torch.empty(B, S, H, D, device=dev, dtype=torch.float32).permute(0, 2, 1, 3)

        def step32():
            ops.fa3_forward(qv, kv, vv, causal=causal, out=out32, out_dtype=torch.float32)
        for _ in range(30):
            step32()
        w32, _k = timed(step32, args.steps, 3)
        a32 = ops.build_args(qv, kv, vv, out32, causal=causal, split_p=True)[0]
        parity_variant = {"ms_per_step": round(statistics.median(w32), 4),
                          "tflops": round(f_rank / (statistics.median(w32) * 1e-3) / 1e12, 2),
                          "kernel": _capi.describe(a32)[0], "what": "fp32 store + P as bf16 hi+lo (PFA_FLAG_SPLIT_P): <= 1e-3 of the reference"}
        del out32
        others = {}
        for name in ("C2", "C4", "C5"):
            if name == args.workload:
                continue
            b2, h2, s2, d2, c2, _w = WORKLOADS[name]
            q2, k2, v2, o2 = make(b2, h2, s2, d2, 77)
            q2v, k2v, v2v, o2v = (t.permute(0, 2, 1, 3) for t in (q2, k2, v2, o2))

            def step2():
                ops.fa3_forward(q2v, k2v, v2v, causal=c2, out=o2v)
            for _ in range(100):
                step2()
            w2, kk2 = timed(step2, args.steps, 3)
            m2 = statistics.median(kk2)
            others[name] = {"ms": round(m2, 4), "tflops": round(flops(b2, h2, s2, d2, c2) / (m2 * 1e-3) / 1e12, 2),
                            "frac": round(flops(b2, h2, s2, d2, c2) / (m2 * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                            "kernel": _capi.describe(ops.build_args(q2v, k2v, v2v, o2v, causal=c2)[0])[0],
                            "shape": f"B={b2} S={s2} H={h2} D={d2} {'causal' if c2 else 'non-causal'}"}
            del q2, k2, v2, o2
        # a [B, Sk] key-padding mask on the same schedule (the *_km_* kernels read the mask bytes themselves; ops hands them the last
        # visible key of every row, so the padded tail is skipped): S = 2048, D = 128, lengths uniform in [S / 2, S]; dense-equivalent
        # flops, next to the unmasked launch of the same shape
        b2, h2, s2, d2 = 16, 16, 2048, 128
        q2, k2, v2, o2 = make(b2, h2, s2, d2, 78)
        q2v, k2v, v2v, o2v = (t.permute(0, 2, 1, 3) for t in (q2, k2, v2, o2))
        lens = torch.randint(s2 // 2, s2 + 1, (b2,), generator=torch.Generator().manual_seed(79)).to(dev)
        kmask = (torch.arange(s2, device=dev)[None, :] < lens[:, None]).to(torch.uint8)
        for tag, kw in (("S2048", {}), ("S2048_key_mask", {"key_mask": kmask})):
            def step3():
                ops.fa3_forward(q2v, k2v, v2v, causal=False, out=o2v, **kw)
            for _ in range(100):
                step3()
            w3, kk3 = timed(step3, args.steps, 3)
            m3 = statistics.median(kk3)
            others[tag] = {"ms": round(m3, 4), "tflops": round(flops(b2, h2, s2, d2, False) / (m3 * 1e-3) / 1e12, 2),
                           # (no fraction of peak for the masked launch: its padded tail is skipped, the flops are dense-equivalent)
                           "frac": None if kw else round(flops(b2, h2, s2, d2, False) / (m3 * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                           "kernel": _capi.describe(ops.build_args(q2v, k2v, v2v, o2v, causal=False, **kw)[0])[0],
                           "shape": f"B={b2} S={s2} H={h2} D={d2} non-causal" + (", key-padding mask (lengths in [S/2, S], dense-equivalent flops)" if kw else "")}
        del q2, k2, v2, o2

    if rank == 0:
        name, nwg = _capi.describe(ops.build_args(qv, kv, vv, outv, causal=causal, variant=args.variant)[0])
        achieved = f_rank / (kern_ms * 1e-3) / 1e12
        line = {
            "metric": "attention_fwd_tflops", "value": round(value, 2), "unit": "TFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.workload}: B={B}/GPU S={S} H={H} D={D} bf16 "
                                   f"{'causal' if causal else 'non-causal'} (BASELINE.json {what})",
                       "global_batch": B * world, "seq_len": S, "heads": H, "head_dim": D, "causal": causal,
                       "sharding": f"batch x head over {world} rank(s), no data-path collective",
                       "flop_convention": "4*B*H*S*S*D, halved for causal", "kernel": name, "workgroups": nwg},
            "timing": {"reps": args.reps, "reps_ms_per_step": [round(x, 4) for x in walls],
                       "reps_kernel_ms": [round(x, 4) for x in kerns], "reported": "median repetition", "prewarm": prewarm},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                         "traffic": PMC_TRAFFIC_BYTES.get(args.workload, (None, None))[0],
                         "traffic_source": PMC_TRAFFIC_BYTES.get(args.workload, (None, None))[1],
                         "algorithmic_bytes": 2 * 2 * 2 * B * H * S * D,
                         "kernel_ms": round(kern_ms, 4),
                         "hbm_algorithmic_GBps": round(2 * 2 * 2 * B * H * S * D / (kern_ms * 1e-3) / 1e9, 1)},
        }
        if gather:
            line["gather"] = gather
            line["end_to_end"] = end_to_end
        if parity_variant:
            line["parity_variant"] = parity_variant
        if others:
            line["other_workloads"] = others
        if not args.no_parity:
            e16, e32 = parity_check(q, k, v, causal, [(0, 0), (B - 1, H - 1)], args.variant)
            line["parity"] = {"benched_kernel_bf16_out_max_abs": round(e16, 6),
                              "parity_variant_fp32_out_max_abs": round(e32, 8), "heads_checked": 2,
                              "oracle": "oracle/fa3_oracle.py (pinned by tests/golden)"}
            if parity_variant:
                parity_variant["max_abs"] = round(e32, 8)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(B, H, S, D, causal)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
