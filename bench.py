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



def pmc_traffic(workload):
    """HBM-side bytes per launch from the PMC passes of tools/profile.sh (separate rocprofv3 --pmc runs; FETCH_SIZE x 2 per the gfx950
    wide-load correction of MI355X_MICROARCH.md section HBM, + WRITE_SIZE, both in KiB).  Not measurable from inside this process, so
    the NEWEST committed profiles/r*_<workload>_rocprof_summary.md is parsed (nothing is pasted here that could go stale); (None, why)
    when there is none."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(REPO, "profiles", f"r*_{workload}_rocprof_summary*.md")):
        m = re.match(r"r(\d+)_", os.path.basename(f))
        if m and (best is None or (int(m.group(1)), os.path.getmtime(f)) > best[0]):
            best = ((int(m.group(1)), os.path.getmtime(f)), f)
    if best is None:
        return None, f"no profiles/r*_{workload}_rocprof_summary.md"
    txt = open(best[1]).read()
    fe, wr = re.search(r"FETCH_SIZE: ([0-9.e+]+)", txt), re.search(r"WRITE_SIZE: ([0-9.e+]+)", txt)
    if not (fe and wr):
        return None, f"{os.path.relpath(best[1], REPO)} holds no FETCH_SIZE / WRITE_SIZE"
    b = (float(fe.group(1)) * 2 + float(wr.group(1))) * 1024
    return b, f"{os.path.relpath(best[1], REPO)}: FETCH_SIZE {float(fe.group(1)):.0f} KiB x 2 + WRITE_SIZE {float(wr.group(1)):.0f} KiB"


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
    ap.add_argument("--no-others", action="store_true", help="skip parity_variant / other_workloads / backward / module (profiling runs)")
    ap.add_argument("--no-probe", action="store_true", help="skip the MFMA ceiling probe")
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
    watchdog = None
    if world > 1:
        # Secondary measurements must never cost the headline line -- not even by HANGING (a collective or an IPC mapping that never
        # returns on hardware this code has only met in a CPU rehearsal): if the gather legs are not through in 180 s, rank 0 prints the
        # line without them and every rank leaves.
        import threading

        def _give_up():
            if rank == 0:
                ach = f_rank / (kern_ms * 1e-3) / 1e12
                print(json.dumps({
                    "metric": "attention_fwd_tflops", "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps,
                    "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
                    "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                    "config": {"workload": f"{args.workload}: B={B}/GPU S={S} H={H} D={D} bf16 {'causal' if causal else 'non-causal'} "
                                           f"(BASELINE.json {what})", "global_batch": B * world, "seq_len": S, "heads": H, "head_dim": D,
                               "causal": causal, "sharding": f"batch x head over {world} rank(s), no data-path collective"},
                    "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": pmc_traffic(args.workload)[0],
                                 "kernel_ms": round(kern_ms, 4)},
                    "gather": {"error": "watchdog: the gather legs did not finish within 180 s; the timed steps above were complete"}}),
                    flush=True)
            os._exit(0)
        watchdog = threading.Timer(180.0, _give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            src = out if backend == "nccl" else out.cpu()
            gather = {"bytes_per_rank": out.numel() * out.element_size(), "backend": backend}
            algos = [a_ for a_ in sharded.GATHER_ALGOS if a_ != "sdma" or backend == "nccl"]      # (copy engines: device tensors only)
            for algo in algos:
                try:
                    gathered, g_ms = sharded.gather_outputs(src, timed=True, algo=algo)   # first call: connection / handle set-up included
                    gathered, g_ms = sharded.gather_outputs(src, timed=True, algo=algo)
                    assert gathered.shape[0] == B * world
                except Exception as exc:   # noqa: BLE001  (every rank fails together: PeerGather agrees on that before it raises)
                    gather[algo + "_error"] = f"{type(exc).__name__}: {exc}"[:200]
                    g_ms = float("inf")
                t = torch.tensor([g_ms], device=red_dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                gather[algo + "_ms"] = round(float(t), 4) if float(t) != float("inf") else None
            algos = [a_ for a_ in algos if gather.get(a_ + "_ms") is not None]
            best = min(algos, key=lambda a_: gather[a_ + "_ms"])
            e2e_step = step if backend == "nccl" else (lambda: (step(), torch.cuda.synchronize()))
            e2e_ms = sharded.overlapped_forward_gather(e2e_step, src, min(args.steps, 50), algo=best)
            t = torch.tensor([e2e_ms], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            end_to_end = {"ms_per_step": round(float(t), 4),
                          "value": round(f_rank * world / (float(t) * 1e-3) / 1e12, 2), "unit": "TFLOP/s", "gather": best,
                          "note": "one gather per forward on a side stream, overlapped with the next forward"
                                  if backend == "nccl" else "gloo rehearsal: host copies, forward and gather alternate"}
            if "sdma" in algos and best != "sdma":      # the copy-engine gather beside the next forward (no CU taken from the persistent kernel)
                e2 = sharded.overlapped_forward_gather(e2e_step, src, min(args.steps, 50), algo="sdma")
                t = torch.tensor([e2], device=red_dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                end_to_end["sdma_ms_per_step"] = round(float(t), 4)
                end_to_end["sdma_value"] = round(f_rank * world / (float(t) * 1e-3) / 1e12, 2)
        except Exception as exc:   # noqa: BLE001
            gather = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        watchdog.cancel()

    # ---- 2b. ceiling probe, same process, clocks settled: what the matrix pipes of THIS device deliver under the tile loop's MFMA + LDS
    # load on random operands without its softmax (csrc/pfa_probe.hip).  The chip lowers its clock under MFMA load, so the nominal
    # 2.5 PFLOP/s is not reachable on random data by any kernel; `frac` stays against the nominal peak, `frac_of_probe` says how much
    # of the achievable rate the kernel holds.
    probe = None
    if not args.no_probe:
        try:
            import ctypes as C
            lib = _capi.load()
            n_cu = torch.cuda.get_device_properties(local_rank).multi_processor_count
            rnd = torch.randn(32768, device=dev).to(torch.bfloat16)
            sink = torch.empty(n_cu * 256, device=dev, dtype=torch.float32)
            fl = C.c_double()
            cs = torch.cuda.current_stream().cuda_stream
            iters = 2500                                   # ~3.5 ms a launch
            for _ in range(10):
                lib.pfa_probe_mfma(rnd.data_ptr(), sink.data_ptr(), iters, local_rank, cs, C.byref(fl))
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    nwg = lib.pfa_probe_mfma(rnd.data_ptr(), sink.data_ptr(), iters, local_rank, cs, C.byref(fl))
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10)
            pm = statistics.median(ts)
            probe = {"tflops": round(fl.value / (pm * 1e-3) / 1e12, 1), "ms": round(pm, 3), "workgroups": int(nwg),
                     "what": "bare v_mfma_f32_32x32x16_bf16 stream, one wave per SIMD on every CU, random operands, A operands re-read "
                             "from LDS as the tile loop reads K / V^T fragments (48 LDS reads per 64 MFMAs); no softmax, no HBM traffic"}
            for _ in range(50):                            # back to the forward's steady state for what follows
                step()
            torch.cuda.synchronize()
        except Exception as exc:   # noqa: BLE001
            probe = {"error": f"{type(exc).__name__}: {exc}"[:200]}

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
        # (+ the same B x H x S at head dim 64, the head dim of every shape the reference's own harness runs: many units per CU)
        for tag, kw in (("S2048", {}), ("S2048_key_mask", {"key_mask": kmask}), ("S2048_D64", {})):
            if tag == "S2048_D64":
                del q2, k2, v2, o2
                d2 = 64
                q2, k2, v2, o2 = make(b2, h2, s2, d2, 81)
                q2v, k2v, v2v, o2v = (t.permute(0, 2, 1, 3) for t in (q2, k2, v2, o2))

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
        # the padded decoder batch: C3's shape under the causal mask with per-batch key counts (seqlens_k, lengths in [S/2, S]) -- round 3:
        # on the persistent ragged kernels, a block behind its batch's cut runs only the key groups that hold visible keys
        b2, h2, s2, d2, c2, _w = WORKLOADS["C3"]
        q2, k2, v2, o2 = make(b2, h2, s2, d2, 80)
        q2v, k2v, v2v, o2v = (t.permute(0, 2, 1, 3) for t in (q2, k2, v2, o2))
        lens3 = [s2, s2 * 9 // 16, s2 * 13 // 16, s2 * 11 // 16][:b2] + [s2 * 3 // 4] * max(0, b2 - 4)
        sl3 = torch.tensor(lens3, dtype=torch.int32, device=dev)

        def step5():
            ops.fa3_forward(q2v, k2v, v2v, causal=True, seqlens_k=sl3, out=o2v)
        for _ in range(100):
            step5()
        w5, kk5 = timed(step5, args.steps, 3)
        m5 = statistics.median(kk5)
        km3 = (torch.arange(s2, device=dev)[None, :] < sl3[:, None])

        def step6():                            # the same padding as a [B, Sk] key mask (what a module's 2-D attention_mask becomes): ops derives the cut
            ops.fa3_forward(q2v, k2v, v2v, causal=True, key_mask=km3, out=o2v)
        for _ in range(100):
            step6()
        w6, kk6 = timed(step6, args.steps, 3)
        m6 = statistics.median(kk6)
        others["C3_key_mask"] = {"ms": round(m6, 4), "tflops": round(flops(b2, h2, s2, d2, True) / (m6 * 1e-3) / 1e12, 2), "frac": None,
                                 "kernel": _capi.describe(ops.build_args(q2v, k2v, v2v, o2v, causal=True, key_mask=km3)[0])[0],
                                 "shape": "C3 under the causal mask with a [B, Sk] key-padding mask of the same lengths (dense-equivalent flops; "
                                          "includes the two small device ops that derive the cut from the mask)"}
        # an ELEMENT mask (flash_attention_3.py:234-236 takes any 4-D mask): a 1024-key sliding window as a [1,1,S,S] mask on the HIP kernels --
        # condensed to mask words + the tile range each Q block can see, so the launch runs what the band leaves visible
        try:
            ii = torch.arange(s2, device=dev)
            band = ((ii[:, None] - ii[None, :] >= 0) & (ii[:, None] - ii[None, :] < 1024))[None, None]

            def step8():
                ops.fa3_forward(q2v, k2v, v2v, mask=band, out=o2v)
            for _ in range(20):
                step8()
            w8, kk8 = timed(step8, args.steps, 3)
            m8 = statistics.median(kk8)
            o8, lse8 = ops.fa3_forward(q2v, k2v, v2v, mask=band, return_lse=True)
            g8 = torch.randn_like(o8)

            def step9():
                ops.fa3_backward(q2v, k2v, v2v, o8, g8, lse8, mask=band)
            for _ in range(3):
                step9()
            w9, kk9 = timed(step9, max(2, args.steps // 4), 3)
            others["C3_window1024_element_mask"] = {
                "ms": round(m8, 4), "backward_ms": round(statistics.median(kk9), 4), "frac": None,
                "tflops": round(flops(b2, h2, s2, d2, True) / (m8 * 1e-3) / 1e12, 2),
                "kernel": _capi.describe(ops.build_args(q2v, k2v, v2v, o2v, mask=band)[0])[0],
                "shape": "C3's shape, causal sliding window of 1024 keys given as a 4-D element mask (tflops: causal-equivalent flops, for scale; "
                         "ms includes the two condensing kernels; backward_ms: pfa_fa3_bwd under the same mask)"}
            del o8, lse8, g8, band
        except Exception as exc:   # noqa: BLE001
            others["C3_window1024_element_mask"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
        others["C3_seqlens"] = {"ms": round(m5, 4), "tflops": round(flops(b2, h2, s2, d2, True) / (m5 * 1e-3) / 1e12, 2), "frac": None,
                                "kernel": _capi.describe(ops.build_args(q2v, k2v, v2v, o2v, causal=True, seqlens_k=sl3)[0])[0],
                                "shape": f"C3 with seqlens_k = {lens3} under the causal mask (dense-equivalent flops: the padding is not computed)"}
        del q2, k2, v2, o2

    # ---- 4. backward (pfa_fa3_bwd: the reference trains through autograd over its eager core, tests/unit/test_flash_attention_3.py:137-160)
    # and the module level (projections + core, core/flash_attention_3.py:49-118), same workload, same process
    backward = module = fp32 = None
    if world == 1 and not args.no_others:
        try:
            lse = torch.empty(B, H, S, device=dev, dtype=torch.float32)
            o_f, lse = ops.fa3_forward(qv, kv, vv, causal=causal, out=outv, return_lse=True)
            dout = torch.randn(B, S, H, D, device=dev, dtype=torch.float32).to(torch.bfloat16).permute(0, 2, 1, 3)

            def step_b():
                ops.fa3_backward(qv, kv, vv, o_f, dout, lse, causal=causal)
            for _ in range(10):
                step_b()
            wb, kb = timed(step_b, max(5, args.steps // 2), 3)
            mb = statistics.median(kb)
            backward = {"ms": round(mb, 4), "tflops": round(2.5 * f_rank / (mb * 1e-3) / 1e12, 2),
                        "frac": round(2.5 * f_rank / (mb * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                        "flop_convention": "2.5 x the forward's algorithmic flops (5 of the 7 executed products)",
                        "kernels": ["fa3_bwd_dq_kernel (+ delta)", "fa3_bwd_dkdv_kernel"], "workload": args.workload}
            del dout
        except Exception as exc:   # noqa: BLE001
            backward = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        try:
            from photonic_flash_attention_amd import FlashAttention3
            module = {}
            for name in dict.fromkeys((args.workload, "C2")):
                b2, h2, s2, d2, c2, _w = WORKLOADS[name]
                e2 = h2 * d2
                m2 = FlashAttention3(e2, h2, dtype=torch.bfloat16, device=dev).eval()
                x2 = torch.randn(b2, s2, e2, device=dev, dtype=torch.float32).to(torch.bfloat16)
                with torch.no_grad():
                    qkv = m2.qkv_proj(x2)
                    q2v, k2v, v2v = (t.view(b2, s2, h2, d2).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
                    att = torch.empty(b2, s2, h2, d2, device=dev, dtype=torch.bfloat16)

                    def f_mod():
                        m2(x2, is_causal=c2)

                    def f_core():
                        ops.fa3_forward(q2v, k2v, v2v, causal=c2, out=att.permute(0, 2, 1, 3))

                    def f_gemm():
                        m2.qkv_proj(x2)
                        m2.out_proj(att.view(b2, s2, e2))
                    res = {}
                    for tag, fn in (("ms", f_mod), ("core_ms", f_core), ("gemm_ms", f_gemm)):
                        for _ in range(20):
                            fn()
                        _w2, k2_ = timed(fn, args.steps, 3)
                        res[tag] = round(statistics.median(k2_), 4)
                res["overhead_frac"] = round(res["ms"] / (res["core_ms"] + res["gemm_ms"]) - 1, 4)
                res["what"] = (f"FlashAttention3({e2}, {h2}) bf16 eval forward at B={b2} S={s2}: ms = whole module; core_ms = the attention "
                               "kernel on the strided QKV views; gemm_ms = qkv_proj + out_proj (hipBLASLt through nn.Linear, bias in the epilogue)")
                module[name] = res
                del m2, x2, qkv, att
        except Exception as exc:   # noqa: BLE001
            module = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        try:   # fp32 operands: the reference's default dtype (flash_attention_3.py:19-27; BASELINE configs[0]) on the exact-fp32 kernels
            fp32 = {}
            for tag, (b2, h2, s2, d2, c2), n_it in (("C1_fp32", (2, 4, 128, 64, False), 50), ("S4096_D128_fp32", (1, 16, 4096, 128, False), 3)):
                q2, k2, v2 = (torch.randn(b2, s2, h2, d2, device=dev, dtype=torch.float32).permute(0, 2, 1, 3) for _ in range(3))
                o2 = torch.empty(b2, s2, h2, d2, device=dev, dtype=torch.float32).permute(0, 2, 1, 3)

                def step4():
                    ops.fa3_forward(q2, k2, v2, causal=c2, out=o2)
                for _ in range(2):
                    step4()
                _w4, k4 = timed(step4, n_it, 3)
                m4 = statistics.median(k4)
                a4 = ops.build_args(q2, k2, v2, o2, causal=c2)[0]
                fp32[tag] = {"ms": round(m4, 4), "tflops": round(flops(b2, h2, s2, d2, c2) / (m4 * 1e-3) / 1e12, 3),
                             "frac_of_fp32_matrix_peak": round(flops(b2, h2, s2, d2, c2) / (m4 * 1e-3) / 1e12 / 157.3, 4),
                             "kernel": _capi.describe(a4)[0], "shape": f"B={b2} S={s2} H={h2} D={d2} fp32 non-causal"}
                # its backward (fa3_bwd_f32_kernel: the same fp32 MFMA, dQ pass + dK/dV pass), 2.5 x forward convention
                o4, lse4 = ops.fa3_forward(q2, k2, v2, causal=c2, return_lse=True)
                g4 = torch.randn(b2, s2, h2, d2, device=dev, dtype=torch.float32).permute(0, 2, 1, 3)

                def step4b():
                    ops.fa3_backward(q2, k2, v2, o4, g4, lse4, causal=c2)
                for _ in range(2):
                    step4b()
                _w4b, k4b = timed(step4b, max(2, n_it // 2), 3)
                m4b = statistics.median(k4b)
                fp32[tag]["backward_ms"] = round(m4b, 4)
                fp32[tag]["backward_tflops"] = round(2.5 * flops(b2, h2, s2, d2, c2) / (m4b * 1e-3) / 1e12, 3)
                del o4, lse4, g4
                del q2, k2, v2, o2
        except Exception as exc:   # noqa: BLE001
            fp32 = {"error": f"{type(exc).__name__}: {exc}"[:300]}

    weights_leg = None
    if world == 1 and not args.no_others:
        try:   # need_weights (flash_attention_3.py:171,180,257-258): the softmax matrix [B,H,Sq,Sk] written by a second pass -- HBM-write bound
            import ctypes as C
            weights_leg = {}
            for tag, (b2, h2, s2, d2, c2) in (("B2_H16_S2048_D128", (2, 16, 2048, 128, False)), ("B4_H16_S4096_D128", (4, 16, 4096, 128, False)),
                                              ("B4_H16_S4096_D128_causal", (4, 16, 4096, 128, True)), ("B16_H12_S2048_D64", (16, 12, 2048, 64, False))):
                q2, k2, v2, o2 = make(b2, h2, s2, d2, 99)
                lse2 = torch.empty((b2, h2, s2), dtype=torch.float32, device=dev)
                a2, keep2 = ops.build_args(*(t.permute(0, 2, 1, 3) for t in (q2, k2, v2, o2)), causal=c2, lse=lse2)
                st2 = torch.cuda.current_stream(dev).cuda_stream
                _capi.check_status(_capi.load().pfa_fa3_fwd(C.byref(a2), C.c_void_p(st2)))
                w2 = torch.empty((b2, h2, s2, s2), dtype=torch.bfloat16, device=dev)

                def step7():
                    _capi.check_status(_capi.load().pfa_fa3_weights(C.byref(a2), C.c_void_p(w2.data_ptr()), _capi.PFA_DTYPE_BF16,
                                                                    w2.stride(0), w2.stride(1), w2.stride(2), C.c_void_p(st2)))
                for _ in range(3):
                    step7()
                _w7, k7 = timed(step7, 10, 3)
                m7 = statistics.median(k7)
                weights_leg[tag] = {"ms": round(m7, 4), "output_MB": round(w2.numel() * 2 / 1e6, 1),
                                    "write_TBps": round(w2.numel() * 2 / (m7 * 1e-3) / 1e12, 3)}
                del q2, k2, v2, o2, w2, lse2, keep2
            weights_leg["what"] = ("pfa_fa3_weights alone (the forward's LSE given): bf16 weights, every element written once; the box's plain fill "
                                   "rate is ~6.8 TB/s (tools/write_rate.py)")
        except Exception as exc:   # noqa: BLE001
            weights_leg = {"error": f"{type(exc).__name__}: {exc}"[:300]}

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
                         "traffic": pmc_traffic(args.workload)[0],
                         "traffic_source": pmc_traffic(args.workload)[1],
                         "algorithmic_bytes": 2 * 2 * 2 * B * H * S * D,
                         "kernel_ms": round(kern_ms, 4),
                         "hbm_algorithmic_GBps": round(2 * 2 * 2 * B * H * S * D / (kern_ms * 1e-3) / 1e9, 1)},
        }
        if gather:
            line["gather"] = gather
            line["end_to_end"] = end_to_end
        if probe:
            line["roofline"]["probe_tflops"] = probe.get("tflops")
            line["roofline"]["frac_of_probe"] = round(achieved / probe["tflops"], 4) if probe.get("tflops") else None
            line["roofline"]["probe"] = probe
        if parity_variant:
            line["parity_variant"] = parity_variant
        if others:
            line["other_workloads"] = others
        if backward:
            line["backward"] = backward
        if module:
            line["module"] = module
        if fp32:
            line["fp32_operands"] = fp32
        if weights_leg:
            line["need_weights"] = weights_leg
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
