"""``pfa-benchmark``: counterpart of the reference's ``photonic-benchmark`` (cli.py:20-145).

Same flags (``--seq-lengths --batch-sizes --embed-dim --num-heads --num-iterations --output -v``) and
the same per-result keys and ``benchmark_info`` envelope, so downstream scripts that read the
reference's JSON keep working.  Differences, all on purpose:

* the module and its input live on the GPU (the reference builds CPU tensors, cli.py:64) and every timed
  call is bracketed by ``torch.cuda.synchronize()`` (the reference times an un-synchronised call);
* adds attention-core FLOP accounting: ``attn_tflops`` (4*B*S*S*E per forward, halved with ``--causal``) and
  ``frac_mfma_peak`` against the 2.5 PFLOP/s dense bf16 peak, plus ``core_ms`` = kernel-only time;
* ``--dtype`` (bf16 default), ``--causal``.

Run: ``python -m photonic_flash_attention_amd.cli benchmark ...`` (or ``device-info``).
"""

from __future__ import annotations

import argparse
import json
import sys
import time

import numpy as np
import torch

from . import HybridFlashAttention, get_device_info, get_version, ops
from .config import get_config
from .utils.logging import get_logger

MFMA_PEAK_TFLOPS = 2500.0


def _parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Benchmark the MI355X attention path")
    p.add_argument("--seq-lengths", nargs="+", type=int, default=[128, 256, 512, 1024, 2048, 4096])
    p.add_argument("--batch-sizes", nargs="+", type=int, default=[1, 2, 4, 8])
    p.add_argument("--embed-dim", type=int, default=768)
    p.add_argument("--num-heads", type=int, default=12)
    p.add_argument("--num-iterations", type=int, default=10)
    p.add_argument("--output", type=str)
    p.add_argument("--verbose", "-v", action="store_true")
    p.add_argument("--dtype", choices=["bf16", "fp16", "fp32"], default="bf16")
    p.add_argument("--causal", action="store_true")
    return p


def benchmark(args=None):
    if args is None:
        args = _parser().parse_args(sys.argv[2:] if len(sys.argv) > 1 and sys.argv[1] == "benchmark" else None)
    logger = get_logger("benchmark")
    if not torch.cuda.is_available():
        raise SystemExit("pfa-benchmark needs an MI355X: this package has no CPU path")
    dev = torch.device("cuda", torch.cuda.current_device())
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    attention = HybridFlashAttention(embed_dim=args.embed_dim, num_heads=args.num_heads, enable_scaling=True,
                                     dtype=dtype).to(dev).eval()
    H, D = args.num_heads, args.embed_dim // args.num_heads
    results = []
    for batch_size in args.batch_sizes:
        for seq_len in args.seq_lengths:
            query = torch.randn(batch_size, seq_len, args.embed_dim, device=dev, dtype=dtype)
            with torch.no_grad():
                for _ in range(3):
                    attention(query, is_causal=args.causal)
                torch.cuda.synchronize()
                latencies = []
                for _ in range(args.num_iterations):
                    t0 = time.perf_counter()
                    attention(query, is_causal=args.causal)
                    torch.cuda.synchronize()
                    latencies.append((time.perf_counter() - t0) * 1000)
                # kernel-only time of the attention core on the same shape
                cd = torch.bfloat16 if dtype == torch.float32 else dtype
                q, k, v = (torch.randn(batch_size, seq_len, H, D, device=dev).to(cd) for _ in range(3))
                ops.fa3_forward_bshd(q, k, v, causal=args.causal)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.num_iterations):
                    ops.fa3_forward_bshd(q, k, v, causal=args.causal)
                e1.record()
                torch.cuda.synchronize()
                core_ms = e0.elapsed_time(e1) / args.num_iterations
            avg = float(np.mean(latencies))
            flops = 4.0 * batch_size * H * seq_len * seq_len * D / (2 if args.causal else 1)
            stats = attention.get_performance_stats()
            results.append({
                "batch_size": batch_size, "seq_length": seq_len, "embed_dim": args.embed_dim, "num_heads": H,
                "avg_latency_ms": avg, "std_latency_ms": float(np.std(latencies)),
                "min_latency_ms": float(np.min(latencies)), "max_latency_ms": float(np.max(latencies)),
                "tokens_per_sec": batch_size * seq_len / (avg / 1000),
                "last_device_used": "gpu",
                "gpu_calls": stats.get("gpu_samples", 0), "photonic_calls": 0, "photonic_usage_ratio": 0.0,
                "core_ms": core_ms, "attn_tflops": flops / core_ms / 1e9,
                "frac_mfma_peak": flops / core_ms / 1e9 / MFMA_PEAK_TFLOPS, "causal": bool(args.causal),
            })
            r = results[-1]
            line = (f"Batch {batch_size:2d}, Seq {seq_len:5d}: module {avg:8.3f} ms, {r['tokens_per_sec']:10.0f} tok/s | "
                    f"core {core_ms:7.3f} ms, {r['attn_tflops']:7.1f} TFLOP/s")
            (logger.info if not args.verbose else print)(line)
            if not args.verbose:
                print(line)
    if args.output:
        with open(args.output, "w") as f:
            json.dump({"benchmark_info": {"version": get_version(), "timestamp": time.time(),
                                          "device_info": get_device_info(), "config": get_config().to_dict()},
                       "results": results}, f, indent=2)
    return results


def device_info(args=None):
    print(json.dumps(get_device_info(), indent=2))


def main():
    cmds = {"benchmark": benchmark, "device-info": device_info}
    if len(sys.argv) < 2 or sys.argv[1] not in cmds:
        raise SystemExit(f"usage: python -m photonic_flash_attention_amd.cli {{{','.join(cmds)}}} [options]")
    cmds[sys.argv[1]]()


if __name__ == "__main__":
    main()
