"""Multi-GPU form of the hot path: batch x head sharding + ONE gather.

Every (batch, head) pair is an independent attention problem (the reference batches over them
in one matmul, flash_attention_3.py:162/231), so ranks need no exchange while computing.  One
process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI; "gloo" in CPU tests);
each rank runs the HIP kernel on its shard, then a single ``all_gather_into_tensor`` assembles
the full output -- the combine the reference's in-process "cluster" does with ``torch.cat(dim=0)``
(scaling/distributed_computing.py:663).

Partitioning (batch-major over the flattened B*H grid):
  * world divides B       -> shard the batch: rank r owns batches [r*B/W, (r+1)*B/W); the gathered
                             tensor is a plain dim-0 concat (C4: B=32 -> 4 per GPU).
  * otherwise, world | H  -> shard the heads: rank r owns heads [r*H/W, (r+1)*H/W) of every batch
                             (C3 on 8 GPUs: 2 heads per rank; C5: 4); gathered as [W,B,S,H/W,D] and
                             permuted back to [B,S,H,D].
xGMI is a point-to-point mesh (7 links per GPU), so the gather is per-link bound: 64 MiB per
rank at C4 is ~0.45 ms per peer link at best -- the same order as the kernel; it is therefore
kept OUT of the per-forward critical path where the caller can consume sharded outputs (e.g. a
row-parallel out_proj), and can be overlapped with the next forward on a side stream.
"""

from __future__ import annotations

import time
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

GATHER_ALGO = "all_gather_into_tensor"
GATHER_ALGOS = ("all_gather_into_tensor", "direct")


def _direct_gather(buf: torch.Tensor, src: torch.Tensor, group=None) -> None:
    """The combine as W-1 concurrent point-to-point exchanges instead of a collective: rank r sends its shard to every peer and
    receives every peer's shard straight into its slot of ``buf`` ([W, ...]), all posted in ONE ``batch_isend_irecv`` group.  xGMI
    is a full mesh of point-to-point links (7 per GPU), so the W-1 transfers of a rank run on W-1 different links at the same
    time -- a ring all-gather pushes the same bytes over one link in W-1 serial steps."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    buf[rank].copy_(src)
    ops = []
    for d in range(1, world):                      # distance d: send to rank+d, receive from rank-d (every pair meets once per d)
        to, frm = (rank + d) % world, (rank - d) % world
        ops.append(dist.P2POp(dist.isend, src, to if group is None else dist.get_global_rank(group, to), group))
        ops.append(dist.P2POp(dist.irecv, buf[frm], frm if group is None else dist.get_global_rank(group, frm), group))
    for w in dist.batch_isend_irecv(ops):
        w.wait()


def shard_plan(B: int, H: int, world: int) -> Tuple[str, int]:
    """-> ("batch" | "head", shard size).  Raises if neither B nor H is divisible by world."""
    if world == 1:
        return "batch", B
    if B % world == 0:
        return "batch", B // world
    if H % world == 0:
        return "head", H // world
    raise ValueError(f"cannot shard B={B}, H={H} evenly over {world} ranks")


def local_slice(t: torch.Tensor, plan: Tuple[str, int], rank: int) -> torch.Tensor:
    """Slice a ``[B,S,H,D]`` tensor down to this rank's shard (a view, no copy)."""
    kind, n = plan
    if kind == "batch":
        return t[rank * n:(rank + 1) * n]
    return t[:, :, rank * n:(rank + 1) * n]


def gather_outputs(out_local: torch.Tensor, plan: Tuple[str, int] = ("batch", 0), group=None,
                   timed: bool = False, algo: str = GATHER_ALGO):
    """Gather the per-rank ``[B_l,S,H_l,D]`` outputs into the full ``[B,S,H,D]`` tensor on every rank: ``algo`` =
    "all_gather_into_tensor" (RCCL's collective) or "direct" (W-1 concurrent peer exchanges, one per xGMI link)."""
    if algo not in GATHER_ALGOS:
        raise ValueError(f"gather algo must be one of {GATHER_ALGOS}")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return (out_local, 0.0) if timed else out_local
    src = out_local.contiguous()
    flat = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    buf = flat.view((world,) + tuple(src.shape))
    t0 = time.perf_counter()
    if src.is_cuda:
        torch.cuda.synchronize(src.device)
        t0 = time.perf_counter()
    if algo == "direct":
        _direct_gather(buf, src, group)
    else:
        dist.all_gather_into_tensor(flat, src, group=group)
    if src.is_cuda:
        torch.cuda.synchronize(src.device)
    ms = (time.perf_counter() - t0) * 1e3
    if plan[0] == "head":
        W, B, S, Hl, D = buf.shape
        full = buf.permute(1, 2, 0, 3, 4).reshape(B, S, W * Hl, D)
    else:
        full = buf.reshape((-1,) + tuple(src.shape[1:]))
    return (full, ms) if timed else full


def sharded_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, causal: bool = False,
                      group=None, gather: bool = True,
                      compute: Optional[Callable] = None, **kw) -> torch.Tensor:
    """Full ``[B,S,H,D]`` operands replicated (or addressable) on every rank -> full output.

    Each rank computes only its shard; ``compute(q,k,v,causal=..., **kw) -> [B_l,Sq,H_l,D]`` defaults
    to the HIP kernel (``ops.fa3_forward_bshd``).  Tests inject a CPU checker there to exercise the
    shard/gather logic under gloo without a GPU."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B, _, H, _ = q.shape
    plan = shard_plan(B, H, world)
    if compute is None:
        from .. import ops

        def compute(a, b, c, **kk):
            return ops.fa3_forward_bshd(a, b, c, **kk)[0]
    ql, kl, vl = (local_slice(t, plan, rank) for t in (q, k, v))
    n = plan[1]
    if "seqlens_k" in kw and kw["seqlens_k"] is not None and plan[0] == "batch":
        kw = dict(kw, seqlens_k=list(kw["seqlens_k"])[rank * n:(rank + 1) * n])
    # masks follow their shard: a [B,Sk] key mask along B; a reference-style mask along its batch / head dimension unless that
    # dimension broadcasts (size 1)
    if kw.get("key_mask") is not None and plan[0] == "batch":
        kw = dict(kw, key_mask=kw["key_mask"][rank * n:(rank + 1) * n])
    if kw.get("mask") is not None:
        m = kw["mask"]
        if m.dim() == 2 and plan[0] == "batch" and m.shape[0] == B:          # [B,Sk]
            m = m[rank * n:(rank + 1) * n]
        elif m.dim() == 3 and plan[0] == "batch" and m.shape[0] == B and B > 1:    # [B,Sq|1,Sk]
            m = m[rank * n:(rank + 1) * n]
        elif m.dim() == 4:
            if plan[0] == "batch" and m.shape[0] == B and B > 1:
                m = m[rank * n:(rank + 1) * n]
            elif plan[0] == "head" and m.shape[1] == H and H > 1:
                m = m[:, rank * n:(rank + 1) * n]
        kw = dict(kw, mask=m)
    out_local = compute(ql, kl, vl, causal=causal, **kw)
    if not gather or world == 1:
        return out_local
    return gather_outputs(out_local, plan, group)


def overlapped_forward_gather(step: Callable[[], None], out_local: torch.Tensor, steps: int, group=None,
                              algo: str = GATHER_ALGO) -> float:
    """Secondary measurement: one gather per forward, issued on a side stream so that the gather of
    step i overlaps the kernel of step i+1 (double-buffered gather target).  Returns ms per step.  With host tensors (a "gloo"
    rehearsal of the control flow) the forward and the gather simply alternate."""
    world = dist.get_world_size(group)
    dev = out_local.device
    if not out_local.is_cuda:
        flat = torch.empty((world * out_local.shape[0],) + tuple(out_local.shape[1:]), dtype=out_local.dtype)
        dist.barrier(group)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
            if algo == "direct":
                _direct_gather(flat.view((world,) + tuple(out_local.shape)), out_local.contiguous(), group)
            else:
                dist.all_gather_into_tensor(flat, out_local.contiguous(), group=group)
        dist.barrier(group)
        return (time.perf_counter() - t0) * 1e3 / steps
    comm = torch.cuda.Stream(device=dev)
    bufs = [torch.empty((world * out_local.shape[0],) + tuple(out_local.shape[1:]), dtype=out_local.dtype, device=dev)
            for _ in range(2)]
    stage = [torch.empty_like(out_local) for _ in range(2)]
    main = torch.cuda.current_stream(dev)
    done = [None, None]
    dist.barrier(group)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        s = i & 1
        if done[s] is not None:
            main.wait_event(done[s])          # stage[s] is free again
        step()
        stage[s].copy_(out_local, non_blocking=True)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(comm):
            comm.wait_event(ready)
            if algo == "direct":
                _direct_gather(bufs[s].view((world,) + tuple(out_local.shape)), stage[s], group)
            else:
                dist.all_gather_into_tensor(bufs[s], stage[s], group=group)
            done[s] = torch.cuda.Event()
            done[s].record(comm)
    torch.cuda.synchronize(dev)
    dist.barrier(group)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) * 1e3 / steps
