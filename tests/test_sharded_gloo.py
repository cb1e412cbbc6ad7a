"""World-size-2 gloo test of the multi-GPU path's host logic (shard plan, local slices, the single
gather, head-shard re-assembly).  The per-shard compute is injected (the oracle as checker): the
HIP kernel itself is covered by the -m gpu tests; this covers the N>1 plumbing on CPU."""

from __future__ import annotations

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, H, S, D, causal, q_out):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import fa3_oracle as orc
        from photonic_flash_attention_amd import synth
        from photonic_flash_attention_amd.parallel import sharded
        torch.set_num_threads(2)
        q, k, v = synth.qkv(B, H, S, S, D, 11, "bf16")
        lens = [S - 3 * b for b in range(B)]
        calls = []

        def compute(a, b_, c, causal=False, seqlens_k=None):
            calls.append(tuple(a.shape))
            return orc.attention_bshd(a, b_, c, causal=causal, seqlens_k=seqlens_k)

        plan = sharded.shard_plan(B, H, world)
        full = sharded.sharded_attention(q, k, v, causal=causal, compute=compute,
                                         seqlens_k=lens if plan[0] == "batch" else None)
        ref = orc.attention_bshd(q, k, v, causal=causal, seqlens_k=lens if plan[0] == "batch" else None)
        ok = bool(torch.equal(full, ref)) or float((full - ref).abs().max()) < 1e-6
        q_out.put((rank, plan, calls[0], ok, tuple(full.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape", [(4, 2, 96, 64, True), (1, 4, 80, 64, False), (3, 6, 70, 64, True)])
def test_shard_compute_gather_equals_single_rank(shape):
    B, H, S, D, causal = shape
    world = 2
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, H, S, D, causal, q_out)) for r in range(world)]
    [p.start() for p in procs]
    res = [q_out.get(timeout=240) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, plan, local_shape, ok, full_shape in res:
        assert ok and full_shape == (B, S, H, D)
        if B % world == 0:
            assert plan == ("batch", B // world) and local_shape == (B // world, S, H, D)
        else:
            assert plan == ("head", H // world) and local_shape == (B, S, H // world, D)


def test_shard_plan_rules():
    from photonic_flash_attention_amd.parallel import sharded
    assert sharded.shard_plan(32, 16, 8) == ("batch", 4)      # C4
    assert sharded.shard_plan(4, 16, 8) == ("head", 2)        # C3 on 8 GPUs
    assert sharded.shard_plan(1, 32, 8) == ("head", 4)        # C5
    assert sharded.shard_plan(4, 16, 1) == ("batch", 4)
    with pytest.raises(ValueError):
        sharded.shard_plan(3, 5, 2)
