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


def _worker_masks(rank, world, port, q_out):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import fa3_oracle as orc
        from photonic_flash_attention_amd import synth
        from photonic_flash_attention_amd.parallel import sharded
        torch.set_num_threads(2)
        res = []
        for (B, H, S, D) in ((4, 2, 72, 64), (1, 4, 72, 64)):              # batch plan, head plan
            q, k, v = synth.qkv(B, H, S, S, D, 5, "bf16")
            km = torch.ones(B, S, dtype=torch.bool)
            km[:, S - 9:] = False
            km[B - 1, S - 20:] = False
            m4 = (torch.from_numpy(synth.normal_f32((B, H, S, S), 9)) > -0.8)
            m4[..., 0] = True
            seen = []

            def compute(a, b_, c, causal=False, key_mask=None, mask=None):
                seen.append((tuple(a.shape), None if key_mask is None else tuple(key_mask.shape), None if mask is None else tuple(mask.shape)))
                mm = mask if mask is not None else key_mask[:, None, None, :]
                if mm.dim() == 3:
                    mm = mm[:, None]
                return orc.flash_attention_forward(a.permute(0, 2, 1, 3).float(), b_.permute(0, 2, 1, 3).float(), c.permute(0, 2, 1, 3).float(),
                                                   mm, D ** -0.5).permute(0, 2, 1, 3)

            m3 = m4[:, 0].contiguous()                                           # a 3-D [B,Sq,Sk] mask follows the batch plan too
            out3 = sharded.sharded_attention(q, k, v, compute=compute, mask=m3)
            assert float((out3 - compute(q, k, v, mask=m3)).abs().max()) <= 1e-6 and seen[0][2][0] == (B // world if B % world == 0 else B)
            seen.clear()
            with pytest.raises(ValueError):                                      # the copy-engine gather maps DEVICE buffers of the peers
                sharded.gather_outputs(q, sharded.shard_plan(B, H, world), algo="sdma")
            for algo in (a_ for a_ in sharded.GATHER_ALGOS if a_ != "sdma"):
                plan = sharded.shard_plan(B, H, world)
                out_l = sharded.sharded_attention(q, k, v, compute=compute, key_mask=km, gather=False)
                full_k = sharded.gather_outputs(out_l, plan, algo=algo)
                full_m = sharded.gather_outputs(sharded.sharded_attention(q, k, v, compute=compute, mask=m4, gather=False), plan, algo=algo)
                ref_k = compute(q, k, v, key_mask=km)
                ref_m = compute(q, k, v, mask=m4)
                res.append((algo, plan[0], float((full_k - ref_k).abs().max()), float((full_m - ref_m).abs().max()), seen[0]))
        q_out.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_masks_follow_their_shard_and_both_gathers_agree():
    """A [B,Sk] key mask / a [B,H,Sq,Sk] mask is sliced with the operands (batch plan: along B, head plan: along H), and the
    direct gather (W-1 point-to-point exchanges) assembles the same tensor as all_gather_into_tensor."""
    world = 2
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_masks, args=(r, world, port, q_out)) for r in range(world)]
    [p.start() for p in procs]
    res = [q_out.get(timeout=240) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, rows in res:
        assert len(rows) == 4
        for algo, kind, ek, em, first in rows:
            assert ek <= 1e-6 and em <= 1e-6, (rank, algo, kind, ek, em)
        assert rows[0][4] == ((2, 72, 2, 64), (2, 72), None)              # batch plan: operands and key mask cut to 2 of 4 batches


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` outside a launcher spawns its ranks itself (before the parent touches a GPU).  No GPU here: the
    children must get as far as bench.py's own "needs MI355X GPUs" assertion, each with its rank environment."""
    import subprocess
    import sys
    env = dict(os.environ, PFA_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    if torch.cuda.is_available():
        assert r.returncode == 0 and '"n_gpus": 2' in r.stdout, r.stderr[-2000:]
    else:
        assert r.returncode != 0 and r.stderr.count("AssertionError: bench.py needs MI355X GPUs") == 2, r.stderr[-2000:]


def test_shard_plan_rules():
    from photonic_flash_attention_amd.parallel import sharded
    assert sharded.shard_plan(32, 16, 8) == ("batch", 4)      # C4
    assert sharded.shard_plan(4, 16, 8) == ("head", 2)        # C3 on 8 GPUs
    assert sharded.shard_plan(1, 32, 8) == ("head", 4)        # C5
    assert sharded.shard_plan(4, 16, 1) == ("batch", 4)
    with pytest.raises(ValueError):
        sharded.shard_plan(3, 5, 2)
