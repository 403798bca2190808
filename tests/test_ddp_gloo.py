"""CPU, world_size 2, gloo: the data-parallel exchange step (batch sharding + one flat all-reduce)
reproduces the single-process full-batch gradient.  Compute is the CPU oracle (the product has no
CPU path); the code under test is nesvor_amd.ddp — the same functions the GPU trainer uses over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import small_args


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat_grads(golden, rows, noise):
    from oracle import hashgrid as hg
    from oracle import nesvor_model as nm

    args = small_args()
    P = {str(k): torch.tensor(golden[f"fw_sd::{k}"]) for k in golden["fw_state_keys"]}
    bb, ax0 = P.pop("inr.bounding_box"), P.pop("axisangle_init")
    base, L = nm.grid_config(bb, args)
    levels = hg.make_levels(L, args.log2_hashmap_size, base, args.level_scale)
    for v in P.values():
        v.requires_grad_(True)
    t = lambda k: torch.tensor(golden[f"fw_{k}"])[rows]
    losses = nm.nesvor_forward(P, levels, args, bb, torch.tensor(golden["fw_psf_sigma"]), ax0, float(golden["fw_delta"]),
                               t("xyz"), t("v"), t("idx"), noise[rows])
    nm.total_loss(losses, args).backward()
    names = sorted(P)
    return torch.cat([P[n].grad.reshape(-1) for n in names]), names


def _worker(rank, world, port, path, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from nesvor_amd import ddp

    r, lr, w = ddp.init_distributed("gloo")
    assert (r, w) == (rank, world)
    golden = np.load(path)
    noise = torch.tensor(golden["fw_noise"])
    B = noise.shape[0]
    batch = {"rows": torch.arange(B)}
    rows = ddp.shard_batch(batch, rank, world)["rows"]
    flat, _ = _flat_grads(golden, rows, noise)
    ddp.make_reduce_hook(n_buckets=3)(flat)  # bucketed all-reduce(sum) in place
    flat /= world
    p = torch.full((5,), float(rank))
    ddp.broadcast_params_(p, src=0)
    assert float(p.sum()) == 0.0
    if rank == 0:
        torch.save(flat, out)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_allreduce_equals_full_batch(golden, tmp_path):
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_golden.npz")
    out = str(tmp_path / "flat.pt")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, path, out), nprocs=2, join=True)
    got = torch.load(out)
    noise = torch.tensor(golden["fw_noise"])
    full, names = _flat_grads(golden, torch.arange(noise.shape[0]), noise)
    # every loss term is a mean over pixels (or a function of replicated params): the average of the two
    # half-batch gradients equals the full-batch gradient up to fp32 summation order
    scale = float(full.abs().max())
    assert float((got - full).abs().max()) < 1e-5 * scale + 1e-9


def test_shard_batch_rows():
    from nesvor_amd.ddp import shard_batch

    b = {"xyz": torch.arange(24.0).view(8, 3), "v": torch.arange(8.0)}
    parts = [shard_batch(b, r, 4) for r in range(4)]
    assert torch.equal(torch.cat([p["v"] for p in parts]), b["v"])
    assert all(p["xyz"].shape == (2, 3) for p in parts)
    assert shard_batch(b, 0, 1) is b
    with pytest.raises(AssertionError):
        shard_batch(b, 0, 3)


def _sharded_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from nesvor_amd import ddp

    ddp.init_distributed("gloo")
    n = 8 * world * 5
    ex = ddp.ShardedExchange(n)
    assert ex.shard == n // world and (ex.lo, ex.hi) == (rank * ex.shard, (rank + 1) * ex.shard)
    g = torch.Generator().manual_seed(7)
    grads = [torch.randn(n, generator=g) for _ in range(world)]  # every rank can rebuild every rank's gradient
    mine = ex.reduce_scatter(grads[rank].clone())
    torch.testing.assert_close(mine, sum(grads)[ex.lo : ex.hi])
    # "optimizer": every rank updates its own slice only, then the slices are gathered
    param = torch.zeros(n)
    param[ex.lo : ex.hi] = -0.1 * mine
    ex.all_gather_(param)
    torch.testing.assert_close(param, -0.1 * sum(grads))
    if rank == 0:
        torch.save(param, out)
    with pytest.raises(ValueError):
        ddp.ShardedExchange(n + 1)
    # two independently sharded parts, the second one exchanged early and asynchronously (the fine hash-grid levels)
    split = 8 * world * 3
    ex2 = ddp.ShardedExchange(n, split=split)
    assert ex2.parts == [(0, split), (split, n)]
    with pytest.raises(ValueError):
        ddp.ShardedExchange(n, split=split + 4)
    g2 = grads[rank].clone()
    late_garbage = g2[:split].clone()
    g2[:split] = float("nan")  # the late part is not complete yet when the early exchange starts: it must not be touched
    mine1, work = ex2.reduce_scatter(g2, 1, async_op=True)
    g2[:split] = late_garbage
    mine0 = ex2.reduce_scatter(g2, 0)
    work.wait()
    total = sum(grads)
    lo0, hi0 = ex2.owned(0)
    lo1, hi1 = ex2.owned(1)
    assert (hi0 - lo0, hi1 - lo1) == (split // world, (n - split) // world)
    torch.testing.assert_close(mine0, total[lo0:hi0])
    torch.testing.assert_close(mine1, total[lo1:hi1])
    param2 = torch.zeros(n)
    param2[lo0:hi0], param2[lo1:hi1] = -0.1 * mine0, -0.1 * mine1
    ex2.all_gather_(param2)
    torch.testing.assert_close(param2, -0.1 * total)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_exchange_equals_allreduce_then_full_update(tmp_path):
    """reduce-scatter -> per-rank update of its slice -> all-gather == all-reduce -> full update (ddp.ShardedExchange; gloo
    has no reduce-scatter, so this exercises the all-reduce-based fallback of the same interface)."""
    port = _free_port()
    out = str(tmp_path / "param.pt")
    mp.spawn(_sharded_worker, args=(2, port, out), nprocs=2, join=True)
    g = torch.Generator().manual_seed(7)
    grads = [torch.randn(80, generator=g) for _ in range(2)]
    torch.testing.assert_close(torch.load(out), -0.1 * sum(grads))


def test_cap_hw_queues_overrides_before_hip_initialises(monkeypatch):
    """``GPU_MAX_HW_QUEUES`` above 4 makes every kernel of the data-parallel step start ~40 us late on MI355X
    (profiles/r04_ddp_queue_probe*.log): ``ddp.init_distributed`` lowers it before the HIP runtime reads it (round 4 only warned)."""
    import torch

    from nesvor_amd import ddp

    if torch.cuda.is_initialized():
        pytest.skip("the HIP runtime is already up in this process")
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    assert ddp.cap_hw_queues() == "lowered" and os.environ["GPU_MAX_HW_QUEUES"] == "4"
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    monkeypatch.setenv("NESVOR_KEEP_HW_QUEUES", "1")  # the opt-out: the caller's export stands
    assert ddp.cap_hw_queues() == "kept" and os.environ["GPU_MAX_HW_QUEUES"] == "8"
    monkeypatch.delenv("NESVOR_KEEP_HW_QUEUES")
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "2")
    assert ddp.cap_hw_queues() is None and os.environ["GPU_MAX_HW_QUEUES"] == "2"
    monkeypatch.delenv("GPU_MAX_HW_QUEUES")
    assert ddp.cap_hw_queues() is None and "GPU_MAX_HW_QUEUES" not in os.environ
