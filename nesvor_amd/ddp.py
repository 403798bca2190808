"""Data-parallel training over the GPUs of one node: one process per GPU,
``torch.distributed`` (backend "nccl" == RCCL over xGMI; "gloo" for CPU tests).

The reference is single-process; this is the build's addition
(SURVEY.md §8e).  Every rank holds all parameters and the whole flat data set,
draws the same permutation (same seed) and takes its own slice of each global
batch; per-rank PSF noise differs (seed + rank).  The only exchange step per
iteration is the sum-all-reduce of the flat gradient buffer (small MLP / per-slice
gradients first, then the hash table, ~30 MB fp32, coarse levels before fine ones).

xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU): a ring all-reduce
of the 30 MB buffer moves 2(W-1)/W x 30 MB per GPU through single links
(~0.34 ms at W=8), comparable to a fast iteration.  The autograd-free step
(nesvor_amd.direct) therefore runs the hash-grid backward in two launches: the fine
levels - half of the table's bytes, the end of the flat buffer - first; their
all-reduce starts at once on RCCL's stream and overlaps the coarse levels' launch;
the rest of the buffer (one contiguous range) follows when the step's gradients are
complete (FusedTrainer.optimizer_step).  NESVOR_DDP_OVERLAP=0 = one all-reduce.
"""
import os
from typing import Dict, Optional

import torch
import torch.distributed as dist


def forced() -> bool:
    """NESVOR_DDP_FORCE=1: run the data-parallel exchange even in a group of ONE rank (every collective is then the
    identity).  Lets the production backend - "nccl" = RCCL - be exercised end to end on a 1-GPU box."""
    return os.environ.get("NESVOR_DDP_FORCE") == "1"


def active(group=None) -> bool:
    """The data-parallel exchange is on: a process group exists and has more than one rank (or forced())."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or forced())


def cap_hw_queues(limit: int = 4) -> Optional[str]:
    """``GPU_MAX_HW_QUEUES`` above ``limit`` is lowered to it - BEFORE the HIP runtime reads it (its first call in this process).
    Measured (tools/ddp_queue_probe.sh, profiles/r04_ddp_queue_probe*.log): with 8 hardware queues and RCCL's queues live EVERY
    kernel of the data-parallel step starts ~40 us late (prologue 5 -> 48 us, loss 17 -> 56 us in a kernel trace; step 1.20 ->
    1.86 ms) whatever the stream layout or stream priorities - the command processor's queue switching, not an ordering problem
    this code could fix.  ROCm's default (4) and 2 do not show it.  Round 4 only warned; a launcher or a harness that exports the
    variable would have made the first scaling curve ever measured read 50 % low, so it is overridden (and logged).  Returns what
    happened ("lowered" / "kept" / "too late: HIP already initialised" / None).  The rewrite is visible to child processes of this one
    (os.environ); ``NESVOR_KEEP_HW_QUEUES=1`` opts out."""
    hwq = os.environ.get("GPU_MAX_HW_QUEUES")
    if hwq is None or not hwq.strip().isdigit() or int(hwq) <= limit:
        return None
    import logging

    if os.environ.get("NESVOR_KEEP_HW_QUEUES") == "1":  # opt-out (round-5 advisor): the caller's export stands, child processes see it unchanged
        logging.warning("GPU_MAX_HW_QUEUES=%s kept (NESVOR_KEEP_HW_QUEUES=1): expect the data-parallel step ~50 %% slower above %d "
                        "hardware queues on MI355X (nesvor_amd/ddp.py)", hwq, limit)
        return "kept"

    if torch.cuda.is_initialized():
        logging.warning("GPU_MAX_HW_QUEUES=%s and the HIP runtime is already initialised: the data-parallel step runs ~50 %% slower "
                        "above %d hardware queues on MI355X (nesvor_amd/ddp.py); export GPU_MAX_HW_QUEUES=%d before launching", hwq, limit, limit)
        return "too late: HIP already initialised"
    os.environ["GPU_MAX_HW_QUEUES"] = str(limit)
    logging.warning("GPU_MAX_HW_QUEUES=%s lowered to %d for this process (above %d every kernel of the data-parallel step starts ~40 us "
                    "late on MI355X: nesvor_amd/ddp.py; NESVOR_KEEP_HW_QUEUES=1 keeps the exported value)", hwq, limit, limit)
    return "lowered"


def init_distributed(backend: Optional[str] = None):
    """Initialise from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, local_rank, world_size).  No-op single-process fallback when WORLD_SIZE is unset."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or forced()) and not dist.is_initialized():
        cap_hw_queues()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # NESVOR_DIST_BACKEND=gloo lets the multi-process path be exercised on a 1-GPU box (every rank
            # then shares device 0, see local_device()); production is nccl (= RCCL over xGMI)
            backend = os.environ.get("NESVOR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def local_device(local_rank: int) -> torch.device:
    """cuda:<local_rank>, or cuda:0 for every rank when NESVOR_SINGLE_DEVICE=1 (1-GPU test boxes)."""
    if os.environ.get("NESVOR_SINGLE_DEVICE") == "1":
        return torch.device("cuda", 0)
    return torch.device("cuda", local_rank)


def shard_batch(batch: Dict[str, torch.Tensor], rank: int, world: int) -> Dict[str, torch.Tensor]:
    """Rank r takes rows [r*B/W, (r+1)*B/W) of a global batch of B rows (B % W == 0)."""
    if world == 1:
        return batch
    out = {}
    for k, v in batch.items():
        b = v.shape[0]
        assert b % world == 0, f"global batch {b} not divisible by world size {world}"
        per = b // world
        out[k] = v[rank * per : (rank + 1) * per]
    return out


def allreduce_flat_(flat_grad: torch.Tensor, group=None, n_buckets: int = 1):
    """Sum-all-reduce the flat gradient buffer in place.  Returns the list of async work handles
    (one per bucket) so the caller may overlap; call .wait() on each (or use wait_all)."""
    if not active(group):
        return []
    n = flat_grad.numel()
    n_buckets = max(1, min(n_buckets, n))
    per = -(-n // n_buckets)
    per = (per + 1023) // 1024 * 1024
    works = []
    for s in range(0, n, per):
        works.append(dist.all_reduce(flat_grad[s : s + per], op=dist.ReduceOp.SUM, group=group, async_op=True))
    return works


def wait_all(works) -> None:
    for w in works:
        w.wait()


def make_reduce_hook(group=None, n_buckets: int = 1):
    """Hook for FusedTrainer.reduce_hook: blocking sum-all-reduce of the flat gradient."""

    def hook(flat_grad: torch.Tensor) -> None:
        wait_all(allreduce_flat_(flat_grad, group, n_buckets))

    return hook


def broadcast_params_(flat_param: torch.Tensor, src: int = 0, group=None) -> None:
    """Make every rank start from rank `src`'s parameters."""
    if active(group):
        dist.broadcast(flat_param, src=src, group=group)


def _backend_has_reduce_scatter(group=None) -> bool:
    return dist.get_backend(group) != "gloo"  # ProcessGroupGloo implements neither reduce_scatter nor its tensor form


class ShardedExchange:
    """Optimizer-state sharding over the data-parallel ranks (SURVEY.md 8e caveat 3): instead of all-reduce + a dense
    AdamW sweep on every rank,

        reduce-scatter(sum) of the flat gradient  ->  AdamW on this rank's 1/W slice of the flat buffers
        ->  all-gather of the updated parameter slices.

    The two collectives are the two halves of a ring all-reduce (same bytes on the wire), but over the xGMI mesh every
    rank exchanges its 1/W slices with all W-1 peers concurrently, and the dense optimizer sweep - the per-iteration cost
    that does not shrink with W - drops to 1/W of the table per rank.  The flat buffers are padded to a multiple of 4 W
    elements by the trainer.

    ``split`` (a multiple of 4 W, or None) cuts the buffer into two independently sharded PARTS, [0, split) and
    [split, numel): part 1 - the fine hash-grid levels, the end of the buffer, whose gradient an iteration finishes first -
    is reduce-scattered EARLY (asynchronously, under the rest of the backward, ``reduce_scatter(.., part=1,
    async_op=True)``), part 0 when the step's gradients are complete; every rank then owns one slice of each part.  This is
    the sharded counterpart of the early all-reduce of ``nesvor_amd.direct``.

    On a backend without reduce-scatter (gloo, CPU tests) the gradient is all-reduced and the slice taken from it - same
    result."""

    def __init__(self, numel: int, group=None, split: Optional[int] = None) -> None:
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if numel % (4 * self.world):
            raise ValueError("flat buffers must be padded to a multiple of 4 x world size")
        if split is not None and (split % (4 * self.world) or not 0 < split < numel):
            raise ValueError("split must be a multiple of 4 x world size inside the buffer")
        self.parts = [(0, numel)] if split is None else [(0, split), (split, numel)]
        self.native = _backend_has_reduce_scatter(group)
        self._mine = [None] * len(self.parts)
        # single-part compatibility attributes
        self.shard = numel // self.world
        self.lo, self.hi = self.owned(0) if split is None else (None, None)

    def owned(self, part: int = 0):
        """[lo, hi) of the flat buffers this rank owns inside ``part``."""
        a, b = self.parts[part]
        n = (b - a) // self.world
        return a + self.rank * n, a + (self.rank + 1) * n

    def reduce_scatter(self, flat_grad: torch.Tensor, part: int = 0, async_op: bool = False):
        """-> this rank's slice of the summed gradient of ``part`` (a buffer owned by the exchange); with ``async_op``
        -> (slice, work handle): the slice is valid after ``work.wait()``."""
        a, b = self.parts[part]
        lo, hi = self.owned(part)
        mine = self._mine[part]
        if mine is None or mine.device != flat_grad.device:
            mine = self._mine[part] = torch.empty(hi - lo, dtype=flat_grad.dtype, device=flat_grad.device)
        if self.native:
            work = dist.reduce_scatter_tensor(mine, flat_grad[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        else:
            work = dist.all_reduce(flat_grad[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            if async_op:
                work = _ThenCopy(work, mine, flat_grad[lo:hi])
            else:
                mine.copy_(flat_grad[lo:hi])
        return (mine, work) if async_op else mine

    def all_gather_(self, flat_param: torch.Tensor) -> None:
        """Every rank's updated slices -> the full parameter buffer on every rank (in place)."""
        for part, (a, b) in enumerate(self.parts):
            lo, hi = self.owned(part)
            if self.native:
                dist.all_gather_into_tensor(flat_param[a:b], flat_param[lo:hi].clone(), group=self.group)
            else:
                pieces = [torch.empty(hi - lo, dtype=flat_param.dtype, device=flat_param.device) for _ in range(self.world)]
                dist.all_gather(pieces, flat_param[lo:hi].clone(), group=self.group)
                flat_param[a:b].copy_(torch.cat(pieces))


class _ThenCopy:
    """Work handle of the all-reduce-based fallback: wait, then take this rank's slice."""

    def __init__(self, work, dst, src):
        self.work, self.dst, self.src = work, dst, src

    def wait(self):
        self.work.wait()
        self.dst.copy_(self.src)
