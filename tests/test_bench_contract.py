"""GPU (-m gpu): the driver's multi-GPU invocation of bench.py, exercised on the one GPU a test box has.

``python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ... bench.py --gpus 2`` with both ranks
on device 0 (``NESVOR_SINGLE_DEVICE=1``) over gloo (``NESVOR_DIST_BACKEND=gloo``): the whole data-parallel path of the bench -
process group, broadcast, sharded batches, split backward with the early exchange, barrier + max over ranks - must come out as
EXACTLY ONE JSON line on stdout that obeys the contract (round-4 verdict, item 6b: until round 5 this only existed as a log under
gpurun_out/)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_emit_one_contract_line(device):
    env = dict(os.environ, NESVOR_SINGLE_DEVICE="1", NESVOR_DIST_BACKEND="gloo", GPU_MAX_HW_QUEUES="8")  # (8: must be lowered, not obeyed)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29531", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
           "--no-cpu-baseline", "--no-extras", "--no-strict", "--small-batches", ""]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["config"]["parallelism"] == "dp2" and d["config"]["global_batch_pixels"] == 2 * 4096
    assert "2^20 points/iter/GPU" in d["config"]["workload"]
    assert d["value"] > 0 and abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]  # whole-job rate: 2 x 2^20 points per step
    assert d["timed_regions"] == 5 and len(d["timed_regions_ms_per_step"]) == 5
    s = d["strong_scaling"]
    assert s is not None and s["scaling"] == "strong" and s["global_batch_pixels"] == 4096 and s["points_per_gpu_per_iter"] == 1 << 19
    assert d["cpu_baseline"] is None and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "lowered to 4" in p.stderr  # ddp.cap_hw_queues
