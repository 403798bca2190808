"""Replay the first iterations of a tests/golden/oracle_run_*.npz fixture through the HIP train() and print both loss tables.
python tools/replay_oracle_run.py oracle_run_c5_nobias.npz [n_iter]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from bench import make_args
from nesvor_amd.phantom import phantom3d, simulate_stacks
from nesvor_amd.train import Dataset, train

dev = torch.device("cuda:0")
gold = np.load(os.path.join(ROOT, "tests", "golden", sys.argv[1]))
n_show = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n_iter, B, S, _, n_stacks = (int(x) for x in gold["config"][:5])
motion_deg, motion_mm, seed, nlb, out_res, stride = (float(x) for x in gold["config_ext"])
vol = torch.tensor(phantom3d(n=128), dtype=torch.float32, device=dev)
slices, _ = simulate_stacks(vol, n_stacks=n_stacks, motion_deg=motion_deg, motion_mm=motion_mm, seed=int(seed))
args = make_args(dev, B, S, 2, n_iter)
args.n_levels_bias, args.output_resolution, args.host_rng = int(nlb), out_res, True
hist = []


class Stop(Exception):
    pass


def cb(i, losses):
    hist.append([float(losses[k]) for k in losses])
    if i >= n_show:
        raise Stop


torch.manual_seed(0)
try:
    train(slices, args, on_iteration=cb)
except Stop:
    pass
keys = [str(k) for k in gold["loss_keys"]]
got, ref = np.array(hist), gold["loss_history"][: len(hist)]
np.set_printoptions(linewidth=200, precision=7)
for j, k in enumerate(keys):
    print(k)
    print("  hip   ", got[:, j])
    print("  oracle", ref[:, j])
    print("  rel   ", np.abs(got[:, j] - ref[:, j]) / (np.abs(ref[:, j]) + 1e-12))
