import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from bench import make_args
from nesvor_amd.phantom import phantom3d, simulate_stacks
from nesvor_amd.train import Dataset, train
from nesvor_amd.models import NeSVoR
dev = torch.device("cuda:0")
vol = torch.tensor(phantom3d(n=128), dtype=torch.float32, device=dev)
torch.manual_seed(0)
slices, _ = simulate_stacks(vol, n_stacks=3)
args = make_args(dev, 4096, 256, 2, 200)
def t(f, name):
    torch.cuda.synchronize(); t0 = time.time(); r = f(); torch.cuda.synchronize(); print(f"{name}: {time.time()-t0:.3f} s", flush=True); return r
ds = t(lambda: Dataset(slices, args), "Dataset()")
ds = t(lambda: Dataset(slices, args), "Dataset() again")
m = t(lambda: NeSVoR(ds.transformation, ds.resolution, ds.mean, ds.bounding_box, args), "NeSVoR()")
t(lambda: ds.mask, "dataset.mask")
t(lambda: train(slices, args), "train(200 it)")
t(lambda: train(slices, args), "train(200 it) again")
