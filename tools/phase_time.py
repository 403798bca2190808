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
from nesvor_amd.sample import sample_slices, sample_volume
args2 = make_args(dev, 4096, 256, 2, 2000)
inr, out_slices, mask = t(lambda: train(slices, args2), "train(2000 it)")
args2.output_resolution = 0.8
v = t(lambda: sample_volume(inr, mask, args2), "sample_volume (0.8 mm)")
print("   volume", tuple(v.image.shape), int(v.mask.sum()), "voxels in mask")
v = t(lambda: sample_volume(inr, mask, args2), "sample_volume again")
s2 = t(lambda: sample_slices(inr, out_slices[:40], mask, args2), "sample_slices (40 slices)")
