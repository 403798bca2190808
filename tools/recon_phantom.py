"""End-to-end check at BASELINE scale: train the INR on 3 simulated stacks of the 128^3 phantom with the
reference's `train()` signature, then report PSNR of the reconstruction against the phantom.
    python tools/recon_phantom.py [n_iter] [motion_deg motion_mm]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_args
from nesvor_amd.phantom import phantom3d, simulate_stacks
from nesvor_amd.train import train

dev = torch.device("cuda:0")
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
motion = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.0, 0.0)
n = 128
vol = torch.tensor(phantom3d(n=n), dtype=torch.float32, device=dev)
torch.manual_seed(0)
slices, true_tf = simulate_stacks(vol, n_stacks=3, motion_deg=motion[0], motion_mm=motion[1])
args = make_args(dev, 4096, 256, 2, n_iter)
args.mlp_bf16 = os.environ.get("NESVOR_MLP_BF16") == "1"  # opt-in mixed precision of the MLPs
t0 = time.time()
inr, out_slices, mask = train(slices, args)
torch.cuda.synchronize()
dt = time.time() - t0
g = torch.arange(n, dtype=torch.float32, device=dev) - (n - 1) / 2
zz, yy, xx = torch.meshgrid(g, g, g, indexing="ij")
pts = torch.stack([xx, yy, zz], -1).reshape(-1, 3)
rec = torch.empty(pts.shape[0], device=dev)
with torch.no_grad():
    for i in range(0, pts.shape[0], 1 << 18):
        rec[i : i + (1 << 18)] = inr(pts[i : i + (1 << 18), None], False).mean(-1)
truth = vol.reshape(-1)
inside = truth > 0
s = float((rec[inside] * truth[inside]).sum() / (rec[inside] ** 2).sum())  # slices were normalised by their 0.99 quantile
mse = float(((rec[inside] * s - truth[inside]) ** 2).mean())
print(f"iters {n_iter}  train wall {dt:.1f} s ({n_iter / dt:.1f} it/s incl. setup)  PSNR {10 * math.log10(float(truth.max()) ** 2 / mse):.2f} dB  (scale {s:.3f}, motion {motion})")
# is the evaluation lattice aligned with the acquisition geometry?  PSNR when the evaluation points are shifted by half a voxel
for sh in ((0, 0, 0), (0.5, 0.5, 0.5), (-0.5, -0.5, -0.5), (0.5, 0, 0), (0, 0, 0.5)):
    with torch.no_grad():
        r = torch.cat([inr(pts[i : i + (1 << 18), None] + torch.tensor(sh, device=dev), False).mean(-1) for i in range(0, pts.shape[0], 1 << 18)])
    s_ = float((r[inside] * truth[inside]).sum() / (r[inside] ** 2).sum())
    print(f"  shift {sh}: PSNR {10 * math.log10(float(truth.max()) ** 2 / float(((r[inside] * s_ - truth[inside]) ** 2).mean())):.2f} dB")
# and with the brightest structure (skull shell, intensity 1.0) excluded
soft = inside & (truth < 0.5)
print(f"  soft tissue only (0 < truth < 0.5): PSNR {10 * math.log10(float(truth.max()) ** 2 / float(((rec[soft] * s - truth[soft]) ** 2).mean())):.2f} dB")
