"""CPU oracle: the NeSVoR training loop and inference sampling.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates
``train()`` (nesvor/nesvor/train.py:123-232: batch sampler :60-75, AdamW two
groups :144-152, MultiStepLR stepped at milestones :154-159,:201-221) and
``sample_points`` (nesvor/nesvor/sample.py:17-33) on the oracle model.  With the
same ``torch.manual_seed`` it draws random numbers in the same order as the
reference (Embedding / Linear initialisers, randperm, randn), so its trajectory
can be compared with fixtures captured from the reference on CPU.

Also the ``cpu_baseline`` leg of bench.py (kind "port").
"""
import math
import time
from argparse import Namespace

import torch
import torch.nn as nn

from . import hashgrid
from . import nesvor_model as nm


class ArrayDataset:
    """train.py:14-75 on plain arrays (already flattened by the caller)."""

    def __init__(self, xyz, v, slice_idx, transformation_mat, resolution):
        self.xyz, self.v, self.slice_idx = xyz, v, slice_idx
        self.transformation_mat = transformation_mat  # (n,3,4) trans_first
        self.resolution = resolution  # (n,3)
        self.count = self.v.shape[0]
        self.epoch = 0

    @property
    def xyz_transformed(self):
        return nm.transform_points_trans_first(self.transformation_mat[self.slice_idx], self.xyz)

    @property
    def bounding_box(self):
        m = 2 * self.resolution.max()
        p = self.xyz_transformed
        return torch.stack([p.amin(0) - m, p.amax(0) + m], 0)

    @property
    def mean(self):
        v = self.v if self.v.numel() < 256**3 else self.v[: 256**3]
        q1, q2 = torch.quantile(v, torch.tensor([0.1, 0.9], dtype=v.dtype))
        return self.v[torch.logical_and(self.v > q1, self.v < q2)].mean().item()

    def get_batch(self, batch_size):
        if self.count + batch_size > self.xyz.shape[0]:
            self.count = 0
            self.epoch += 1
            idx = torch.randperm(self.xyz.shape[0])
            self.xyz, self.v, self.slice_idx = self.xyz[idx], self.v[idx], self.slice_idx[idx]
        s = slice(self.count, self.count + batch_size)
        self.count += batch_size
        return self.xyz[s], self.v[s], self.slice_idx[s]


def init_params_like_reference(n_slices, bounding_box, args, axisangle_init):
    """Parameter dict initialised by the same torch initialisers, in the same
    order, as NeSVoR.build_network (models.py:221-258) -> same RNG stream."""
    base, L = nm.grid_config(bounding_box, args)
    levels = hashgrid.make_levels(L, args.log2_hashmap_size, base, args.level_scale)
    Fe = args.n_features_per_level
    P = {}

    def seq(prefix, shapes):
        for i, (a, b) in enumerate(shapes):
            lin = nn.Linear(a, b)
            P[f"{prefix}.{2 * i}.weight"] = lin.weight.detach().clone()
            P[f"{prefix}.{2 * i}.bias"] = lin.bias.detach().clone()

    if args.n_features_slice:
        P["slice_embedding.weight"] = nn.Embedding(n_slices, args.n_features_slice).weight.detach().clone()
    if not args.no_slice_scale:
        P["logit_coef"] = torch.zeros(n_slices)
    if not args.no_slice_variance:
        P["log_var_slice"] = torch.zeros(n_slices)
    g = torch.Generator().manual_seed(1337)  # the encoding owns its generator (tinycudann seed)
    P["inr.encoding.params"] = (torch.rand(hashgrid.n_params(levels, Fe), generator=g) * 2 - 1) * 1e-4
    seq("inr.density_net", nm.mlp_shapes(L * Fe, 1 + args.n_features_z, args.width, args.depth))
    if not args.no_pixel_variance:
        seq("sigma_net", nm.mlp_shapes(args.n_features_slice + args.n_features_z, 1, args.width, args.depth))
    if args.n_levels_bias:
        seq("b_net", nm.mlp_shapes(args.n_levels_bias * Fe + args.n_features_slice, 1, args.width, args.depth))
    P["axisangle"] = axisangle_init.clone()
    return P, levels


def train(ds: ArrayDataset, args: Namespace, n_iter=None, log=None, time_from_iter=0):
    """Returns (P, levels, bounding_box, info).  info['iters_per_s'] is wall-clock
    over iterations > time_from_iter."""
    n_iter = args.n_iter if n_iter is None else n_iter
    bb = ds.bounding_box
    v_mean = ds.mean
    n = ds.resolution.shape[0]
    ax_init = nm.tc.mat2axisangle_forward(ds.transformation_mat)
    psf_sigma = nm.resolution2sigma(ds.resolution)
    P, levels = init_params_like_reference(n, bb, args, ax_init)
    trainable = [k for k in P if not (k == "axisangle" and args.no_transformation_optimization)]
    for k in trainable:
        P[k].requires_grad_(True)
    # same grouping rule as train.py:136-141 ("_net" in name); both groups get wd 1e-2
    net = [P[k] for k in trainable if "_net" in k]
    enc = [P[k] for k in trainable if "_net" not in k]
    opt = torch.optim.AdamW(
        [{"params": enc}, {"params": net, "weight_decay": 1e-2}], lr=args.learning_rate, betas=(0.9, 0.99), eps=1e-15
    )
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=list(range(1, len(args.milestones) + 1)), gamma=args.gamma)
    decay = [int(m * args.n_iter) for m in args.milestones]
    delta = args.delta * v_mean
    hist = []
    t0 = None
    for i in range(1, n_iter + 1):
        if i == time_from_iter + 1:
            t0 = time.time()
        xyz, v, idx = ds.get_batch(args.batch_size)
        noise = torch.randn(xyz.shape[0], args.n_samples, 3, dtype=xyz.dtype)
        losses = nm.nesvor_forward(P, levels, args, bb, psf_sigma, ax_init, delta, xyz, v, idx, noise)
        nm.total_loss(losses, args).backward()
        opt.step()
        opt.zero_grad()
        hist.append({k: float(val.detach()) for k, val in losses.items()})
        if log is not None:
            log(i, hist[-1])
        if (decay and i >= decay[0]) or i == args.n_iter:
            if i < args.n_iter:
                decay.pop(0)
                sched.step()
    dt = time.time() - t0 if t0 is not None else float("nan")
    info = {"iters_per_s": (n_iter - time_from_iter) / dt if dt > 0 else float("nan"), "history": hist,
            "psf_sigma": psf_sigma, "axisangle_init": ax_init, "delta": delta}
    return {k: v.detach() for k, v in P.items()}, levels, bb, info


def sample_points(P, levels, args, bb, xyz, chunk=None):
    """sample.py:17-33: PSF-averaged density at world points."""
    out = torch.empty(xyz.shape[0], dtype=torch.float32)
    chunk = chunk or args.inference_batch_size
    sigma = args.output_resolution * nm.GAUSSIAN_FWHM
    S = 0 if args.no_output_psf else args.n_inference_samples
    with torch.no_grad():
        for i in range(0, xyz.shape[0], chunk):
            pts = xyz[i : i + chunk]
            noise = torch.randn(pts.shape[0], S, 3, dtype=pts.dtype) if S > 1 else None
            out[i : i + chunk] = nm.sample_points(P, levels, args, bb, pts, noise, sigma)
    return out
