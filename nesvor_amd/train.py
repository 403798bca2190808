"""Training data set and loop: host-side mirror of ``nesvor.nesvor.train``
(nesvor/nesvor/train.py).  ``train(slices, args) -> (INR, List[Slice], Volume)``.

Differences from the reference that do not change results:
* losses are averaged on the device; the per-iteration ``.item()`` syncs of
  train.py:199-200 are gone (one sync when a log line is produced);
* ``args.device`` may be any HIP device (the reference hard-codes device 0);
* with ``torch.distributed`` initialised the batch is sharded over ranks and
  gradients are averaged over RCCL (see ``nesvor_amd.ddp``).
"""
import logging
import time
from argparse import Namespace
from typing import Dict, List, Tuple

import torch

from .image import Slice, Volume
from .models import B_REG, D_LOSS, I_REG, INR, S_LOSS, T_REG, NeSVoR
from .transform import RigidTransform, transform_points
from .utils import MovingAverage, gaussian_blur


class Dataset(object):
    """All masked slice pixels flattened into SoA device arrays (train.py:14-120)."""

    def __init__(self, slices: List[Slice], args: Namespace) -> None:
        self.mask_threshold = args.mask_threshold
        xyz, v, idx, tfm, res = [], [], [], [], []
        for i, s in enumerate(slices):
            vi = s.v_masked
            xyz.append(s.xyz_masked_untransformed)
            v.append(vi)
            idx.append(torch.full(vi.shape, i, device=vi.device))
            tfm.append(s.transformation)
            res.append(s.resolution_xyz)
        self.xyz = torch.cat(xyz)
        self.v = torch.cat(v)
        self.slice_idx = torch.cat(idx)
        self.transformation = RigidTransform.cat(tfm)
        self.resolution = torch.stack(res, 0)
        # count == len  =>  the first get_batch() already shuffles (train.py:40,61-63)
        self.count = self.v.shape[0]
        self.epoch = 0

    @property
    def xyz_transformed(self) -> torch.Tensor:
        return transform_points(self.transformation[self.slice_idx], self.xyz)

    @property
    def bounding_box(self) -> torch.Tensor:
        margin = 2 * self.resolution.max()
        pts = self.xyz_transformed
        return torch.stack([pts.amin(0) - margin, pts.amax(0) + margin], 0)

    @property
    def mean(self) -> float:
        cap = 256 * 256 * 256
        sample = self.v if self.v.numel() < cap else self.v[:cap]
        q = torch.tensor([0.1, 0.9], dtype=self.v.dtype, device=self.v.device)
        q1, q2 = torch.quantile(sample, q)
        return self.v[torch.logical_and(self.v > q1, self.v < q2)].mean().item()

    def get_batch(self, batch_size: int, device, generator=None, host_rng: bool = False) -> Dict[str, torch.Tensor]:
        if self.count + batch_size > self.xyz.shape[0]:  # epoch boundary: reshuffle, drop the partial batch
            self.count = 0
            self.epoch += 1
            if host_rng:  # the permutation a CPU run of the reference draws (train.py:64), then moved to the device
                perm = torch.randperm(self.xyz.shape[0], generator=generator).to(device)
            else:
                perm = torch.randperm(self.xyz.shape[0], device=device, generator=generator)
            self.xyz, self.v, self.slice_idx = self.xyz[perm], self.v[perm], self.slice_idx[perm]
        sl = slice(self.count, self.count + batch_size)
        self.count += batch_size
        return {"xyz": self.xyz[sl], "v": self.v[sl], "slice_idx": self.slice_idx[sl]}

    @property
    def mask(self) -> Volume:
        """Occupancy of the transformed pixels -> blurred -> thresholded volume mask (train.py:81-120)."""
        with torch.no_grad():
            r_min, r_max = self.resolution.min(), self.resolution.max()
            xyz = self.xyz_transformed
            lo = xyz.amin(0) - r_max * 10
            hi = xyz.amax(0) + r_max * 10
            shape_xyz = ((hi - lo) / r_min).ceil().long()
            shape = (int(shape_xyz[2]), int(shape_xyz[1]), int(shape_xyz[0]))
            kji = ((xyz - lo) / r_min).round().long()
            lin = kji[..., 0] + shape[2] * kji[..., 1] + shape[2] * shape[1] * kji[..., 2]
            occ = torch.bincount(lin, minlength=shape[0] * shape[1] * shape[2]).view((1, 1) + shape).float()
            thr = self.mask_threshold * r_min**3 / self.resolution.log().mean().exp() ** 3
            thr = thr * (occ.sum() / (occ > 0).sum())
            mask = (gaussian_blur(occ, (r_max / r_min).item(), 3) > thr)[0, 0]
            centre = lo + (shape_xyz - 1) / 2 * r_min
            return Volume(mask.float(), mask, RigidTransform(torch.cat([0 * centre, centre])[None], True),
                          r_min, r_min, r_min)


def loss_weights(args: Namespace) -> Dict[str, float]:
    return {D_LOSS: 1, S_LOSS: 1, T_REG: args.weight_transformation, B_REG: args.weight_bias, I_REG: args.weight_image}


def build_optimizer(model: NeSVoR, args: Namespace):
    """AdamW with two groups; both end up with weight decay 1e-2 (train.py:134-152)."""
    net, enc = [], []
    for name, p in model.named_parameters():
        if p.numel() > 0:
            (net if "_net" in name else enc).append(p)
    opt = torch.optim.AdamW(
        params=[{"name": "encoding", "params": enc}, {"name": "net", "params": net, "weight_decay": 1e-2}],
        lr=args.learning_rate, betas=(0.9, 0.99), eps=1e-15,
    )
    sched = torch.optim.lr_scheduler.MultiStepLR(
        optimizer=opt, milestones=list(range(1, len(args.milestones) + 1)), gamma=args.gamma
    )
    return opt, sched


def train(slices: List[Slice], args: Namespace, on_iteration=None) -> Tuple[INR, List[Slice], Volume]:
    """``on_iteration(i, losses)`` (optional, not in the reference) sees every iteration's loss dict (0-d device tensors).
    ``args.host_rng = True`` draws the batch permutation and the PSF noise from the HOST generator in the order a CPU run
    of the reference does (train.py:64, models.py:270) and uploads them: with the same ``torch.manual_seed`` the run then
    replays the reference's trajectory (tests/test_gpu_model.py holds it to the reference's own 20-iteration fixture)."""
    dataset = Dataset(slices, args)
    model = NeSVoR(dataset.transformation, dataset.resolution, dataset.mean, dataset.bounding_box, args)
    from . import direct

    # fused trainer: the single-precision model, and the half-precision model structure when the autograd-free step
    # covers it (bf16 matrix operands, fp32 accumulation: no GradScaler); otherwise the module path below
    use_fused = getattr(args, "fused", True) and (args.dtype == torch.float32 or direct.supported(model))
    # data parallel (one process per GPU): args.batch_size is the GLOBAL batch, every rank draws the same
    # permutation (seed the global RNG identically before calling train) and takes its slice of each batch
    import torch.distributed as dist

    from . import ddp

    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    parallel = ddp.active()  # world > 1, or a forced single-rank group (ddp.forced)
    if parallel and not use_fused:
        raise RuntimeError("data-parallel training needs the fused trainer (flat gradient buffer)")
    if use_fused:
        from .fused import FusedTrainer

        trainer = FusedTrainer(model, args, world_size=world, distributed=parallel)
        if parallel:
            ddp.broadcast_params_(trainer.flat.param)
            trainer.reduce_hook = ddp.make_reduce_hook()
        # nothing but the losses is read between iterations (a callback might read the model: then every step joins)
        trainer.defer_table_join = on_iteration is None
    perm_gen = None
    if parallel:
        # identical batch permutations on every rank from a dedicated generator; the global stream (PSF
        # noise) becomes rank-specific
        perm_gen = torch.Generator(device=args.device)
        perm_gen.manual_seed(torch.initial_seed())
        torch.manual_seed(torch.initial_seed() + 7919 * (rank + 1))
    if not use_fused:
        optimizer, scheduler = build_optimizer(model, args)
    decay_milestones = [int(m * args.n_iter) for m in args.milestones]
    weights = loss_weights(args)
    model.train()
    average = MovingAverage(1 - 0.001)
    logging.info("NeSVoR training starts.")
    t0 = time.time()
    host_rng = bool(getattr(args, "host_rng", False))
    if host_rng and world > 1:
        raise RuntimeError("args.host_rng replays a single-process CPU run; it is not defined under data parallelism")
    for i in range(1, args.n_iter + 1):
        batch = ddp.shard_batch(dataset.get_batch(args.batch_size, args.device, perm_gen, host_rng), rank, world)
        noise = None
        if host_rng:
            noise = torch.randn(batch["xyz"].shape[0], args.n_samples, 3, dtype=batch["xyz"].dtype).to(args.device)
        if use_fused:
            losses = trainer.step(**batch, noise=noise)
        else:
            losses = model(**batch) if noise is None else model.forward_with_noise(batch["xyz"], batch["v"], batch["slice_idx"], noise)
            loss = 0
            for k in losses:
                if k in weights and weights[k]:
                    loss = loss + weights[k] * losses[k]
            loss.backward()
            optimizer.step()
            optimizer.zero_grad()
        average.update_all(losses)  # stays on the device: no per-iteration sync, two small launches
        if on_iteration is not None:
            on_iteration(i, losses)
        if (decay_milestones and i >= decay_milestones[0]) or i == args.n_iter:
            logging.info(
                "time %.1fs epoch %d iter %d %s", time.time() - t0, dataset.epoch, i,
                " ".join("%s=%.3e" % (k, float(average[k])) for k in losses),
            )
            if i < args.n_iter:
                decay_milestones.pop(0)
                if use_fused:
                    trainer.decay_lr(args.gamma)
                else:
                    scheduler.step()
    if use_fused:
        trainer.finish()
    transformation = model.transformation
    dataset.transformation = transformation
    mask = dataset.mask
    output_slices = []
    for i, s in enumerate(slices):
        out = s.clone()
        out.transformation = transformation[i]
        output_slices.append(out)
    return model.inr, output_slices, mask
