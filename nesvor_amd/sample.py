"""Inference from a trained INR: host-side mirror of ``nesvor.nesvor.sample``
(nesvor/nesvor/sample.py:10-64)."""
from argparse import Namespace
from typing import List

import torch

from .image import Slice, Volume
from .models import INR
from .transform import transform_points
from .utils import meshgrid, resolution2sigma


def sample_volume(model: INR, mask: Volume, args: Namespace) -> Volume:
    model.eval()
    img = mask.resample(args.output_resolution, None)
    img.image[img.mask] = sample_points(model, img.xyz_masked, args)
    return img


def sample_points(model: INR, xyz: torch.Tensor, args: Namespace) -> torch.Tensor:
    """PSF-averaged density at world points, in chunks of args.inference_batch_size."""
    shape = xyz.shape[:-1]
    xyz = xyz.view(-1, 3)
    out = torch.empty(xyz.shape[0], dtype=torch.float32, device=args.device)
    sigma = resolution2sigma(args.output_resolution, isotropic=True)
    n_samples = 0 if args.no_output_psf else args.n_inference_samples
    step = args.inference_batch_size
    with torch.no_grad():
        for i in range(0, xyz.shape[0], step):
            pts = model.sample_batch(xyz[i : i + step], None, sigma, n_samples)
            out[i : i + step] = model(pts, False).mean(-1)
    return out.view(shape)


def sample_slice(model: INR, slice: Slice, mask: Volume, args: Namespace) -> Slice:
    out = slice.clone()
    out.image = torch.zeros_like(out.image)
    out.mask = torch.zeros_like(out.mask)
    xyz = meshgrid(out.shape_xyz, out.resolution_xyz).view(-1, 3)
    inside = mask.sample_points(transform_points(out.transformation, xyz)) > 0
    if inside.any():
        pts = model.sample_batch(
            xyz[inside], out.transformation, resolution2sigma(out.resolution_xyz, isotropic=False),
            0 if args.no_output_psf else args.n_inference_samples,
        )
        v = model(pts, False).mean(-1)
        out.mask = inside.view(out.mask.shape)
        out.image[out.mask] = v.to(out.image.dtype)
    return out


def sample_slices(model: INR, slices: List[Slice], mask: Volume, args: Namespace) -> List[Slice]:
    model.eval()
    with torch.no_grad():
        return [sample_slice(model, s, mask, args) for s in slices]
