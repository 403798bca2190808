"""Differentiable slice acquisition A and its adjoint A^T: the functions of ``nesvor.slice_acquisition``
(slice_acquisition/slice_acq.py:22-211) on the dispatcher ops ``torch.ops.nesvor.slice_acq_forward`` /
``slice_acq_adjoint_forward``, whose autograd formulas (``nesvor_amd.ops``) call the matching backward ops.
float32 / float64, both interpolation modes.
"""
import torch

from . import ops as _ops  # noqa: F401  (registers torch.ops.nesvor)


def _or_empty(mask, like):
    return mask if mask is not None else torch.empty(0, device=like.device)


def slice_acquisition(transforms, vol, vol_mask, slices_mask, psf, slice_shape, res_slice, need_weight, interp_psf):
    """(n,3,4) poses in voxel units, (1,1,D,H,W) volume, (d,h,w) PSF -> slices (n,1,h,w) [, PSF weight per pixel]."""
    out = torch.ops.nesvor.slice_acq_forward(
        transforms.contiguous(), vol.contiguous(), _or_empty(vol_mask, vol), _or_empty(slices_mask, vol), psf.contiguous(),
        [int(s) for s in slice_shape], float(res_slice), bool(need_weight), bool(interp_psf))
    return (out[0], out[1]) if need_weight else out[0]


def slice_acquisition_adjoint(transforms, psf, slices, slices_mask, vol_mask, vol_shape, res_slice, interp_psf, equalize):
    """Back-projection of slices into a (1,1,D,H,W) volume; ``equalize`` divides by the back-projected PSF weight."""
    vol, _ = torch.ops.nesvor.slice_acq_adjoint_forward(
        transforms.contiguous(), psf.contiguous(), slices.contiguous(), _or_empty(slices_mask, slices), _or_empty(vol_mask, slices),
        [int(s) for s in vol_shape], float(res_slice), bool(interp_psf), bool(equalize))
    return vol
