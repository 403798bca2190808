"""Autograd wrappers over the slice-acquisition native module.  Mirrors
``nesvor.slice_acquisition`` (slice_acquisition/slice_acq.py:22-211).
A (forward + backward) and A^T (forward) are built; the backward of A^T is §8(f) "next".
"""
import torch
from torch.autograd import Function

from . import slice_acq_cuda as _backend


class SliceAcqFunction(Function):
    @staticmethod
    def forward(ctx, transforms, vol, vol_mask, slices_mask, psf, slice_shape, res_slice, need_weight, interp_psf):
        if vol_mask is None:
            vol_mask = torch.empty(0, device=vol.device)
        if slices_mask is None:
            slices_mask = torch.empty(0, device=vol.device)
        outputs = _backend.forward(
            transforms.contiguous(), vol.contiguous(), vol_mask, slices_mask, psf.contiguous(),
            slice_shape, res_slice, need_weight, interp_psf,
        )
        ctx.save_for_backward(transforms, vol, vol_mask, slices_mask, psf)
        ctx.interp_psf, ctx.res_slice, ctx.need_weight = interp_psf, res_slice, need_weight
        return (outputs[0], outputs[1]) if need_weight else outputs[0]

    @staticmethod
    def backward(ctx, *grads):
        transforms, vol, vol_mask, slices_mask, psf = ctx.saved_tensors
        outputs = _backend.backward(
            transforms, vol, vol_mask, psf, grads[0].contiguous(), slices_mask, ctx.res_slice, ctx.interp_psf,
            ctx.needs_input_grad[1], ctx.needs_input_grad[0],
        )
        grad_vol, grad_transforms = outputs
        return grad_transforms, grad_vol, None, None, None, None, None, None, None


def slice_acquisition(transforms, vol, vol_mask, slices_mask, psf, slice_shape, res_slice, need_weight, interp_psf):
    return SliceAcqFunction.apply(
        transforms, vol, vol_mask, slices_mask, psf, slice_shape, res_slice, need_weight, interp_psf
    )


class SliceAcqAdjointFunction(Function):
    """slice_acq.py:86-163"""

    @staticmethod
    def forward(ctx, transforms, psf, slices, slices_mask, vol_mask, vol_shape, res_slice, interp_psf, equalize):
        if vol_mask is None:
            vol_mask = torch.empty(0, device=slices.device)
        if slices_mask is None:
            slices_mask = torch.empty(0, device=slices.device)
        vol, vol_weight = _backend.adjoint_forward(
            transforms.contiguous(), psf.contiguous(), slices.contiguous(), slices_mask, vol_mask, vol_shape, res_slice,
            interp_psf, equalize)
        ctx.set_materialize_grads(False)
        return vol

    @staticmethod
    def backward(ctx, grad_vol):
        if grad_vol is None:
            return (None,) * 9
        _backend.adjoint_backward()  # raises: SURVEY.md §8(f)


def slice_acquisition_adjoint(transforms, psf, slices, slices_mask, vol_mask, vol_shape, res_slice, interp_psf, equalize):
    return SliceAcqAdjointFunction.apply(
        transforms, psf, slices, slices_mask, vol_mask, vol_shape, res_slice, interp_psf, equalize)
