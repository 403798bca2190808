"""Autograd wrappers over the slice-acquisition native module.  Mirrors
``nesvor.slice_acquisition`` (slice_acquisition/slice_acq.py:22-211).
A and A^T, forward and backward, in the default linear-interpolation mode.
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
        transforms, psf, slices = transforms.contiguous(), psf.contiguous(), slices.contiguous()
        vol, vol_weight = _backend.adjoint_forward(
            transforms, psf, slices, slices_mask, vol_mask, vol_shape, res_slice, interp_psf, equalize)
        if equalize:
            ctx.save_for_backward(transforms, psf, slices, slices_mask, vol_mask, vol, vol_weight)
        else:
            ctx.save_for_backward(transforms, psf, slices, slices_mask, vol_mask)
        ctx.res_slice, ctx.interp_psf, ctx.equalize = res_slice, interp_psf, equalize
        return vol

    @staticmethod
    def backward(ctx, grad_vol):
        if ctx.equalize:
            transforms, psf, slices, slices_mask, vol_mask, vol, vol_weight = ctx.saved_tensors
        else:
            transforms, psf, slices, slices_mask, vol_mask = ctx.saved_tensors
            vol = vol_weight = None
        # the native op equalises grad_vol in place (as the reference does): work on a private contiguous copy
        grad_vol = grad_vol.contiguous().clone() if ctx.equalize else grad_vol.contiguous()
        grad_slices, grad_transforms = _backend.adjoint_backward(
            transforms, grad_vol, vol_weight, vol_mask, psf, slices, slices_mask, vol, ctx.res_slice, ctx.interp_psf,
            ctx.equalize, ctx.needs_input_grad[2], ctx.needs_input_grad[0])
        return grad_transforms, None, grad_slices, None, None, None, None, None, None


def slice_acquisition_adjoint(transforms, psf, slices, slices_mask, vol_mask, vol_shape, res_slice, interp_psf, equalize):
    return SliceAcqAdjointFunction.apply(
        transforms, psf, slices, slices_mask, vol_mask, vol_shape, res_slice, interp_psf, equalize)
