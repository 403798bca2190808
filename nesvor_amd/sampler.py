"""PSF sampling + rigid transform op (autograd Function over the HIP sampler kernels)."""
import torch
from torch.autograd import Function

from . import _lib


def forward_raw(mat, slice_idx, xyz, psf_sigma, noise, bb):
    """One launch, no autograd: -> x (B,S,3), u (B*S,3).  All inputs contiguous device tensors."""
    _lib.require_device(mat, xyz, psf_sigma, noise, bb, dtype=torch.float32, name="psf_transform input")
    _lib.require_device(slice_idx, dtype=torch.int64, name="slice_idx")
    B, S = noise.shape[0], noise.shape[1]
    x = torch.empty((B, S, 3), dtype=torch.float32, device=noise.device)
    u = torch.empty((B * S, 3), dtype=torch.float32, device=noise.device)
    with torch.cuda.device(noise.device), _lib.kernel_timer.span("psf_transform_fwd"):
        err = _lib.load().nesvor_psf_transform_forward(
            _lib.ptr(mat), _lib.ptr(slice_idx), _lib.ptr(xyz), _lib.ptr(psf_sigma), _lib.ptr(noise), _lib.ptr(bb),
            _lib.ptr(x), _lib.ptr(u), B, S, _lib.stream_ptr())
    _lib.check(err, "psf_transform forward")
    return x, u


def backward_raw(mat, slice_idx, xyz, psf_sigma, noise, bb, dx, du):
    """-> dpix (B,3,4): gradient w.r.t. each PIXEL's slice matrix (the caller index-adds pixels into slices)."""
    B, S = noise.shape[0], noise.shape[1]
    dpix = torch.empty((B, 3, 4), dtype=torch.float32, device=noise.device)
    with torch.cuda.device(noise.device), _lib.kernel_timer.span("psf_transform_bwd"):
        err = _lib.load().nesvor_psf_transform_backward(
            _lib.ptr(mat), _lib.ptr(slice_idx), _lib.ptr(xyz), _lib.ptr(psf_sigma), _lib.ptr(noise), _lib.ptr(bb),
            _lib.ptr(dx), _lib.ptr(du), _lib.ptr(dpix), B, S, _lib.stream_ptr())
    _lib.check(err, "psf_transform backward")
    return dpix


class PsfTransformFunction(Function):
    """(mat (n,3,4), slice_idx, xyz, psf_sigma (n,3), noise (B,S,3), bounding_box (2,3)) -> x (B,S,3), u (B*S,3)."""

    @staticmethod
    def forward(ctx, mat, slice_idx, xyz, psf_sigma, noise, bounding_box):
        mat_c, xyz, psf_sigma, noise, bb = (t.contiguous() for t in (mat, xyz, psf_sigma, noise, bounding_box))
        slice_idx = slice_idx.contiguous()
        x, u = forward_raw(mat_c, slice_idx, xyz, psf_sigma, noise, bb)
        ctx.save_for_backward(mat_c, slice_idx, xyz, psf_sigma, noise, bb)
        return x, u

    @staticmethod
    def backward(ctx, dx, du):
        mat, slice_idx, xyz, psf_sigma, noise, bb = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None, None
        dx = None if dx is None else dx.contiguous()
        du = None if du is None else du.contiguous()
        dpix = backward_raw(mat, slice_idx, xyz, psf_sigma, noise, bb, dx, du)
        dmat = torch.zeros_like(mat).index_add_(0, slice_idx, dpix)
        return dmat, None, None, None, None, None


def psf_transform(mat, slice_idx, xyz, psf_sigma, noise, bounding_box):
    return PsfTransformFunction.apply(mat, slice_idx, xyz, psf_sigma, noise, bounding_box)
