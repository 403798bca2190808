"""PSF sampling + rigid transform: raw launches of the HIP sampler kernels and the differentiable ``psf_transform``."""
import torch

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


def psf_transform(mat, slice_idx, xyz, psf_sigma, noise, bounding_box):
    """(mat (n,3,4), slice_idx, xyz, psf_sigma (n,3), noise (B,S,3), bounding_box (2,3)) -> x (B,S,3), u (B*S,3);
    differentiable in ``mat``: the dispatcher op ``torch.ops.nesvor.psf_transform`` (``nesvor_amd.ops``)."""
    return torch.ops.nesvor.psf_transform(mat.contiguous(), slice_idx.contiguous(), xyz.contiguous(), psf_sigma.contiguous(),
                                          noise.contiguous(), bounding_box.contiguous())
