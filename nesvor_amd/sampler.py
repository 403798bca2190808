"""PSF sampling + rigid transform: raw launches of the HIP sampler kernels and the differentiable ``psf_transform``."""
import torch

from . import _lib


def _seed64(seed: int) -> int:
    return int(seed) & 0xFFFFFFFFFFFFFFFF


def forward_raw(mat, slice_idx, xyz, psf_sigma, noise, bb, rng=None, n_samples=None, need_x=True):
    """One launch, no autograd: -> x (B,S,3) | None, u (B*S,3).  All inputs contiguous device tensors.
    ``noise`` (B,S,3): explicit N(0,1) draws; or ``noise=None`` with ``rng = (seed, offset)`` and ``n_samples``: the kernel
    draws them itself (counter-based generator, csrc/sampler.hip) - pass the same pair to ``backward_raw``."""
    _lib.require_device(mat, xyz, psf_sigma, bb, dtype=torch.float32, name="psf_transform input")
    _lib.require_device(slice_idx, dtype=torch.int64, name="slice_idx")
    if noise is not None:
        _lib.require_device(noise, dtype=torch.float32, name="psf_transform noise")
        B, S = noise.shape[0], noise.shape[1]
    else:
        B, S = xyz.shape[0], int(n_samples)
    dev = xyz.device
    x = torch.empty((B, S, 3), dtype=torch.float32, device=dev) if need_x else None
    u = torch.empty((B * S, 3), dtype=torch.float32, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev), _lib.kernel_timer.span("psf_transform_fwd"):
        if noise is not None:
            err = lib.nesvor_psf_transform_forward(
                _lib.ptr(mat), _lib.ptr(slice_idx), _lib.ptr(xyz), _lib.ptr(psf_sigma), _lib.ptr(noise), _lib.ptr(bb),
                _lib.ptr(x), _lib.ptr(u), B, S, _lib.stream_ptr())
        else:
            err = lib.nesvor_psf_transform_forward_rng(
                _lib.ptr(mat), _lib.ptr(slice_idx), _lib.ptr(xyz), _lib.ptr(psf_sigma), _seed64(rng[0]), _seed64(rng[1]),
                _lib.ptr(bb), _lib.ptr(x), _lib.ptr(u), B, S, _lib.stream_ptr())
    _lib.check(err, "psf_transform forward")
    return x, u


def backward_raw(mat, slice_idx, xyz, psf_sigma, noise, bb, dx, du, rng=None, n_samples=None):
    """-> dpix (B,3,4): gradient w.r.t. each PIXEL's slice matrix (the caller index-adds pixels into slices)."""
    if noise is not None:
        B, S = noise.shape[0], noise.shape[1]
    else:
        B, S = xyz.shape[0], int(n_samples)
    dev = xyz.device
    dpix = torch.empty((B, 3, 4), dtype=torch.float32, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev), _lib.kernel_timer.span("psf_transform_bwd"):
        if noise is not None:
            err = lib.nesvor_psf_transform_backward(
                _lib.ptr(mat), _lib.ptr(slice_idx), _lib.ptr(xyz), _lib.ptr(psf_sigma), _lib.ptr(noise), _lib.ptr(bb),
                _lib.ptr(dx), _lib.ptr(du), _lib.ptr(dpix), B, S, _lib.stream_ptr())
        else:
            err = lib.nesvor_psf_transform_backward_rng(
                _lib.ptr(mat), _lib.ptr(slice_idx), _lib.ptr(xyz), _lib.ptr(psf_sigma), _seed64(rng[0]), _seed64(rng[1]),
                _lib.ptr(bb), _lib.ptr(dx), _lib.ptr(du), _lib.ptr(dpix), B, S, _lib.stream_ptr())
    _lib.check(err, "psf_transform backward")
    return dpix


def psf_noise(seed: int, offset: int, n_samples: int, device) -> torch.Tensor:
    """The (n_samples, 3) N(0,1) draws the kernels generate for ``rng = (seed, offset)`` (tests, debugging)."""
    out = torch.empty((n_samples, 3), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        err = _lib.load().nesvor_psf_noise(_seed64(seed), _seed64(offset), _lib.ptr(out), n_samples, _lib.stream_ptr())
    _lib.check(err, "psf_noise")
    return out


def psf_transform(mat, slice_idx, xyz, psf_sigma, noise, bounding_box):
    """(mat (n,3,4), slice_idx, xyz, psf_sigma (n,3), noise (B,S,3), bounding_box (2,3)) -> x (B,S,3), u (B*S,3);
    differentiable in ``mat``: the dispatcher op ``torch.ops.nesvor.psf_transform`` (``nesvor_amd.ops``)."""
    return torch.ops.nesvor.psf_transform(mat.contiguous(), slice_idx.contiguous(), xyz.contiguous(), psf_sigma.contiguous(),
                                          noise.contiguous(), bounding_box.contiguous())
