"""Drop-in for the reference's pybind module ``nesvor.slice_acq_cuda``
(nesvor/slice_acquisition/slice_acq_cuda.cpp:156-161).

``forward`` is implemented (gfx950 HIP kernel).  ``backward``,
``adjoint_forward`` and ``adjoint_backward`` are SURVEY.md §8(f) rank-1 "next"
rows: they raise until built, they never fall back.
"""
import torch

from . import _lib


def forward(transforms, vol, vol_mask, slices_mask, psf, slice_shape, res_slice, need_weight, interp_psf):
    """-> [slices (n,1,h,w)] or [slices, weight]; "None" masks are passed as empty tensors
    (numel()==0), as the reference's wrapper does (slice_acq.py:36-39)."""
    _lib.require_device(transforms, vol, psf, dtype=torch.float32, name="transforms/vol/psf")
    vm = vol_mask if (vol_mask is not None and vol_mask.numel() > 0) else None
    sm = slices_mask if (slices_mask is not None and slices_mask.numel() > 0) else None
    for m in (vm, sm):
        if m is not None:
            _lib.require_device(m, dtype=torch.bool, name="mask")
    n = transforms.shape[0]
    h, w = int(slice_shape[0]), int(slice_shape[1])
    D, H, W = (int(s) for s in vol.shape[-3:])
    d_p, h_p, w_p = (int(s) for s in psf.shape)
    slices = torch.zeros((n, 1, h, w), dtype=vol.dtype, device=vol.device)
    weight = torch.zeros((n, 1, h, w), dtype=vol.dtype, device=vol.device) if need_weight else None
    with torch.cuda.device(vol.device):
        err = _lib.load().nesvor_slice_acq_forward(
            _lib.ptr(transforms), _lib.ptr(vol), _lib.ptr(vm), _lib.ptr(sm), _lib.ptr(psf), _lib.ptr(slices),
            _lib.ptr(weight), D, H, W, d_p, h_p, w_p, n, h, w, float(res_slice), int(bool(interp_psf)),
            _lib.stream_ptr(),
        )
    _lib.check(err, "slice_acq forward")
    return [slices, weight] if need_weight else [slices]


def backward(*args, **kwargs):
    raise NotImplementedError("slice_acq backward: SURVEY.md §8(f) rank 1, not built yet (no fallback)")


def adjoint_forward(*args, **kwargs):
    raise NotImplementedError("slice_acq adjoint_forward: SURVEY.md §8(f) rank 1, not built yet (no fallback)")


def adjoint_backward(*args, **kwargs):
    raise NotImplementedError("slice_acq adjoint_backward: SURVEY.md §8(f) rank 1, not built yet (no fallback)")
