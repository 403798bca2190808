"""Drop-in for the reference's pybind module ``nesvor.slice_acq_cuda``
(nesvor/slice_acquisition/slice_acq_cuda.cpp:156-161).

``forward``, ``backward`` and ``adjoint_forward`` run gfx950 HIP kernels (the latter two in the
default linear-interpolation mode) and so does ``adjoint_backward`` (only reached from SVoRT training in
the reference).  ``interp_psf=True`` is never used by the reference's own callers: it raises, nothing
ever falls back.
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


def _mask(m):
    m = m if (m is not None and m.numel() > 0) else None
    if m is not None:
        _lib.require_device(m, dtype=torch.bool, name="mask")
    return m


def backward(transforms, vol, vol_mask, psf, grad_slices, slices_mask, res_slice, interp_psf, need_vol_grad,
             need_transforms_grad):
    """-> [grad_vol | None, grad_transforms | None] (slice_acq_cuda_kernel.cu:173-470, host :993-1027)."""
    if interp_psf:
        raise NotImplementedError("slice_acq backward: interp_psf=True is not built (no fallback)")
    _lib.require_device(transforms, vol, psf, grad_slices, dtype=torch.float32, name="slice_acq backward input")
    vm, sm = _mask(vol_mask), _mask(slices_mask)
    n, h, w = grad_slices.shape[0], grad_slices.shape[-2], grad_slices.shape[-1]
    D, H, W = (int(s) for s in vol.shape[-3:])
    d_p, h_p, w_p = (int(s) for s in psf.shape)
    grad_vol = torch.empty_like(vol) if need_vol_grad else None
    grad_tf = torch.empty_like(transforms) if need_transforms_grad else None
    scratch = torch.empty(n * h * w, dtype=torch.float32, device=vol.device)
    with torch.cuda.device(vol.device):
        err = _lib.load().nesvor_slice_acq_backward(
            _lib.ptr(transforms), _lib.ptr(vol), _lib.ptr(vm), _lib.ptr(psf), _lib.ptr(grad_slices), _lib.ptr(sm),
            _lib.ptr(grad_vol), _lib.ptr(grad_tf), _lib.ptr(scratch), D, H, W, d_p, h_p, w_p, n, h, w, float(res_slice),
            _lib.stream_ptr())
    _lib.check(err, "slice_acq backward")
    return [grad_vol, grad_tf]


def adjoint_forward(transforms, psf, slices, slices_mask, vol_mask, vol_shape, res_slice, interp_psf, equalize):
    """A^T -> [vol (1,1,D,H,W), vol_weight (same shape, or an empty tensor when not equalising)]
    (slice_acq_cuda_kernel.cu:472-693, host :1029-1077)."""
    if interp_psf:
        raise NotImplementedError("slice_acq adjoint_forward: interp_psf=True is not built (no fallback)")
    _lib.require_device(transforms, psf, slices, dtype=torch.float32, name="slice_acq adjoint input")
    vm, sm = _mask(vol_mask), _mask(slices_mask)
    n, h, w = slices.shape[0], slices.shape[-2], slices.shape[-1]
    D, H, W = (int(s) for s in vol_shape)
    d_p, h_p, w_p = (int(s) for s in psf.shape)
    vol = torch.empty((1, 1, D, H, W), dtype=slices.dtype, device=slices.device)
    vol_weight = torch.empty_like(vol) if equalize else None
    scratch = torch.empty(2 * n * h * w, dtype=torch.float32, device=slices.device)
    with torch.cuda.device(slices.device):
        err = _lib.load().nesvor_slice_acq_adjoint_forward(
            _lib.ptr(transforms), _lib.ptr(psf), _lib.ptr(slices), _lib.ptr(sm), _lib.ptr(vm), _lib.ptr(vol),
            _lib.ptr(vol_weight), _lib.ptr(scratch), D, H, W, d_p, h_p, w_p, n, h, w, float(res_slice), int(bool(equalize)),
            _lib.stream_ptr())
    _lib.check(err, "slice_acq adjoint_forward")
    return [vol, vol_weight if equalize else torch.empty(0, device=slices.device)]


def adjoint_backward(transforms, grad_vol, vol_weight, vol_mask, psf, slices, slices_mask, vol, res_slice, interp_psf,
                     equalize, need_slices_grad, need_transforms_grad):
    """-> [grad_slices | None, grad_transforms | None] (slice_acq_cuda_kernel.cu:695-950, host :1079-1131).
    Like the reference, ``grad_vol`` is equalised IN PLACE when ``equalize`` is set."""
    if interp_psf:
        raise NotImplementedError("slice_acq adjoint_backward: interp_psf=True is not built (no fallback)")
    _lib.require_device(transforms, grad_vol, psf, slices, dtype=torch.float32, name="slice_acq adjoint_backward input")
    if equalize:
        _lib.require_device(vol_weight, vol, dtype=torch.float32, name="slice_acq adjoint_backward vol/vol_weight")
    vm, sm = _mask(vol_mask), _mask(slices_mask)
    n, h, w = slices.shape[0], slices.shape[-2], slices.shape[-1]
    D, H, W = (int(s) for s in grad_vol.shape[-3:])
    d_p, h_p, w_p = (int(s) for s in psf.shape)
    grad_slices = torch.zeros_like(slices) if need_slices_grad else None
    grad_tf = torch.empty_like(transforms) if need_transforms_grad else None
    with torch.cuda.device(slices.device):
        err = _lib.load().nesvor_slice_acq_adjoint_backward(
            _lib.ptr(transforms), _lib.ptr(grad_vol), _lib.ptr(vol_weight if equalize else None), _lib.ptr(vm), _lib.ptr(psf),
            _lib.ptr(slices), _lib.ptr(sm), _lib.ptr(vol if equalize else None), _lib.ptr(grad_slices), _lib.ptr(grad_tf),
            D, H, W, d_p, h_p, w_p, n, h, w, float(res_slice), int(bool(equalize)), _lib.stream_ptr())
    _lib.check(err, "slice_acq adjoint_backward")
    return [grad_slices, grad_tf]
