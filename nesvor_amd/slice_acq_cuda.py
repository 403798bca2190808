"""Drop-in for the reference's pybind module ``nesvor.slice_acq_cuda``
(nesvor/slice_acquisition/slice_acq_cuda.cpp:156-161): ``forward``, ``backward``, ``adjoint_forward``,
``adjoint_backward`` with the reference's argument lists and list returns.

The functions are the dispatcher ops ``torch.ops.nesvor.slice_acq_*`` (``nesvor_amd.ops``).  "None" masks are passed as
empty tensors (slice_acq.py:36-39); a result that was not requested is ``None`` in the returned list where the
reference returns an undefined Tensor.  float32 and float64; both interpolation modes (``interp_psf=True`` runs the
``*_interp`` kernels of the backward / adjoint operators) - nothing ever falls back.
"""
import torch

from . import ops as _ops  # noqa: F401  (registers torch.ops.nesvor)


def _or_empty(mask, like):
    return mask if mask is not None else torch.empty(0, device=like.device)


def _or_none(t):
    return t if t.numel() > 0 else None


def forward(transforms, vol, vol_mask, slices_mask, psf, slice_shape, res_slice, need_weight, interp_psf):
    """-> [slices (n,1,h,w)] or [slices, weight]."""
    return list(torch.ops.nesvor.slice_acq_forward(
        transforms, vol, _or_empty(vol_mask, vol), _or_empty(slices_mask, vol), psf, [int(s) for s in slice_shape],
        float(res_slice), bool(need_weight), bool(interp_psf)))


def backward(transforms, vol, vol_mask, psf, grad_slices, slices_mask, res_slice, interp_psf, need_vol_grad,
             need_transforms_grad):
    """-> [grad_vol | None, grad_transforms | None] (slice_acq_cuda_kernel.cu:173-470, host :993-1027)."""
    gv, gt = torch.ops.nesvor.slice_acq_backward(
        transforms, vol, _or_empty(vol_mask, vol), psf, grad_slices, _or_empty(slices_mask, vol), float(res_slice),
        bool(interp_psf), bool(need_vol_grad), bool(need_transforms_grad))
    return [_or_none(gv), _or_none(gt)]


def adjoint_forward(transforms, psf, slices, slices_mask, vol_mask, vol_shape, res_slice, interp_psf, equalize):
    """A^T -> [vol (1,1,D,H,W), vol_weight (same shape, or an empty tensor when not equalising)]
    (slice_acq_cuda_kernel.cu:472-693, host :1029-1077)."""
    return list(torch.ops.nesvor.slice_acq_adjoint_forward(
        transforms, psf, slices, _or_empty(slices_mask, slices), _or_empty(vol_mask, slices), [int(s) for s in vol_shape],
        float(res_slice), bool(interp_psf), bool(equalize)))


def adjoint_backward(transforms, grad_vol, vol_weight, vol_mask, psf, slices, slices_mask, vol, res_slice, interp_psf,
                     equalize, need_slices_grad, need_transforms_grad):
    """-> [grad_slices | None, grad_transforms | None] (slice_acq_cuda_kernel.cu:695-950, host :1079-1131).
    Like the reference, ``grad_vol`` is equalised IN PLACE when ``equalize`` is set."""
    e = torch.empty(0, device=slices.device)
    gs, gt = torch.ops.nesvor.slice_acq_adjoint_backward(
        transforms, grad_vol, vol_weight if vol_weight is not None else e, _or_empty(vol_mask, slices), psf, slices,
        _or_empty(slices_mask, slices), vol if vol is not None else e, float(res_slice), bool(interp_psf), bool(equalize),
        bool(need_slices_grad), bool(need_transforms_grad))
    return [_or_none(gs), _or_none(gt)]
