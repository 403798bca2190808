"""Drop-in for the reference's pybind module ``nesvor.transform_convert_cuda``
(nesvor/transform/transform_convert_cuda.cpp:64-69).

Same four entry points, same conventions: device tensors only, contiguous,
float or double, each call returns a *list of length 1* holding a newly
allocated output.  Backed by libnesvor_hip.so (gfx950 HIP kernels).
"""
import torch

from . import _lib


def _sfx(t):
    if t.dtype == torch.float32:
        return ""
    if t.dtype == torch.float64:
        return "_f64"
    raise RuntimeError(f"transform_convert: unsupported dtype {t.dtype}")


def axisangle2mat_forward(axisangle):
    _lib.require_device(axisangle, name="axisangle")
    n = axisangle.shape[0]
    mat = torch.zeros((n, 3, 4), dtype=axisangle.dtype, device=axisangle.device)
    fn = getattr(_lib.load(), "nesvor_axisangle2mat_forward" + _sfx(axisangle))
    with torch.cuda.device(axisangle.device):
        _lib.check(fn(_lib.ptr(axisangle), _lib.ptr(mat), n, _lib.stream_ptr()), "axisangle2mat_forward")
    return [mat]


def axisangle2mat_backward(grad_mat, axisangle):
    _lib.require_device(grad_mat, axisangle, name="grad_mat/axisangle")
    n = axisangle.shape[0]
    grad_ax = torch.zeros((n, 6), dtype=axisangle.dtype, device=axisangle.device)
    fn = getattr(_lib.load(), "nesvor_axisangle2mat_backward" + _sfx(axisangle))
    with torch.cuda.device(axisangle.device):
        _lib.check(fn(_lib.ptr(grad_mat), _lib.ptr(axisangle), _lib.ptr(grad_ax), n, _lib.stream_ptr()), "axisangle2mat_backward")
    return [grad_ax]


def mat2axisangle_forward(mat):
    _lib.require_device(mat, name="mat")
    n = mat.shape[0]
    ax = torch.zeros((n, 6), dtype=mat.dtype, device=mat.device)
    fn = getattr(_lib.load(), "nesvor_mat2axisangle_forward" + _sfx(mat))
    with torch.cuda.device(mat.device):
        _lib.check(fn(_lib.ptr(mat), _lib.ptr(ax), n, _lib.stream_ptr()), "mat2axisangle_forward")
    return [ax]


def mat2axisangle_backward(mat, grad_axisangle):
    _lib.require_device(mat, grad_axisangle, name="mat/grad_axisangle")
    n = mat.shape[0]
    grad_mat = torch.zeros((n, 3, 4), dtype=mat.dtype, device=mat.device)
    fn = getattr(_lib.load(), "nesvor_mat2axisangle_backward" + _sfx(mat))
    with torch.cuda.device(mat.device):
        _lib.check(fn(_lib.ptr(mat), _lib.ptr(grad_axisangle), _lib.ptr(grad_mat), n, _lib.stream_ptr()), "mat2axisangle_backward")
    return [grad_mat]
