"""Drop-in for the reference's pybind module ``nesvor.transform_convert_cuda``
(nesvor/transform/transform_convert_cuda.cpp:64-69).

Same four entry points, same conventions: device tensors only, contiguous, float or double, each call returns a *list of
length 1* holding a newly allocated output.  The functions are the dispatcher ops ``torch.ops.nesvor.*`` registered in
``nesvor_amd.ops`` (HIP kernels of libnesvor_hip.so under the CUDA dispatch key; a host tensor is refused by the
dispatcher, a strided one by the op).
"""
import torch

from . import ops as _ops  # noqa: F401  (registers torch.ops.nesvor)


def axisangle2mat_forward(axisangle):
    return [torch.ops.nesvor.axisangle2mat_forward(axisangle)]


def axisangle2mat_backward(grad_mat, axisangle):
    return [torch.ops.nesvor.axisangle2mat_backward(grad_mat, axisangle)]


def mat2axisangle_forward(mat):
    return [torch.ops.nesvor.mat2axisangle_forward(mat)]


def mat2axisangle_backward(mat, grad_axisangle):
    return [torch.ops.nesvor.mat2axisangle_backward(mat, grad_axisangle)]
