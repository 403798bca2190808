"""Imaging model + losses (nesvor_imaging_loss): argument block of the kernel and the differentiable ``imaging_loss``.

Takes the network outputs of a batch and returns the loss terms of NeSVoR.forward
(models.py:286-325) as 0-d tensors that carry autograd: the forward launch computes the per-pixel
partial sums, the backward launch receives the upstream gradients of the individual terms (whatever
weights the caller combines them with) and writes every input gradient in one pass.
"""
import ctypes

import torch

from . import _lib

REG_TYPES = {"edge": 0, "TV": 1, "L2": 2}


def _fill(z0, log_var, log_bias, x, v, slice_idx, c, lvs, lb_mean, reg_type, delta):
    a = _lib.LossT()
    for name, t in (("z0", z0), ("log_var", log_var), ("log_bias", log_bias), ("x", x), ("v", v),
                    ("slice_idx", slice_idx), ("c", c), ("log_var_slice", lvs), ("log_bias_mean", lb_mean)):
        setattr(a, name, None if t is None else t.data_ptr())
    a.B, a.S = x.shape[0], x.shape[1]
    a.reg_type, a.delta = reg_type, float(delta)
    return a


def imaging_loss(z0, log_var, log_bias, x, v, slice_idx, c, log_var_slice, reg_name, delta):
    """-> (MSE, logVar, imageReg, biasReg) 0-d tensors that carry autograd: the dispatcher op
    ``torch.ops.nesvor.imaging_loss`` (``nesvor_amd.ops``)."""
    cont = lambda t: None if t is None else t.contiguous()
    out = torch.ops.nesvor.imaging_loss(cont(z0), cont(log_var), cont(log_bias), cont(x), cont(v), cont(slice_idx), cont(c),
                                        cont(log_var_slice), REG_TYPES[reg_name], float(delta))
    return out[0], out[1], out[2], out[3]
