"""Imaging model + losses op (autograd Function over nesvor_imaging_loss).

Takes the network outputs of a batch and returns the loss terms of NeSVoR.forward
(models.py:286-325) as 0-d tensors that carry autograd: the forward launch computes the per-pixel
partial sums, the backward launch receives the upstream gradients of the individual terms (whatever
weights the caller combines them with) and writes every input gradient in one pass.
"""
import ctypes

import torch
from torch.autograd import Function

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


class ImagingLossFunction(Function):
    @staticmethod
    def forward(ctx, z0, log_var, log_bias, x, v, slice_idx, c, log_var_slice, reg_type, delta):
        cont = lambda t: None if t is None else t.contiguous()
        z0, log_var, log_bias, x, v, slice_idx, c, log_var_slice = map(
            cont, (z0, log_var, log_bias, x, v, slice_idx, c, log_var_slice))
        _lib.require_device(z0, log_var, log_bias, x, v, c, log_var_slice, dtype=torch.float32, name="imaging loss input")
        _lib.require_device(slice_idx, dtype=torch.int64, name="slice_idx")
        B, S = x.shape[0], x.shape[1]
        lb_mean = log_bias.mean().reshape(1) if log_bias is not None else None
        loss_pix = torch.empty((B, 3), dtype=torch.float32, device=x.device)
        a = _fill(z0, log_var, log_bias, x, v, slice_idx, c, log_var_slice, lb_mean, reg_type, delta)
        a.loss_pix = loss_pix.data_ptr()
        with torch.cuda.device(x.device), _lib.kernel_timer.span("imaging_loss_fwd"):
            err = _lib.load().nesvor_imaging_loss(ctypes.byref(a), _lib.stream_ptr())
        _lib.check(err, "imaging loss forward")
        sums = loss_pix.sum(0)
        mse, logvar = sums[0] / B, sums[1] / B
        mean_term = sums[2] / (B * S)
        ireg = delta * (mean_term - 1) if reg_type == 0 else mean_term
        breg = lb_mean[0] ** 2 if log_bias is not None else torch.zeros((), device=x.device)
        ctx.save_for_backward(z0, log_var, log_bias, x, v, slice_idx, c, log_var_slice, lb_mean)
        ctx.cfg = (reg_type, delta)
        return mse, logvar, ireg, breg

    @staticmethod
    def backward(ctx, g_mse, g_logvar, g_ireg, g_breg):
        z0, log_var, log_bias, x, v, slice_idx, c, log_var_slice, lb_mean = ctx.saved_tensors
        reg_type, delta = ctx.cfg
        dev = x.device
        B, S = x.shape[0], x.shape[1]
        gw = torch.stack([g_mse, g_logvar, g_ireg, g_breg]).to(torch.float32).contiguous()
        dz0 = torch.empty_like(z0)
        dlv = torch.empty_like(log_var) if log_var is not None else None
        dlb = torch.empty_like(log_bias) if log_bias is not None else None
        dx = torch.empty_like(x) if ctx.needs_input_grad[3] else None
        dc_pix = torch.empty(B, dtype=torch.float32, device=dev) if c is not None else None
        dlvs_pix = torch.empty(B, dtype=torch.float32, device=dev) if log_var_slice is not None else None
        a = _fill(z0, log_var, log_bias, x, v, slice_idx, c, log_var_slice, lb_mean, reg_type, delta)
        a.gw = gw.data_ptr()
        for name, t in (("dz0", dz0), ("dlog_var", dlv), ("dlog_bias", dlb), ("dx", dx), ("dc_pix", dc_pix), ("dlvs_pix", dlvs_pix)):
            setattr(a, name, None if t is None else t.data_ptr())
        with torch.cuda.device(dev), _lib.kernel_timer.span("imaging_loss_bwd"):
            err = _lib.load().nesvor_imaging_loss(ctypes.byref(a), _lib.stream_ptr())
        _lib.check(err, "imaging loss backward")
        dc = torch.zeros_like(c).index_add_(0, slice_idx, dc_pix) if c is not None else None
        dlvs = torch.zeros_like(log_var_slice).index_add_(0, slice_idx, dlvs_pix) if log_var_slice is not None else None
        return dz0, dlv, dlb, dx, None, None, dc, dlvs, None, None


def imaging_loss(z0, log_var, log_bias, x, v, slice_idx, c, log_var_slice, reg_name, delta):
    """-> (MSE, logVar, imageReg, biasReg) 0-d tensors."""
    return ImagingLossFunction.apply(z0, log_var, log_bias, x, v, slice_idx, c, log_var_slice, REG_TYPES[reg_name], float(delta))
