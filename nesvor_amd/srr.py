"""Classical super-resolution reconstruction on the slice-acquisition operator: conjugate gradients on
A^T A and the gradient-descent SRR of the reference (nesvor/svort/srr.py:12-160).  Pure host logic on top
of the native A / A^T ops."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .slice_acquisition import slice_acquisition, slice_acquisition_adjoint
from .transform import axisangle2mat


def dot(x, y):
    return torch.dot(x.flatten(), y.flatten())


def _safe_div(a, b):
    """a / b with 0 where b == 0 (no host sync).  The native A / A^T here are deterministic, so a CG started
    at the exact solution has a residual of exactly 0 where the reference (atomic adds) has round-off;
    0/0 must then mean "no update" instead of NaN (SURVEY.md §4 gotcha)."""
    return torch.where(b != 0, a / torch.where(b != 0, b, torch.ones_like(b)), torch.zeros_like(b))


def CG(A, b, x0, n_iter, tol=0.0):
    """Conjugate gradients for A x = b (A symmetric positive definite, given as a callable); srr.py:12-34."""
    if x0 is None:
        x, r = 0, b
    else:
        x, r = x0, b - A(x0)
    p = r
    rr = dot(r, r)
    i = 0
    while True:
        Ap = A(p)
        alpha = _safe_div(rr, dot(p, Ap))
        x = x + alpha * p
        i += 1
        if i == n_iter:
            return x
        r = r - alpha * Ap
        rr_new = dot(r, r)
        if rr_new <= tol:
            return x
        p = r + _safe_div(rr_new, rr) * p
        rr = rr_new


def PSFreconstruction(transforms, slices, slices_mask, vol_mask, params):
    """Equalised back-projection A^T y / A^T 1 (srr.py:37-48)."""
    return slice_acquisition_adjoint(
        transforms, params["psf"], slices, slices_mask, vol_mask, params["volume_shape"],
        params["res_s"] / params["res_r"], params["interp_psf"], True)


class SRR(nn.Module):
    """min_x |A x - y|^2 (+ edge-preserving prior), by CG on the normal equations or by gradient descent
    (srr.py:51-160)."""

    def __init__(self, n_iter=10, use_CG=False, alpha=0.5, beta=0.02, delta=0.1, tol=0.0):
        super().__init__()
        self.n_iter, self.use_CG, self.alpha, self.delta, self.tol = n_iter, use_CG, alpha, delta, tol
        self.beta = beta * delta * delta

    def forward(self, theta, slices, volume, params, p=None, mu=0, z=None, vol_mask=None, slices_mask=None):
        transforms = axisangle2mat(theta) if theta.ndim == 2 else theta
        rs = params["res_s"] / params["res_r"]

        def A(x):
            return slice_acquisition(transforms, x, vol_mask, slices_mask, params["psf"], params["slice_shape"], rs,
                                     False, params["interp_psf"])

        def At(y):
            return slice_acquisition_adjoint(transforms, params["psf"], y, slices_mask, vol_mask, params["volume_shape"],
                                             rs, params["interp_psf"], False)

        def AtA(x):
            y = A(x)
            if p is not None:
                y = y * p
            v = At(y)
            if mu and z is not None:
                v = v + mu * x
            return v

        x = volume
        if self.use_CG:
            b = At(slices * p if p is not None else slices)
            if mu and z is not None:
                b = b + mu * z
            x = CG(AtA, b, volume, self.n_iter, self.tol)
        else:
            for _ in range(self.n_iter):
                err = A(x) - slices
                if p is not None:
                    err = p * err
                g = At(err)
                if self.beta:
                    g.add_(self.dR(x, self.delta), alpha=self.beta)
                x.add_(g, alpha=-self.alpha)
        return F.relu(x, True)

    @staticmethod
    def dR(v, delta):
        """Gradient of the edge-preserving (Charbonnier-like) prior over the 26-neighbourhood (srr.py:134-160)."""
        g = torch.zeros_like(v)
        D, H, W = v.shape[-3:]
        core = v[:, :, 1 : D - 1, 1 : H - 1, 1 : W - 1]
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    if dx == 0 and dy == 0 and dz == 0:
                        continue
                    nb = v[:, :, 1 + dz : D - 1 + dz, 1 + dy : H - 1 + dy, 1 + dx : W - 1 + dx]
                    dv = core - nb
                    dv_ = dv * (1 / (dx * dx + dy * dy + dz * dz) / (delta * delta))
                    g[:, :, 1 : D - 1, 1 : H - 1, 1 : W - 1] += dv_ / torch.sqrt(1 + dv * dv_)
        return g
