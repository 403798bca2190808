"""Classical super-resolution reconstruction (SRR) on the native slice-acquisition operator - the solver family of
``nesvor.svort.srr`` (nesvor/svort/srr.py:12-160) behind the same names: ``CG``, ``PSFreconstruction``, ``SRR``.

Built around ``AcquisitionOperator``: the linear map A (volume -> slices, PSF-weighted sampling at the slices' poses)
with its exact adjoint A^T, both native HIP kernels (csrc/slice_acq.hip).  On top of it
* ``conjugate_gradient`` solves the (weighted, optionally Tikhonov-damped) normal equations; every scalar of the
  recurrence stays a 0-d device tensor, so an n-iteration solve issues no host synchronisation unless ``tol`` > 0;
* ``SRR`` with ``use_CG=False`` runs the reference's gradient descent with the edge-preserving prior; the prior's
  gradient is evaluated for all 26 neighbours at once instead of one sliced pass per neighbour.
Host logic only; SURVEY.md 8(f) rank 1.
"""
from typing import Callable, Optional

import torch
import torch.nn as nn

from .slice_acquisition import slice_acquisition, slice_acquisition_adjoint
from .transform import axisangle2mat


class AcquisitionOperator:
    """y = A x for one set of slice poses: ``forward`` (A), ``adjoint`` (A^T), ``normal`` (A^T W A + mu I)."""

    def __init__(self, transforms: torch.Tensor, params: dict, vol_mask=None, slices_mask=None) -> None:
        self.transforms = transforms
        self.psf = params["psf"]
        self.slice_shape = params["slice_shape"]
        self.volume_shape = params["volume_shape"]
        self.ratio = params["res_s"] / params["res_r"]  # slice pixel size in volume voxels
        self.interp_psf = params["interp_psf"]
        self.vol_mask, self.slices_mask = vol_mask, slices_mask

    def forward(self, volume: torch.Tensor) -> torch.Tensor:
        return slice_acquisition(self.transforms, volume, self.vol_mask, self.slices_mask, self.psf, self.slice_shape,
                                 self.ratio, False, self.interp_psf)

    def adjoint(self, slices: torch.Tensor, equalize: bool = False) -> torch.Tensor:
        return slice_acquisition_adjoint(self.transforms, self.psf, slices, self.slices_mask, self.vol_mask,
                                         self.volume_shape, self.ratio, self.interp_psf, equalize)

    def normal(self, volume: torch.Tensor, weight: Optional[torch.Tensor] = None, mu: float = 0.0) -> torch.Tensor:
        y = self.forward(volume)
        if weight is not None:
            y = y * weight
        out = self.adjoint(y)
        return out + mu * volume if mu else out


def _inner(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return (a * b).sum()


def _ratio(num: torch.Tensor, den: torch.Tensor) -> torch.Tensor:
    """num / den as a 0-d device tensor, 0 where den == 0.  The native A / A^T are deterministic (no atomics), so a
    solve started at the exact solution has a residual of exactly zero where the reference sees round-off noise; a
    vanished search direction must then mean "stay", not NaN."""
    ok = den != 0
    return torch.where(ok, num / torch.where(ok, den, torch.ones_like(den)), torch.zeros_like(den))


def conjugate_gradient(apply: Callable[[torch.Tensor], torch.Tensor], rhs: torch.Tensor, start: Optional[torch.Tensor],
                       n_iter: int, tol: float = 0.0) -> torch.Tensor:
    """``n_iter`` conjugate-gradient steps on ``apply(x) = rhs`` (``apply`` symmetric positive definite), from ``start``
    (``None``: from zero).  Stops early once the squared residual norm drops to ``tol`` (checked only when tol > 0: the
    check reads a device scalar)."""
    if start is None:
        x = torch.zeros_like(rhs)
        residual = rhs.clone()
    else:
        x = start.clone()
        residual = rhs - apply(start)
    direction = residual.clone()
    rr = _inner(residual, residual)
    for it in range(n_iter):
        a_dir = apply(direction)
        step = _ratio(rr, _inner(direction, a_dir))
        x.addcmul_(direction, step)
        if it + 1 == n_iter:
            break
        residual.addcmul_(a_dir, -step)
        rr_next = _inner(residual, residual)
        if tol > 0 and float(rr_next) <= tol:
            break
        direction.mul_(_ratio(rr_next, rr)).add_(residual)
        rr = rr_next
    return x


def CG(A, b, x0, n_iter, tol=0.0):
    """The reference's entry point (srr.py:12-34): ``A`` callable, ``x0`` may be None."""
    return conjugate_gradient(A, b, x0, n_iter, tol)


def PSFreconstruction(transforms, slices, slices_mask, vol_mask, params):
    """Equalised back-projection A^T y / A^T 1 (srr.py:37-48)."""
    return AcquisitionOperator(transforms, params, vol_mask, slices_mask).adjoint(slices, equalize=True)


# the 26 neighbour offsets (dz, dy, dx) and their squared lengths
_OFFSETS = [(dz, dy, dx) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dz, dy, dx) != (0, 0, 0)]


def edge_prior_gradient(volume: torch.Tensor, delta: float) -> torch.Tensor:
    """Gradient of sum over neighbour pairs of delta^2 (sqrt(1 + (d / (|o| delta))^2) - 1)-type edge-preserving
    penalty as the reference defines it (srr.py:134-160): for every interior voxel the sum over its 26 neighbours of
    t / sqrt(1 + d t), d = v - v_neighbour, t = d / (|o|^2 delta^2); border voxels get 0."""
    out = torch.zeros_like(volume)
    D, H, W = volume.shape[-3:]
    if min(D, H, W) < 3:
        return out
    core = volume[..., 1 : D - 1, 1 : H - 1, 1 : W - 1]
    shifted = torch.stack([volume[..., 1 + dz : D - 1 + dz, 1 + dy : H - 1 + dy, 1 + dx : W - 1 + dx] for dz, dy, dx in _OFFSETS])
    inv = torch.tensor([1.0 / ((dz * dz + dy * dy + dx * dx) * delta * delta) for dz, dy, dx in _OFFSETS],
                       dtype=volume.dtype, device=volume.device).view(-1, *([1] * volume.ndim))
    diff = core[None] - shifted
    scaled = diff * inv
    out[..., 1 : D - 1, 1 : H - 1, 1 : W - 1] = (scaled * torch.rsqrt(1 + diff * scaled)).sum(0)
    return out


class SRR(nn.Module):
    """min_x |A x - y|_W^2 (+ mu |x - z|^2) by CG on the normal equations (``use_CG``), or ``n_iter`` steps of gradient
    descent x -= alpha (A^T W (A x - y) + beta delta^2 dR(x)) with the edge-preserving prior; result clamped at 0
    (srr.py:51-132).  ``theta``: (n, 6) axis-angle poses or (n, 3, 4) matrices, in voxel units of the volume."""

    def __init__(self, n_iter=10, use_CG=False, alpha=0.5, beta=0.02, delta=0.1, tol=0.0):
        super().__init__()
        self.n_iter, self.use_CG, self.alpha, self.delta, self.tol = n_iter, use_CG, alpha, delta, tol
        self.beta = beta * delta * delta

    def forward(self, theta, slices, volume, params, p=None, mu=0, z=None, vol_mask=None, slices_mask=None):
        op = AcquisitionOperator(axisangle2mat(theta) if theta.ndim == 2 else theta, params, vol_mask, slices_mask)
        damped = bool(mu) and z is not None
        if self.use_CG:
            rhs = op.adjoint(slices if p is None else slices * p)
            if damped:
                rhs = rhs + mu * z
            x = conjugate_gradient(lambda v: op.normal(v, p, mu if damped else 0.0), rhs, volume, self.n_iter, self.tol)
        else:
            x = volume  # updated in place, as the reference does
            for _ in range(self.n_iter):
                misfit = op.forward(x) - slices
                grad = op.adjoint(misfit if p is None else misfit * p)
                if self.beta:
                    grad.add_(edge_prior_gradient(x, self.delta), alpha=self.beta)
                x.sub_(grad, alpha=self.alpha)
        return x.clamp_(min=0)

    @staticmethod
    def dR(v, delta):
        return edge_prior_gradient(v, delta)
