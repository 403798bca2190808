"""Rigid transforms: host-side mirror of ``nesvor.transform``
(nesvor/transform/transform.py, transform_convert.py).

Conventions (identical to the reference):
* a transform is either ``(n,6)`` axis-angle ``[rotvec | t]`` or ``(n,3,4)``
  matrix ``[R | t]``;
* ``trans_first=True``  means  x' = R (x + t);  ``False`` means x' = R x + t;
* all algebra (inv/compose/cat) is done on trans_first matrices.

The two conversions are the differentiable dispatcher ops ``torch.ops.nesvor.axisangle2mat_forward`` /
``mat2axisangle_forward`` (``nesvor_amd.ops``); there is no CPU implementation here.
"""
from __future__ import annotations

import math
from typing import Iterable

import torch
from . import transform_convert_cuda as _backend  # noqa: F401  (the reference's module name; registers torch.ops.nesvor)


def axisangle2mat(axisangle: torch.Tensor) -> torch.Tensor:
    """(n,6) -> (n,3,4); differentiable (transform_convert.py:20-34,52)."""
    return torch.ops.nesvor.axisangle2mat_forward(axisangle.contiguous())


def mat2axisangle(mat: torch.Tensor) -> torch.Tensor:
    """(n,3,4) -> (n,6); differentiable (transform_convert.py:36-49,53)."""
    return torch.ops.nesvor.mat2axisangle_forward(mat.contiguous())


def trans_loss_raw(axisangle, axisangle_init):
    """One launch, no autograd: -> (per-slice loss terms (n), d loss / d axisangle (n,6))."""
    from . import _lib

    ax, ax0 = axisangle.contiguous(), axisangle_init.contiguous()
    _lib.require_device(ax, ax0, dtype=torch.float32, name="trans_loss input")
    n = ax.shape[0]
    per = torch.empty(n, dtype=torch.float32, device=ax.device)
    grad = torch.empty((n, 6), dtype=torch.float32, device=ax.device)
    with torch.cuda.device(ax.device):
        err = _lib.load().nesvor_trans_loss(_lib.ptr(ax), _lib.ptr(ax0), _lib.ptr(per), _lib.ptr(grad), n, _lib.stream_ptr())
    _lib.check(err, "trans_loss")
    return per, grad


def trans_loss_fused(axisangle: torch.Tensor, axisangle_init: torch.Tensor) -> torch.Tensor:
    """mean(err_R^2) + 1e-3 mean(err_T^2), err = axisangle(inv(init) o cur): one fused launch (models.py:357-363)."""
    return torch.ops.nesvor.trans_loss(axisangle.contiguous(), axisangle_init.contiguous())[0]


def _split(mat):
    return mat[..., :3], mat[..., 3:]


def mat_first2last(mat: torch.Tensor) -> torch.Tensor:
    """[R|t] with x'=R(x+t)  ->  [R|Rt] with x'=Rx+t'."""
    R, t = _split(mat)
    return torch.cat([R, R @ t], -1)


def mat_last2first(mat: torch.Tensor) -> torch.Tensor:
    R, t = _split(mat)
    return torch.cat([R, R.transpose(-2, -1) @ t], -1)


def ax_first2last(axisangle: torch.Tensor) -> torch.Tensor:
    return mat2axisangle(mat_first2last(axisangle2mat(axisangle)))


def ax_last2first(axisangle: torch.Tensor) -> torch.Tensor:
    return mat2axisangle(mat_last2first(axisangle2mat(axisangle)))


class RigidTransform(object):
    """Batch of n rigid transforms (transform.py:8-128)."""

    def __init__(self, data: torch.Tensor, trans_first: bool = True, device=None) -> None:
        if device is not None:
            data = data.to(device)
        self.trans_first = trans_first
        self._axisangle = None
        self._matrix = None
        if data.ndim == 2 and data.shape[1] == 6:
            self._axisangle = data
        elif data.ndim == 3 and data.shape[1] == 3:
            self._matrix = data
        else:
            raise Exception("Unknown format for rigid transform!")

    # -- representations ---------------------------------------------------
    def matrix(self, trans_first: bool = True) -> torch.Tensor:
        mat = self._matrix if self._matrix is not None else axisangle2mat(self._axisangle)
        if self.trans_first and not trans_first:
            mat = mat_first2last(mat)
        elif not self.trans_first and trans_first:
            mat = mat_last2first(mat)
        return mat

    def axisangle(self, trans_first: bool = True) -> torch.Tensor:
        ax = self._axisangle if self._axisangle is not None else mat2axisangle(self._matrix)
        if self.trans_first and not trans_first:
            ax = ax_first2last(ax)
        elif not self.trans_first and trans_first:
            ax = ax_last2first(ax)
        return ax

    # -- algebra -----------------------------------------------------------
    def inv(self) -> RigidTransform:
        R, t = _split(self.matrix(True))
        return RigidTransform(torch.cat([R.transpose(-2, -1), -(R @ t)], -1), trans_first=True)

    def compose(self, other: RigidTransform) -> RigidTransform:
        """self ∘ other (other applied first)."""
        R1, t1 = _split(self.matrix(True))
        R2, t2 = _split(other.matrix(True))
        return RigidTransform(torch.cat([R1 @ R2, t2 + R2.transpose(-2, -1) @ t1], -1), trans_first=True)

    # -- container protocol ------------------------------------------------
    def _data(self) -> torch.Tensor:
        if self._axisangle is not None:
            return self._axisangle
        if self._matrix is not None:
            return self._matrix
        raise Exception("Both data are None!")

    def __getitem__(self, idx) -> RigidTransform:
        full = self._data()
        data = full[idx]
        if data.ndim < full.ndim:
            data = data.unsqueeze(0)
        return RigidTransform(data, self.trans_first)

    def detach(self) -> RigidTransform:
        return RigidTransform(self._data().detach(), self.trans_first)

    def clone(self) -> RigidTransform:
        return RigidTransform(self._data().clone(), self.trans_first)

    @property
    def device(self):
        return self._data().device

    def __len__(self) -> int:
        return self._data().shape[0]

    @staticmethod
    def cat(transforms: Iterable[RigidTransform]) -> RigidTransform:
        return RigidTransform(torch.cat([t.matrix(True) for t in transforms], 0), trans_first=True)


# -- resolution rescaling (translation part only) --------------------------
def mat_update_resolution(mat: torch.Tensor, res_from, res_to) -> torch.Tensor:
    assert mat.dim() == 3
    fac = torch.ones_like(mat[:1, :1])
    fac[..., 3] = res_from / res_to
    return mat * fac


def ax_update_resolution(ax: torch.Tensor, res_from, res_to) -> torch.Tensor:
    assert ax.dim() == 2
    fac = torch.ones_like(ax[:1])
    fac[:, 3:] = res_from / res_to
    return ax * fac


# -- Euler / three-point parameterisations (transform.py:162-256) ----------
def mat2euler(mat: torch.Tensor) -> torch.Tensor:
    """-> (n,6) [tx,ty,tz,rx,ry,rz] with angles in degrees."""
    ry = torch.asin(-mat[:, 0, 2])
    gimbal = torch.cos(ry).abs() <= 1e-6
    rx = torch.atan2(mat[:, 1, 2], mat[:, 2, 2])
    rz = torch.atan2(mat[:, 0, 1], mat[:, 0, 0])
    rx_g = torch.atan2(-mat[:, 0, 2] * mat[:, 1, 0], -mat[:, 0, 2] * mat[:, 2, 0])
    rx = torch.where(gimbal, rx_g, rx)
    rz = torch.where(gimbal, torch.zeros_like(rz), rz)
    deg = 180 / math.pi
    return torch.stack([mat[:, 0, 3], mat[:, 1, 3], mat[:, 2, 3], rx * deg, ry * deg, rz * deg], -1)


def euler2mat(p: torch.Tensor) -> torch.Tensor:
    rad = p[:, 3:] * (math.pi / 180.0)
    (cx, cy, cz), (sx, sy, sz) = torch.cos(rad).unbind(-1), torch.sin(rad).unbind(-1)
    rows = [
        [cy * cz, cy * sz, -sy, p[:, 0]],
        [sx * sy * cz - cx * sz, sx * sy * sz + cx * cz, sx * cy, p[:, 1]],
        [cx * sy * cz + sx * sz, cx * sy * sz - sx * cz, cx * cy, p[:, 2]],
    ]
    return torch.stack([torch.stack(r, -1) for r in rows], -2)


def point2mat(p: torch.Tensor) -> torch.Tensor:
    """Three points (left-bottom, centre, right-bottom of a slice) -> [R|t]."""
    p = p.view(-1, 3, 3)
    p1, p2, p3 = p[:, 0], p[:, 1], p[:, 2]
    ex = p3 - p1
    ez = torch.cross(ex, p2 - p1, dim=-1)
    ey = torch.cross(ez, ex, dim=-1)
    R = torch.stack([ex, ey, ez], -1)
    R = R / torch.linalg.norm(R, ord=2, dim=-2, keepdim=True)
    T = R.transpose(-2, -1) @ p2.unsqueeze(-1)
    return torch.cat([R, T], -1)


def mat2point(mat: torch.Tensor, sx, sy, rs) -> torch.Tensor:
    hx, hy = (sx - 1) / 2 * rs, (sy - 1) / 2 * rs
    corners = torch.tensor([[-hx, -hy, 0], [0, 0, 0], [hx, -hy, 0]], dtype=mat.dtype, device=mat.device)
    R, T = mat[:, None, :, :3], mat[:, None, :, 3:]
    pts = R @ (corners[None, :, :, None] + T)
    return pts.reshape(-1, 9)


# -- applying transforms to points -----------------------------------------
def mat_transform_points(mat: torch.Tensor, x: torch.Tensor, trans_first: bool) -> torch.Tensor:
    """mat (*,3,4), x (*,3) with broadcasting (transform.py:259-271): R (x + T) for ``trans_first``, else R x + T.
    Evaluated without the broadcast batched matmul the formula suggests: one 3x3 matrix for all points is a plain
    (n,3) x (3,3) product, per-point matrices are nine broadcast multiply-adds.  (rocBLAS runs a batch of 3x3 by 3x1
    products at ~7 ms per million and faults at batch counts beyond ~2^24 - the 0.5 mm lattice of a 128^3 volume.)"""
    R, T = mat[..., :3], mat[..., 3]
    y = x + T if trans_first else x
    if R.numel() == 9:
        out = y @ R.reshape(3, 3).t()
    else:
        out = torch.stack([(R[..., i, :] * y).sum(-1) for i in range(3)], -1)
    return out if trans_first else out + T


def ax_transform_points(ax: torch.Tensor, x: torch.Tensor, trans_first: bool) -> torch.Tensor:
    mat = axisangle2mat(ax.reshape(-1, 6)).view(ax.shape[:-1] + (3, 4))
    return mat_transform_points(mat, x, trans_first)


def transform_points(transform: RigidTransform, x: torch.Tensor) -> torch.Tensor:
    assert x.ndim == 2 and x.shape[-1] == 3
    return mat_transform_points(transform.matrix(transform.trans_first), x, transform.trans_first)
