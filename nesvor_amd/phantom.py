"""Synthetic inputs for BASELINE.json's configs: the 3-D modified Shepp-Logan
phantom and slice stacks simulated from it with the slice-acquisition operator.

``phantom3d`` reproduces the arrays of the reference's test generator
(tests/phantom3d.py:7-96) bit for bit — including its grid quirk: the
coordinate grid has (n-1)^3 points whose flat indices are written into an n^3
buffer — because BASELINE's configs are defined on exactly those arrays
(fixture hashes in tests/golden/).

``simulate_stacks`` follows the recipe of the reference's slice-acquisition test
(tests/slice_acquisition/test_slice_acq.py:13-74) and wraps the result as
``Slice`` objects for ``train``.
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .image import Slice
from .slice_acquisition import slice_acquisition
from .transform import RigidTransform, mat_update_resolution
from .utils import get_PSF

# Toft's "modified Shepp-Logan" head phantom: additive intensity A, semi-axes a b c,
# centre x0 y0 z0, Euler angles phi theta psi (degrees)
_MODIFIED_SHEPP_LOGAN = np.array(
    [
        [1.0, 0.6900, 0.920, 0.810, 0.00, 0.0000, 0.00, 0, 0, 0],
        [-0.8, 0.6624, 0.874, 0.780, 0.00, -0.0184, 0.00, 0, 0, 0],
        [-0.2, 0.1100, 0.310, 0.220, 0.22, 0.0000, 0.00, -18, 0, 10],
        [-0.2, 0.1600, 0.410, 0.280, -0.22, 0.0000, 0.00, 18, 0, 10],
        [0.1, 0.2100, 0.250, 0.410, 0.00, 0.3500, -0.15, 0, 0, 0],
        [0.1, 0.0460, 0.046, 0.050, 0.00, 0.1000, 0.25, 0, 0, 0],
        [0.1, 0.0460, 0.046, 0.050, 0.00, -0.1000, 0.25, 0, 0, 0],
        [0.1, 0.0460, 0.023, 0.050, -0.08, -0.6050, 0.00, 0, 0, 0],
        [0.1, 0.0230, 0.023, 0.020, 0.00, -0.6060, 0.00, 0, 0, 0],
        [0.1, 0.0230, 0.046, 0.020, 0.06, -0.6050, 0.00, 0, 0, 0],
    ]
)
_SHEPP_LOGAN_A = [1, -0.98, -0.02, -0.02, 0.01, 0.01, 0.01, 0.01, 0.01, 0.01]
_YU_YE_WANG = np.array(
    [
        [1.0, 0.6900, 0.920, 0.900, 0.00, 0.000, 0.000, 0, 0, 0],
        [-0.8, 0.6624, 0.874, 0.880, 0.00, 0.000, 0.000, 0, 0, 0],
        [-0.2, 0.4100, 0.160, 0.210, -0.22, 0.000, -0.250, 108, 0, 0],
        [-0.2, 0.3100, 0.110, 0.220, 0.22, 0.000, -0.250, 72, 0, 0],
        [0.2, 0.2100, 0.250, 0.500, 0.00, 0.350, -0.250, 0, 0, 0],
        [0.2, 0.0460, 0.046, 0.046, 0.00, 0.100, -0.250, 0, 0, 0],
        [0.1, 0.0460, 0.023, 0.020, -0.08, -0.650, -0.250, 0, 0, 0],
        [0.1, 0.0460, 0.023, 0.020, 0.06, -0.650, -0.250, 90, 0, 0],
        [0.2, 0.0560, 0.040, 0.100, 0.06, -0.105, 0.625, 90, 0, 0],
        [-0.2, 0.0560, 0.056, 0.100, 0.00, 0.100, 0.625, 0, 0, 0],
    ]
)


def _ellipsoids(name: str) -> np.ndarray:
    if name == "modified-shepp-logan":
        return _MODIFIED_SHEPP_LOGAN.copy()
    if name == "shepp_logan":
        e = _MODIFIED_SHEPP_LOGAN.copy()
        e[:, 0] = _SHEPP_LOGAN_A
        return e
    if name == "yu_ye_wang":
        return _YU_YE_WANG.copy()
    raise TypeError('phantom type "%s" not recognized' % name)


def _euler_zxz(phi: float, theta: float, psi: float) -> np.ndarray:
    cp, sp, ct, st, cs, ss = np.cos(phi), np.sin(phi), np.cos(theta), np.sin(theta), np.cos(psi), np.sin(psi)
    return np.array(
        [
            [cs * cp - ct * sp * ss, cs * sp + ct * cp * ss, ss * st],
            [-ss * cp - ct * sp * cs, -ss * sp + ct * cp * cs, cs * st],
            [st * sp, -st * cp, ct],
        ]
    )


def phantom3d(phantom: str = "modified-shepp-logan", n: int = 64) -> np.ndarray:
    """(n,n,n) float64 sum of ellipsoid indicator functions (see module docstring)."""
    table = _ellipsoids(phantom)
    half = (n - 1) / 2
    axis = (np.arange(0, n - 1) - half) / half  # n-1 grid points, as the reference has it
    gx, gy, gz = np.meshgrid(axis, axis, axis)
    coord = np.vstack((gx.flatten(), gy.flatten(), gz.flatten()))
    out = np.zeros(n**3)
    for A, a, b, c, x0, y0, z0, phi, theta, psi in table:
        rot = _euler_zxz(phi * np.pi / 180, theta * np.pi / 180, psi * np.pi / 180)
        p = np.dot(rot, coord)
        inside = (p[0, :] - x0) ** 2.0 / a**2 + (p[1, :] - y0) ** 2.0 / b**2 + (p[2, :] - z0) ** 2.0 / c**2 <= 1
        idx = np.nonzero(inside)[0]
        out[idx] = out[idx] + A
    return out.reshape((n, n, n))


# stack orientations of the reference's slice-acquisition test (test_slice_acq.py:24-41)
STACK_ANGLES = [
    [0, 0, 0], [np.pi / 2, 0, 0], [0, np.pi / 2, 0], [0, 0, np.pi / 2],
    [np.pi / 4, np.pi / 4, 0], [0, np.pi / 4, np.pi / 4], [np.pi / 4, 0, np.pi / 4],
    [np.pi / 3, np.pi / 3, 0], [0, np.pi / 3, np.pi / 3], [np.pi / 3, 0, np.pi / 3],
    [2 * np.pi / 3, 2 * np.pi / 3, 0], [0, 2 * np.pi / 3, 2 * np.pi / 3], [2 * np.pi / 3, 0, 2 * np.pi / 3],
    [np.pi / 5, np.pi / 5, 0], [0, np.pi / 5, np.pi / 5], [np.pi / 5, 0, np.pi / 5],
]


def stack_geometry(vs: int, res: float, res_s: float, gap: float) -> Tuple[int, int]:
    """(n_slice, slice_size) covering the volume diagonal (test_slice_acq.py:18-19)."""
    n_slice = int((np.sqrt(3) * vs) / gap) + 4
    ss = int((np.sqrt(3) * vs) / res_s) + 4
    return n_slice, ss


def stack_transforms(angle: Sequence[float], n_slice: int, gap: float, device) -> RigidTransform:
    rot = torch.tensor([list(angle)], dtype=torch.float32, device=device).expand(n_slice, -1)
    tz = (torch.arange(0, n_slice, dtype=torch.float32, device=device) - (n_slice - 1) / 2.0) * gap
    txy = torch.ones_like(tz) * 0.5
    return RigidTransform(torch.cat((rot, torch.stack((txy, txy, tz), -1)), -1), trans_first=True)


def simulate_stacks(
    volume: torch.Tensor,  # (D,H,W) float32 device tensor, isotropic `res` mm
    n_stacks: int = 3,
    res: float = 1.0,
    res_s: float = 1.5,
    s_thick: float = 3.0,
    gap: Optional[float] = None,
    motion_deg: float = 0.0,
    motion_mm: float = 0.0,
    seed: int = 0,
    normalize: bool = True,
) -> Tuple[List[Slice], RigidTransform]:
    """Simulate `n_stacks` orthogonal/oblique stacks.  Returns (slices with the NOMINAL
    poses, true poses).  With motion_* > 0 each slice is acquired at a perturbed pose
    (rotvec ~ N(0, motion_deg^2), t ~ N(0, motion_mm^2)) — BASELINE config 4."""
    device = volume.device
    gap = s_thick if gap is None else gap
    vs = volume.shape[-1]
    n_slice, ss = stack_geometry(vs, res, res_s, gap)
    psf = get_PSF(res_ratio=(res_s / res, res_s / res, s_thick / res), device=device)
    g = torch.Generator().manual_seed(seed)
    slices: List[Slice] = []
    true_tf = []
    vol5 = volume[None, None].contiguous()
    for si in range(n_stacks):
        nominal = stack_transforms(STACK_ANGLES[si], n_slice, gap, device)
        actual = nominal
        if motion_deg > 0 or motion_mm > 0:
            d = torch.cat(
                [torch.randn(n_slice, 3, generator=g) * (motion_deg * np.pi / 180), torch.randn(n_slice, 3, generator=g) * motion_mm], -1
            ).to(device)
            actual = RigidTransform(d, trans_first=True).compose(nominal)
        mat = mat_update_resolution(actual.matrix(), 1, res)
        imgs = slice_acquisition(mat, vol5, None, None, psf, (ss, ss), res_s / res, False, False)
        true_tf.append(actual)
        for k in range(n_slice):
            img = imgs[k]
            slices.append(Slice(img, img > 0, nominal[k], res_s, res_s, s_thick, stack_idx=si, slice_idx=k))
    if normalize:  # intensities / 0.99-quantile, as svort/inference.py:558 does
        allv = torch.cat([s.image[s.mask] for s in slices])
        k = max(int(0.99 * allv.numel()), 1)
        q = torch.kthvalue(allv, k).values
        for s in slices:
            s.image = s.image / q
    return slices, RigidTransform.cat(true_tf)
