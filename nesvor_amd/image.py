"""Image containers of the NeSVoR path: ``Slice`` (training input), ``Volume`` (mask / sampled output) and
``Stack`` (a series of parallel slices).  Same public surface as the container part of ``nesvor.image``
(image/image.py:17-250: constructor arguments, ``shape_xyz`` / ``resolution_xyz`` / ``xyz_masked`` /
``v_masked`` / ``rescale`` / ``clone`` / ``Volume.sample_points`` / ``Volume.resample``), written around one
idea: an image is a regular lattice in its own frame — voxel ``(k, j, i)`` sits at
``(index_xyz - (shape_xyz - 1) / 2) * resolution_xyz`` — placed in the world by a rigid pose.  File formats live in
``nesvor_amd.image_io``.
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
import torch.nn.functional as F

from .transform import RigidTransform, transform_points
from .utils import meshgrid

Scalar = Union[float, torch.Tensor]


def _identity_pose(device) -> RigidTransform:
    return RigidTransform(torch.zeros((1, 6), dtype=torch.float32, device=device))


class Image(object):
    """A 3-D array + boolean mask on a lattice with voxel size (rx, ry, rz), posed by ``transformation``."""

    def __init__(self, image: torch.Tensor, mask: Optional[torch.Tensor] = None,
                 transformation: Optional[RigidTransform] = None,
                 resolution_x: Scalar = 1.0, resolution_y: Scalar = 1.0, resolution_z: Scalar = 1.0) -> None:
        if image.ndim != 3:
            raise AssertionError("an Image holds a (D, H, W) array")
        self.image = image
        self.mask = mask if mask is not None else torch.ones_like(image, dtype=torch.bool)
        self.transformation = transformation if transformation is not None else _identity_pose(image.device)
        self.resolution_x, self.resolution_y, self.resolution_z = resolution_x, resolution_y, resolution_z

    # ---- lattice geometry ----------------------------------------------------------------------------------------
    @property
    def shape_xyz(self) -> torch.Tensor:
        d, h, w = self.image.shape
        return torch.tensor((w, h, d), device=self.image.device)

    @property
    def resolution_xyz(self) -> torch.Tensor:
        return torch.tensor([self.resolution_x, self.resolution_y, self.resolution_z], device=self.image.device)

    def _half_extent(self) -> torch.Tensor:
        """Distance (mm) from the lattice centre to its outermost voxel centres, per axis."""
        return (self.shape_xyz - 1) * self.resolution_xyz / 2

    @property
    def xyz_masked_untransformed(self) -> torch.Tensor:
        """(M, 3) lattice-frame coordinates (mm) of the masked voxels, in the order ``image[mask]`` lists them."""
        index_xyz = torch.nonzero(self.mask).flip(-1)  # (k, j, i) -> (x, y, z)
        return (index_xyz - (self.shape_xyz - 1) / 2) * self.resolution_xyz

    @property
    def xyz_masked(self) -> torch.Tensor:
        return transform_points(self.transformation, self.xyz_masked_untransformed)

    @property
    def v_masked(self) -> torch.Tensor:
        return self.image[self.mask]

    # ---- content ---------------------------------------------------------------------------------------------------
    def rescale(self, intensity_mean: Scalar) -> None:
        """Scale intensities in place so that the masked mean becomes ``intensity_mean``."""
        self.image *= intensity_mean / self.v_masked.mean()

    def _copy_fields(self, zero: bool) -> dict:
        make = torch.zeros_like if zero else torch.clone
        return dict(image=make(self.image), mask=make(self.mask), transformation=self.transformation.clone(),
                    resolution_x=float(self.resolution_x), resolution_y=float(self.resolution_y),
                    resolution_z=float(self.resolution_z))

    def clone(self, zero: bool = False):
        raise NotImplementedError

    def save(self, path: str, masked: bool = True) -> None:
        """NIfTI file of the (optionally masked) image (image.py:64-78)."""
        from .image_io import save_image

        save_image(self, path, masked)


class Slice(Image):
    def __init__(self, image, mask=None, transformation=None, resolution_x=1.0, resolution_y=1.0, resolution_z=1.0,
                 stack_idx: Optional[int] = None, slice_idx: Optional[int] = None) -> None:
        super().__init__(image, mask, transformation, resolution_x, resolution_y, resolution_z)
        self.stack_idx, self.slice_idx = stack_idx, slice_idx

    def clone(self, zero: bool = False) -> Slice:
        return Slice(**self._copy_fields(zero), stack_idx=self.stack_idx, slice_idx=self.slice_idx)


def _axis_aligned_cover(points: torch.Tensor, step: torch.Tensor, margin_voxels: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Smallest lattice of pitch ``step`` that covers ``points`` plus a margin: (corner voxel centre, shape_xyz)."""
    lo = points.amin(0) - margin_voxels * step
    hi = points.amax(0) + margin_voxels * step
    return lo, ((hi - lo) / step).ceil().long()


class Volume(Image):
    def sample_points(self, xyz: torch.Tensor) -> torch.Tensor:
        """Trilinear interpolation of the volume at world points (zero outside); image.py:124-132."""
        lead = xyz.shape[:-1]
        local = transform_points(self.transformation.inv(), xyz.reshape(-1, 3))
        unit = (local / self._half_extent()).view(1, 1, 1, -1, 3)  # [-1, 1] spans first..last voxel centre
        return F.grid_sample(self.image[None, None], unit, align_corners=True).view(lead)

    def resample(self, resolution_new: Optional[Scalar], transformation_new: Optional[RigidTransform]) -> Volume:
        """The volume on a new lattice: orientation of ``transformation_new`` (default: unchanged), voxel size
        ``resolution_new`` (default: unchanged), covering the masked voxels with a 10-voxel margin (image.py:134-177)."""
        pose = self.transformation if transformation_new is None else transformation_new
        rot = pose.matrix()[0, :, :3]
        if resolution_new is None:
            step = self.resolution_xyz
        elif torch.is_tensor(resolution_new) and resolution_new.numel() == 3:
            step = resolution_new.to(rot)
        else:
            step = torch.full((3,), float(resolution_new), dtype=rot.dtype, device=rot.device)
        # masked voxel centres in the new orientation (rows times R = R^T applied to column vectors)
        corner, shape_xyz = _axis_aligned_cover(self.xyz_masked.reshape(-1, 3) @ rot, step, 10)
        lattice = meshgrid(shape_xyz, step, corner, rot.device, True)
        values = self.sample_points(lattice @ rot.t())
        # a trans_first pose: the translation is the lattice centre expressed in the lattice's own orientation
        centre = corner + (shape_xyz - 1) / 2 * step
        placed = torch.cat([rot, centre[:, None]], -1)[None]
        rx, ry, rz = (float(s) for s in step)
        return Volume(values, values > 0, RigidTransform(placed, trans_first=True), rx, ry, rz)

    def clone(self, zero: bool = False) -> Volume:
        return Volume(**self._copy_fields(zero))


class Stack(object):
    """n parallel slices (n, 1, h, w) with common in-plane resolution, thickness and gap (image.py:183-250).  Without
    explicit poses slice k sits at z = (k - n / 2) * gap."""

    def __init__(self, slices: torch.Tensor, mask=None, transformation=None, score: float = 0.0,
                 resolution_x: float = 1.0, resolution_y: float = 1.0, thickness: float = 1.0, gap: float = 1.0) -> None:
        self.slices = slices
        self.mask = mask if mask is not None else torch.ones_like(slices, dtype=torch.bool)
        if transformation is None:
            n = slices.shape[0]
            pose = torch.zeros((n, 6), dtype=torch.float32, device=slices.device)
            pose[:, 5] = gap * (torch.arange(n, dtype=torch.float32, device=slices.device) - n / 2)
            transformation = RigidTransform(pose)
        self.transformation = transformation
        self.score = score
        self.resolution_x, self.resolution_y, self.thickness, self.gap = resolution_x, resolution_y, thickness, gap

    def __len__(self) -> int:
        return self.slices.shape[0]

    def __getitem__(self, idx):
        """One ``Slice`` for an integer index, a list of them for a slice / index tensor."""
        if self.slices.ndim != 4:
            raise AssertionError("a Stack holds (n, 1, h, w) slices")
        images, masks, poses = self.slices[idx], self.mask[idx], self.transformation[idx]
        as_slice = lambda im, mk, tf: Slice(im, mk, tf, self.resolution_x, self.resolution_y, self.thickness)
        if images.ndim == 3:
            return as_slice(images, masks, poses)
        return [as_slice(images[k], masks[k], poses[k]) for k in range(len(poses))]
