"""Image containers on the hot path: ``Slice`` (input), ``Volume`` (mask / output),
``Stack``.  Mirrors the container part of ``nesvor.image`` (image/image.py:17-250).
File I/O (NIfTI, checkpoints) lives in ``nesvor_amd.image_io``; ``Image.save`` forwards to it.
"""
from __future__ import annotations

from typing import Dict, Optional, Union

import torch
import torch.nn.functional as F

from .transform import RigidTransform, transform_points
from .utils import meshgrid


class Image(object):
    def __init__(
        self,
        image: torch.Tensor,
        mask: Optional[torch.Tensor] = None,
        transformation: Optional[RigidTransform] = None,
        resolution_x: Union[float, torch.Tensor] = 1.0,
        resolution_y: Union[float, torch.Tensor] = 1.0,
        resolution_z: Union[float, torch.Tensor] = 1.0,
    ) -> None:
        assert image.ndim == 3
        self.image = image
        self.mask = torch.ones_like(image, dtype=torch.bool) if mask is None else mask
        if transformation is None:
            transformation = RigidTransform(torch.zeros((1, 6), dtype=torch.float32, device=image.device))
        self.transformation = transformation
        self.resolution_x = resolution_x
        self.resolution_y = resolution_y
        self.resolution_z = resolution_z

    def clone(self, zero: bool = False):
        raise NotImplementedError

    def save(self, path: str, masked: bool = True) -> None:
        """image.py:64-78"""
        from .image_io import save_image

        save_image(self, path, masked)

    def _clone_image(self, zero: bool = False) -> Dict:
        return {
            "image": torch.zeros_like(self.image) if zero else self.image.clone(),
            "mask": torch.zeros_like(self.mask) if zero else self.mask.clone(),
            "transformation": self.transformation.clone(),
            "resolution_x": float(self.resolution_x),
            "resolution_y": float(self.resolution_y),
            "resolution_z": float(self.resolution_z),
        }

    @property
    def shape_xyz(self) -> torch.Tensor:
        return torch.tensor(self.image.shape[::-1], device=self.image.device)

    @property
    def resolution_xyz(self) -> torch.Tensor:
        return torch.tensor([self.resolution_x, self.resolution_y, self.resolution_z], device=self.image.device)

    @property
    def xyz_masked_untransformed(self) -> torch.Tensor:
        """Physical coordinates (slice frame, mm) of the masked pixels, centre at 0."""
        kji = torch.flip(torch.nonzero(self.mask), (-1,))
        return (kji - (self.shape_xyz - 1) / 2) * self.resolution_xyz

    @property
    def xyz_masked(self) -> torch.Tensor:
        return transform_points(self.transformation, self.xyz_masked_untransformed)

    @property
    def v_masked(self) -> torch.Tensor:
        return self.image[self.mask]

    def rescale(self, intensity_mean: Union[float, torch.Tensor]) -> None:
        self.image *= intensity_mean / self.image[self.mask].mean()


class Slice(Image):
    def __init__(self, image, mask=None, transformation=None, resolution_x=1.0, resolution_y=1.0, resolution_z=1.0,
                 stack_idx: Optional[int] = None, slice_idx: Optional[int] = None) -> None:
        super().__init__(image, mask, transformation, resolution_x, resolution_y, resolution_z)
        self.stack_idx = stack_idx
        self.slice_idx = slice_idx

    def clone(self, zero: bool = False) -> Slice:
        return Slice(stack_idx=self.stack_idx, slice_idx=self.slice_idx, **self._clone_image(zero))


class Volume(Image):
    def sample_points(self, xyz: torch.Tensor) -> torch.Tensor:
        """Trilinear lookup of the volume at world points (image.py:124-132)."""
        shape = xyz.shape[:-1]
        xyz = transform_points(self.transformation.inv(), xyz.view(-1, 3))
        xyz = xyz / ((self.shape_xyz - 1) * self.resolution_xyz / 2)
        return F.grid_sample(self.image[None, None], xyz.view(1, 1, 1, -1, 3), align_corners=True).view(shape)

    def resample(self, resolution_new, transformation_new: Optional[RigidTransform]) -> Volume:
        """Resample onto a new grid covering the masked region + 10 voxels margin (image.py:134-177)."""
        if transformation_new is None:
            transformation_new = self.transformation
        R = transformation_new.matrix()[0, :3, :3]
        if resolution_new is None:
            resolution_new = self.resolution_xyz
        elif isinstance(resolution_new, (float, int)) or resolution_new.numel() == 1:
            resolution_new = torch.tensor([float(resolution_new)] * 3, dtype=R.dtype, device=R.device)
        xyz = torch.matmul(torch.inverse(R), self.xyz_masked.view(-1, 3, 1))[..., 0]
        xyz_min = xyz.amin(0) - resolution_new * 10
        xyz_max = xyz.amax(0) + resolution_new * 10
        shape_xyz = ((xyz_max - xyz_min) / resolution_new).ceil().long()
        mat = torch.zeros((1, 3, 4), dtype=R.dtype, device=R.device)
        mat[0, :, :3] = R
        mat[0, :, -1] = xyz_min + (shape_xyz - 1) / 2 * resolution_new
        grid = meshgrid(shape_xyz, resolution_new, xyz_min, R.device, True)
        grid = torch.matmul(R, grid[..., None])[..., 0]
        v = self.sample_points(grid)
        return Volume(v, v > 0, RigidTransform(mat, trans_first=True),
                      resolution_new[0].item(), resolution_new[1].item(), resolution_new[2].item())

    def clone(self, zero: bool = False) -> Volume:
        return Volume(**self._clone_image(zero))


class Stack(object):
    """A stack of parallel slices (n,1,h,w) sharing resolution/thickness/gap (image.py:183-250)."""

    def __init__(self, slices: torch.Tensor, mask=None, transformation=None, score: float = 0.0,
                 resolution_x: float = 1.0, resolution_y: float = 1.0, thickness: float = 1.0, gap: float = 1.0) -> None:
        self.slices = slices
        self.mask = torch.ones_like(slices, dtype=torch.bool) if mask is None else mask
        if transformation is None:
            n = slices.shape[0]
            t = torch.zeros((n, 6), dtype=torch.float32, device=slices.device)
            t[:, -1] = (torch.arange(n, dtype=torch.float32, device=slices.device) - n / 2) * gap
            transformation = RigidTransform(t)
        self.transformation = transformation
        self.score = score
        self.resolution_x, self.resolution_y = resolution_x, resolution_y
        self.thickness, self.gap = thickness, gap

    def __len__(self) -> int:
        return self.slices.shape[0]

    def _slice(self, img, msk, tr) -> Slice:
        return Slice(img, msk, tr, self.resolution_x, self.resolution_y, self.thickness)

    def __getitem__(self, idx):
        assert self.slices.ndim == 4
        imgs, msks, trs = self.slices[idx], self.mask[idx], self.transformation[idx]
        if imgs.ndim < self.slices.ndim:
            return self._slice(imgs, msks, trs)
        return [self._slice(imgs[i], msks[i], trs[i]) for i in range(len(trs))]
