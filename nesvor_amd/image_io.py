"""Volume / slice / stack file I/O and the affine <-> rigid-transform conventions of ``nesvor.image``
(image/image.py:251-393, image/image_utils.py:8-85), over the NumPy NIfTI codec in ``nesvor_amd.nifti``
(the reference uses ``nibabel``).  SURVEY.md §8(f) rank 2.

Conventions (identical to the reference):
* tensors are (D, H, W) = (z, y, x); files store [x, y, z];
* a volume's ``transformation`` maps voxel-centred mm coordinates (origin at the image centre in x, y and at the
  FIRST slice in z for stacks) to world mm with trans_first = True;  affine = [R diag(res) | R (t - centre)].
"""
import os
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import nifti
from .image import Slice, Stack, Volume
from .transform import RigidTransform


def compare_resolution_affine(r1, a1, r2, a2, s1, s2) -> bool:
    """image_utils.py:8-24"""
    r1, a1, r2, a2 = np.array(r1), np.array(a1), np.array(r2), np.array(a2)
    if s1 != s2 or r1.shape != r2.shape or a1.shape != a2.shape:
        return False
    return bool(np.amax(np.abs(r1 - r2)) <= 1e-3 and np.amax(np.abs(a1 - a2)) <= 1e-3)


def affine2transformation(volume: torch.Tensor, mask: torch.Tensor, resolutions: np.ndarray,
                          affine: np.ndarray) -> Tuple[torch.Tensor, torch.Tensor, RigidTransform]:
    """File affine -> one rigid transform per slice (image_utils.py:27-66).  A left-handed affine flips x."""
    device = volume.device
    d, h, w = volume.shape
    resolutions = np.asarray(resolutions, dtype=np.float64)
    R = affine[:3, :3]
    negative_det = np.linalg.det(R) < 0
    T = affine[:3, -1:]
    R = R @ np.linalg.inv(np.diag(resolutions))
    T0 = np.array([(w - 1) / 2 * resolutions[0], (h - 1) / 2 * resolutions[1], 0.0])
    T = np.linalg.inv(R) @ T + T0.reshape(3, 1)
    tz = torch.arange(0, d, device=device, dtype=torch.float32) * float(resolutions[2]) + float(T[2].item())
    tx = torch.ones_like(tz) * float(T[0].item())
    ty = torch.ones_like(tz) * float(T[1].item())
    t = torch.stack((tx, ty, tz), -1).view(-1, 3, 1)
    Rt = torch.tensor(R, device=device, dtype=torch.float32).unsqueeze(0).repeat(d, 1, 1)
    if negative_det:
        volume = torch.flip(volume, (-1,))
        mask = torch.flip(mask, (-1,))
        t[:, 0, -1] *= -1
        Rt[:, :, 0] *= -1
    return volume, mask, RigidTransform(torch.cat((Rt, t), -1).to(torch.float32), trans_first=True)


def transformation2affine(volume: torch.Tensor, transformation: RigidTransform, resolution_x: float,
                          resolution_y: float, resolution_z: float) -> np.ndarray:
    """image_utils.py:69-85"""
    mat = transformation.matrix(trans_first=True).detach().cpu().numpy().astype(np.float64)
    assert mat.shape[0] == 1
    R, T = mat[0, :, :-1], mat[0, :, -1:].copy()
    d, h, w = volume.shape
    T[0] -= (w - 1) / 2 * resolution_x
    T[1] -= (h - 1) / 2 * resolution_y
    T[2] -= (d - 1) / 2 * resolution_z
    affine = np.eye(4)
    affine[:3, :] = np.concatenate((R @ np.diag([resolution_x, resolution_y, resolution_z]), R @ T.reshape(3, 1)), -1)
    return affine


def save_nii_volume(path: str, volume, affine) -> None:
    """image.py:251-271: (D,H,W) or (D,1,H,W) tensor/array -> NIfTI [x,y,z], qform 'aligned' + sform 'scanner', mm."""
    assert len(volume.shape) == 3 or (len(volume.shape) == 4 and volume.shape[1] == 1)
    if len(volume.shape) == 4:
        volume = volume.squeeze(1)
    if isinstance(volume, torch.Tensor):
        volume = volume.detach().cpu().numpy()
    if isinstance(affine, torch.Tensor):
        affine = affine.detach().cpu().numpy()
    if affine is None:
        affine = np.eye(4)
    nifti.save(path, np.asarray(volume).transpose(2, 1, 0), affine, qform_code=2, sform_code=1, xyzt_units=2)


def load_nii_volume(path: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """image.py:274-294 -> (volume (D,H,W) float32, resolutions (x,y,z), affine (4,4))."""
    data, pixdim, sform, qform, hdr = nifti.load(path)
    dim = hdr["dim"]
    if not (dim[0] == 3 or (dim[0] > 3 and all(d == 1 for d in dim[4:]))):
        raise AssertionError("Expect a 3D volume but the input is %dD" % dim[0])
    while data.ndim > 3:
        data = data.squeeze(-1)
    affine = sform
    if np.any(np.isnan(affine)):
        affine = qform
    if affine is None:
        # neither sform nor qform coded (common for masks and converted data): the header's base affine, as nibabel's
        # ``img.affine`` falls back to - voxel sizes on the diagonal, origin at the centre voxel
        zooms = np.asarray(pixdim, dtype=np.float64)
        affine = np.diag(np.append(zooms, 1.0))
        affine[:3, 3] = -(np.asarray(data.shape[:3], dtype=np.float64) - 1) / 2 * zooms
    return np.ascontiguousarray(data.transpose(2, 1, 0)), pixdim, affine


def save_image(img, path: str, masked: bool = True) -> None:
    """Image.save (image.py:64-78)"""
    affine = transformation2affine(img.image, img.transformation, float(img.resolution_x), float(img.resolution_y),
                                   float(img.resolution_z))
    out = img.image * img.mask.to(img.image.dtype) if masked else img.image
    save_nii_volume(path, out, affine)


def save_slices(folder: str, images: List[Slice]) -> None:
    for i, image in enumerate(images):
        save_image(image, os.path.join(folder, f"{i}.nii.gz"), True)


def load_slices(folder: str, device=torch.device("cpu")) -> List[Slice]:
    slices, ids = [], []
    for f in os.listdir(folder):
        if not (f.endswith("nii") or f.endswith("nii.gz")):
            continue
        ids.append(int(f.split(".nii")[0]))
        arr, resolutions, affine = load_nii_volume(os.path.join(folder, f))
        t = torch.tensor(arr, device=device)
        t, m, transformation = affine2transformation(t, t > 0, resolutions, affine)
        slices.append(Slice(image=t, mask=m, transformation=transformation, resolution_x=float(resolutions[0]),
                            resolution_y=float(resolutions[1]), resolution_z=float(resolutions[2])))
    return [s for _, s in sorted(zip(ids, slices), key=lambda p: p[0])]


def _load_with_mask(path_vol, path_mask):
    vol, resolutions, affine = load_nii_volume(path_vol)
    if path_mask is None:
        mask = vol > 0
    else:
        mask, resolutions_m, affine_m = load_nii_volume(path_mask)
        mask = mask > 0
        if not compare_resolution_affine(resolutions, affine, resolutions_m, affine_m, vol.shape, mask.shape):
            raise Exception("Error: the sizes/resolutions/affine transformations of the input stack and stack mask do not match!")
    return vol, mask, resolutions, affine


def load_stack(path_vol: str, path_mask: Optional[str] = None, device=torch.device("cpu")) -> Stack:
    vol, mask, resolutions, affine = _load_with_mask(path_vol, path_mask)
    t, m, transformation = affine2transformation(torch.tensor(vol, device=device), torch.tensor(mask, device=device),
                                                 resolutions, affine)
    return Stack(slices=t.unsqueeze(1), mask=m.unsqueeze(1), transformation=transformation,
                 resolution_x=float(resolutions[0]), resolution_y=float(resolutions[1]), thickness=float(resolutions[2]),
                 gap=float(resolutions[2]))


def load_volume(path_vol: str, path_mask: Optional[str] = None, device=torch.device("cpu")) -> Volume:
    vol, mask, resolutions, affine = _load_with_mask(path_vol, path_mask)
    t, m, transformation = affine2transformation(torch.tensor(vol, device=device), torch.tensor(mask, device=device),
                                                 resolutions, affine)
    transformation = RigidTransform(transformation.axisangle().mean(0, keepdim=True))
    return Volume(image=t, mask=m, transformation=transformation, resolution_x=float(resolutions[0]),
                  resolution_y=float(resolutions[1]), resolution_z=float(resolutions[2]))


# ---- checkpoint layout of `nesvor reconstruct --output-model` (cli/io.py:33-59) ------------------------------------
def save_model(path: str, model, mask: Volume, args) -> None:
    """torch.save({'model': state_dict, 'mask': Volume, 'args': Namespace}) as the reference writes it (cli/io.py:38-47).
    The mask's classes are recorded under the reference's import paths (``_ckpt_pickle``), so the file also loads in an
    environment that has the reference installed."""
    from . import _ckpt_pickle

    torch.save({"model": model.state_dict(), "mask": mask, "args": args}, path, pickle_module=_ckpt_pickle)


def load_model(path: str, device, args=None):
    """-> (INR with the stored weights, mask Volume moved to `device`, stored args (device / dtype overridden as the
    reference does in cli/io.py:36-45)).  Accepts checkpoints written by the reference (its ``nesvor.image`` /
    ``nesvor.transform`` classes resolve to this package's) as well as this package's own."""
    from . import _ckpt_pickle
    from .models import INR

    cp = torch.load(path, map_location=device, weights_only=False, pickle_module=_ckpt_pickle)
    stored = cp["args"]
    stored.device = device
    if args is not None:
        stored.dtype = getattr(args, "dtype", stored.dtype)
    inr = INR(cp["model"]["bounding_box"].to(device), stored)
    inr.load_state_dict(cp["model"])
    mask = cp["mask"]
    return inr.to(device), mask, stored
