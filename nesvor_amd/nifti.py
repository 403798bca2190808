"""Minimal NIfTI-1 single-file (.nii / .nii.gz) reader and writer in NumPy: what the reference gets from
``nibabel`` for its volume I/O (nesvor/image/image.py:251-296).  Written against the NIfTI-1 standard
(348-byte header, data at vox_offset 352, Fortran-ordered voxels, sform rows + qform quaternion).

Only what the path needs: 3-D (or trailing-singleton 4-D+) scalar images, the common numeric data types on
read, float32 on write, scl_slope / scl_inter scaling, either byte order on read.
"""
import gzip
import struct
from typing import Tuple

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
           768: np.uint32, 1024: np.int64, 1280: np.uint64}


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def quaternion_from_rotation(R: np.ndarray) -> Tuple[float, float, float, float]:
    """Rotation matrix (proper, det +1) -> unit quaternion (a, b, c, d) with a >= 0 (the qform convention)."""
    R = np.asarray(R, dtype=np.float64)
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        a = 0.5 * np.sqrt(1.0 + tr)
        b, c, d = (R[2, 1] - R[1, 2]) / (4 * a), (R[0, 2] - R[2, 0]) / (4 * a), (R[1, 0] - R[0, 1]) / (4 * a)
    else:
        i = int(np.argmax([R[0, 0], R[1, 1], R[2, 2]]))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = 2.0 * np.sqrt(max(1.0 + R[i, i] - R[j, j] - R[k, k], 1e-30))
        v = [0.0, 0.0, 0.0]
        v[i] = 0.25 * s
        v[j] = (R[j, i] + R[i, j]) / s
        v[k] = (R[k, i] + R[i, k]) / s
        a = (R[k, j] - R[j, k]) / s
        b, c, d = v
    if a < 0:
        a, b, c, d = -a, -b, -c, -d
    return float(a), float(b), float(c), float(d)


def rotation_from_quaternion(b: float, c: float, d: float) -> np.ndarray:
    a = np.sqrt(max(1.0 - (b * b + c * c + d * d), 0.0))
    return np.array([
        [a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
        [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
        [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c],
    ])


def save(path: str, data_xyz: np.ndarray, affine: np.ndarray, qform_code: int = 2, sform_code: int = 1,
         xyzt_units: int = 2) -> None:
    """Write a 3-D float32 image.  data_xyz is indexed [x, y, z] (the NIfTI voxel order)."""
    data = np.asarray(data_xyz, dtype=np.float32)
    assert data.ndim == 3
    affine = np.asarray(affine, dtype=np.float64)
    RZS = affine[:3, :3]
    zooms = np.sqrt((RZS * RZS).sum(0))
    zooms[zooms == 0] = 1.0
    R = RZS / zooms
    qfac = 1.0
    if np.linalg.det(R) < 0:  # improper: the qform stores the flip of the third axis in pixdim[0]
        R = R.copy()
        R[:, 2] *= -1
        qfac = -1.0
    # nearest proper rotation (the affine may carry round-off / slight shear)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    _, qb, qc, qd = quaternion_from_rotation(R)
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    dim = [3, data.shape[0], data.shape[1], data.shape[2], 1, 1, 1, 1]
    struct.pack_into("<8h", hdr, 40, *dim)
    struct.pack_into("<h", hdr, 70, 16)  # datatype float32
    struct.pack_into("<h", hdr, 72, 32)  # bitpix
    struct.pack_into("<8f", hdr, 76, qfac, float(zooms[0]), float(zooms[1]), float(zooms[2]), 1.0, 1.0, 1.0, 1.0)
    struct.pack_into("<f", hdr, 108, 352.0)  # vox_offset
    struct.pack_into("<f", hdr, 112, 1.0)  # scl_slope
    struct.pack_into("<f", hdr, 116, 0.0)  # scl_inter
    struct.pack_into("<B", hdr, 123, xyzt_units)
    struct.pack_into("<h", hdr, 252, qform_code)
    struct.pack_into("<h", hdr, 254, sform_code)
    struct.pack_into("<3f", hdr, 256, qb, qc, qd)
    struct.pack_into("<3f", hdr, 268, affine[0, 3], affine[1, 3], affine[2, 3])
    struct.pack_into("<4f", hdr, 280, *affine[0])
    struct.pack_into("<4f", hdr, 296, *affine[1])
    struct.pack_into("<4f", hdr, 312, *affine[2])
    hdr[344:348] = b"n+1\0"
    with _open(path, "wb") as f:
        f.write(bytes(hdr))
        f.write(b"\0\0\0\0")  # no header extensions
        f.write(np.asfortranarray(data).tobytes(order="F"))


def load(path: str):
    """-> (data [x,y,z,...] float32 with scaling applied, pixdim[1:4], sform affine (4,4; NaN if absent),
    qform affine (4,4) | None, header dict)."""
    with _open(path, "rb") as f:
        raw = f.read()
    endian = "<"
    if struct.unpack_from("<i", raw, 0)[0] != 348:
        endian = ">"
        if struct.unpack_from(">i", raw, 0)[0] != 348:
            raise ValueError(f"{path}: not a NIfTI-1 file")
    if raw[344:347] != b"n+1":
        raise ValueError(f"{path}: only single-file NIfTI-1 (magic n+1) is supported")
    dim = struct.unpack_from(endian + "8h", raw, 40)
    datatype = struct.unpack_from(endian + "h", raw, 70)[0]
    pixdim = struct.unpack_from(endian + "8f", raw, 76)
    vox_offset = int(struct.unpack_from(endian + "f", raw, 108)[0])
    slope, inter = struct.unpack_from(endian + "2f", raw, 112)
    qform_code, sform_code = struct.unpack_from(endian + "2h", raw, 252)
    qb, qc, qd, qx, qy, qz = struct.unpack_from(endian + "6f", raw, 256)
    srow = np.array(struct.unpack_from(endian + "12f", raw, 280), dtype=np.float64).reshape(3, 4)
    if datatype not in _DTYPES:
        raise ValueError(f"{path}: unsupported NIfTI datatype {datatype}")
    shape = tuple(int(d) for d in dim[1 : 1 + dim[0]])
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(endian)
    n = int(np.prod(shape))
    data = np.frombuffer(raw, dtype=dt, count=n, offset=max(vox_offset, 352)).reshape(shape, order="F")
    data = data.astype(np.float32)
    if slope not in (0.0,) and not np.isnan(slope) and (slope != 1.0 or inter != 0.0):
        data = data * np.float32(slope) + np.float32(inter)
    sform = np.full((4, 4), np.nan)
    if sform_code > 0:
        sform = np.eye(4)
        sform[:3, :] = srow
    qform = None
    if qform_code > 0:
        R = rotation_from_quaternion(qb, qc, qd)
        qfac = -1.0 if pixdim[0] < 0 else 1.0
        qform = np.eye(4)
        qform[:3, :3] = R @ np.diag([pixdim[1], pixdim[2], pixdim[3] * qfac])
        qform[:3, 3] = [qx, qy, qz]
    header = {"dim": dim, "pixdim": pixdim, "datatype": datatype, "qform_code": qform_code, "sform_code": sform_code}
    return data, np.array(pixdim[1:4], dtype=np.float64), sform, qform, header
