"""File formats: the NumPy NIfTI-1 codec and the affine <-> RigidTransform conventions of
nesvor.image, checked as the reference's tests/image/test_image.py does (save -> load round trips over the transform
table), on CPU with the oracle standing in for the native transform ops and on the GPU with the HIP ops."""
import os

import numpy as np
import pytest
import torch

from conftest import TRANSFORM_TABLE


def test_nifti_codec_roundtrip(tmp_path):
    from nesvor_amd import nifti

    rng = np.random.default_rng(0)
    data = rng.standard_normal((7, 5, 3)).astype(np.float32)
    for k, (ext, flip) in enumerate([(".nii", False), (".nii.gz", False), (".nii.gz", True)]):
        A = np.eye(4)
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        if (np.linalg.det(q) < 0) != flip:
            q[:, 0] *= -1
        A[:3, :3] = q @ np.diag([0.8, 1.1, 2.5])
        A[:3, 3] = [10.5, -3.25, 7.0]
        path = str(tmp_path / f"v{k}{ext}")
        nifti.save(path, data, A)
        got, pixdim, sform, qform, hdr = nifti.load(path)
        np.testing.assert_array_equal(got, data)
        np.testing.assert_allclose(pixdim, [0.8, 1.1, 2.5], rtol=1e-6)
        np.testing.assert_allclose(sform, A, atol=1e-5)
        np.testing.assert_allclose(qform, A, atol=1e-4)  # quaternion + qfac reproduce the (possibly left-handed) affine
        assert hdr["qform_code"] == 2 and hdr["sform_code"] == 1 and tuple(hdr["dim"][:4]) == (3, 7, 5, 3)
    raw = open(str(tmp_path / "v0.nii"), "rb").read()
    assert len(raw) == 352 + data.size * 4 and raw[344:348] == b"n+1\x00"  # single-file layout, data at vox_offset 352


def test_nifti_quaternion_roundtrip():
    from scipy.spatial.transform import Rotation

    from nesvor_amd import nifti

    for rv in np.array(TRANSFORM_TABLE)[:, :3]:
        R = Rotation.from_rotvec(rv).as_matrix()
        a, b, c, d = nifti.quaternion_from_rotation(R)
        assert a >= 0 and abs(a * a + b * b + c * c + d * d - 1) < 1e-12
        np.testing.assert_allclose(nifti.rotation_from_quaternion(b, c, d), R, atol=1e-12)


def _image_cases(device, is_volume, small):
    """tests/image/test_image.py:12-38 (sizes reduced when `small`)."""
    from nesvor_amd.image import Slice, Volume
    from nesvor_amd.transform import RigidTransform

    out = []
    for i, row in enumerate(TRANSFORM_TABLE):
        ax = torch.tensor([row], dtype=torch.float32, device=device)
        tf = RigidTransform(ax, trans_first=i % 2 == 1)
        d0, h0, w0 = (12, 13, 24) if small else (128, 128, 256)
        image = torch.full(((d0 - i) if is_volume else 1, h0 + i, w0 + i), float(i), dtype=torch.float32, device=device)
        res = (0.5 + 0.1 * i, 0.5 + 0.2 * i, 0.5 + 0.3 * i)
        C = Volume if is_volume else Slice
        out.append((C(image, None, tf, *res), image, tf, res))
    return out


def _check_slices(tmp_path, device, small):
    from nesvor_amd.image_io import load_slices, save_slices

    cases = _image_cases(device, False, small)
    folder = str(tmp_path / "slices")
    os.makedirs(folder)
    save_slices(folder, [c[0] for c in cases])
    loaded = load_slices(folder, device)
    assert len(loaded) == len(cases)
    for s, (obj, image, tf, res) in zip(loaded, cases):
        assert abs(res[0] - s.resolution_x) < 1e-3 and abs(res[1] - s.resolution_y) < 1e-3 and abs(res[2] - s.resolution_z) < 1e-3
        torch.testing.assert_close(s.transformation.axisangle(), tf.axisangle(), atol=1e-4, rtol=1e-3)
        torch.testing.assert_close(s.image, image)


def _check_volumes(tmp_path, device, small):
    from nesvor_amd.image_io import load_volume

    for i, (v, image, tf, res) in enumerate(_image_cases(device, True, small)):
        path = str(tmp_path / f"{i}.nii.gz")
        v.save(path)
        v_ = load_volume(path, device=device)
        assert abs(v_.resolution_x - res[0]) < 1e-3 and abs(v_.resolution_y - res[1]) < 1e-3 and abs(v_.resolution_z - res[2]) < 1e-3
        torch.testing.assert_close(v_.transformation.axisangle(), v.transformation.axisangle(), atol=1e-4, rtol=1e-3)
        torch.testing.assert_close(v_.image, v.image)


def test_save_load_slices_cpu(tmp_path, oracle_backend):
    _check_slices(tmp_path, torch.device("cpu"), True)


def test_save_load_volume_cpu(tmp_path, oracle_backend):
    _check_volumes(tmp_path, torch.device("cpu"), True)


def test_stack_mask_mismatch_raises(tmp_path, oracle_backend):
    from nesvor_amd.image_io import load_stack, save_nii_volume

    a = np.eye(4)
    save_nii_volume(str(tmp_path / "s.nii.gz"), torch.rand(4, 6, 5), a)
    save_nii_volume(str(tmp_path / "m.nii.gz"), torch.ones(4, 6, 5), a)
    st = load_stack(str(tmp_path / "s.nii.gz"), str(tmp_path / "m.nii.gz"))
    assert st.slices.shape == (4, 1, 6, 5) and bool(st.mask.all())
    b = a.copy()
    b[0, 3] = 2.0
    save_nii_volume(str(tmp_path / "m2.nii.gz"), torch.ones(4, 6, 5), b)
    with pytest.raises(Exception, match="do not match"):
        load_stack(str(tmp_path / "s.nii.gz"), str(tmp_path / "m2.nii.gz"))


@pytest.mark.gpu
def test_save_load_slices_gpu(tmp_path, device):
    _check_slices(tmp_path, device, False)


@pytest.mark.gpu
def test_save_load_volume_gpu(tmp_path, device):
    _check_volumes(tmp_path, device, True)


@pytest.mark.gpu
def test_checkpoint_roundtrip_gpu(tmp_path, device, golden):
    """cli/io.py:38-46 / :52-58: {'model','mask','args'} -> INR with identical outputs."""
    from conftest import small_args
    from nesvor_amd.image import Volume
    from nesvor_amd.image_io import load_model, save_model
    from nesvor_amd.models import INR

    args = small_args(device=device)
    bbox = torch.tensor(golden["fw_sd::inr.bounding_box"]).to(device)
    torch.manual_seed(0)
    inr = INR(bbox, args).to(device)
    mask = Volume(torch.ones(4, 5, 6, device=device), None, None, 1.0, 1.0, 1.0)
    path = str(tmp_path / "model.pt")
    save_model(path, inr, mask, args)
    inr2, mask2, args2 = load_model(path, device)
    assert list(inr2.state_dict().keys()) == list(inr.state_dict().keys())
    x = bbox[0] + (bbox[1] - bbox[0]) * torch.rand(64, 3, device=device)
    torch.testing.assert_close(inr2(x), inr(x))
    assert mask2.image.shape == (4, 5, 6) and args2.n_features_z == args.n_features_z
