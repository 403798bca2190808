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


def test_checkpoint_is_interchangeable_with_the_reference(tmp_path):
    """cli/io.py:33-59: the reference pickles the mask as ``nesvor.image.image.Volume`` holding a
    ``nesvor.transform.transform.RigidTransform``.  (a) A checkpoint written here records exactly those class paths
    (none of this package's), so the reference can unpickle it; (b) a checkpoint whose pickle stream names the
    reference's classes - as one written by the reference does - loads here without the reference installed."""
    import io
    import pickle
    import zipfile
    from argparse import Namespace

    from conftest import small_args
    from nesvor_amd.image import Volume
    from nesvor_amd.image_io import load_model, save_model
    from nesvor_amd.models import INR
    from nesvor_amd.transform import RigidTransform

    args = small_args()
    bbox = torch.tensor([[-20.0, -22.0, -24.0], [20.0, 22.0, 24.0]])
    torch.manual_seed(0)
    inr = INR(bbox, args)
    pose = RigidTransform(torch.tensor([[0.1, -0.2, 0.3, 1.0, 2.0, 3.0]]), trans_first=True)
    mask = Volume(torch.rand(4, 5, 6), torch.rand(4, 5, 6) > 0.5, pose, 0.8, 0.9, 1.1)
    path = str(tmp_path / "model.pt")
    save_model(path, inr, mask, args)
    with zipfile.ZipFile(path) as zf:
        stream = zf.read([n for n in zf.namelist() if n.endswith("data.pkl")][0])
    assert b"nesvor.image.image\nVolume" in stream and b"nesvor.transform.transform\nRigidTransform" in stream
    assert b"nesvor_amd" not in stream
    # (b): this stream IS what the reference writes for the mask; additionally build one by hand with the shorter
    # package-level paths the reference's __init__ re-exports
    inr2, mask2, args2 = load_model(path, torch.device("cpu"))
    assert type(mask2) is Volume and type(mask2.transformation) is RigidTransform
    torch.testing.assert_close(mask2.image, mask.image)
    assert torch.equal(mask2.mask, mask.mask) and float(mask2.resolution_z) == 1.1
    torch.testing.assert_close(mask2.transformation._axisangle, pose._axisangle)
    for (k1, v1), (k2, v2) in zip(inr.state_dict().items(), inr2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    assert args2.n_features_z == args.n_features_z and args2.device == torch.device("cpu")

    from nesvor_amd import _ckpt_pickle

    class _Ref:  # stands for an object of a reference class: pickled by hand under the reference's path
        pass

    by_hand = (b"\x80\x02cnesvor.image\nVolume\nq\x00)\x81q\x01}q\x02(X\x0c\x00\x00\x00resolution_xq\x03G?\xe0\x00\x00\x00\x00\x00\x00"
               b"X\x05\x00\x00\x00imageq\x04Nub.")
    obj = _ckpt_pickle.load(io.BytesIO(by_hand))
    assert type(obj) is Volume and obj.resolution_x == 0.5
    with pytest.raises((ModuleNotFoundError, ImportError)):
        pickle.loads(by_hand)  # the stock unpickler needs the reference installed


def test_nifti_without_sform_or_qform_gets_the_centred_base_affine(tmp_path):
    """image.py:274-294 reads ``img.affine``; for a header with sform_code == qform_code == 0 nibabel falls back to
    diag(pixdim) with the origin at the centre voxel.  Such files (masks, converted data) must load, not raise."""
    from nesvor_amd import nifti
    from nesvor_amd.image_io import load_nii_volume

    data = np.arange(4 * 5 * 6, dtype=np.float32).reshape(4, 5, 6)
    A = np.diag([0.8, 1.1, 2.5, 1.0])
    path = str(tmp_path / "nocodes.nii.gz")
    nifti.save(path, data, A, qform_code=0, sform_code=0)
    vol, res, affine = load_nii_volume(path)
    assert vol.shape == (6, 5, 4)
    np.testing.assert_allclose(res, [0.8, 1.1, 2.5], rtol=1e-6)
    expect = np.diag([0.8, 1.1, 2.5, 1.0])
    expect[:3, 3] = -(np.array([4, 5, 6]) - 1) / 2 * np.array([0.8, 1.1, 2.5])
    np.testing.assert_allclose(affine, expect, rtol=1e-6, atol=1e-6)
