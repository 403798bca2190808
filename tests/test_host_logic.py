"""CPU: the product's HOST logic (no native compute) against fixtures captured from the
reference's Python.  The native transform ops are stood in by the oracle (fixture
`oracle_backend`) — the product itself has no CPU path."""
import hashlib
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, scipy_table, small_args


def _t(a):
    return torch.tensor(a)


def test_phantom_bit_exact(golden):
    from nesvor_amd.phantom import phantom3d

    for n in (32, 64):
        sha = hashlib.sha1(phantom3d(n=n).astype(np.float32).tobytes()).digest()
        assert np.frombuffer(sha, dtype=np.uint8).tolist() == golden[f"phantom_sha1_{n}"].tolist()


def test_psf_and_utils(golden):
    from nesvor_amd.utils import gaussian_blur, get_PSF, meshgrid, resolution2sigma

    psf = get_PSF(res_ratio=(1.5, 1.5, 3.0))
    assert tuple(psf.shape) == (9, 5, 5) and int((psf > 0).sum()) == 153
    np.testing.assert_array_equal(psf.numpy(), golden["psf_15_15_3"])
    np.testing.assert_array_equal(get_PSF(res_ratio=(1.0, 1.0, 1.0)).numpy(), golden["psf_1_1_1"])
    np.testing.assert_array_equal(
        resolution2sigma(torch.tensor([[1.5, 1.5, 3.0], [0.8, 0.8, 0.8]])).numpy(), golden["sigma_aniso"])
    assert resolution2sigma(0.8, isotropic=True) == float(golden["sigma_iso"])
    np.testing.assert_array_equal(gaussian_blur(_t(golden["blur_in"]), 1.5, 3).numpy(), golden["blur_out"])
    np.testing.assert_array_equal(meshgrid((4, 3, 2), (1.5, 1.5, 3.0)).numpy(), golden["meshgrid"])


def test_rigid_transform_algebra(golden, oracle_backend):
    from nesvor_amd.transform import (RigidTransform, euler2mat, mat2euler, mat2point, mat_update_resolution,
                                      point2mat, transform_points)

    ax = _t(golden["tf_ax"])
    A = RigidTransform(ax, trans_first=True)
    B = RigidTransform(ax.flip(0).clone(), trans_first=False)
    close = lambda a, k: np.testing.assert_allclose(a.numpy(), golden[k], rtol=1e-6, atol=1e-6, err_msg=k)
    close(A.matrix(True), "tf_mat_first")
    close(A.matrix(False), "tf_mat_last")
    close(A.axisangle(False), "tf_ax_last")
    close(A.inv().matrix(True), "tf_inv")
    close(A.compose(B).matrix(True), "tf_compose")
    close(B.axisangle(True), "tf_B_first_ax")
    close(transform_points(A, _t(golden["tf_pts"])), "tf_pts_out")
    close(mat2euler(A.matrix(True)), "tf_euler")
    close(euler2mat(mat2euler(A.matrix(True))), "tf_euler2mat")
    close(mat2point(A.matrix(True), 128, 96, 0.8), "tf_mat2point")
    close(point2mat(mat2point(A.matrix(True), 128, 96, 0.8)), "tf_point2mat")
    close(mat_update_resolution(A.matrix(True), 1.0, 0.8), "tf_update_res")
    assert len(A[2:5]) == 3 and len(A[3]) == 1 and len(RigidTransform.cat([A, B])) == 22


def test_compose_inv_reference_test(oracle_backend):
    """tests/transform/test_transform.py:7-23, verbatim in structure (mixed trans_first)."""
    from nesvor_amd.transform import RigidTransform

    ax, mat = scipy_table()
    zeros = torch.zeros(1, 6)
    n = len(ax)
    for i in range(n):
        ax_a, mat_a = ax[i : i + 1], mat[i : i + 1]
        ax_b, mat_b = ax[n - 1 - i : n - i], mat[n - 1 - i : n - i]
        ab = RigidTransform(ax_a, trans_first=i % 2 == 0).compose(RigidTransform(mat_b, trans_first=i % 2 == 1))
        inv_b_inv_a = RigidTransform(ax_b, trans_first=i % 2 == 1).inv().compose(
            RigidTransform(mat_a, trans_first=i % 2 == 0).inv())
        err = ab.compose(inv_b_inv_a).axisangle()
        # reference tolerance on the rotation; the translations of this table reach 300 mm, where one
        # fp32 ulp of an intermediate is 3.05e-5 (CPU matmul rounding differs from the GPU's FMA order)
        torch.testing.assert_close(err[:, :3], zeros[:, :3], atol=2e-5, rtol=1e-3)
        torch.testing.assert_close(err[:, 3:], zeros[:, 3:], atol=1e-4, rtol=1e-3)


def test_roundtrips_reference_tests():
    """tests/transform/test_transform_convert.py:23-33 (pure-PyTorch helpers)."""
    from nesvor_amd.transform import euler2mat, mat2euler, mat2point, point2mat

    _, mat = scipy_table()
    for i in range(len(mat)):
        m = mat[i : i + 1]
        torch.testing.assert_close(point2mat(mat2point(m, 128 + 2 * i, 128 + 4 * i, 0.5 + 0.1 * i)), m)
        torch.testing.assert_close(euler2mat(mat2euler(m)), m)


def _golden_slices(golden):
    from nesvor_amd.image import Slice
    from nesvor_amd.transform import RigidTransform

    vs, res, res_s, s_thick, gap, n_slice, ss = golden["sim_geom"]
    imgs = _t(golden["sim_stacks"])
    tf = RigidTransform(_t(golden["sim_transforms"]), trans_first=True)
    return [Slice(imgs[k], imgs[k] > 0, tf[k], float(res_s), float(res_s), float(s_thick)) for k in range(imgs.shape[0])]


def test_dataset_fields_bbox_mean_mask(golden, oracle_backend):
    from nesvor_amd.train import Dataset

    args = small_args()
    ds = Dataset(_golden_slices(golden), args)
    np.testing.assert_array_equal(ds.xyz.numpy(), golden["ds_xyz"])
    np.testing.assert_array_equal(ds.v.numpy(), golden["ds_v"])
    np.testing.assert_array_equal(ds.slice_idx.numpy(), golden["ds_slice_idx"])
    np.testing.assert_allclose(ds.transformation.matrix().numpy(), golden["ds_transformation"], atol=1e-6)
    np.testing.assert_array_equal(ds.resolution.numpy(), golden["ds_resolution"])
    np.testing.assert_allclose(ds.bounding_box.numpy(), golden["ds_bounding_box"], rtol=1e-6, atol=1e-5)
    assert abs(ds.mean - float(golden["ds_mean"])) < 1e-6
    m = ds.mask
    np.testing.assert_array_equal(m.mask.numpy(), golden["ds_mask"])
    np.testing.assert_allclose(m.transformation.matrix().numpy(), golden["ds_mask_tf"], rtol=1e-6, atol=1e-5)
    assert abs(float(m.resolution_x) - float(golden["ds_mask_res"])) < 1e-7
    # first get_batch shuffles (count starts at len) and the epoch counter becomes 1
    torch.manual_seed(0)
    ds2 = Dataset(_golden_slices(golden), args)
    b = ds2.get_batch(64, "cpu")
    np.testing.assert_array_equal(b["xyz"].numpy(), golden["ds_batch_xyz"])
    np.testing.assert_array_equal(b["slice_idx"].numpy(), golden["ds_batch_idx"])
    assert [ds2.epoch, ds2.count] == golden["ds_epoch_count"].tolist()


def test_dataset_partial_batch_dropped(golden, oracle_backend):
    from nesvor_amd.train import Dataset

    ds = Dataset(_golden_slices(golden)[:4], small_args())
    M = ds.v.shape[0]
    bs = M // 2 + 1  # two batches do not fit -> second call must reshuffle, never return a short batch
    a = ds.get_batch(bs, "cpu")
    b = ds.get_batch(bs, "cpu")
    assert a["v"].shape[0] == bs and b["v"].shape[0] == bs and ds.epoch == 2


def test_grid_spec_and_inr_hyperparameters(golden):
    from nesvor_amd.grid import HashGridSpec
    from nesvor_amd.models import grid_hyperparameters
    from oracle import hashgrid as hg

    for ext, scale, n_levels, base in golden["inr_levels_table"]:
        a = small_args(level_scale=float(scale), finest_resolution=0.5)
        bb = torch.tensor([[0.0, 0, 0], [ext, ext * 0.8, ext * 0.5]])
        assert grid_hyperparameters(bb, a) == (int(base), int(n_levels))
    # defaults give L=12; --level-scale 1.26 gives L=16 for a 130 mm box
    bb = torch.tensor([[0.0, 0, 0], [130.0, 130, 130]])
    assert grid_hyperparameters(bb, small_args(finest_resolution=0.5))[1] == 12
    assert grid_hyperparameters(bb, small_args(finest_resolution=0.5, level_scale=1.26)) == (9, 16)
    spec = HashGridSpec(16, 2, 19, 9, 1.26)
    ref = hg.make_levels(16, 19, 9, 1.26)
    assert spec.n_params == 7854240 == hg.n_params(ref, 2)
    for a_, b_ in zip(spec.levels, ref):
        assert (a_.res, a_.size, a_.offset, a_.hashed) == (b_.res, b_.size, b_.offset, b_.hashed)
        assert np.float32(a_.scale) == np.float32(b_.scale)


def test_state_dict_layout_matches_reference(golden):
    """INR.state_dict() keys/shapes (checkpoint contract, cli/io.py:24-46) without touching the GPU."""
    from nesvor_amd.models import INR

    args = small_args()
    bb = _t(golden["fw_sd::inr.bounding_box"])
    inr = INR(bb, args)
    ref_keys = [str(k)[4:] for k in golden["fw_state_keys"] if str(k).startswith("inr.")]
    sd = inr.state_dict()
    assert list(sd.keys()) == ref_keys
    for k in ref_keys:
        assert tuple(sd[k].shape) == golden["fw_sd::inr." + k].shape, k


def test_moving_average_semantics():
    from nesvor_amd.utils import MovingAverage

    ema = MovingAverage(0.999)
    for x in (1.0, 2.0, 4.0):
        ema("k", x)
    v = ((1.0 * 0.001) * 0.999 + 2.0 * 0.001) * 0.999 + 4.0 * 0.001
    assert abs(ema["k"] - v / (1 - 0.999**3)) < 1e-12 and ema["missing"] == 0


def test_moving_average_update_all_matches_per_key_calls():
    from nesvor_amd.utils import MovingAverage

    a, b = MovingAverage(0.999), MovingAverage(0.999)
    g = torch.Generator().manual_seed(0)
    for _ in range(5):
        vals = {"x": torch.rand((), generator=g, dtype=torch.float64), "y": torch.rand((), generator=g, dtype=torch.float64)}
        for k, v in vals.items():
            a(k, v)
        b.update_all(vals)
    for k in ("x", "y"):
        assert abs(float(a[k]) - float(b[k])) < 1e-12
    assert a.value[0] == b.value[0] == 5


def test_cg_vs_scipy_reference_test():
    """tests/svort/test_cg.py:9-20 — CG on a 5x5 Hankel system against scipy.sparse.linalg.cg (pure host logic)."""
    import scipy.linalg
    import scipy.sparse.linalg

    from nesvor_amd.srr import CG

    n = 5
    A = scipy.linalg.hankel(np.arange(1, n + 1, dtype=np.float64)) + np.eye(n) * 20  # SPD, as the reference builds it
    b = np.arange(n, dtype=np.float64)
    x_ref, _ = scipy.sparse.linalg.cg(A, b, np.zeros_like(b), maxiter=n, atol=1e-12)
    At = torch.tensor(A)
    x = CG(lambda v: At @ v, torch.tensor(b), torch.zeros(n, dtype=torch.float64), n, 0.0)
    np.testing.assert_allclose(x.numpy(), x_ref, rtol=1e-8, atol=1e-10)
    # started at the exact solution the residual is exactly zero: no NaN (deterministic operators)
    x2 = CG(lambda v: At @ v, At @ torch.tensor(x_ref), torch.tensor(x_ref), 3, 0.0)
    assert torch.isfinite(x2).all()


def test_volume_resample_vs_reference_fixture(golden, oracle_backend):
    """``Volume.resample`` (image/image.py:134-177) as ``sample_volume`` uses it: the reference resampled the mask of
    its trained data set (slices at the optimised poses, ``train_out_tf``) to ``output_resolution`` = 2 mm; lattice
    shape, boolean mask and pose of the result are fixtures (``train_volume*``)."""
    from nesvor_amd.image import Slice
    from nesvor_amd.train import Dataset
    from nesvor_amd.transform import RigidTransform

    vs, res, res_s, s_thick, gap, n_slice, ss = golden["sim_geom"]
    imgs = _t(golden["sim_stacks"])
    tf = RigidTransform(_t(golden["train_out_tf"]), trans_first=True)
    slices = [Slice(imgs[k], imgs[k] > 0, tf[k], float(res_s), float(res_s), float(s_thick)) for k in range(imgs.shape[0])]
    args = small_args()
    out = Dataset(slices, args).mask.resample(args.output_resolution, None)
    assert tuple(out.image.shape) == golden["train_volume"].shape
    np.testing.assert_array_equal(out.mask.numpy(), golden["train_volume_mask"])
    np.testing.assert_allclose(out.transformation.matrix().numpy(), golden["train_volume_tf"], rtol=1e-6, atol=1e-5)
    assert float(out.resolution_x) == float(out.resolution_y) == float(out.resolution_z) == 2.0
    # an explicit target orientation: the lattice is rebuilt in that frame and still covers every masked voxel
    rot = RigidTransform(torch.tensor([[0.3, -0.2, 0.5, 0.0, 0.0, 0.0]]), trans_first=True)
    turned = out.resample(None, rot)
    np.testing.assert_allclose(turned.transformation.matrix()[0, :, :3].numpy(), rot.matrix()[0, :, :3].numpy(), atol=1e-6)
    back = turned.sample_points(out.xyz_masked)
    assert float((back > 0).float().mean()) > 0.95


def test_edge_prior_gradient_is_the_derivative_of_the_charbonnier_penalty():
    """``SRR.dR`` (svort/srr.py:134-160): for every interior voxel a, sum over the 26 neighbours o of
    d/dv_a sqrt(1 + (v_a - v_(a+o))^2 / (|o|^2 delta^2)) with the neighbour held fixed; border voxels get 0.
    Checked against autograd of exactly that expression in fp64."""
    from nesvor_amd.srr import SRR, edge_prior_gradient

    torch.manual_seed(0)
    v = torch.rand(2, 1, 6, 7, 8, dtype=torch.float64)
    delta = 0.3
    g = edge_prior_gradient(v, delta)
    va = v.clone().requires_grad_(True)
    D, H, W = v.shape[-3:]
    energy = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if (dz, dy, dx) == (0, 0, 0):
                    continue
                w = 1.0 / ((dz * dz + dy * dy + dx * dx) * delta * delta)
                nb = v[..., 1 + dz : D - 1 + dz, 1 + dy : H - 1 + dy, 1 + dx : W - 1 + dx]  # constant
                energy = energy + torch.sqrt(1 + w * (va[..., 1:-1, 1:-1, 1:-1] - nb) ** 2).sum()
    energy.backward()
    torch.testing.assert_close(g, va.grad, rtol=1e-12, atol=1e-12)
    assert float(g[..., 0, :, :].abs().max()) == 0.0 and float(g[..., :, :, -1].abs().max()) == 0.0
    torch.testing.assert_close(SRR.dR(v, delta), g)


def test_committed_bench_line_carries_the_contract_fields():
    """The JSON line `bench.py` printed for the committed round profile (profiles/r03_bench_n1.json, produced by
    tools/gpu_check.sh on the GPU box): every field the measurement contract names, with consistent arithmetic."""
    import json

    path = os.path.join(ROOT, "profiles", "r03_bench_n1.json")
    d = json.load(open(path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] == pytest.approx(1e3 / d["ms_per_step"], rel=1e-6)  # 2^20-point iterations per second on one GPU
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["launch_ms"] * 1e-3) / 1e9, rel=1e-6)
    assert 0 < r["traffic"] < r["algorithmic_bytes_per_launch"]  # PMC bytes: the table is cache-resident, the scatter merged on chip
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0


def test_queue_capacities_saved_by_one_process_are_loaded_by_the_next(tmp_path, monkeypatch):
    """``save_queue_scales`` / ``NESVOR_HASHGRID_QUEUE=load:<file>`` (encoding.QueueSizer): a sizer made under the load policy
    starts from the capacities written for its key (clamped to [START, 1]), any other key from START; no GPU involved."""
    import torch

    from nesvor_amd import encoding
    from nesvor_amd.encoding import QueueSizer
    from nesvor_amd.grid import HashGridSpec

    spec = HashGridSpec(16, 2, 19, 9, 1.26)
    dev = torch.device("cpu")
    monkeypatch.setattr(encoding, "_SIZERS", {})
    monkeypatch.setattr(QueueSizer, "policy", "adaptive")
    sz = encoding.queue_sizer(spec, 1 << 20, dev)
    assert all(abs(sz.scale[l] - QueueSizer.START) < 1e-9 for l in range(16))
    sz.scale[3], sz.scale[15] = 0.25, 1.0
    path = str(tmp_path / "scales.json")
    encoding.save_queue_scales(path)
    monkeypatch.setattr(encoding, "_SIZERS", {})
    monkeypatch.setattr(QueueSizer, "policy", "load:" + path)
    again = encoding.queue_sizer(spec, 1 << 20, dev)
    assert [round(float(again.scale[l]), 4) for l in (0, 3, 15)] == [round(QueueSizer.START, 4), 0.25, 1.0]
    other = encoding.queue_sizer(spec, 1 << 17, dev)  # another batch size: nothing saved for it
    assert all(abs(other.scale[l] - QueueSizer.START) < 1e-9 for l in range(16))
    monkeypatch.setattr(QueueSizer, "policy", "load:" + str(tmp_path / "missing.json"))
    monkeypatch.setattr(encoding, "_SIZERS", {})
    assert abs(encoding.queue_sizer(spec, 1 << 20, dev).scale[3] - QueueSizer.START) < 1e-9
