"""GPU (-m gpu): end-to-end parity of the HIP training / inference path.

* the reference's own 20-iteration ``train()`` run and its ``sample_volume`` output (fixtures captured from the
  reference's Python, tests/golden/make_golden.py) replayed through the HIP path with the host random stream
  (``args.host_rng``);
* a shared-noise HIP-vs-oracle training run held to the north-star tolerance (PSNR within 0.1 dB);
* the BASELINE configurations that are not bench lines: C2 (the real L=16 / T=2^19 model end to end), C4 (per-slice
  motion, joint pose + INR optimisation), C5-shaped (0.5 mm grid, bias field with n_levels_bias=4).
"""
import math

import numpy as np
import pytest
import torch

from conftest import small_args

pytestmark = pytest.mark.gpu


def _golden_slices(golden, device, poses=None):
    from nesvor_amd.image import Slice
    from nesvor_amd.transform import RigidTransform

    vs, res, res_s, s_thick, gap, n_slice, ss = golden["sim_geom"]
    imgs = torch.tensor(golden["sim_stacks"]).to(device)
    tf = RigidTransform(torch.tensor(golden["sim_transforms"] if poses is None else poses).to(device), trans_first=True)
    return [Slice(imgs[k], imgs[k] > 0, tf[k], float(res_s), float(res_s), float(s_thick)) for k in range(imgs.shape[0])]


def _deviation(got, ref):
    """(max |diff| / max |ref|, fraction of entries off by more than 1e-3 of max |ref|)."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    scale = max(float(np.abs(ref).max()), 1e-12)
    d = np.abs(got - ref) / scale
    return float(d.max()), float((d > 1e-3).mean())


def test_train_replays_reference_trajectory(device, golden):
    """The reference's ``train()`` (nesvor/nesvor/train.py:123-232) ran 20 iterations on CPU from
    ``torch.manual_seed(0)`` (AdamW, lr decays at iterations 10 / 15 / 18); its final INR state_dict and slice poses
    are fixtures.  The HIP ``train()`` replays it with the same host random stream (initialisers, batch permutation,
    PSF noise).  Tolerance: fp32 with another summation order through 20 AdamW steps - AdamW divides by sqrt(v) + 1e-15,
    so a table entry whose tiny gradient changes sign under reordering can move by up to 2 x lr per step.  Measured on
    MI355X: hash table within 2.6e-4 of its range (no entry beyond 1e-3), MLP weights within 5e-7; asserted: every tensor
    within 2e-3 of its range, the MLPs within 1e-5."""
    from nesvor_amd.train import train
    from nesvor_amd.transform import RigidTransform

    args = small_args(device=device, n_iter=20, batch_size=64, n_samples=8, host_rng=True)
    torch.manual_seed(0)
    inr, out_slices, mask = train(_golden_slices(golden, device), args)
    sd = inr.state_dict()
    keys = [k[len("train_sd::"):] for k in golden.files if k.startswith("train_sd::")]
    assert sorted(keys) == sorted(sd.keys())
    report = {}
    for k in keys:
        ref = golden["train_sd::" + k]
        got = sd[k].detach().cpu().numpy()
        assert got.shape == ref.shape, k
        report[k] = _deviation(got, ref)
    print("train() replay: max deviation / outlier fraction per tensor:", report)
    for k, (dmax, frac) in report.items():
        assert dmax <= (2e-3 if k == "encoding.params" else 1e-5), (k, dmax, frac)
    np.testing.assert_allclose(np.asarray(sd["bounding_box"].cpu()), golden["train_sd::bounding_box"], rtol=1e-6, atol=1e-5)
    tf = RigidTransform.cat([s.transformation for s in out_slices]).matrix().cpu().numpy()
    np.testing.assert_allclose(tf, golden["train_out_tf"], rtol=1e-4, atol=2e-4)


def _trained_reference_inr(golden, device):
    from nesvor_amd.models import INR

    args = small_args(device=device)
    bb = torch.tensor(golden["train_sd::bounding_box"])
    inr = INR(bb, args)
    inr.load_state_dict({k[len("train_sd::"):]: torch.tensor(golden[k]) for k in golden.files if k.startswith("train_sd::")})
    return inr.to(device), args


def test_sample_volume_vs_reference_fixture(device, golden):
    """``sample_volume`` (nesvor/nesvor/sample.py:10-33, image/image.py:134-177) of the reference's trained INR on the
    reference's mask, ``torch.manual_seed(5)``: lattice shape, mask and pose exactly; intensities to fp32 tolerance
    (rtol 1e-4 of the volume's range: the 16 PSF samples per voxel are the reference's own draws)."""
    from nesvor_amd.sample import sample_volume
    from nesvor_amd.train import Dataset

    inr, args = _trained_reference_inr(golden, device)
    ds = Dataset(_golden_slices(golden, device, golden["train_out_tf"]), args)
    mask = ds.mask
    args.host_rng = True
    torch.manual_seed(5)
    vol = sample_volume(inr, mask, args)
    assert tuple(vol.image.shape) == golden["train_volume"].shape
    np.testing.assert_array_equal(vol.mask.cpu().numpy(), golden["train_volume_mask"])
    np.testing.assert_allclose(vol.transformation.matrix().cpu().numpy(), golden["train_volume_tf"], rtol=1e-6, atol=1e-5)
    ref = golden["train_volume"]
    np.testing.assert_allclose(vol.image.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * float(np.abs(ref).max()))
    # the module-level call the reference's API offers gives the same numbers as the fused sampler
    args.host_rng = False
    args.no_output_psf = True
    from nesvor_amd.sample import sample_points

    pts = vol.xyz_masked[:777]
    with torch.no_grad():
        direct = inr(pts[:, None], False).mean(-1)
    torch.testing.assert_close(sample_points(inr, pts, args), direct, rtol=1e-5, atol=1e-6)


def test_sample_slices_match_module_path(device, golden):
    """``sample_slices`` (sample.py:36-64): a slice simulated from the INR through the fused sampler equals the
    op-by-op evaluation (``INR.sample_batch`` + ``INR.forward``) on the same noise-free points; pixels outside the
    mask stay zero / unmasked."""
    from nesvor_amd.sample import sample_slices
    from nesvor_amd.train import Dataset
    from nesvor_amd.transform import transform_points
    from nesvor_amd.utils import meshgrid

    inr, args = _trained_reference_inr(golden, device)
    slices = _golden_slices(golden, device, golden["train_out_tf"])
    mask = Dataset(slices, args).mask
    args.no_output_psf = True
    picked = slices[5:8]
    out = sample_slices(inr, picked, mask, args)
    assert len(out) == 3
    for s_in, s_out in zip(picked, out):
        assert s_out.image.shape == s_in.image.shape and s_out.mask.dtype == torch.bool
        lattice = meshgrid(s_out.shape_xyz, s_out.resolution_xyz).view(-1, 3)
        world = transform_points(s_out.transformation, lattice)
        inside = (mask.sample_points(world) > 0).view(s_out.mask.shape)
        assert torch.equal(inside, s_out.mask)
        assert float(s_out.image[~s_out.mask].abs().max()) == 0.0
        with torch.no_grad():
            ref = inr(world[inside.view(-1)][:, None], False).mean(-1)
        torch.testing.assert_close(s_out.image[s_out.mask], ref, rtol=1e-5, atol=1e-6)


def _psnr(a, b, peak):
    return 10 * math.log10(peak**2 / float(((a - b) ** 2).mean()))


def _phantom_points(n, device):
    g = torch.arange(n, dtype=torch.float32, device=device) - (n - 1) / 2
    zz, yy, xx = torch.meshgrid(g, g, g, indexing="ij")
    return torch.stack([xx, yy, zz], -1).reshape(-1, 3)


def _fit_psnr(rec, truth):
    """Slice intensities were normalised by their 0.99 quantile: fit one global scale, PSNR over the object."""
    inside = truth > 0
    s = float((rec[inside] * truth[inside]).sum() / (rec[inside] ** 2).sum())
    return _psnr(rec[inside] * s, truth[inside], float(truth.max()))


def _eval_inr(inr, n, device, chunk=1 << 18):
    pts = _phantom_points(n, device)
    rec = torch.empty(pts.shape[0], device=device)
    with torch.no_grad():
        for i in range(0, pts.shape[0], chunk):
            rec[i : i + chunk] = inr(pts[i : i + chunk, None], False).mean(-1)
    return rec


def test_shared_noise_training_matches_oracle_within_0p1_db(device):
    """North-star parity statement, at a size the CPU oracle affords: the same 3-stack 32^3 phantom is reconstructed by
    the HIP ``train()`` and by the oracle's restatement of the reference loop FROM THE SAME RANDOM STREAM (host
    generator, seed 0: initialisers, permutations, PSF noise), default configuration incl. pose optimisation.
    * per-iteration losses: rtol 1e-4 over the first 10 iterations (identical parameters up to fp32 summation order;
      later the two fp32 trajectories separate - AdamW with eps 1e-15 amplifies sign flips of vanishing gradients - and
      the deviation is printed, not asserted);
    * reconstruction quality after 300 iterations: |PSNR(HIP) - PSNR(oracle)| <= 0.1 dB, both against the phantom."""
    from nesvor_amd.phantom import phantom3d, simulate_stacks
    from nesvor_amd.train import Dataset, train
    from oracle import nesvor_model as nm
    from oracle import train_loop as otl

    n = 32
    vol = torch.tensor(phantom3d(n=n), dtype=torch.float32, device=device)
    slices, _ = simulate_stacks(vol, n_stacks=3)
    args = small_args(device=device, n_iter=300, batch_size=512, n_samples=16, finest_resolution=1.0, log2_hashmap_size=14,
                      host_rng=True)
    ds = Dataset(slices, args)
    cds = otl.ArrayDataset(ds.xyz.cpu(), ds.v.cpu(), ds.slice_idx.cpu(), ds.transformation.matrix().cpu(), ds.resolution.cpu())
    hist = []
    torch.manual_seed(0)
    inr, _, _ = train(slices, args, on_iteration=lambda i, losses: hist.append(torch.stack([losses[k].detach() for k in losses])))
    keys = None
    torch.manual_seed(0)
    P, levels, bb, info = otl.train(cds, small_args(**{**vars(args), "device": torch.device("cpu")}))
    keys = list(info["history"][0].keys())
    got = torch.stack(hist).cpu().double().numpy()
    ref = np.array([[h[k] for k in keys] for h in info["history"]])
    assert got.shape == ref.shape
    rel = np.abs(got - ref) / (np.abs(ref) + 1e-7)
    print("loss deviation HIP vs oracle, max over keys, at iterations 1/10/50/100/300:",
          [float(rel[i - 1].max()) for i in (1, 10, 50, 100, 300)], keys)
    print("first 3 iterations, HIP :", got[:3].tolist())
    print("first 3 iterations, oracle:", ref[:3].tolist())
    # transReg starts at exactly 0 and imageReg = delta (mean sqrt(1 + eps) - 1) starts as a cancellation of order 1e-8
    # in fp32: both carry an absolute floor of 1e-6 next to the relative tolerance
    for j, k in enumerate(keys):
        tol = 1e-4 * np.abs(ref[:10, j]) + (1e-6 if k in ("transReg", "imageReg") else 1e-7)
        assert (np.abs(got[:10, j] - ref[:10, j]) <= tol).all(), (k, got[:10, j], ref[:10, j])
    truth = vol.reshape(-1)
    rec = _eval_inr(inr, n, device)
    with torch.no_grad():
        rec_o = nm.sample_points(P, levels, args, bb, _phantom_points(n, torch.device("cpu")), None, 0.0)
    p_hip, p_cpu = _fit_psnr(rec, truth), _fit_psnr(rec_o.to(device), truth)
    print(f"PSNR hip {p_hip:.3f} dB, cpu-oracle {p_cpu:.3f} dB")
    assert p_hip > 8.0 and abs(p_hip - p_cpu) <= 0.1


@pytest.mark.parametrize("width,depth", [(40, 1), (32, 2), (64, 3), (128, 1), (64, 4), (96, 5)])
def test_other_widths_and_depths_match_oracle_losses(device, width, depth):
    """``--width`` / ``--depth`` are free in the reference (cli/main.py:68-73).  Widths below 64 run zero-padded on the
    64-wide kernels (nesvor_amd.mlp.kernel_params: the same function, evaluated exactly), three hidden layers run on the
    separate dX / dW kernels; wider / deeper networks (128 x 1, 64 x 4, 96 x 5) run on the hand-written wide kernels
    (csrc/mlp_wide.hip, round 6; rounds 3-5: library GEMMs) - no network of this test may reach ``library_mlp``.  All train
    through the autograd path.  Held to the oracle's restatement of the reference loop from the same random stream: every loss of the
    first 10 iterations to rtol 1e-4."""
    from nesvor_amd.phantom import phantom3d, simulate_stacks
    from nesvor_amd.train import Dataset, train
    from oracle import train_loop as otl

    vol = torch.tensor(phantom3d(n=24), dtype=torch.float32, device=device)
    slices, _ = simulate_stacks(vol, n_stacks=3)
    args = small_args(device=device, n_iter=10, batch_size=256, n_samples=16, finest_resolution=1.0, log2_hashmap_size=14,
                      host_rng=True, width=width, depth=depth)
    ds = Dataset(slices, args)
    cds = otl.ArrayDataset(ds.xyz.cpu(), ds.v.cpu(), ds.slice_idx.cpu(), ds.transformation.matrix().cpu(), ds.resolution.cpu())
    hist = []
    torch.manual_seed(0)
    inr, _, _ = train(slices, args, on_iteration=lambda i, losses: hist.append(torch.stack([losses[k].detach() for k in losses])))
    assert [l.out_features for l in inr.density_net if hasattr(l, "out_features")][:-1] == [width] * depth
    from nesvor_amd import mlp as mlp_mod

    assert mlp_mod.supported(inr.density_net) or mlp_mod.wide_supported(inr.density_net)
    assert not mlp_mod._warned_library, mlp_mod._warned_library  # the library-GEMM fallback stayed unreachable
    torch.manual_seed(0)
    _, _, _, info = otl.train(cds, small_args(**{**vars(args), "device": torch.device("cpu")}))
    keys = list(info["history"][0].keys())
    got = torch.stack(hist).cpu().double().numpy()
    ref = np.array([[h[k] for k in keys] for h in info["history"]])
    assert got.shape == ref.shape
    for j, k in enumerate(keys):
        tol = 1e-4 * np.abs(ref[:, j]) + (1e-6 if k in ("transReg", "imageReg") else 1e-7)
        assert (np.abs(got[:, j] - ref[:, j]) <= tol).all(), (k, got[:, j], ref[:, j])
    # inference (raw kernel calls on the padded parameters) against the module path
    from nesvor_amd.sample import sample_points

    args.no_output_psf = True
    pts = inr.bounding_box[0] + (inr.bounding_box[1] - inr.bounding_box[0]) * torch.rand(4096, 3, device=device)
    with torch.no_grad():
        ref_v = inr(pts[:, None], False).mean(-1)
    torch.testing.assert_close(sample_points(inr, pts, args), ref_v, rtol=1e-5, atol=1e-6)


def test_config_c2_real_model_end_to_end(device):
    """BASELINE C2: 3 stacks of the 128^3 phantom (77 slices of 151^2 each), the real model - L=16 levels at scale 1.26
    down to 0.5 mm, T=2^19, two hidden layers of 64 - at B=1024 x S=256 = 2^18 samples per iteration, poses optimised.
    2000 iterations; the reconstruction evaluated at the phantom's voxel centres must reach the PSNR this configuration
    gives (15-17 dB: the error is dominated by the thin bright skull shell, which 3 mm slices do not resolve; floor
    15 dB) and all parameters stay finite."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_args
    from nesvor_amd.phantom import phantom3d, simulate_stacks
    from nesvor_amd.train import train

    n = 128
    vol = torch.tensor(phantom3d(n=n), dtype=torch.float32, device=device)
    torch.manual_seed(0)
    slices, _ = simulate_stacks(vol, n_stacks=3)
    assert len(slices) == 3 * 77 and tuple(slices[0].image.shape[-2:]) == (151, 151)
    args = make_args(device, 1024, 256, 2, 2000)
    inr, out_slices, mask = train(slices, args)
    assert inr.n_levels == 16 and inr.encoding.spec.levels[-1].size == 1 << 19
    assert all(torch.isfinite(p).all() for p in inr.parameters())
    p = _fit_psnr(_eval_inr(inr, n, device), vol.reshape(-1))
    print(f"C2 PSNR {p:.2f} dB")
    assert p >= 15.0


def _pose_errors(est, true):
    """Per-slice rotation (deg) and translation (mm) of true^-1 o est."""
    err = true.inv().compose(est).axisangle(True)
    return err[:, :3].norm(dim=-1) * 180 / math.pi, err[:, 3:].norm(dim=-1)


def test_config_c4_motion_pose_recovery(device):
    """BASELINE C4: every slice acquired at a perturbed pose (rotvec ~ N(0, (2 deg)^2), t ~ N(0, (1 mm)^2)), training
    starts from the unperturbed poses and optimises poses and INR jointly (models.py:193-210, 357-363).  64^3 phantom.
    * the mean pose error against the true poses shrinks (rotation and translation);
    * the reconstruction is not more than 0.5 dB below the one with poses frozen at the nominal values + 0, i.e. pose
      optimisation must pay for itself, and within 1.5 dB of the motion-free reconstruction."""
    from nesvor_amd.phantom import phantom3d, simulate_stacks
    from nesvor_amd.train import train
    from nesvor_amd.transform import RigidTransform

    n = 64
    vol = torch.tensor(phantom3d(n=n), dtype=torch.float32, device=device)
    truth = vol.reshape(-1)
    base = dict(device=device, n_iter=1500, batch_size=1024, n_samples=64, finest_resolution=0.5, level_scale=1.26,
                log2_hashmap_size=19, depth=2)
    results = {}
    for tag, motion, freeze in (("still", 0.0, False), ("motion", 2.0, False), ("motion_frozen", 2.0, True)):
        slices, true_tf = simulate_stacks(vol, n_stacks=3, motion_deg=motion, motion_mm=motion / 2, seed=0)
        args = small_args(**base, no_transformation_optimization=freeze)
        torch.manual_seed(0)
        inr, out_slices, _ = train(slices, args)
        est = RigidTransform.cat([s.transformation for s in out_slices])
        nominal = RigidTransform.cat([s.transformation for s in slices])
        results[tag] = dict(psnr=_fit_psnr(_eval_inr(inr, n, device), truth), before=_pose_errors(nominal, true_tf),
                            after=_pose_errors(est, true_tf))
    m = results["motion"]
    # slices that see the object (the outermost ones image empty space and carry no pose information)
    print({k: round(v["psnr"], 2) for k, v in results.items()},
          "rot deg %.3f -> %.3f, trans mm %.3f -> %.3f" % (m["before"][0].mean(), m["after"][0].mean(),
                                                           m["before"][1].mean(), m["after"][1].mean()))
    assert float(m["after"][0].mean()) < float(m["before"][0].mean())
    assert float(m["after"][1].mean()) < float(m["before"][1].mean())
    assert m["psnr"] >= results["motion_frozen"]["psnr"] - 0.1
    assert m["psnr"] >= results["still"]["psnr"] - 1.5


def test_config_c5_shape_bias_field_fine_grid(device):
    """BASELINE C5's model shape on one GPU: finest hash resolution and output resolution 0.5 mm, bias field on the 4
    coarsest levels (models.py:248-258, 341-346, 322-323), two hidden layers.  64^3 phantom, 600 iterations: losses and
    parameters finite, ``sample_volume`` at 0.5 mm produces a finite volume on the mask's lattice, and the reconstruction
    is within 0.5 dB of the same run without the bias field (the phantom has no bias: the field must stay neutral)."""
    from nesvor_amd.phantom import phantom3d, simulate_stacks
    from nesvor_amd.sample import sample_volume
    from nesvor_amd.train import train

    n = 64
    vol = torch.tensor(phantom3d(n=n), dtype=torch.float32, device=device)
    slices, _ = simulate_stacks(vol, n_stacks=3)
    psnr = {}
    for nb in (0, 4):
        args = small_args(device=device, n_iter=600, batch_size=1024, n_samples=64, finest_resolution=0.5, level_scale=1.26,
                          log2_hashmap_size=19, depth=2, n_levels_bias=nb, output_resolution=0.5)
        last = {}
        torch.manual_seed(0)
        inr, out_slices, mask = train(slices, args, on_iteration=lambda i, losses: last.update(losses))
        assert all(bool(torch.isfinite(v)) for v in last.values()), last
        assert ("biasReg" in last) == (nb > 0)
        assert all(torch.isfinite(p).all() for p in inr.parameters())
        psnr[nb] = _fit_psnr(_eval_inr(inr, n, device), vol.reshape(-1))
        if nb:
            out = sample_volume(inr, mask, args)
            assert abs(float(out.resolution_x) - 0.5) < 1e-6 and bool(torch.isfinite(out.image).all())
            assert float(out.image[out.mask].mean()) > 0 and float(out.image[~out.mask].abs().max()) == 0.0
    print(f"C5-shape PSNR without bias field {psnr[0]:.2f} dB, with n_levels_bias=4 {psnr[4]:.2f} dB")
    assert psnr[4] >= psnr[0] - 0.5
