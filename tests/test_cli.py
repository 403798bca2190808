"""Command-line surface: flag names / defaults of the reference's parser, and an end-to-end
`reconstruct` -> `sample-volume` -> `sample-slices` run on NIfTI stacks written to disk."""
import os

import numpy as np
import pytest
import torch


def test_parser_flags_and_defaults_match_reference_table():
    """cli/main.py:27-326 as tabulated in SURVEY.md §8d."""
    from nesvor_amd.cli import build_parser

    a = build_parser().parse_args(["reconstruct", "--input-slices", "x", "--output-volume", "v.nii.gz"])
    expect = dict(
        n_features_per_level=2, log2_hashmap_size=19, level_scale=1.3819, coarsest_resolution=16.0, finest_resolution=0.5,
        n_levels_bias=0, depth=1, width=64, n_features_z=15, n_features_slice=16, no_transformation_optimization=False,
        no_slice_scale=False, no_pixel_variance=False, no_slice_variance=False, single_precision=False,
        weight_transformation=0.1, weight_bias=100.0, image_regularization="edge", weight_image=2.0, delta=0.2,
        learning_rate=5e-3, gamma=0.33, milestones=[0.5, 0.75, 0.9], n_iter=6000, batch_size=4096, n_samples=256,
        output_resolution=0.8, output_intensity_mean=700.0, mask_threshold=1.0, no_output_psf=False,
        inference_batch_size=None, n_inference_samples=None, svort_version="v1",
    )
    for k, v in expect.items():
        assert getattr(a, k) == v, k
    b = build_parser().parse_args(["sample-volume", "--input-model", "m.pt", "--output-volume", "v.nii.gz"])
    assert b.output_resolution == 0.8 and b.output_intensity_mean == 700.0
    with pytest.raises(SystemExit):
        build_parser().parse_args(["sample-slices", "--input-model", "m.pt"])  # --input-slices / --simulated-slices required


def test_merge_args_new_overrides_old():
    from argparse import Namespace

    from nesvor_amd.cli import merge_args

    m = merge_args(Namespace(a=1, b=2), Namespace(b=3, c=4))
    assert (m.a, m.b, m.c) == (1, 3, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", [["--single-precision"], []])  # [] = the reference's default half-precision structure
def test_cli_reconstruct_sample_roundtrip(tmp_path, device, precision):
    from nesvor_amd import cli
    from nesvor_amd.image import Volume
    from nesvor_amd.image_io import load_slices, load_volume
    from nesvor_amd.phantom import phantom3d, simulate_stacks, stack_geometry
    from nesvor_amd.transform import RigidTransform

    vs, res_s, gap = 32, 1.5, 3.0
    vol = torch.tensor(phantom3d(n=vs), dtype=torch.float32, device=device)
    torch.manual_seed(0)
    slices, _ = simulate_stacks(vol, n_stacks=3, res_s=res_s, s_thick=gap, normalize=False)
    n_slice, _ = stack_geometry(vs, 1.0, res_s, gap)
    paths = []
    for i in range(3):
        ss = slices[i * n_slice : (i + 1) * n_slice]
        img = torch.cat([s.image for s in ss], 0)  # (n, h, w)
        ax = torch.cat([s.transformation.axisangle() for s in ss], 0).mean(0, keepdim=True)  # stack centre pose
        p = str(tmp_path / f"stack{i}.nii.gz")
        Volume(img, img > 0, RigidTransform(ax), res_s, res_s, gap).save(p, masked=False)
        paths.append(p)
    out_vol, out_model = str(tmp_path / "recon.nii.gz"), str(tmp_path / "model.pt")
    small = [*precision, "--n-iter", "80", "--batch-size", "512", "--n-samples", "32", "--log2-hashmap-size", "12",
             "--finest-resolution", "2.0", "--output-resolution", "2.0", "--seed", "0", "--verbose", "0"]
    cli.main(["reconstruct", "--input-stacks", *paths, "--thicknesses", "3", "3", "3", "--output-volume", out_vol,
              "--output-model", out_model, "--output-slices", str(tmp_path / "out_slices"),
              "--simulated-slices", str(tmp_path / "sim_slices"), *small])
    v = load_volume(out_vol, device=device)
    assert v.image.ndim == 3 and abs(v.resolution_x - 2.0) < 1e-3 and torch.isfinite(v.image).all()
    assert abs(float(v.image[v.mask].mean()) - 700.0) < 1.0  # --output-intensity-mean
    n_out = len(load_slices(str(tmp_path / "out_slices"), device))
    assert n_out == len(load_slices(str(tmp_path / "sim_slices"), device)) > 20
    out2 = str(tmp_path / "resampled.nii.gz")
    cli.main(["sample-volume", "--input-model", out_model, "--output-volume", out2, "--output-resolution", "2.0", "--verbose", "0"])
    v2 = load_volume(out2, device=device)
    assert v2.image.shape == v.image.shape
    a, b = v.image.flatten(), v2.image.flatten()
    assert float(torch.corrcoef(torch.stack([a, b]))[0, 1]) > 0.98  # same model, different PSF noise
    cli.main(["sample-slices", "--input-model", out_model, "--input-slices", str(tmp_path / "out_slices"),
              "--simulated-slices", str(tmp_path / "sim2"), "--verbose", "0"])
    assert len(os.listdir(str(tmp_path / "sim2"))) == n_out
    with pytest.raises(NotImplementedError):
        cli.main(["reconstruct", "--input-stacks", *paths, "--registration", "svort", "--output-volume", out_vol, *small])
    if precision:  # --registration stack on stacks that are already aligned: the reconstruction must stay as good
        out3 = str(tmp_path / "recon_stackreg.nii.gz")
        def fit(out_dir, sim_dir):  # correlation between the acquired slices and the slices simulated from the model
            a = torch.cat([s.image[s.mask] for s in load_slices(out_dir, device)])
            b = torch.cat([s.image[m.mask] for s, m in zip(load_slices(sim_dir, device), load_slices(out_dir, device))])
            return float(torch.corrcoef(torch.stack([a, b]))[0, 1])

        cli.main(["reconstruct", "--input-stacks", *paths, "--thicknesses", "3", "3", "3", "--registration", "stack",
                  "--output-volume", out3, "--output-slices", str(tmp_path / "out_slices3"),
                  "--simulated-slices", str(tmp_path / "sim_slices3"), *small])
        v3 = load_volume(out3, device=device)
        assert torch.isfinite(v3.image).all() and abs(float(v3.image[v3.mask].mean()) - 700.0) < 1.0
        c_none, c_stack = fit(str(tmp_path / "out_slices"), str(tmp_path / "sim_slices")), fit(str(tmp_path / "out_slices3"), str(tmp_path / "sim_slices3"))
        print(f"slice fit (correlation): --registration none {c_none:.4f}, stack {c_stack:.4f}")
        # the `register` command alone: stacks -> registered slices on disk, usable as --input-slices
        cli.main(["register", "--input-stacks", *paths, "--thicknesses", "3", "3", "3", "--output-slices", str(tmp_path / "reg_slices"),
                  "--verbose", "0"])
        assert len(load_slices(str(tmp_path / "reg_slices"), device)) == n_out
        assert c_stack > c_none - 0.03  # a mis-registered stack could not be fitted by one volume
