"""Generate the golden fixtures in tests/golden/*.npz from the REFERENCE's Python.

Run in the build container only (needs /root/reference, never runs on the GPU box):

    python tests/golden/make_golden.py

The reference's Python-level code (nesvor.transform, nesvor.utils, nesvor.image,
nesvor.slice_acquisition, nesvor.nesvor.{models,train,sample}) is imported with
its three native back ends replaced by stub modules built from this repo's CPU
oracle (never import the reference un-stubbed on ROCm):

    nesvor.transform_convert_cuda  <- oracle.transform_convert
    nesvor.slice_acq_cuda          <- oracle.slice_acq
    tinycudann                     <- oracle.hashgrid
    nibabel                        <- empty stub (NIfTI I/O is not exercised)

What the fixtures pin: everything the reference computes *in Python* around those
back ends — RigidTransform algebra, PSF/blur/meshgrid utilities, Dataset fields /
bounding box / mean / mask, INR hyper-parameter derivation, state_dict layout,
NeSVoR.forward's loss dict and gradients, the AdamW/MultiStepLR trajectory of
train(), sample_volume.  Only inputs and outputs are stored; no reference source.
"""
import os
import sys
import types
from argparse import Namespace

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.nn as nn

from oracle import hashgrid as o_hg
from oracle import slice_acq as o_sa
from oracle import transform_convert as o_tc

REF = "/root/reference"


# --------------------------------------------------------------------- stubs
def install_stubs():
    tcm = types.ModuleType("nesvor.transform_convert_cuda")
    tcm.axisangle2mat_forward = lambda ax: [o_tc.axisangle2mat_forward(ax)]
    tcm.axisangle2mat_backward = lambda g, ax: [o_tc.axisangle2mat_backward(g, ax)]
    tcm.mat2axisangle_forward = lambda m: [o_tc.mat2axisangle_forward(m)]
    tcm.mat2axisangle_backward = lambda m, g: [o_tc.mat2axisangle_backward(m, g)]
    sam = types.ModuleType("nesvor.slice_acq_cuda")

    def sa_forward(transforms, vol, vol_mask, slices_mask, psf, slice_shape, res_slice, need_weight, interp_psf):
        vm = vol_mask if vol_mask.numel() > 0 else None
        sm = slices_mask if slices_mask.numel() > 0 else None
        out = o_sa.slice_acquisition_forward(transforms, vol, vm, sm, psf, slice_shape, res_slice, need_weight, interp_psf)
        return list(out) if need_weight else [out]

    sam.forward = sa_forward
    tcnn = types.ModuleType("tinycudann")

    class Encoding(nn.Module):
        def __init__(self, n_input_dims, encoding_config, dtype=torch.float32):
            super().__init__()
            c = encoding_config
            self.levels = o_hg.make_levels(c["n_levels"], c["log2_hashmap_size"], c["base_resolution"], c["per_level_scale"])
            self.F = c["n_features_per_level"]
            g = torch.Generator().manual_seed(1337)
            self.params = nn.Parameter((torch.rand(o_hg.n_params(self.levels, self.F), generator=g) * 2 - 1) * 1e-4)

        def forward(self, x):
            return o_hg.encode(x, self.params, self.levels, self.F)

    tcnn.Encoding = Encoding
    nib = types.ModuleType("nibabel")
    nib.nifti1 = types.ModuleType("nibabel.nifti1")
    sys.modules["nesvor.transform_convert_cuda"] = tcm
    sys.modules["nesvor.slice_acq_cuda"] = sam
    sys.modules["tinycudann"] = tcnn
    sys.modules["nibabel"] = nib
    sys.modules["nibabel.nifti1"] = nib.nifti1
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "tests"))


def small_args(**over):
    a = Namespace(
        n_features_per_level=2, log2_hashmap_size=12, level_scale=1.3819, coarsest_resolution=16.0,
        finest_resolution=2.0, n_levels_bias=0, depth=1, width=64, n_features_z=15, n_features_slice=16,
        no_transformation_optimization=False, no_slice_scale=False, no_pixel_variance=False,
        no_slice_variance=False, single_precision=True, weight_transformation=0.1, weight_bias=100.0,
        image_regularization="edge", weight_image=2.0, delta=0.2, learning_rate=5e-3, gamma=0.33,
        milestones=[0.5, 0.75, 0.9], n_iter=20, batch_size=64, n_samples=8, output_resolution=2.0,
        output_intensity_mean=700.0, mask_threshold=1.0, no_output_psf=False, debug=False,
        device=torch.device("cpu"), dtype=torch.float32,
    )
    for k, v in over.items():
        setattr(a, k, v)
    a.inference_batch_size = 8 * a.batch_size
    a.n_inference_samples = 2 * a.n_samples
    return a


def np_(t):
    return t.detach().cpu().numpy()


def main():
    install_stubs()
    import phantom3d as ref_phantom
    from nesvor.image import Slice, Volume
    from nesvor.nesvor import models as ref_models
    from nesvor.nesvor import sample as ref_sample
    from nesvor.nesvor import train as ref_train
    from nesvor.slice_acquisition import slice_acquisition
    from nesvor.transform import (RigidTransform, ax_transform_points, euler2mat, mat2euler, mat2point,
                                  mat_update_resolution, point2mat, transform_points)
    from nesvor.utils import gaussian_blur, get_PSF, meshgrid, resolution2sigma

    out = {}
    # ---- (6) RigidTransform algebra on the reference's 11-case table -------
    ax = torch.tensor(
        [[0, 0, 0, 0, 0, 0], [np.pi / 2, 0, 0, 1, 2, 3], [0, -np.pi / 2, 0, -1.1, -10, 100.5],
         [0, 0, np.pi - 0.01, 2, 1, 10.5], [0, -np.pi + 0.01, 0, 2, 1, 10.5], [0.1, 0.1, 0.1, 0.1, 0.1, 0.1],
         [-0.1, 0, -0.4, 0.1, 0.5, 0.1], [-0.2, 0.2, -0.1, -100, 200, -159], [-0.12, -0.01, 0.1, -100, 200, -159],
         [np.pi / 4, np.pi / 4, np.pi / 4, 0.1, 0.1, 0.1], [np.pi / 3, -np.pi / 4, np.pi / 5, 100, 200, -300]],
        dtype=torch.float32)
    A = RigidTransform(ax, trans_first=True)
    Bt = RigidTransform(ax.flip(0).clone(), trans_first=False)
    out["tf_ax"] = np_(ax)
    out["tf_mat_first"] = np_(A.matrix(True))
    out["tf_mat_last"] = np_(A.matrix(False))
    out["tf_ax_last"] = np_(A.axisangle(False))
    out["tf_inv"] = np_(A.inv().matrix(True))
    out["tf_compose"] = np_(A.compose(Bt).matrix(True))
    out["tf_B_first_ax"] = np_(Bt.axisangle(True))
    pts = torch.linspace(-30, 30, 33).view(11, 3)
    out["tf_pts"] = np_(pts)
    out["tf_pts_out"] = np_(transform_points(A, pts))
    out["tf_euler"] = np_(mat2euler(A.matrix(True)))
    out["tf_euler2mat"] = np_(euler2mat(mat2euler(A.matrix(True))))
    out["tf_mat2point"] = np_(mat2point(A.matrix(True), 128, 96, 0.8))
    out["tf_point2mat"] = np_(point2mat(mat2point(A.matrix(True), 128, 96, 0.8)))
    out["tf_update_res"] = np_(mat_update_resolution(A.matrix(True), 1.0, 0.8))

    # ---- (4) utils ----------------------------------------------------------
    psf = get_PSF(res_ratio=(1.5, 1.5, 3.0))
    out["psf_15_15_3"] = np_(psf)
    out["psf_1_1_1"] = np_(get_PSF(res_ratio=(1.0, 1.0, 1.0)))
    out["sigma_aniso"] = np_(resolution2sigma(torch.tensor([[1.5, 1.5, 3.0], [0.8, 0.8, 0.8]])))
    out["sigma_iso"] = np.array(resolution2sigma(0.8, isotropic=True))
    gb_in = torch.zeros(1, 1, 9, 10, 11)
    gb_in[0, 0, 4, 5, 5] = 1.0
    gb_in[0, 0, 2, 2, 8] = 2.0
    out["blur_in"] = np_(gb_in)
    out["blur_out"] = np_(gaussian_blur(gb_in, 1.5, 3))
    out["meshgrid"] = np_(meshgrid((4, 3, 2), (1.5, 1.5, 3.0)))

    # ---- (5) phantom hashes -------------------------------------------------
    import hashlib
    for n in (32, 64, 128):
        arr = ref_phantom.phantom3d(n=n).astype(np.float32)
        out[f"phantom_sha1_{n}"] = np.frombuffer(hashlib.sha1(arr.tobytes()).digest(), dtype=np.uint8)

    # ---- (d) tiny 3-stack scenario through the reference's own wrappers ------
    vs, res, res_s, s_thick = 24, 1.0, 1.5, 3.0
    gap = s_thick
    n_slice = int((np.sqrt(3) * vs) / gap) + 4
    ss = int((np.sqrt(3) * vs) / res_s) + 4
    volume = torch.tensor(ref_phantom.phantom3d(n=vs), dtype=torch.float32)[None, None]
    angles = [[0, 0, 0], [np.pi / 2, 0, 0], [0, np.pi / 2, 0]]
    slices, stack_imgs, stack_tf = [], [], []
    for ang in angles:
        angle = torch.tensor([ang], dtype=torch.float32).expand(n_slice, -1)
        tz = (torch.arange(0, n_slice, dtype=torch.float32) - (n_slice - 1) / 2.0) * gap
        txy = torch.ones_like(tz) * 0.5
        tf = RigidTransform(torch.cat((angle, torch.stack((txy, txy, tz), -1)), -1), trans_first=True)
        mat = mat_update_resolution(tf.matrix(), 1, res)
        imgs = slice_acquisition(mat, volume, None, None, psf, (ss, ss), res_s / res, False, False)
        stack_imgs.append(imgs)
        stack_tf.append(tf.matrix())
        for k in range(n_slice):
            slices.append(Slice(imgs[k], imgs[k] > 0, tf[k], res_s, res_s, s_thick))
    out["sim_volume"] = np_(volume[0, 0])
    out["sim_stacks"] = np_(torch.cat(stack_imgs, 0))
    out["sim_transforms"] = np_(torch.cat(stack_tf, 0))
    out["sim_geom"] = np.array([vs, res, res_s, s_thick, gap, n_slice, ss])

    args = small_args()
    torch.manual_seed(0)
    ds = ref_train.Dataset(slices, args)
    out["ds_xyz"], out["ds_v"], out["ds_slice_idx"] = np_(ds.xyz), np_(ds.v), np_(ds.slice_idx)
    out["ds_transformation"] = np_(ds.transformation.matrix())
    out["ds_resolution"] = np_(ds.resolution)
    out["ds_bounding_box"] = np_(ds.bounding_box)
    out["ds_mean"] = np.array(ds.mean)
    m = ds.mask
    out["ds_mask"] = np_(m.mask)
    out["ds_mask_tf"] = np_(m.transformation.matrix())
    out["ds_mask_res"] = np.array(float(m.resolution_x))
    b1 = ds.get_batch(64, "cpu")
    out["ds_batch_xyz"], out["ds_batch_idx"] = np_(b1["xyz"]), np_(b1["slice_idx"])
    out["ds_epoch_count"] = np.array([ds.epoch, ds.count])

    # ---- (7) INR hyper-parameters for a table of boxes ----------------------
    rows = []
    for ext in (40.0, 64.0, 100.0, 130.0, 180.0, 240.0):
        for scale in (1.3819, 1.26):
            a = small_args(level_scale=scale, finest_resolution=0.5, log2_hashmap_size=4)
            inr = ref_models.INR(torch.tensor([[0.0, 0, 0], [ext, ext * 0.8, ext * 0.5]]), a)
            rows.append([ext, scale, len(inr.encoding.levels), int(round((inr.encoding.levels[0].scale + 1)))])
    out["inr_levels_table"] = np.array(rows)

    # ---- (1),(8) NeSVoR.forward loss dict + grads for fixed noise ------------
    for tag, over in (("", {}), ("_bias", {"n_levels_bias": 2, "depth": 2})):
        args = small_args(**over)
        torch.manual_seed(1)
        model = ref_models.NeSVoR(ds.transformation, ds.resolution, ds.mean, ds.bounding_box, args)
        with torch.no_grad():  # move parameters off their trivial init
            model.logit_coef.normal_(0, 0.1)
            model.log_var_slice.normal_(0, 0.1)
            model.axisangle.add_(torch.randn_like(model.axisangle) * 0.02)
            model.inr.encoding.params.mul_(2000.0)
        sd = model.state_dict()
        out[f"fw{tag}_state_keys"] = np.array(list(sd.keys()))
        for k, v in sd.items():
            out[f"fw{tag}_sd::{k}"] = np_(v)
        out[f"fw{tag}_psf_sigma"] = np_(model.psf_sigma)
        out[f"fw{tag}_delta"] = np.array(model.delta)
        B, S = 48, args.n_samples
        idx = torch.randperm(ds.xyz.shape[0])[:B]
        xyz, v, sidx = ds.xyz[idx], ds.v[idx], ds.slice_idx[idx]
        noise = torch.randn(B, S, 3)
        real_randn = torch.randn
        torch.randn = lambda *a, **k: noise.clone()
        try:
            losses = model(xyz, v, sidx)
        finally:
            torch.randn = real_randn
        w = {"MSE": 1, "logVar": 1, "transReg": args.weight_transformation, "biasReg": args.weight_bias,
             "imageReg": args.weight_image}
        total = sum(w[k] * losses[k] for k in losses if k in w and w[k])
        total.backward()
        out[f"fw{tag}_xyz"], out[f"fw{tag}_v"], out[f"fw{tag}_idx"], out[f"fw{tag}_noise"] = np_(xyz), np_(v), np_(sidx), np_(noise)
        out[f"fw{tag}_loss_keys"] = np.array(list(losses.keys()))
        out[f"fw{tag}_loss_vals"] = np.array([float(losses[k]) for k in losses])
        for n_, p in model.named_parameters():
            out[f"fw{tag}_grad::{n_}"] = np_(p.grad)

    # ---- (3) train(): AdamW + MultiStepLR trajectory + sample_volume ---------
    args = small_args(n_iter=20, batch_size=64, n_samples=8)
    torch.manual_seed(0)
    inr, out_slices, mask = ref_train.train(slices, args)
    for k, v in inr.state_dict().items():
        out[f"train_sd::{k}"] = np_(v)
    out["train_out_tf"] = np_(RigidTransform.cat([s.transformation for s in out_slices]).matrix())
    torch.manual_seed(5)
    vol = ref_sample.sample_volume(inr, mask, args)
    out["train_volume"] = np_(vol.image)
    out["train_volume_mask"] = np_(vol.mask)
    out["train_volume_tf"] = np_(vol.transformation.matrix())

    path = os.path.join(HERE, "reference_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024), len(out), "arrays")


if __name__ == "__main__":
    main()
