"""INR + NeSVoR imaging model: host-side mirror of ``nesvor.nesvor.models``
(nesvor/nesvor/models.py) on the MI355X-native ops.

Same constructor signatures, same ``args`` fields, same ``state_dict`` keys
(``bounding_box``, ``encoding.params``, ``density_net.{0,2,..}.{weight,bias}``…)
and the same loss-dict keys, so code written against the reference's model API
binds to this module unchanged.  The native backend is swapped in exactly where
the reference reaches tinycudann: ``build_encoding`` / ``build_network``.

Two execution paths share the parameters:
* ``NeSVoR.forward`` — autograd over the dispatcher ops (fused sampler, hash grid, fused MLPs,
  fused imaging loss): the parity surface for the loss dict and every gradient;
* ``nesvor_amd.fused`` / ``nesvor_amd.direct`` — the autograd-free training step used by ``train()``.

Networks outside the fused kernels' shapes (``--width`` > 64, ``--depth`` > 3, ...; the reference accepts
any, cli/main.py:68-73) keep every other stage on the HIP kernels and evaluate their matrix products on
library GEMMs (``nesvor_amd.mlp.apply_net``), with a warning about speed.
"""
from argparse import Namespace
from math import log2
from typing import Any, Dict, Optional, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import mlp as fused_mlp_mod
from . import tinycudann as tcnn
from .encoding import hashgrid_encode
from .loss import imaging_loss
from .sampler import psf_transform
from .transform import RigidTransform, ax_transform_points, axisangle2mat, mat_transform_points, trans_loss_fused
from .utils import resolution2sigma

# loss / regulariser keys (models.py:14-19)
D_LOSS = "MSE"
S_LOSS = "logVar"
DS_LOSS = "MSE+logVar"
B_REG = "biasReg"
T_REG = "transReg"
I_REG = "imageReg"


def build_encoding(**config):
    """models.py:22-25"""
    n_input_dims = config.pop("n_input_dims")
    dtype = config.pop("dtype")
    return tcnn.Encoding(n_input_dims=n_input_dims, encoding_config=config, dtype=dtype)


def build_network(**config):
    """models.py:28-69 — fp16: bias-free tcnn.Network; fp32: nn.Linear stack with biases."""
    dtype = config.pop("dtype")
    if dtype == torch.float16:
        return tcnn.Network(
            n_input_dims=config["n_input_dims"],
            n_output_dims=config["n_output_dims"],
            network_config={
                "otype": "CutlassMLP",
                "activation": config["activation"],
                "output_activation": config["output_activation"],
                "n_neurons": config["n_neurons"],
                "n_hidden_layers": config["n_hidden_layers"],
            },
        )
    if dtype != torch.float32:
        raise ValueError("unknown dtype")
    # any --width / --depth builds (cli/main.py:68-73): shapes the fused kernels cover (ReLU, width <= 64 - narrower
    # ones zero-padded, exact -, 1-3 hidden layers, <= 64 inputs, <= 16 outputs) run on them, the rest on library GEMMs
    # through nesvor_amd.mlp.apply_net, which says so once
    act = None if config["activation"] == "None" else getattr(nn, config["activation"])
    out_act = None if config["output_activation"] == "None" else getattr(nn, config["output_activation"])
    dims = [config["n_input_dims"]] + [config["n_neurons"]] * config["n_hidden_layers"] + [config["n_output_dims"]]
    layers = []
    for li in range(len(dims) - 1):
        if li > 0 and act is not None:
            layers.append(act())
        layers.append(nn.Linear(dims[li], dims[li + 1]))
    if out_act is not None:
        layers.append(out_act())
    return nn.Sequential(*layers)


def grid_hyperparameters(bounding_box: torch.Tensor, args: Namespace):
    """(base_resolution, n_levels) from the bounding box (models.py:79-101)."""
    extent = (bounding_box[1] - bounding_box[0]).max()
    base_resolution = int((extent / args.coarsest_resolution).ceil().int().item())
    n_levels = int(
        (torch.log2(extent / args.finest_resolution / base_resolution) / log2(args.level_scale) + 1).ceil().int().item()
    )
    return base_resolution, n_levels


class INR(nn.Module):
    def __init__(self, bounding_box: torch.Tensor, args: Namespace) -> None:
        super().__init__()
        self.register_buffer("bounding_box", bounding_box)
        base_resolution, n_levels = grid_hyperparameters(self.bounding_box, args)
        self.base_resolution, self.n_levels = base_resolution, n_levels
        self.encoding = build_encoding(
            n_input_dims=3,
            otype="HashGrid",
            n_levels=n_levels,
            n_features_per_level=args.n_features_per_level,
            log2_hashmap_size=args.log2_hashmap_size,
            base_resolution=base_resolution,
            per_level_scale=args.level_scale,
            dtype=args.dtype,
        )
        self.density_net = build_network(
            n_input_dims=n_levels * args.n_features_per_level,
            n_output_dims=1 + args.n_features_z,
            activation="ReLU",
            output_activation="None",
            n_neurons=args.width,
            n_hidden_layers=args.depth,
            dtype=args.dtype,
        )

    def forward(self, x: torch.Tensor, return_all: bool = True):
        x = (x - self.bounding_box[0]) / (self.bounding_box[1] - self.bounding_box[0])
        prefix_shape = x.shape[:-1]
        enc = self.encoding
        if (not torch.is_grad_enabled() and enc.dtype == torch.float16 and fused_mlp_mod.supported(self.density_net)):
            # half-precision model structure at inference (sample_volume / sample_slices): the same two kernels with
            # bf16 matrix operands; outputs stay fp32
            from .encoding import hashgrid_forward

            pe_fm = hashgrid_forward(enc.spec, x.reshape(-1, 3).float().contiguous(), enc.params, _lib.LAYOUT_FEATURE_MAJOR)
            net = fused_mlp_mod.NetParams(self.density_net)
            z_fm, _ = fused_mlp_mod.forward_raw(net.weights, net.biases, None, pe_fm, 0, pe_fm.shape[0], 1, False,
                                                bf16=fused_mlp_mod.HALF_OPERANDS[0])
        else:
            # feature-major hash grid -> fused MLP; (N,E)/(N,16) views are returned for API parity
            pe_fm = hashgrid_encode(x.reshape(-1, 3).float(), enc.params, enc.spec, _lib.LAYOUT_FEATURE_MAJOR, enc.grad_accum)
            z_fm = fused_mlp_mod.apply_net(self.density_net, None, pe_fm, 0, pe_fm.shape[0], 1)
        density = F.softplus(z_fm[0].view(prefix_shape))
        return (density, pe_fm.t(), z_fm.t()) if return_all else density

    def sample_batch(
        self,
        xyz: torch.Tensor,
        transformation: Optional[RigidTransform],
        psf_sigma: Union[float, torch.Tensor],
        n_samples: int,
    ) -> torch.Tensor:
        """(M,3) -> (M,S,3): Gaussian PSF cloud around each point, optionally moved by a rigid transform."""
        if n_samples > 1:
            if isinstance(psf_sigma, torch.Tensor):
                psf_sigma = psf_sigma.view(-1, 1, 3)
            noise = torch.randn(xyz.shape[0], n_samples, 3, dtype=xyz.dtype, device=xyz.device)
            xyz = xyz[:, None] + noise * psf_sigma
        else:
            xyz = xyz[:, None]
        if transformation is not None:
            tf = transformation.trans_first
            xyz = mat_transform_points(transformation.matrix(tf)[:, None], xyz, tf)
        return xyz


class NeSVoR(nn.Module):
    def __init__(
        self,
        transformation: RigidTransform,
        resolution: torch.Tensor,
        v_mean: float,
        bounding_box: torch.Tensor,
        args: Namespace,
    ) -> None:
        super().__init__()
        self.args = args
        self.n_slices = 0
        self.trans_first = True
        self.transformation = transformation
        self.psf_sigma = resolution2sigma(resolution, isotropic=False)
        self.delta = args.delta * v_mean
        self.image_regularization = {"TV": tv_reg, "edge": edge_reg, "L2": l2_reg}[args.image_regularization]
        self.build_network(bounding_box)
        self.to(args.device)
        self.psf_sigma = self.psf_sigma.to(args.device)

    @property
    def transformation(self) -> RigidTransform:
        return RigidTransform(self.axisangle.detach(), self.trans_first)

    @transformation.setter
    def transformation(self, value: RigidTransform) -> None:
        if self.n_slices == 0:
            self.n_slices = len(value)
        else:
            assert self.n_slices == len(value)
        axisangle = value.axisangle(self.trans_first).detach().clone()
        self.register_buffer("axisangle_init", axisangle.clone())
        if not self.args.no_transformation_optimization:
            self.axisangle = nn.Parameter(axisangle.clone())
        else:
            self.register_buffer("axisangle", axisangle.clone())

    def build_network(self, bounding_box) -> None:
        a = self.args
        if a.n_features_slice:
            self.slice_embedding = nn.Embedding(self.n_slices, a.n_features_slice)
        if not a.no_slice_scale:
            self.logit_coef = nn.Parameter(torch.zeros(self.n_slices, dtype=torch.float32))
        if not a.no_slice_variance:
            self.log_var_slice = nn.Parameter(torch.zeros(self.n_slices, dtype=torch.float32))
        self.inr = INR(bounding_box, a)
        head = dict(n_output_dims=1, activation="ReLU", output_activation="None", n_neurons=a.width,
                    n_hidden_layers=a.depth, dtype=a.dtype)
        if not a.no_pixel_variance:
            self.sigma_net = build_network(n_input_dims=a.n_features_slice + a.n_features_z, **head)
        if a.n_levels_bias:
            self.b_net = build_network(n_input_dims=a.n_levels_bias * a.n_features_per_level + a.n_features_slice, **head)

    def forward(self, xyz: torch.Tensor, v: torch.Tensor, slice_idx: torch.Tensor) -> Dict[str, Any]:
        """One batch of slice pixels -> dict of scalar losses (models.py:260-327)."""
        a = self.args
        B, S = xyz.shape[0], a.n_samples
        noise = torch.randn(B, S, 3, dtype=xyz.dtype, device=xyz.device)
        return self.forward_with_noise(xyz, v, slice_idx, noise)

    def forward_with_noise(self, xyz, v, slice_idx, noise) -> Dict[str, Any]:
        """``forward`` with the PSF draws handed in (B, S, 3): the fused sampler turns the per-slice matrices (n is a few
        hundred) into x and the box-normalised u in one launch, then hash grid + networks + fused imaging loss."""
        mat = axisangle2mat(self.axisangle)
        x, u = psf_transform(mat, slice_idx, xyz, self.psf_sigma, noise, self.inr.bounding_box)
        return self.fused_losses(x, u, v, slice_idx)

    def fused_losses(self, x, u, v, slice_idx) -> Dict[str, Any]:
        """Imaging model + losses (models.py:286-325) through the fused loss kernel; same dict, same order."""
        a = self.args
        z, log_var, log_bias = self.fused_outputs(u, slice_idx, x.shape[1])
        c = F.softmax(self.logit_coef, 0) * self.n_slices if not a.no_slice_scale else None
        lvs = self.log_var_slice if not a.no_slice_variance else None
        mse, logvar, ireg, breg = imaging_loss(z[0], log_var, log_bias, x, v, slice_idx, c, lvs,
                                               a.image_regularization, self.delta)
        losses = {D_LOSS: mse}
        if not (a.no_pixel_variance and a.no_slice_variance):
            losses[S_LOSS] = logvar
            losses[DS_LOSS] = mse + logvar
        if not a.no_transformation_optimization:
            losses[T_REG] = self.trans_loss(trans_first=self.trans_first)
        if a.n_levels_bias:
            losses[B_REG] = breg
        losses[I_REG] = ireg
        return losses

    def use_fused_mlp(self) -> bool:
        """Every network of the single-precision model inside the fused kernels' shapes (then the autograd-free step
        applies, nesvor_amd.direct); otherwise ``apply_net`` picks per network."""
        a = self.args
        nets = [self.inr.density_net]
        if not a.no_pixel_variance:
            nets.append(self.sigma_net)
        if a.n_levels_bias:
            nets.append(self.b_net)
        return (getattr(a, "fused_mlp", True) and a.dtype == torch.float32 and self.axisangle.is_cuda
                and all(fused_mlp_mod.supported(n) for n in nets))

    def fused_outputs(self, u: torch.Tensor, slice_idx: torch.Tensor, S: int):
        """Hash grid (feature-major) + the three fused MLPs -> raw network outputs:
        z (1 + n_features_z, N), log_var (N) | None, log_bias (N) | None.  No (N,E) row-major pe, no expanded
        slice embedding, no concatenated MLP inputs are materialised (models.py:329-355 builds all three)."""
        a = self.args
        inr = self.inr
        enc = inr.encoding
        pe = hashgrid_encode(u, enc.params, enc.spec, _lib.LAYOUT_FEATURE_MAJOR, enc.grad_accum)  # (E, N)
        z = fused_mlp_mod.apply_net(inr.density_net, None, pe, 0, pe.shape[0], S)
        se = self.slice_embedding(slice_idx) if a.n_features_slice else None  # (B, n_features_slice)
        log_bias = log_var = None
        if a.n_levels_bias:
            kb = a.n_levels_bias * a.n_features_per_level
            log_bias = fused_mlp_mod.apply_net(self.b_net, se, pe, 0, kb, S)[0]
        if not a.no_pixel_variance:
            log_var = fused_mlp_mod.apply_net(self.sigma_net, se, z, 1, a.n_features_z, S)[0]
        return z, log_var, log_bias

    def trans_loss(self, trans_first: bool = True) -> torch.Tensor:
        if trans_first and self.axisangle.is_cuda and self.axisangle.dtype == torch.float32 and getattr(self.args, "fused_mlp", True):
            return trans_loss_fused(self.axisangle, self.axisangle_init)
        cur = RigidTransform(self.axisangle, trans_first=trans_first)
        init = RigidTransform(self.axisangle_init, trans_first=trans_first)
        err = init.inv().compose(cur).axisangle(trans_first=trans_first)
        return torch.mean(err[:, :3] ** 2) + 1e-3 * torch.mean(err[:, 3:] ** 2)


def _pair_diffs(density: torch.Tensor, xyz: torch.Tensor):
    """Sample j of a pixel is paired with sample S-1-j (models.py:367-368)."""
    d_density = density - torch.flip(density, (1,))
    dx2 = ((xyz - torch.flip(xyz, (1,))) ** 2).sum(-1) + 1e-6
    return d_density, dx2


def tv_reg(density: torch.Tensor, xyz: torch.Tensor, delta: float):
    dd, dx2 = _pair_diffs(density, xyz)
    return torch.abs(dd / dx2.sqrt()).mean()


def edge_reg(density: torch.Tensor, xyz: torch.Tensor, delta: float):
    dd, dx2 = _pair_diffs(density, xyz)
    return delta * ((1 + dd**2 / dx2 / (delta * delta)).sqrt().mean() - 1)


def l2_reg(density: torch.Tensor, xyz: torch.Tensor, delta: float):
    dd, dx2 = _pair_diffs(density, xyz)
    return (dd**2 / dx2).mean()
