"""CPU oracle: the NeSVoR imaging model, losses and rigid-transform algebra.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional PyTorch-CPU
restatement (autograd provides the backward) of

* ``INR.__init__`` level derivation            nesvor/nesvor/models.py:79-101
* ``INR.forward``                              models.py:142-152
* ``build_network`` fp32 branch (nn.Linear)    models.py:42-67
* ``NeSVoR.forward`` / ``net_forward``         models.py:260-355
* ``trans_loss``                               models.py:357-363
* ``tv_reg`` / ``edge_reg`` / ``l2_reg``       models.py:366-384
* ``RigidTransform.inv/.compose``              nesvor/transform/transform.py:44-63
* ``mat_transform_points`` (trans_first)       transform.py:259-271
* ``resolution2sigma``                         nesvor/utils/psf.py:5-34

Parameters live in a flat ``dict`` keyed like ``NeSVoR.state_dict()``.
"""
import math
from argparse import Namespace
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import hashgrid
from . import transform_convert as tc

GAUSSIAN_FWHM = 1 / (2 * math.sqrt(2 * math.log(2)))
SINC_FWHM = 1.206709128803223 * GAUSSIAN_FWHM

D_LOSS, S_LOSS, DS_LOSS = "MSE", "logVar", "MSE+logVar"
B_REG, T_REG, I_REG = "biasReg", "transReg", "imageReg"


# ---------------------------------------------------------------- transforms
class _Ax2Mat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ax):
        ctx.save_for_backward(ax)
        return tc.axisangle2mat_forward(ax)

    @staticmethod
    def backward(ctx, g):
        (ax,) = ctx.saved_tensors
        return tc.axisangle2mat_backward(g.contiguous(), ax)


class _Mat2Ax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mat):
        ctx.save_for_backward(mat)
        return tc.mat2axisangle_forward(mat)

    @staticmethod
    def backward(ctx, g):
        (mat,) = ctx.saved_tensors
        return tc.mat2axisangle_backward(mat, g.contiguous())


def axisangle2mat(ax):
    return _Ax2Mat.apply(ax)


def mat2axisangle(mat):
    return _Mat2Ax.apply(mat)


def mat_inv(mat):
    """x' = R (x + t)  ->  inverse [R^T | -R t]   (transform.py:44-49)."""
    R, t = mat[:, :, :3], mat[:, :, 3:]
    return torch.cat((R.transpose(-2, -1), -torch.matmul(R, t)), -1)


def mat_compose(m1, m2):
    """self=m1, other=m2 (transform.py:51-63): R = R1 R2, t = t2 + R2^T t1."""
    R1, t1, R2, t2 = m1[:, :, :3], m1[:, :, 3:], m2[:, :, :3], m2[:, :, 3:]
    return torch.cat((torch.matmul(R1, R2), t2 + torch.matmul(R2.transpose(-2, -1), t1)), -1)


def transform_points_trans_first(mat, x):
    """mat (*,3,4), x (*,3): R (x + t)   (transform.py:259-271)."""
    R, T = mat[..., :-1], mat[..., -1:]
    return torch.matmul(R, x[..., None] + T)[..., 0]


def resolution2sigma(res: torch.Tensor, isotropic=False):
    if isotropic:
        return res * GAUSSIAN_FWHM
    return res * torch.tensor([SINC_FWHM, SINC_FWHM, GAUSSIAN_FWHM], dtype=res.dtype)


# ---------------------------------------------------------------- networks
def grid_config(bounding_box: torch.Tensor, args: Namespace):
    """(base_resolution, n_levels) as INR.__init__ derives them (models.py:79-101)."""
    ext = (bounding_box[1] - bounding_box[0]).max()
    base_resolution = int((ext / args.coarsest_resolution).ceil().int().item())
    n_levels = int(
        (torch.log2(ext / args.finest_resolution / base_resolution) / math.log2(args.level_scale) + 1)
        .ceil()
        .int()
        .item()
    )
    return base_resolution, n_levels


def mlp_shapes(n_in, n_out, width, depth):
    """Layer (in,out) list of build_network's fp32 branch (models.py:53-64)."""
    if depth <= 0:
        return [(n_in, n_out)]
    dims = [n_in] + [width] * depth + [n_out]
    return list(zip(dims[:-1], dims[1:]))


def mlp_forward(P: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor, n_layers: int):
    """Linear(+bias) / ReLU stack; state-dict keys ``prefix.{0,2,4,..}.{weight,bias}``."""
    for i in range(n_layers):
        x = F.linear(x, P[f"{prefix}.{2 * i}.weight"], P[f"{prefix}.{2 * i}.bias"])
        if i < n_layers - 1:
            x = F.relu(x)
    return x


def init_params(n_slices, bounding_box, args, axisangle_init, seed=0):
    """Random-init parameter dict with the reference's state_dict names/shapes."""
    g = torch.Generator().manual_seed(seed)
    base, L = grid_config(bounding_box, args)
    levels = hashgrid.make_levels(L, args.log2_hashmap_size, base, args.level_scale)
    Fe = args.n_features_per_level
    P = {}
    P["inr.encoding.params"] = (torch.rand(hashgrid.n_params(levels, Fe), generator=g) * 2 - 1) * 1e-4

    def lin(prefix, shapes):
        for i, (a, b) in enumerate(shapes):
            k = 1 / math.sqrt(a)
            P[f"{prefix}.{2 * i}.weight"] = (torch.rand(b, a, generator=g) * 2 - 1) * k
            P[f"{prefix}.{2 * i}.bias"] = (torch.rand(b, generator=g) * 2 - 1) * k

    lin("inr.density_net", mlp_shapes(L * Fe, 1 + args.n_features_z, args.width, args.depth))
    if not args.no_pixel_variance:
        lin("sigma_net", mlp_shapes(args.n_features_slice + args.n_features_z, 1, args.width, args.depth))
    if args.n_levels_bias:
        lin("b_net", mlp_shapes(args.n_levels_bias * Fe + args.n_features_slice, 1, args.width, args.depth))
    if args.n_features_slice:
        P["slice_embedding.weight"] = torch.randn(n_slices, args.n_features_slice, generator=g)
    if not args.no_slice_scale:
        P["logit_coef"] = torch.zeros(n_slices)
    if not args.no_slice_variance:
        P["log_var_slice"] = torch.zeros(n_slices)
    P["axisangle"] = axisangle_init.clone()
    return P, levels


def inr_forward(P, levels, args, bounding_box, x):
    """models.py:142-152 — returns density, pe, z for x (...,3) in mm."""
    u = (x - bounding_box[0]) / (bounding_box[1] - bounding_box[0])
    shape = u.shape[:-1]
    pe = hashgrid.encode(u.reshape(-1, 3), P["inr.encoding.params"], levels, args.n_features_per_level)
    z = mlp_forward(P, "inr.density_net", pe, args.depth + 1)
    density = F.softplus(z[..., 0].view(shape))
    return density, pe, z


def edge_reg(density, xyz, delta):
    dd = density - torch.flip(density, (1,))
    dx2 = ((xyz - torch.flip(xyz, (1,))) ** 2).sum(-1) + 1e-6
    return delta * ((1 + dd**2 / dx2 / (delta * delta)).sqrt().mean() - 1)


def tv_reg(density, xyz, delta):
    dd = density - torch.flip(density, (1,))
    dx2 = ((xyz - torch.flip(xyz, (1,))) ** 2).sum(-1) + 1e-6
    return torch.abs(dd / dx2.sqrt()).mean()


def l2_reg(density, xyz, delta):
    dd = density - torch.flip(density, (1,))
    dx2 = ((xyz - torch.flip(xyz, (1,))) ** 2).sum(-1) + 1e-6
    return (dd**2 / dx2).mean()


IMAGE_REG = {"edge": edge_reg, "TV": tv_reg, "L2": l2_reg}


def trans_loss(axisangle, axisangle_init):
    """models.py:357-363."""
    x = axisangle2mat(axisangle)
    y = axisangle2mat(axisangle_init)
    err = mat2axisangle(mat_compose(mat_inv(y), x))
    return torch.mean(err[:, :3] ** 2) + 1e-3 * torch.mean(err[:, 3:] ** 2)


def nesvor_forward(
    P: Dict[str, torch.Tensor],
    levels,
    args: Namespace,
    bounding_box: torch.Tensor,
    psf_sigma: torch.Tensor,  # (n,3)
    axisangle_init: torch.Tensor,  # (n,6)
    delta: float,  # args.delta * v_mean
    xyz: torch.Tensor,  # (B,3)
    v: torch.Tensor,  # (B,)
    slice_idx: torch.Tensor,  # (B,) int64
    noise: torch.Tensor,  # (B,S,3) standard normal  (torch.randn at models.py:269)
    return_aux: bool = False,
):
    """NeSVoR.forward (models.py:260-327) with the PSF noise passed in."""
    n = psf_sigma.shape[0]
    S = noise.shape[1]
    sig = psf_sigma[slice_idx][:, None]
    t = P["axisangle"][slice_idx]
    mat = axisangle2mat(t)[:, None]  # (B,1,3,4)
    x = transform_points_trans_first(mat, xyz[:, None] + noise * sig)  # (B,S,3)
    se = None
    if args.n_features_slice:
        se = P["slice_embedding.weight"][slice_idx][:, None].expand(-1, S, -1)
    density, pe, z = inr_forward(P, levels, args, bounding_box, x)
    zs = []
    if se is not None:
        zs.append(se.reshape(-1, se.shape[-1]))
    log_bias = None
    if args.n_levels_bias:
        pe_bias = pe[..., : args.n_levels_bias * args.n_features_per_level]
        log_bias = mlp_forward(P, "b_net", torch.cat(zs + [pe_bias], -1), args.depth + 1).view(density.shape)
    log_var = None
    if not args.no_pixel_variance:
        zs.append(z[..., 1:])
        log_var = mlp_forward(P, "sigma_net", torch.cat(zs, -1), args.depth + 1).view(density.shape)
    if log_bias is not None:
        bias = log_bias.exp()
        bias_detach = bias.detach()
    else:
        bias, bias_detach = 1, 1
    var = log_var.exp() if log_var is not None else 1
    if not args.no_slice_scale:
        c = F.softmax(P["logit_coef"], 0)[slice_idx] * n
    else:
        c = 1
    v_out = c * (bias * density).mean(-1)
    if not args.no_pixel_variance:
        var = (bias_detach * var).mean(-1)
        var = (c.detach() if torch.is_tensor(c) else c) * var
        var = var**2
    if not args.no_slice_variance:
        var = var + P["log_var_slice"].exp()[slice_idx]
    losses = {D_LOSS: ((v_out - v) ** 2 / (2 * var)).mean()}
    if not (args.no_pixel_variance and args.no_slice_variance):
        losses[S_LOSS] = 0.5 * var.log().mean()
        losses[DS_LOSS] = losses[D_LOSS] + losses[S_LOSS]
    if not args.no_transformation_optimization:
        losses[T_REG] = trans_loss(P["axisangle"], axisangle_init)
    if args.n_levels_bias:
        losses[B_REG] = log_bias.mean() ** 2
    losses[I_REG] = IMAGE_REG[args.image_regularization](density, x, delta)
    if return_aux:
        return losses, {"x": x, "density": density, "pe": pe, "z": z, "log_var": log_var, "log_bias": log_bias, "v_out": v_out, "var": var}
    return losses


def total_loss(losses, args):
    w = {D_LOSS: 1, S_LOSS: 1, T_REG: args.weight_transformation, B_REG: args.weight_bias, I_REG: args.weight_image}
    tot = 0
    for k, val in losses.items():
        if k in w and w[k]:
            tot = tot + w[k] * val
    return tot


def sample_points(P, levels, args, bounding_box, xyz, noise: Optional[torch.Tensor], sigma_iso: float):
    """sample.py:17-33 + INR.sample_batch (models.py:154-174), no transformation."""
    if noise is not None:
        x = xyz[:, None] + noise * sigma_iso
    else:
        x = xyz[:, None]
    density, _, _ = inr_forward(P, levels, args, bounding_box, x)
    return density.mean(-1)
