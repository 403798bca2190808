"""Inference from a trained INR: ``sample_volume`` / ``sample_points`` / ``sample_slice(s)`` with the call
signatures of ``nesvor.nesvor.sample`` (nesvor/nesvor/sample.py:10-64).

What the reference evaluates per chunk of query points (sample.py:22-32 -> models.py:154-174 -> models.py:142-152) -
draw a Gaussian cloud around every point, optionally move it by a rigid pose, normalise to the bounding box, encode,
run the density network, softplus, average the cloud - is done here by ``PsfAveragedDensity`` in four native launches
(fused sampler -> hash grid, feature-major -> fused MLP forward without activation save -> softplus + cloud mean);
no (M, S, 3) -> (M*S, E) row-major intermediates are built.
"""
from argparse import Namespace
from typing import List, Optional, Sequence, Union

import torch
import torch.nn.functional as F

from . import _lib, mlp as mlp_mod, sampler
from .encoding import hashgrid_forward, points_are_ordered
from .image import Slice, Volume
from .models import INR
from .transform import RigidTransform, transform_points
from .utils import meshgrid, resolution2sigma


class PsfAveragedDensity:
    _chunks = 0  # chunks evaluated so far in this process: the offset of the next chunk's noise stream

    """v(p) = mean_s softplus(density_net(encode(T (p + sigma * xi_s))))[0] for chunks of points.

    ``args.n_inference_samples`` cloud samples per point (one noiseless sample with ``args.no_output_psf``),
    ``args.inference_batch_size`` points per chunk (commands.py:94-98).  ``args.host_rng`` draws the noise from the
    host generator in the reference's order (models.py:161) - the replay mode the parity tests use."""

    def __init__(self, model: INR, args: Namespace) -> None:
        self.model, self.args = model, args
        self.device = model.bounding_box.device
        self.n_samples = 0 if args.no_output_psf else int(args.n_inference_samples)
        self.chunk = int(args.inference_batch_size)
        self.host_rng = bool(getattr(args, "host_rng", False))
        self.kernel_noise = __import__("os").environ.get("NESVOR_PSF_NOISE", "kernel") != "tensor"
        if self.device.type != "cuda":
            raise RuntimeError("inference runs on the HIP kernels: the INR must live on a HIP device (no CPU path)")
        self.operands = mlp_mod.inference_operands(model, args)  # None: a network the kernels do not cover (library GEMMs)
        if self.operands is None:
            return
        self.net = mlp_mod.NetParams(model.density_net)
        # only output 0 (the density logit) of the density network is used here: its other rows (the features of the
        # variance / bias networks) are neither computed into HBM nor stored
        self.weights = list(self.net.weights[:-1]) + [self.net.weights[-1][:1].contiguous()]
        self.biases = list(self.net.biases[:-1]) + [self.net.biases[-1][:1].contiguous()]

    def _noise(self, m: int, s: int):
        """-> (noise tensor | None, rng | None): explicit draws (no PSF: zeros; replay mode: the host generator's), or the
        (seed, offset) pair the sampler kernel draws from itself (one offset per chunk)."""
        if s <= 1:
            return torch.zeros((m, 1, 3), dtype=torch.float32, device=self.device), None
        if self.host_rng:
            return torch.randn(m, s, 3, dtype=torch.float32).to(self.device), None
        if not self.kernel_noise:
            return torch.randn(m, s, 3, dtype=torch.float32, device=self.device), None
        PsfAveragedDensity._chunks += 1
        return None, (torch.initial_seed() ^ 0x2545F4914F6CDD1D, PsfAveragedDensity._chunks)

    @torch.no_grad()
    def __call__(self, xyz: torch.Tensor, pose: Optional[RigidTransform], sigma: Union[float, Sequence, torch.Tensor]) -> torch.Tensor:
        model, dev = self.model, self.device
        xyz = xyz.reshape(-1, 3).to(device=dev, dtype=torch.float32).contiguous()
        out = torch.empty(xyz.shape[0], dtype=torch.float32, device=dev)
        if pose is None:
            mat = torch.eye(3, 4, dtype=torch.float32, device=dev)[None].contiguous()
        else:
            if len(pose) != 1:
                raise ValueError("one rigid pose per call")
            mat = pose.matrix(True).to(dev).contiguous()  # the sampler applies x = R (p + t): trans_first matrices
        sig = torch.as_tensor(sigma, dtype=torch.float32, device=dev).reshape(-1)
        sig = (sig.expand(3) if sig.numel() == 1 else sig).reshape(1, 3).contiguous()
        s = max(self.n_samples, 1)
        enc = model.encoding
        bb = model.bounding_box.contiguous()
        ordered = None  # (decided on the first chunk of a call with fewer than 128 samples per point)
        for begin in range(0, xyz.shape[0], self.chunk):
            pts = xyz[begin : begin + self.chunk]
            m = pts.shape[0]
            which = torch.zeros(m, dtype=torch.int64, device=dev)
            noise, rng = self._noise(m, s)
            _, u = sampler.forward_raw(mat, which, pts, sig, noise, bb, rng, s, need_x=False)
            if s < 128 and ordered is None:
                # one point per voxel (--no-output-psf, sample.py:29 of the reference): a raster-ordered voxel list is as good
                # as a batch of clouds for the per-cloud kernel (0.06 against 0.09 ms per 2^20 points), a shuffled one is not
                ordered = points_are_ordered(u)
            pe = hashgrid_forward(enc.spec, u, enc.params, _lib.LAYOUT_FEATURE_MAJOR, clustered=s >= 128 or bool(ordered))
            if self.operands is None:
                z = mlp_mod.apply_net(model.density_net, None, pe, 0, pe.shape[0], s)
            else:
                z, _ = mlp_mod.forward_raw(self.weights, self.biases, None, pe, 0, pe.shape[0], s, False, self.operands)
            out[begin : begin + m] = F.softplus(z[0].view(m, s)).mean(-1)
        return out


def sample_points(model: INR, xyz: torch.Tensor, args: Namespace) -> torch.Tensor:
    """PSF-averaged density at world points (any leading shape), isotropic output PSF (sample.py:17-33)."""
    sigma = resolution2sigma(args.output_resolution, isotropic=True)
    return PsfAveragedDensity(model, args)(xyz, None, sigma).view(xyz.shape[:-1])


def sample_volume(model: INR, mask: Volume, args: Namespace) -> Volume:
    """The INR on the mask's lattice at ``args.output_resolution``; voxels outside the mask stay 0 (sample.py:10-14)."""
    model.eval()
    out = mask.resample(args.output_resolution, None)
    out.image[out.mask] = sample_points(model, out.xyz_masked, args)
    return out


def sample_slice(model: INR, slice: Slice, mask: Volume, args: Namespace) -> Slice:
    """Simulate ``slice`` from the INR: its pixels inside the volume mask get the density averaged over the slice's own
    anisotropic PSF at the slice's pose; everything else is 0 and unmasked (sample.py:36-52)."""
    out = slice.clone(zero=True)
    lattice = meshgrid(out.shape_xyz, out.resolution_xyz).view(-1, 3)
    inside = mask.sample_points(transform_points(out.transformation, lattice)) > 0
    if bool(inside.any()):
        sigma = resolution2sigma(out.resolution_xyz, isotropic=False)
        values = PsfAveragedDensity(model, args)(lattice[inside], out.transformation, sigma)
        out.mask = inside.view(out.mask.shape)
        out.image[out.mask] = values.to(out.image.dtype)
    return out


def sample_slices(model: INR, slices: List[Slice], mask: Volume, args: Namespace) -> List[Slice]:
    model.eval()
    return [sample_slice(model, s, mask, args) for s in slices]
