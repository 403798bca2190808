"""Small host-side utilities on the hot path: PSF model, meshgrid, separable
Gaussian blur, EMA of losses.  Mirrors ``nesvor.utils`` (psf.py, misc.py) —
pure PyTorch, device-agnostic, no native code involved.
"""
import collections
import math
from typing import Any, Collection, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

# FWHM -> sigma factors (nesvor/utils/psf.py:5-6): Gaussian through-plane, sinc-like in-plane
GAUSSIAN_FWHM = 1 / (2 * math.sqrt(2 * math.log(2)))
SINC_FWHM = 1.206709128803223 * GAUSSIAN_FWHM


def resolution2sigma(rx, ry=None, rz=None, /, isotropic=False):
    """Voxel size -> PSF sigma (psf.py:9-34).  Accepts a scalar, a 3-sequence,
    a (...,3) tensor, or three scalars."""
    fxy = GAUSSIAN_FWHM if isotropic else SINC_FWHM
    fz = GAUSSIAN_FWHM
    assert (ry is None) == (rz is None)
    if ry is not None:
        return fxy * rx, fxy * ry, fz * rz
    if isinstance(rx, (float, int)):
        return fxy * rx if isotropic else (fxy * rx, fxy * rx, fz * rx)
    if isinstance(rx, torch.Tensor):
        if isotropic:
            return fxy * rx
        assert rx.shape[-1] == 3
        return rx * torch.tensor([fxy, fxy, fz], dtype=rx.dtype, device=rx.device)
    if isinstance(rx, (list, tuple)):
        assert len(rx) == 3
        return resolution2sigma(rx[0], rx[1], rx[2], isotropic=isotropic)
    raise Exception(str(type(rx)))


def get_PSF(
    r_max: Optional[int] = None,
    res_ratio: Tuple[float, float, float] = (1, 1, 3),
    threshold: float = 1e-3,
    device=torch.device("cpu"),
) -> torch.Tensor:
    """Discrete anisotropic Gaussian PSF (d,h,w), thresholded, cropped to its
    support and normalised to sum 1 (psf.py:37-65)."""
    sig = resolution2sigma(res_ratio, isotropic=False)
    if r_max is None:
        r_max = max(max(int(2 * s + 1) for s in sig), 4)
    ax = torch.linspace(-r_max, r_max, 2 * r_max + 1, dtype=torch.float32, device=device)
    gz, gy, gx = torch.meshgrid(ax, ax, ax, indexing="ij")
    psf = torch.exp(-0.5 * (gx**2 / sig[0] ** 2 + gy**2 / sig[1] ** 2 + gz**2 / sig[2] ** 2))
    psf[psf.abs() < threshold] = 0
    lo = []
    for dims in ((0, 1), (0, 2), (1, 2)):  # first non-empty index along x, y, z
        lo.append(int(torch.nonzero(psf.sum(dims) > 0)[0, 0].item()))
    hi = [2 * r_max + 1 - v for v in lo]
    psf = psf[lo[2] : hi[2], lo[1] : hi[1], lo[0] : hi[0]].contiguous()
    return psf / psf.sum()


def meshgrid(
    shape_xyz: Collection,
    resolution_xyz: Collection,
    min_xyz: Optional[Collection] = None,
    device=None,
    stack_output: bool = True,
):
    """Regular grid of physical coordinates, indexed [z,y,x], last dim (x,y,z) (misc.py:29-60)."""
    assert len(shape_xyz) == len(resolution_xyz)
    if min_xyz is None:
        min_xyz = tuple(-(s - 1) * r / 2 for s, r in zip(shape_xyz, resolution_xyz))
    else:
        assert len(shape_xyz) == len(min_xyz)
    if device is None:
        if isinstance(shape_xyz, torch.Tensor):
            device = shape_xyz.device
        elif isinstance(resolution_xyz, torch.Tensor):
            device = resolution_xyz.device
        else:
            device = torch.device("cpu")
    axes = [
        torch.arange(s, dtype=torch.float32, device=device) * r + m
        for s, r, m in zip(shape_xyz, resolution_xyz, min_xyz)
    ]
    grids = torch.meshgrid(axes[::-1], indexing="ij")[::-1]
    return torch.stack(grids, -1) if stack_output else grids


def gaussian_1d_kernel(sigma: float, truncated: float, device) -> torch.Tensor:
    """erf-integrated Gaussian taps (misc.py:83-88)."""
    tail = int(max(sigma * truncated, 0.5) + 0.5)
    x = torch.arange(-tail, tail + 1, dtype=torch.float, device=device)
    t = 0.70710678 / sigma
    return (0.5 * ((t * (x + 0.5)).erf() - (t * (x - 0.5)).erf())).clamp(min=0)


def gaussian_blur(x: torch.Tensor, sigma, truncated: float) -> torch.Tensor:
    """Separable Gaussian blur of (N,C,*spatial), zero padding (misc.py:63-80).

    The reference runs one depthwise convolution per axis, and so does this function on the CPU (bit-identical to the
    reference's fixture there).  On a HIP device the first convolution of a process costs seconds of MIOpen kernel search
    (3 s for the volume mask at the end of ``train``: a third of a 6000-iteration reconstruction), so each axis is
    evaluated as what it is - a dozen shifted multiply-adds of the whole array: out[i] = sum_j k[j] x[i + j - r]."""
    nd = x.ndim - 2
    if not isinstance(sigma, collections.abc.Iterable):
        sigma = [sigma] * nd
    if not x.is_cuda:
        conv = [F.conv1d, F.conv2d, F.conv3d][nd - 1]
        c = x.shape[1]
        for d, s in enumerate(sigma):
            k = gaussian_1d_kernel(s, truncated, x.device)
            shape = [1] * x.ndim
            shape[d + 2] = -1
            k = k.reshape(shape).repeat(*([c, 1] + [1] * nd))
            pad = [0] * nd
            pad[d] = (k.shape[d + 2] - 1) // 2
            x = conv(x, k, padding=pad, groups=c)
        return x
    for d, s in enumerate(sigma):
        # (a few numbers: computed on the host, no device sync)
        x = _filter_axis(x, gaussian_1d_kernel(float(s), truncated, "cpu").tolist(), d + 2)
    return x


def _filter_axis(x: torch.Tensor, taps, dim: int) -> torch.Tensor:
    """out[i] = sum_j taps[j] x[i + j - r] along ``dim`` with zero padding (r = len(taps) // 2): a 1-D cross-correlation as
    len(taps) shifted multiply-adds of the whole array - no convolution library involved."""
    r = (len(taps) - 1) // 2
    n = x.shape[dim]
    out = torch.zeros_like(x)
    for j, w in enumerate(taps):
        shift = j - r
        lo, hi = max(0, -shift), min(n, n - shift)
        if lo < hi and w != 0.0:
            out.narrow(dim, lo, hi - lo).add_(x.narrow(dim, lo + shift, hi - lo), alpha=w)
    return out


def ncc_loss(I: torch.Tensor, J: torch.Tensor, mask: Optional[torch.Tensor] = None, win: Optional[int] = 9, level: int = 0,
             eps: float = 1e-6, reduction: str = "none") -> torch.Tensor:
    """Negative squared normalised cross-correlation of (N,C,*spatial) images (nesvor/utils/loss.py:6-70):
    ``win=None``: one value per (N,C) over the whole (optionally masked) image; otherwise local statistics in a
    box window of ``2 * int(win / 2^level / 2) + 1`` voxels (zero padding), one value per voxel."""
    nd = I.ndim - 2
    if mask is not None:
        I, J = I * mask, J * mask
    c = I.shape[1]
    if win is None:
        I, J = torch.flatten(I, 1), torch.flatten(J, 1)
        if mask is not None:
            n = torch.flatten(mask, 1).sum(-1) + eps
            avg = lambda t: t.sum(-1) / n
        else:
            avg = lambda t: t.mean(-1)
    else:
        I, J = I.reshape(-1, 1, *I.shape[2:]), J.reshape(-1, 1, *J.shape[2:])
        w = 2 * int(win / 2**level / 2) + 1
        if I.is_cuda and not (torch.is_grad_enabled() and (I.requires_grad or J.requires_grad)):
            # the box window is separable: w shifted adds per axis instead of a convolution (see gaussian_blur).  Under
            # autograd the convolution below is kept: one graph node per window instead of one per tap

            def avg(t):
                for d in range(nd):
                    t = _filter_axis(t, [1.0 / w] * w, d + 2)
                return t
        else:
            box = torch.ones([1, 1] + [w] * nd, device=I.device, dtype=I.dtype) / w**nd
            conv = [F.conv1d, F.conv2d, F.conv3d][nd - 1]
            avg = lambda t: conv(t, box, stride=1, padding=w // 2)
    mi, mj = avg(I), avg(J)
    cross = avg(I * J) - mi * mj
    cc = cross * cross / ((avg(I * I) - mi * mi) * (avg(J * J) - mj * mj) + eps)
    if reduction == "mean":
        return -cc.mean()
    if reduction == "sum":
        return -cc.sum()
    return -cc.view(-1, c) if win is None else -cc.view(-1, c, *I.shape[2:])


class MovingAverage:
    """Bias-corrected EMA (alpha>0) or running mean (alpha==0) per key (misc.py:91-145)."""

    def __init__(self, alpha: float) -> None:
        assert 0 <= alpha < 1
        self.alpha = alpha
        self._value: Dict[str, Any] = dict()

    def __call__(self, key: str, value) -> None:
        num, v = self._value.get(key, (0, 0))
        num += 1
        v = v * self.alpha + value * (1 - self.alpha) if self.alpha else v + value
        self._value[key] = (num, v)

    def update_all(self, values: Dict[str, Any]) -> None:
        """Same update as calling the object once per key, for 0-d device tensors: one stack + one lerp launch per call
        instead of three launches per key (the training loop calls this every iteration)."""
        keys = tuple(values.keys())
        vec = torch.stack([values[k].detach() for k in keys])
        if getattr(self, "_keys", None) != keys:
            self._keys, self._num, self._vec = keys, 0, torch.zeros_like(vec)
        self._num += 1
        self._vec = torch.lerp(vec, self._vec, self.alpha) if self.alpha else self._vec + vec
        for i, k in enumerate(keys):
            self._value[k] = (self._num, self._vec[i])

    def __getitem__(self, key: str) -> Any:
        if key not in self._value:
            return 0
        num, v = self._value[key]
        return v / (1 - self.alpha**num) if self.alpha else v / num

    def to_dict(self) -> Dict[str, Any]:
        return {"alpha": self.alpha, "value": self._value}

    def from_dict(self, d: Dict) -> None:
        self.alpha, self._value = d["alpha"], d["value"]

    @property
    def value(self) -> List:
        vals = [self[k] for k in self._value]
        if self._value:
            return [next(iter(self._value.values()))[0]] + vals
        return vals

    def __str__(self) -> str:
        s = "".join("%s = %.3e  " % (k, self[k]) for k in self._value)
        if self._value:
            return ("iter = %d  " % list(self._value.values())[-1][0]) + s
        return s
