"""CPU oracle: multi-resolution hash-grid encoding (Instant-NGP / tiny-cuda-nn).

TEST INFRASTRUCTURE (see oracle/__init__.py).

**PARITY UNPINNED.**  The reference reaches this arithmetic through the
external module ``tinycudann`` (``tcnn.Encoding`` at
nesvor/nesvor/models.py:25, configured at models.py:102-111 with
``otype="HashGrid"``, ``n_levels``, ``n_features_per_level``,
``log2_hashmap_size``, ``base_resolution``, ``per_level_scale``).  tinycudann is
not vendored and not version-pinned (requirements.txt:5 is a commented-out git
URL), its source is absent from /root/reference and no reference test touches
it.  This file restates the published tiny-cuda-nn ``GridEncoding`` algorithm
(``HashGrid`` type, linear interpolation, "coherent prime" hash):

* per level l:  scale_l = exp2(l * log2(per_level_scale)) * base_resolution - 1   (fp32)
                res_l   = ceil(scale_l) + 1
                size_l  = min(round_up(res_l^3, 8), 2^log2_hashmap_size)   entries
* pos = fma(scale_l, u, 0.5); cell = floor(pos); w = pos - cell
* corner c in {0,1}^3: weight prod(c_d ? w_d : 1 - w_d); index
      idx = sum_d (cell_d + c_d) * res^d          accumulated while stride <= size_l
      idx = (x*1) ^ (y*2654435761) ^ (z*805459861)   (uint32) if the level is hashed
            (res^3 > size_l)
      idx %= size_l
* output (N, L*F), level-major; params: one flat fp32 tensor, levels
  concatenated, F contiguous per entry.
* gradients: d/dparams = scatter-add of weight * dy; d/du from the piecewise
  linear interpolant (what torch autograd gives for this formulation).
"""
from dataclasses import dataclass
from typing import List

import numpy as np
import torch

PRIME_Y = 2654435761
PRIME_Z = 805459861
U32 = 0xFFFFFFFF


@dataclass
class Level:
    scale: float  # fp32 value
    res: int
    size: int  # entries in this level
    offset: int  # entry offset into the flat table
    hashed: bool


def make_levels(n_levels: int, log2_hashmap_size: int, base_resolution: int, per_level_scale: float) -> List[Level]:
    levels = []
    offset = 0
    log2s = np.log2(np.float32(per_level_scale)).astype(np.float32)
    for l in range(n_levels):
        scale = np.float32(np.exp2(np.float32(l) * log2s).astype(np.float32) * np.float32(base_resolution) - np.float32(1.0))
        res = int(np.ceil(scale)) + 1
        dense = res**3
        size = min((min(dense, 2**31 - 1) + 7) // 8 * 8, 1 << log2_hashmap_size)
        levels.append(Level(float(scale), res, size, offset, dense > size))
        offset += size
    return levels


def n_params(levels: List[Level], n_features: int) -> int:
    return (levels[-1].offset + levels[-1].size) * n_features


def _corner_index(lv: Level, gx, gy, gz):
    """int64 tensors of grid coordinates -> entry index within the level."""
    if lv.hashed:
        h = (gx & U32) ^ ((gy * PRIME_Y) & U32) ^ ((gz * PRIME_Z) & U32)
        return h % lv.size
    # dense: strides 1, res, res^2 (all <= size because res^3 <= size here)
    idx = gx + gy * lv.res + gz * (lv.res * lv.res)
    return idx % lv.size


def _pos(u: torch.Tensor, scale: float) -> torch.Tensor:
    if u.dtype == torch.float32:
        # emulate fmaf(scale, u, 0.5f): exact product in fp64, one final rounding
        p64 = u.detach().double() * float(np.float32(scale)) + 0.5
        p32 = p64.float()
        if u.requires_grad:
            # same value, gradient d pos / d u = scale
            return p32 + (u - u.detach()) * scale
        return p32
    return u * scale + 0.5


def encode(u: torch.Tensor, table: torch.Tensor, levels: List[Level], n_features: int) -> torch.Tensor:
    """u (N,3) in [0,1], table flat -> (N, L*F).  Differentiable w.r.t. both."""
    N = u.shape[0]
    tab = table.view(-1, n_features)
    outs = []
    for lv in levels:
        pos = _pos(u, lv.scale)
        cell = torch.floor(pos.detach())
        w = pos - cell
        g = cell.long()
        feat = torch.zeros(N, n_features, dtype=table.dtype)
        for cz in (0, 1):
            for cy in (0, 1):
                for cx in (0, 1):
                    wgt = (
                        (w[:, 0] if cx else 1 - w[:, 0])
                        * (w[:, 1] if cy else 1 - w[:, 1])
                        * (w[:, 2] if cz else 1 - w[:, 2])
                    )
                    idx = _corner_index(lv, g[:, 0] + cx, g[:, 1] + cy, g[:, 2] + cz) + lv.offset
                    feat = feat + wgt[:, None] * tab[idx]
        outs.append(feat)
    return torch.cat(outs, -1)


def encode_backward(u, table, levels, n_features, dy, need_input_grad=True):
    """Returns (grad_table flat, grad_u or None) for upstream dy (N, L*F)."""
    u_ = u.detach().clone().requires_grad_(need_input_grad)
    t_ = table.detach().clone().requires_grad_(True)
    y = encode(u_, t_, levels, n_features)
    grads = torch.autograd.grad(y, [t_, u_] if need_input_grad else [t_], dy)
    return grads[0], (grads[1] if need_input_grad else None)


def encode_backward_explicit(u, table, levels, n_features, dy):
    """Explicit (no autograd) restatement of the tiny-cuda-nn backward scheme:
    param grad = scatter-add(weight * dy); input grad per dim d =
    scale * sum_{other corners} w_other * (f(c_d=1) - f(c_d=0)) . dy.
    Used to check that autograd of ``encode`` equals the published formulas."""
    N = u.shape[0]
    tab = table.view(-1, n_features)
    gt = torch.zeros_like(tab)
    gu = torch.zeros(N, 3, dtype=u.dtype)
    for li, lv in enumerate(levels):
        pos = _pos(u, lv.scale)
        cell = torch.floor(pos)
        w = pos - cell
        g = cell.long()
        dyl = dy[:, li * n_features : (li + 1) * n_features]
        for cz in (0, 1):
            for cy in (0, 1):
                for cx in (0, 1):
                    c = (cx, cy, cz)
                    wd = [w[:, d] if c[d] else 1 - w[:, d] for d in range(3)]
                    idx = _corner_index(lv, g[:, 0] + cx, g[:, 1] + cy, g[:, 2] + cz) + lv.offset
                    gt.index_add_(0, idx, (wd[0] * wd[1] * wd[2])[:, None] * dyl)
                    fdy = (tab[idx] * dyl).sum(-1)
                    for d in range(3):
                        o = [k for k in range(3) if k != d]
                        sign = 1.0 if c[d] else -1.0
                        gu[:, d] += lv.scale * sign * wd[o[0]] * wd[o[1]] * fdy
    return gt.view(-1), gu
