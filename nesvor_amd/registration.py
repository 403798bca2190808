"""Stack-to-stack rigid registration (the reference's ``nesvor/svort/registration.py:189-284``
``VVR`` + ``resample`` and the registration-only branch of ``svort/inference.py``: ``parse_data`` :176-248,
``stack_registration`` :308-367, ``run_svort`` :447-552 with ``svort=False, vvr=True`` = ``--registration stack``).

Volume-to-volume registration by normalised gradient descent on a similarity loss over a coarse-to-fine pyramid:

* level l (num_levels-1 .. 0): both volumes are blurred (sigma = 0.5 * 2^l voxels of the finest axis) and resampled
  to isotropic voxels of size ``min(res) * 2^l``; the target's non-zero voxels become a point list in physical
  coordinates, the source is sampled at the rigidly moved points (trilinear);
* a level runs ``num_steps`` rounds, round r with step length ``step_size * 2^l / 2^r``; a round repeats up to
  ``max_iter`` times: gradient (autograd, or central differences with the step length as offset), momentum
  buffer, normalise the direction, move by the step length, keep the move only where the loss decreased - the
  first rejected move ends the round for that batch entry;
* the parameter vector is (rotation vector in DEGREES, translation in mm) during the descent, so one step length
  serves both parts.

Written for this code base (explicit pyramid level object, one objective function); behaviour follows the reference,
whose own test (``tests/svort/test_vvr.py``) is re-stated in ``tests/test_registration.py``.

Two evaluation paths.  Generic (any loss callable, autograd gradients, any device the transform ops support):
PyTorch ``grid_sample`` + the loss, as the reference does - one objective evaluation is ~25 small launches and a
finite-difference gradient needs 13 of them.  Fused (HIP, ``nesvor_vvr_similarity``): when the loss is given by name
(global NCC or MSE) and the gradient is by finite differences, ONE launch samples the source under all 13 poses and
returns the moment sums the loss is a function of (fp64); ``stack_registration`` uses it.
"""
import ctypes
import logging
import math
import time
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import torch
import torch.nn.functional as F

from . import _lib
from .transform import RigidTransform, mat_transform_points
from .utils import gaussian_blur, meshgrid, ncc_loss


def resample(x: torch.Tensor, res_xyz_old: Sequence[float], res_xyz_new: Sequence[float]) -> torch.Tensor:
    """Linear resampling of (N,C,*spatial) from voxel size ``res_xyz_old`` to ``res_xyz_new`` (both listed x first),
    keeping the first and the last sample position fixed relative to the centre (registration.py:267-284)."""
    nd = x.ndim - 2
    assert len(res_xyz_old) == len(res_xyz_new) == nd
    axes = []
    for d in range(nd):  # d = 0 is x = the last tensor dimension
        size = x.shape[-1 - d]
        fac = res_xyz_old[d] / res_xyz_new[d]
        size_new = int(size * fac)
        reach = (size_new - 1) / fac / (size - 1)  # normalised half-extent covered by the new samples
        axes.append(torch.linspace(-reach, reach, size_new, dtype=x.dtype, device=x.device))
    mesh = torch.meshgrid(*axes[::-1], indexing="ij")[::-1]  # each of shape (..., ny, nx); element order x, y, (z)
    grid = torch.stack(mesh, -1)[None].expand(x.shape[0], *([-1] * (nd + 1)))
    return F.grid_sample(x, grid, align_corners=True)


class _Level:
    """One pyramid level: the source volume on the level's grid, the target as a masked point list."""

    def __init__(self, source: torch.Tensor, target: torch.Tensor, voxel: float):
        self.source = source
        keep = (target > 0).reshape(-1)
        shape_xyz = (target.shape[-1], target.shape[-2], target.shape[-3])
        self.points = meshgrid(shape_xyz, (voxel, voxel, voxel), device=target.device).reshape(-1, 3)[keep]
        self.values = target.reshape(-1)[keep]
        # physical (mm, centred) -> grid_sample's normalised coordinates of the source
        unit = [2.0 / (source.shape[-1] - 1) / voxel, 2.0 / (source.shape[-2] - 1) / voxel, 2.0 / (source.shape[-3] - 1) / voxel]
        self.to_unit = torch.tensor(unit, dtype=source.dtype, device=source.device)
        self.to_unit_host = (ctypes.c_float * 3)(*unit)
        self.target_sums = None  # (sum J, sum J^2) of the fused path, filled by its first call


class VVR:
    """``VVR(num_levels, num_steps, step_size, max_iter, optimizer, loss, auto_grad)`` then
    ``theta_out, loss = vvr(theta, source, target, params, transform_t, trans_first)``:

    theta (B,6) axis-angle [rad | mm] of the source in the ``trans_first`` convention, source / target volumes
    (1,1,D,H,W), ``params = {"res_s": in-plane, "s_thick": through-plane}`` voxel sizes, ``transform_t`` the target's
    pose.  ``optimizer = {"name": "gd", "momentum": m}``; ``loss`` = {"name": "mse" | "ncc", ...} or a callable
    ``loss(self, warped, target) -> per-element or per-batch loss``.
    """

    def __init__(self, num_levels: int, num_steps: int, step_size: float, max_iter: int, optimizer: Dict,
                 loss: Union[Dict, Callable], auto_grad: bool) -> None:
        self.num_levels, self.num_steps, self.step_size, self.max_iter = num_levels, num_steps, step_size, max_iter
        self.auto_grad = auto_grad
        self.current_level = num_levels - 1
        if optimizer.get("name") != "gd":
            raise Exception("unknown optimizer")
        self.momentum = float(optimizer.get("momentum", 0))
        self._fused_kind, self._eps = None, 1e-6
        if isinstance(loss, dict):
            kw = dict(loss)
            name = kw.pop("name")
            if name == "mse":
                self._loss = lambda x, y: F.mse_loss(x, y, reduction="none", **kw)
                self._fused_kind = "mse" if not kw else None
            elif name == "ncc":
                self._loss = lambda x, y: ncc_loss(x, y, reduction="none", level=self.current_level, **kw)
                if kw.get("win", 9) is None and set(kw) <= {"win", "eps"}:  # global NCC: a function of five moment sums
                    self._fused_kind, self._eps = "ncc", float(kw.get("eps", 1e-6))
            else:
                raise Exception("unknown loss")
        elif callable(loss):
            self._loss = lambda x, y: loss(self, x, y)
        else:
            raise Exception("unknown loss")
        self.theta_t: Optional[RigidTransform] = None
        self.trans_first = True

    # ---- units: the descent works on (degrees, mm) -----------------------------------------------------------
    @staticmethod
    def _unit(theta: torch.Tensor) -> torch.Tensor:
        d = math.pi / 180
        return torch.tensor([d, d, d, 1, 1, 1], dtype=theta.dtype, device=theta.device).view(1, 6)

    # ---- pyramid -------------------------------------------------------------------------------------------
    def _level(self, level: int, source: torch.Tensor, target: torch.Tensor) -> _Level:
        f = 2.0**level
        sigma = [0.5 * f / r for r in self.relative_res]  # per tensor dimension (z, y, x)
        vols = []
        for v in (source, target):
            v = gaussian_blur(v, sigma, truncated=4.0)
            vols.append(resample(v, self.relative_res[::-1], [f] * 3))
        return _Level(vols[0], vols[1], self.res * f)

    # ---- objective -------------------------------------------------------------------------------------------
    def _warp(self, theta_deg: torch.Tensor, lv: _Level) -> torch.Tensor:
        """Source sampled at the target's points moved by  inv(T(theta)) o T_target  ->  (B, M)."""
        pose = RigidTransform(theta_deg * self._unit(theta_deg), trans_first=self.trans_first)
        mat = pose.inv().compose(self.theta_t).matrix()  # (B,3,4), translation-first convention
        moved = mat_transform_points(mat[:, None], lv.points[None], True)  # (B,M,3)
        grid = (moved * lv.to_unit).view(moved.shape[0], -1, 1, 1, 3)
        src = lv.source.expand(moved.shape[0], -1, -1, -1, -1)
        return F.grid_sample(src, grid, align_corners=True).view(moved.shape[0], -1)

    def _objective(self, theta_deg: torch.Tensor, lv: _Level) -> torch.Tensor:
        warped = self._warp(theta_deg, lv)
        loss = self._loss(warped[:, None], lv.values.view(1, 1, -1).expand(warped.shape[0], -1, -1))
        return loss.reshape(loss.shape[0], -1).mean(1)

    def _fused(self, theta: torch.Tensor, lv: _Level) -> bool:
        return (self._fused_kind is not None and not self.auto_grad and theta.shape[0] == 1 and lv.source.is_cuda
                and lv.source.dtype == torch.float32 and lv.source.shape[0] == 1)

    def _objective_fused(self, thetas_deg: torch.Tensor, lv: _Level) -> torch.Tensor:
        """Losses of K poses of ONE registration problem in one launch -> (K,)."""
        K = thetas_deg.shape[0]
        pose = RigidTransform(thetas_deg * self._unit(thetas_deg), trans_first=self.trans_first)
        mats = pose.inv().compose(self.theta_t).matrix().contiguous()
        dev = lv.source.device
        sums = torch.empty((K, 3), dtype=torch.float64, device=dev)
        first = lv.target_sums is None
        if first:
            lv.target_sums = torch.empty(2, dtype=torch.float64, device=dev)
        src = lv.source.contiguous()
        D, H, W = src.shape[-3:]
        M = lv.values.numel()
        with torch.cuda.device(dev):
            err = _lib.load().nesvor_vvr_similarity(_lib.ptr(src), D, H, W, _lib.ptr(lv.points), _lib.ptr(lv.values), _lib.ptr(mats),
                                                    lv.to_unit_host, M, K, _lib.ptr(sums),
                                                    _lib.ptr(lv.target_sums) if first else None, _lib.stream_ptr())
        _lib.check(err, "vvr similarity")
        sI, sII, sIJ = sums[:, 0] / M, sums[:, 1] / M, sums[:, 2] / M
        sJ, sJJ = lv.target_sums[0] / M, lv.target_sums[1] / M
        if self._fused_kind == "mse":
            return (sII - 2 * sIJ + sJJ).float()
        cross = sIJ - sI * sJ
        return (-(cross * cross) / ((sII - sI * sI) * (sJJ - sJ * sJ) + self._eps)).float()

    def _gradient(self, theta: torch.Tensor, lv: _Level, h: float) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._fused(theta, lv):  # the pose and its 12 perturbations in one launch
            e = torch.zeros((13, 6), dtype=theta.dtype, device=theta.device)
            idx = torch.arange(6, device=theta.device)
            e[1 + 2 * idx, idx] = h
            e[2 + 2 * idx, idx] = -h
            l = self._objective_fused(theta + e, lv)
            return l[:1], (l[1::2] - l[2::2]).view(1, 6)
        if self.auto_grad:
            with torch.enable_grad():
                t = theta.detach().requires_grad_(True)
                loss = self._objective(t, lv)
                (grad,) = torch.autograd.grad([loss.sum()], [t])
            return loss.detach(), grad
        loss = self._objective(theta, lv)
        grad = torch.zeros_like(theta)
        for j in range(theta.shape[1]):  # central differences with the step length as offset, not divided by it
            e = torch.zeros_like(theta)
            e[:, j] = h
            grad[:, j] = self._objective(theta + e, lv) - self._objective(theta - e, lv)
        return loss, grad

    # ---- descent -----------------------------------------------------------------------------------------------
    def _round(self, theta: torch.Tensor, lv: _Level, step: float, state: Dict) -> Tuple[torch.Tensor, torch.Tensor]:
        """Up to max_iter accepted moves of length `step`; an entry leaves the active set at its first rejected move."""
        active = torch.ones(theta.shape[0], dtype=torch.bool, device=theta.device)
        loss_all = torch.zeros(theta.shape[0], dtype=theta.dtype, device=theta.device)
        for _ in range(self.max_iter):
            idx = torch.nonzero(active).flatten()
            cur = theta[idx]
            loss, grad = self._gradient(cur, lv, step)
            loss_all[idx] = loss
            if self.momentum:
                if "buf" not in state:
                    state["buf"] = grad.clone()  # (first use: all entries are active)
                else:
                    state["buf"][idx] = state["buf"][idx] * self.momentum + grad
                direction = state["buf"][idx]
            else:
                direction = grad
            move = direction / (torch.linalg.norm(direction, dim=-1, keepdim=True) + 1e-6) * (-step)
            better = (self._objective_fused(cur + move, lv) if self._fused(cur, lv) else self._objective(cur + move, lv)) < loss
            active[idx] = better
            if not bool(better.any()):
                break
            theta[idx[better]] += move[better]
        return theta, loss_all

    @torch.no_grad()
    def __call__(self, theta: torch.Tensor, source: torch.Tensor, target: torch.Tensor, params: Dict,
                 transform_t: RigidTransform, trans_first: bool) -> Tuple[torch.Tensor, torch.Tensor]:
        self.theta_t, self.trans_first = transform_t, trans_first
        res = [params["s_thick"], params["res_s"], params["res_s"]]  # voxel size per tensor dimension (z, y, x)
        self.res = min(res)
        self.relative_res = [r / self.res for r in res]
        unit = self._unit(theta)
        theta0 = theta.clone()
        cur = (theta.detach() / unit).clone()
        loss = torch.zeros(theta.shape[0], dtype=theta.dtype, device=theta.device)
        for level in range(self.num_levels - 1, -1, -1):
            self.current_level = level
            lv = self._level(level, source, target)
            step = self.step_size * 2**level
            state: Dict = {}  # the momentum buffer lives for one level
            for _ in range(self.num_steps):
                cur, loss = self._round(cur, lv, step, state)
                step /= 2
        return theta0 + (cur * unit - theta0), loss


# ---------------------------------------------------------------------------------------------------------------------
# --registration stack
# ---------------------------------------------------------------------------------------------------------------------
def _mean_pose(t: RigidTransform) -> RigidTransform:
    return RigidTransform(t.axisangle().mean(0, keepdim=True))


def stack_registration(transforms_list: List[List[RigidTransform]], transform_target: RigidTransform,
                       stacks: List[torch.Tensor], res_s: float, s_thick: float) -> List[RigidTransform]:
    """Register every stack (as a volume of its slices) to stack 0 and return per-slice poses
    (svort/inference.py:308-367): stack j starts from each candidate in ``transforms_list`` (mean pose of its
    slices, expressed relative to the target's), the best final NCC wins; the output poses put the slices of stack
    j at  centre o registered_j o (0,0,(i - (n-1)/2) * s_thick)."""
    device = transform_target.device
    t_target = _mean_pose(transform_target)
    candidates = [[_mean_pose(t) for t in ts] for ts in transforms_list]
    # global NCC given by name: on a HIP device VVR then evaluates a gradient's 13 poses in one launch
    vvr = VVR(num_levels=3, num_steps=4, step_size=2, max_iter=20, optimizer={"name": "gd", "momentum": 0.1},
              loss={"name": "ncc", "win": None}, auto_grad=False)
    trans_first = False
    registered = [t_target]
    target = stacks[0].squeeze(1)[None, None]
    for j in range(1, len(stacks)):
        source = stacks[j].squeeze(1)[None, None]
        best, best_ax = float("inf"), None
        for cand in candidates:
            ax = t_target.compose(cand[0].inv()).compose(cand[j]).axisangle(trans_first=trans_first)
            ax, ncc = vvr(ax, source, target, {"res_s": res_s, "s_thick": s_thick}, t_target, trans_first)
            if float(ncc) < best:
                best, best_ax = float(ncc), ax
        registered.append(RigidTransform(best_ax, trans_first=trans_first))
    centre = registered[0].axisangle(trans_first=False).clone()
    centre[..., :3] = 0
    centre[..., 3:] *= -1
    centre = RigidTransform(centre)
    out = []
    for j, stack in enumerate(stacks):
        n = stack.shape[0]
        t = torch.zeros((n, 6), dtype=torch.float32, device=device)
        t[:, -1] = (torch.arange(n, dtype=torch.float32, device=device) - (n - 1) / 2) * s_thick
        out.append(centre.compose(registered[j]).compose(RigidTransform(t)))
    return out


def register_stacks(dataset: List, res_s: float = 1.0) -> List:
    """``--registration stack``: in-plane resampling of the masked stacks to ``res_s`` mm (parse_data), stack
    registration to the first stack, new per-slice poses written into the stacks (run_svort with svort=False, vvr=True).
    As in the reference, the slice spacing of the output poses is the MEAN thickness of the input stacks."""
    t0 = time.time()
    stacks = [resample(s.slices * s.mask, (s.resolution_x, s.resolution_y), (res_s, res_s)) for s in dataset]
    transforms = [s.transformation for s in dataset]
    s_thick = float(sum(s.thickness for s in dataset) / len(dataset))
    out = stack_registration([transforms], transforms[0], stacks, res_s, s_thick)
    logging.debug("time for stack registration: %f s", time.time() - t0)
    for s, t in zip(dataset, out):
        s.transformation = t
    return dataset
