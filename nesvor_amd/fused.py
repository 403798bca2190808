"""Fused training step for the NeSVoR INR (the loop body of train.py:179-198).

All trainable tensors of the model are re-homed into ONE flat fp32 parameter
buffer with matching flat gradient and Adam-moment buffers, so that
* the optimiser + zero_grad is one HBM-streaming launch (nesvor_adamw_step),
* data-parallel training needs exactly one RCCL collective per iteration over
  one contiguous buffer (nesvor_amd.ddp),
* the hash-grid backward scatters straight into the flat gradient.
"""
from argparse import Namespace
from typing import Dict, Optional

import os

import torch

from . import ops as _ops  # noqa: F401  (registers torch.ops.nesvor)
from .models import NeSVoR
from .train import loss_weights


class FlatParams:
    """Re-home every parameter of `module` into one flat buffer (params become views)."""

    def __init__(self, module: torch.nn.Module, pad_to: int = 4):
        """``pad_to``: the buffers' length is rounded up to a multiple of it (4 W for W ranks sharding the optimizer)."""
        named = [(n, p) for n, p in module.named_parameters() if p.numel() > 0]
        # everything in definition order (so each MLP's W0,b0,W1,b1,... stay adjacent: nesvor_amd.direct sums the
        # kernels' partial gradients straight into that segment), the big table LAST: its fine levels - the end of the
        # buffer - are the part a data-parallel step all-reduces early, and what remains is one contiguous range;
        # 16-byte aligned segments so float4 access never straddles a tensor
        biggest = max(p.numel() for _, p in named)
        named.sort(key=lambda kv: 1 if kv[1].numel() == biggest else 0)
        self.names, self.offsets, total = [], {}, 0
        for n, p in named:
            self.offsets[n] = (total, p.numel())
            self.names.append(n)
            total += (p.numel() + 3) // 4 * 4
        dev = named[0][1].device
        total = -(-total // pad_to) * pad_to
        self.param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        for n, p in named:
            off, cnt = self.offsets[n]
            self.param[off : off + cnt].copy_(p.data.reshape(-1))
            p.data = self.param[off : off + cnt].view(p.shape)
            p.grad = self.grad[off : off + cnt].view(p.shape)
        self.numel = total

    def grad_view(self, name):
        off, cnt = self.offsets[name]
        return self.grad[off : off + cnt]


class LossScaler:
    """``torch.cuda.amp.GradScaler`` as the reference configures it (nesvor/nesvor/train.py:161-164: ``init_scale=1.0,
    growth_factor=2.0, backoff_factor=0.5``, PyTorch's default ``growth_interval=2000``), host side: the scale is a Python
    float, the verdict of an iteration (all gradients finite?) is the one device value read per step."""

    def __init__(self, init_scale: float = 1.0, growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000):
        self.scale, self.growth_factor, self.backoff_factor, self.growth_interval = float(init_scale), growth_factor, backoff_factor, growth_interval
        self.growth_tracker, self.skipped = 0, 0

    def update(self, found_inf: bool) -> None:
        if found_inf:
            self.scale *= self.backoff_factor
            self.growth_tracker = 0
            self.skipped += 1
        else:
            self.growth_tracker += 1
            if self.growth_tracker == self.growth_interval:
                self.scale *= self.growth_factor
                self.growth_tracker = 0

    def state_dict(self):
        return {"scale": self.scale, "growth_tracker": self.growth_tracker, "skipped": self.skipped}


class FusedTrainer:
    def __init__(self, model: NeSVoR, args: Namespace, world_size: int = 1, distributed: Optional[bool] = None):
        if next(model.parameters()).device.type != "cuda":
            raise RuntimeError("FusedTrainer needs the model on a HIP device (no CPU path)")
        self.model, self.args = model, args
        # optimizer-state sharding over the ranks (ddp.ShardedExchange): opt-in, args.ddp_sharded_optimizer / NESVOR_DDP_SHARDED=1

        distributed = world_size > 1 if distributed is None else distributed
        self.sharded = distributed and bool(getattr(args, "ddp_sharded_optimizer", os.environ.get("NESVOR_DDP_SHARDED") == "1"))
        self.flat = FlatParams(model, 4 * world_size if self.sharded else 4)
        self._exchange = None
        enc = model.inr.encoding
        enc.grad_accum = self.flat.grad_view("inr.encoding.params")
        self.weights = loss_weights(args)
        self.lr = float(args.learning_rate)
        self.betas, self.eps, self.weight_decay = (0.9, 0.99), 1e-15, 1e-2
        self.t = 0
        self.world_size = world_size
        self._reduce_hook = None  # set by ddp: callable(flat_grad) performing the all-reduce(sum)
        # True (set by a training loop that touches nothing but the losses between steps): a single-process native step may
        # return while the hash table's AdamW update is still running on the side stream; the next step waits for it where it
        # reads the table (csrc/step.hip, NESVOR_STEP_DEFER_JOIN), everyone else calls join() first
        self.defer_table_join = False
        self._opt_stream = None  # data parallel: AdamW of the early-exchanged part of the gradient (optimizer_step)
        self._early_updated = False
        self._late_join = os.environ.get("NESVOR_OWNER_JOIN_LATE", "1") != "0"  # 0: join the owner pass before the step's epilogue (A/B)
        # autograd-free evaluation of the iteration when the configuration allows it (nesvor_amd.direct)
        from . import direct

        self.direct = direct.DirectStep(model, self.flat, self.weights) if direct.supported(model) else None
        # the reference's default numerics, opt-in (round 6): fp16 matrix operands + its GradScaler (train.py:161-164)
        self.scaler = None
        if getattr(args, "fp16_loss_scaling", False):
            if self.direct is None or not direct.half_precision_model(model) or distributed:
                raise RuntimeError("args.fp16_loss_scaling: the half-precision model structure (no --single-precision) on the "
                                   "autograd-free step, single process (the reference's loop, which it restates, has no data parallelism)")
            from . import mlp as _mlp

            _mlp.HALF_OPERANDS[0] = _mlp.FP16  # (the module path of tinycudann.Network: inference between / after training)
            self.scaler = LossScaler()
        if (getattr(args, "mlp_bf16", False) or getattr(args, "mlp_fp16", False)) and self.direct is None:
            raise RuntimeError("args.mlp_bf16 / args.mlp_fp16 need the autograd-free step (fused fp32 model, MLPs of at most two hidden layers)")

    @property
    def reduce_hook(self):
        return self._reduce_hook

    def _scaled_step(self, xyz, v, slice_idx, noise=None) -> Dict[str, torch.Tensor]:
        """One iteration under the loss scaler (``args.fp16_loss_scaling``): what the reference's loop does around its optimizer
        (train.py:190-196: ``scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()``) - every gradient carries the
        scale, a step whose gradients are not all finite is SKIPPED (parameters, moments and step count untouched, gradients
        dropped) and halves the scale, ``growth_interval`` finite steps in a row double it.  Like ``GradScaler.step`` this reads one
        device flag per iteration (a host synchronisation the fp32 path does not have)."""
        sc = self.scaler
        self.direct.set_loss_scale(sc.scale)
        losses = self._forward_backward(xyz, v, slice_idx, noise)
        self.direct.join_owner()
        finite = bool(torch.isfinite(self.flat.grad).all())
        if finite:
            self.t += 1
            f = self.flat
            torch.ops.nesvor.adamw_step_(f.param, f.grad, f.exp_avg, f.exp_avg_sq, self.lr, self.betas[0], self.betas[1], self.eps,
                                         self.weight_decay, self.t, 1.0 / (self.world_size * sc.scale), True)
        else:
            self.flat.grad.zero_()
        sc.update(found_inf=not finite)
        return losses

    @reduce_hook.setter
    def reduce_hook(self, hook) -> None:
        self._reduce_hook = hook
        if self.direct is not None:
            # the fine hash-grid levels' gradient is exchanged early, under the rest of the backward: an all-reduce of that
            # range, or - with optimizer sharding - its reduce-scatter (ddp.ShardedExchange part 1)
            self.direct.set_overlap(hook is not None)
            if self.sharded and self.direct.split_level:
                self.direct.early_exchange = self._early_reduce_scatter
            elif hook is not None and self.direct.split_level and os.environ.get("NESVOR_DDP_EARLY_ADAMW", "1") != "0":
                self.direct.early_update = self._early_adamw

    def _sharded_exchange(self):
        from . import ddp

        if self._exchange is None:
            split = None
            if self.direct is not None and self.direct.split_level:
                split = self.direct.early_range()[0]
            self._exchange = ddp.ShardedExchange(self.flat.numel, split=split)
        return self._exchange

    def _early_adamw(self, works, lo: int, hi: int) -> None:
        """DirectStep.early_update: on the side stream, behind the all-reduce of flat.grad[lo:hi] (the fine levels of the table),
        while the main stream runs the coarse levels' backward - AdamW of that range with this step's t."""
        for w in works:
            w.wait()
        self.t += 1
        self._adamw(lo, hi)
        self._early_updated = True

    def _early_reduce_scatter(self, lo: int, hi: int):
        """Called by the step once the fine levels' gradient is complete: start part 1's reduce-scatter (asynchronous, on the
        collective's stream) -> ((slice, work), lo, hi) for ``optimizer_step``."""
        ex = self._sharded_exchange()
        assert (lo, hi) == ex.parts[1]
        return ex.reduce_scatter(self.flat.grad, 1, async_op=True), lo, hi

    def decay_lr(self, gamma: float) -> None:
        self.lr *= gamma

    def _forward_backward(self, xyz, v, slice_idx, noise=None) -> Dict[str, torch.Tensor]:
        if self.direct is not None:
            return self.direct.run(xyz, v, slice_idx, noise, defer_owner_join=self._late_join)  # joined in optimizer_step
        losses = self.model(xyz, v, slice_idx) if noise is None else self.model.forward_with_noise(xyz, v, slice_idx, noise)
        loss = 0
        for k, val in losses.items():
            if k in self.weights and self.weights[k]:
                loss = loss + self.weights[k] * val
        loss.backward()
        return losses

    def step(self, xyz, v, slice_idx, noise=None) -> Dict[str, torch.Tensor]:
        if self.scaler is not None:
            return self._scaled_step(xyz, v, slice_idx, noise)
        if self.direct is not None and self.reduce_hook is None and not self.sharded and self.direct.native_ready(noise):
            # single process: forward, backward AND AdamW behind one native call (csrc/step.hip)
            from . import _lib

            t = self.t + 1
            adam = _lib.AdamwT(self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, 1 - self.betas[0] ** t,
                               1 - self.betas[1] ** t, 1.0 / self.world_size)
            losses = self.direct.run(xyz, v, slice_idx, None, defer_owner_join=self._late_join, adam=adam,
                                     defer_table_join=self.defer_table_join)
            if self.direct.ran_optimizer:
                self.t = t
                return losses
            self.optimizer_step()
            return losses
        losses = self._forward_backward(xyz, v, slice_idx, noise)
        self.optimizer_step()
        return losses

    def _adamw(self, lo: int, hi: int, grad=None) -> None:
        f = self.flat
        torch.ops.nesvor.adamw_step_(f.param[lo:hi], f.grad[lo:hi] if grad is None else grad, f.exp_avg[lo:hi], f.exp_avg_sq[lo:hi],
                                     self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.t,
                                     1.0 / self.world_size, True)

    def optimizer_step(self) -> None:
        if self.direct is not None:
            self.direct.join_owner()  # the table gradient is complete behind the owner pass on the side stream
        if self.sharded:
            ex = self._sharded_exchange()
            early = self.direct.take_early_reduce() if self.direct is not None else None
            slices = {}
            if early is not None:  # part 1 (the fine levels) is already being reduce-scattered (nesvor_amd.direct)
                slices[1] = early[0][0]
            for part in range(len(ex.parts)):
                if part not in slices:
                    slices[part] = ex.reduce_scatter(self.flat.grad, part)
            if early is not None:
                early[0][1].wait()
            self.t += 1
            for part, mine in slices.items():
                lo, hi = ex.owned(part)
                self._adamw(lo, hi, mine)
            self.flat.grad.zero_()  # this rank's partial sums: the next step accumulates from zero
            ex.all_gather_(self.flat.param)
            return
        if self.reduce_hook is not None:
            early = self.direct.take_early_reduce() if self.direct is not None else None
            if early is not None:  # [start, end) of the flat gradient is already being all-reduced (nesvor_amd.direct)
                works, start, end = early
                if self._early_updated:
                    # the early part took its AdamW step on the side stream, right behind its all-reduce (_early_adamw): joined
                    # by join_owner() above; t is already this step's
                    self._early_updated = False
                else:
                    self.t += 1
                    # (Python-issued step) the early part - the fine levels, two thirds of the table - takes its AdamW step on a
                    # stream of its own as soon as its all-reduce is done; nothing else touches that range of the flat buffers
                    main = torch.cuda.current_stream(self.flat.param.device)
                    if self._opt_stream is None:
                        self._opt_stream = torch.cuda.Stream(device=self.flat.param.device)
                    with torch.cuda.stream(self._opt_stream):
                        for w in works:
                            w.wait()
                        self._adamw(start, end)
                    main.wait_stream(self._opt_stream)
                if start > 0:
                    self.reduce_hook(self.flat.grad[:start])
                    self._adamw(0, start)
                if end < self.flat.numel:
                    self.reduce_hook(self.flat.grad[end:])
                    self._adamw(end, self.flat.numel)
                return
            self.reduce_hook(self.flat.grad)
        self.t += 1
        self._adamw(0, self.flat.numel)

    def join(self) -> None:
        """Make the current stream wait for a table update left on the side stream (``defer_table_join``)."""
        if self.direct is not None:
            self.direct.join_owner()

    def finish(self) -> None:
        self.join()
        self.model.inr.encoding.grad_accum = None
