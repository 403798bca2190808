"""Autograd-free evaluation of one NeSVoR training iteration (the loop body of train.py:179-194 around
``NeSVoR.forward`` models.py:260-327 and ``loss.backward()``).

The autograd path (models.NeSVoR.fused_losses) already runs every heavy stage on the HIP kernels, but the
engine around them costs ~70 small launches per iteration (gradient accumulation adds, index_put sorts,
zero-fills, the loss weighting) and the host falls behind the GPU.  Here the same kernels are called
back-to-back in a fixed order and every parameter gradient is written straight into the flat gradient
buffer.  Results equal the autograd path up to summation order (tests/test_gpu_model.py).

Two issue paths for the same sequence of launches: ``_run_native`` hands the whole iteration to ONE C entry point
(``nesvor_step_run``, csrc/step.hip: every launch enqueued by the library into buffers allocated once - the host side
of an iteration drops from ~0.42 ms to a few tens of microseconds, which is what bounds the small per-GPU batches of
BASELINE C2 / C3); ``run`` issues the launches from Python one by one (explicit PSF noise = the replay mode of the parity
tests, per-kernel event timing, the half-precision model structure, NESVOR_STEP_NATIVE=0).

Supported configurations: the fused single-precision model and the half-precision model structure (bias-free
tinycudann networks, evaluated with bf16 matrix operands), MLPs of at most two hidden layers; anything else keeps the
autograd path (FusedTrainer decides through ``supported``).
"""
import ctypes
from typing import Dict

import torch

from . import _lib, loss as loss_mod, mlp as mlp_mod, sampler
from .encoding import hashgrid_backward, hashgrid_forward
from .models import B_REG, D_LOSS, DS_LOSS, I_REG, S_LOSS, T_REG, NeSVoR
from .transform import trans_loss_raw


def _nets(model: NeSVoR):
    a = model.args
    return [model.inr.density_net] + ([] if a.no_pixel_variance else [model.sigma_net]) + ([model.b_net] if a.n_levels_bias else [])


def half_precision_model(model: NeSVoR) -> bool:
    """The reference's default structure (args.dtype == float16, models.py:28-41): bias-free tinycudann networks.  Here
    it trains on the same kernels with bf16 matrix operands and fp32 accumulation / master weights - at least the
    precision of tinycudann's fp16 path, and no loss scaling is needed."""
    from .tinycudann import Network

    return model.args.dtype == torch.float16 and all(isinstance(n, Network) for n in _nets(model))


def supported(model: NeSVoR) -> bool:
    a = model.args
    if not (getattr(a, "fused_loss", True) and getattr(a, "direct_step", True) and getattr(a, "fused_mlp", True)):
        return False
    if not (model.axisangle.is_cuda and mlp_mod.FUSED_BACKWARD):
        return False
    if half_precision_model(model):
        # the bf16-operand kernels exist for the wave-specialised layout only: a 16-sample group inside one pixel,
        # pixel features in whole 16-blocks, first layer of at most 32 inputs unless there is a single hidden layer
        ks = a.n_features_slice if (not a.no_pixel_variance or a.n_levels_bias) else 0
        if not (a.n_samples % 16 == 0 and ks % 16 == 0 and all(
                mlp_mod.supported(n) and (n.shapes[0][1] <= 32 or len(n.shapes) == 2) for n in _nets(model))):
            return False
    elif not model.use_fused_mlp():
        return False
    # (narrower networks run zero-padded on the module path: mlp.kernel_params)
    return all(mlp_mod.n_hidden_layers(n) <= 2 and mlp_mod.native_width(n) for n in _nets(model))


class DirectStep:
    def __init__(self, model: NeSVoR, flat, weights: Dict[str, float]):
        self.model, self.flat = model, flat
        a = model.args
        dev = flat.param.device
        self.opt_T = not a.no_transformation_optimization
        self.has_lv = not a.no_pixel_variance
        self.has_c = not a.no_slice_scale
        self.has_lvs = not a.no_slice_variance
        self.has_b = bool(a.n_levels_bias)
        self.kb_bias = a.n_levels_bias * a.n_features_per_level  # rows of pe the bias field sees
        self.ks = a.n_features_slice if (self.has_lv or self.has_b) else 0  # slice embedding feeds sigma_net / b_net
        self.has_var = self.has_lv or self.has_lvs
        w = weights
        # upstream gradients of the loss kernel's four terms; DS_LOSS is reported but not weighted (train.py:183-186)
        self.gw = torch.tensor([w.get(D_LOSS, 0), w.get(S_LOSS, 0) if self.has_var else 0, w.get(I_REG, 0),
                                w.get(B_REG, 0) if self.has_b else 0.0], dtype=torch.float32, device=dev)
        self.w_T = float(w.get(T_REG, 0))
        self._gw_base, self._w_T_base, self.loss_scale = self.gw.clone(), self.w_T, 1.0
        self.reg_type = loss_mod.REG_TYPES[a.image_regularization]
        self.delta = float(model.delta)
        # the parameters were re-homed into `flat` before this point: the views taken here stay valid
        self.d_net = mlp_mod.NetParams(model.inr.density_net)
        self.s_net = mlp_mod.NetParams(model.sigma_net) if self.has_lv else None
        self.b_net = mlp_mod.NetParams(model.b_net) if self.has_b else None
        # side stream: the pose regulariser (a serial chain per slice) at the start of the step; at its end the owner pass of
        # the hash-grid backward (latency-bound, finishes the table gradient only) while the main stream runs the
        # sampler backward and the per-slice bookkeeping
        # NESVOR_SIDE_STREAM_PRIORITY (A/B switch): HIP stream priority of the side stream (lower number = higher priority)
        _prio = __import__("os").environ.get("NESVOR_SIDE_STREAM_PRIORITY")
        self.side = torch.cuda.Stream(device=dev) if _prio is None else torch.cuda.Stream(device=dev, priority=int(_prio))
        self._overlap_owner = __import__("os").environ.get("NESVOR_OWNER_OVERLAP", "1") != "0"
        # the table's AdamW step inside the owner pass (nesvor_hashgrid_backward_adamw) whenever one native call covers
        # gradient and update (no data-parallel exchange in between): the table gradient then never goes through HBM
        self._adamw_in_owner = __import__("os").environ.get("NESVOR_ADAMW_IN_OWNER", "1") != "0"
        # evaluation of the MLP matrix products (mlp.operand_mode): bf16-rounded operands for the half-precision model
        # structure and, opt-in, for the fp32 model (args.mlp_bf16); otherwise fp32 - the split-fp16 default, or the
        # plain fp32 MFMAs with args.mlp_fp32_mfma
        if half_precision_model(model) and getattr(a, "fp16_loss_scaling", False):
            self.bf16 = mlp_mod.FP16  # the reference's default arithmetic, under its loss scaler (fused.LossScaler)
        elif bool(getattr(a, "mlp_bf16", False)) or half_precision_model(model):
            self.bf16 = True
        elif bool(getattr(a, "mlp_fp16", False)):
            self.bf16 = mlp_mod.FP16S  # scaled fp16 operands on the split mode's kernels: one MFMA per product (opt-in)
        else:
            self.bf16 = mlp_mod.MFMA_FP32 if getattr(a, "mlp_fp32_mfma", False) else False
        import os

        import torch.distributed as dist

        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        from . import ddp

        self.parallel = ddp.active()
        # Data parallel: the hash-grid backward runs in two launches - the fine levels (about half of the table's bytes,
        # the END of the flat buffer) first; their all-reduce is started at once and overlaps the coarse levels' launch
        # (NESVOR_DDP_OVERLAP=0: one launch, one all-reduce after the step)
        self._early = None
        self._owner_pending = False  # an owner pass of the hash-grid backward is running on the side stream
        self._pending_state = None   # ... left there by this native context (its next run joins it by itself)
        self._kernel_noise = os.environ.get("NESVOR_PSF_NOISE", "kernel") != "tensor"
        self._noise_stream, self._noise_calls = 0x5851F42D4C957F2D, 0  # stream id of the training draws, calls so far
        self._native = {}  # batch size -> (StepT, handle, buffers) of the one-call iteration (csrc/step.hip)
        self._native_timing, self._last_state, self._native_shapes_ok = False, None, None
        self._native_on = os.environ.get("NESVOR_STEP_NATIVE", "1") != "0"
        self.split_level = 0  # 0 = one launch; set by set_overlap() once a gradient all-reduce is installed
        self._split_candidate = 0
        if self.parallel and os.environ.get("NESVOR_DDP_OVERLAP", "1") != "0":
            spec = model.inr.encoding.spec
            F = spec.n_features
            total = spec.n_params // F
            if spec.n_levels > 1:
                self._split_candidate = min(range(1, spec.n_levels), key=lambda l: abs(spec.levels[l].offset - total / 2))
                off, cnt = flat.offsets["inr.encoding.params"]
                self._early_range = (off + spec.levels[self._split_candidate].offset * F, off + cnt)
        self.early_exchange = None  # callable(lo, hi) -> handle for take_early_reduce(); default: all-reduce of flat.grad[lo:hi]
        # callable(works, lo, hi) run on the side stream right behind the early all-reduce (FusedTrainer: the AdamW step of that
        # range, under the coarse levels' backward; the side stream is idle there and maps to a hardware queue of its own -
        # a fresh stream shared the main stream's queue and its kernel ran after the whole iteration)
        self.early_update = None

    def set_overlap(self, on: bool) -> None:
        """Split the hash-grid backward and start the fine levels' all-reduce early (FusedTrainer turns this on when a
        reduce hook is installed: the hook's owner then finishes the reduction in optimizer_step)."""
        self.split_level = self._split_candidate if on else 0

    def early_range(self):
        """[lo, hi) of the flat gradient that is complete after the fine levels' backward.  For the sharded exchange the range
        must be cut into W equal, 16-byte aligned slices: lo is rounded UP to a multiple of 4 W (the few fine-level entries
        below it travel with the late part) and hi is the padded end of the buffer."""
        lo, hi = self._early_range
        if self.early_exchange is not None:
            q = 4 * self.world
            lo, hi = -(-lo // q) * q, self.flat.numel
        return lo, hi

    def _start_early(self):
        from . import ddp

        lo, hi = self.early_range()
        if self.early_exchange is not None:
            return self.early_exchange(lo, hi)
        return ddp.allreduce_flat_(self.flat.grad[lo:hi]), lo, hi

    def take_early_reduce(self):
        """(work handles, start, end) of the flat-gradient range whose all-reduce this step has already started, or None."""
        early, self._early = self._early, None
        return early

    # ------------------------------------------------------------------------------------------------ one-call iteration
    def native_ready(self, noise=None) -> bool:
        """The iteration can go through ``nesvor_step_run``: PSF noise drawn in the kernels, no per-kernel event timing, every
        network's gradient one contiguous segment of the flat buffer (the single-precision model under FusedTrainer), and - for
        the bias field under data parallelism, whose global mean needs the host's all-reduce mid-step - not that combination."""
        if not (self._native_on and noise is None and self._kernel_noise and not _lib.kernel_timer.enabled):
            return False
        if (self.bf16 is True or self.bf16 == mlp_mod.FP16) and half_precision_model(self.model):
            return False
        if self.loss_scale != 1.0:
            return False  # (the one-call step's descriptors hold the unscaled pose-regulariser weight)
        nets = [self.d_net] + ([self.s_net] if self.has_lv else []) + ([self.b_net] if self.has_b else [])
        if not all((not p.flat_params) and p.segment is not None and p.segment.numel() == sum(
                w.numel() + b.numel() for w, b in zip(p.weights, p.biases)) for p in nets):
            return False
        if self.has_b and self.parallel:
            return False
        if mlp_mod.operand_mode(self.bf16) in (mlp_mod.BF16, mlp_mod.FP16) and not self._fused_backward_takes_all():
            return False  # (the dX + dW launch pair that other shapes fall back to has no bf16-operand form; the scaled modes 2 / 4 run it on fp32 MFMAs)
        return True

    def set_loss_scale(self, scale: float) -> None:
        """Every gradient of the step times ``scale`` (the loss scaler of the fp16 mode, train.py:190 of the reference:
        ``scaler.scale(loss).backward()``): the loss kernel's four upstream weights and the pose regulariser's.  A host float
        that changes on growth / backoff events only - no device traffic per step.  The reported loss VALUES are unscaled."""
        if scale != self.loss_scale:
            self.loss_scale = float(scale)
            self.gw.copy_(self._gw_base * self.loss_scale)
            self.w_T = self._w_T_base * self.loss_scale

    def _fused_backward_ok(self, N=None):
        """Per network (density, sigma | None, bias | None): does the wave-specialised fused MLP backward take it at N points per
        iteration?  (``nesvor_mlp_backward_fused_ok``; the answer depends on the shapes and - through the 32-bit row offsets of
        its scalar-base addressing - on N: asked once per N, with the REAL N of the batch, round-5 advisor.)  A network it refuses
        runs as a dX launch + a dW launch through ``nesvor_step_t.dpre_scratch`` and returns its pixel-feature gradient per
        sample, not per 16-sample group."""
        m, a = self.model, self.model.args
        S = a.n_samples
        N = 16 * S if N is None else int(N)
        if self._native_shapes_ok is None:
            self._native_shapes_ok = {}
        key = (N, mlp_mod.operand_mode(self.bf16))  # (bench.py switches the evaluation mode of a live trainer)
        if key not in self._native_shapes_ok:
            E = m.inr.encoding.spec.n_output_dims
            ok = lambda dd: bool(_lib.load().nesvor_mlp_backward_fused_ok(ctypes.byref(dd), N))
            self._native_shapes_ok[key] = (
                ok(mlp_mod.dims_desc(len(self.d_net.weights) - 1, 1 + a.n_features_z, 0, E, 0, S, self.bf16)),
                ok(mlp_mod.dims_desc(len(self.s_net.weights) - 1, 1, self.ks, a.n_features_z, 1, S, self.bf16)) if self.has_lv else None,
                ok(mlp_mod.dims_desc(len(self.b_net.weights) - 1, 1, self.ks, self.kb_bias, 0, S, self.bf16)) if self.has_b else None)
        return self._native_shapes_ok[key]

    def _fused_backward_takes_all(self, N=None) -> bool:
        """Whether the fused MLP backward takes EVERY network of the model at N points per iteration; otherwise the one-call step
        hands the networks a shared dpre scratch (``nesvor_step_t.dpre_scratch``)."""
        return all(v is not False for v in self._fused_backward_ok(N))

    def _native_state(self, B: int):
        key = (B, mlp_mod.operand_mode(self.bf16), self._overlap_owner, self._adamw_in_owner)  # (bench.py switches the evaluation mode of a live trainer)
        st = self._native.get(key)
        if st is not None:
            return st
        m, a, f = self.model, self.model.args, self.flat
        dev = f.param.device
        S, n = a.n_samples, m.n_slices
        N = B * S
        n_pad = (N + 15) // 16 * 16
        enc = m.inr.encoding
        E = enc.spec.n_output_dims
        zr = 1 + a.n_features_z
        buf = {}

        def new(name, *shape):
            buf[name] = torch.empty(shape, dtype=torch.float32, device=dev)
            return buf[name].data_ptr()

        d = _lib.StepT()
        d.grid = enc.spec.c_struct
        mode = self.bf16
        d.density = mlp_mod._desc(self.d_net.weights, self.d_net.biases, 0, E, 0, S, mode)
        n_par = lambda p: sum(w.numel() + b.numel() for w, b in zip(p.weights, p.biases))
        d.n_density_params, d.g_density = n_par(self.d_net), self.d_net.segment.data_ptr()
        largest = d.n_density_params
        if self.has_lv:
            d.sigma = mlp_mod._desc(self.s_net.weights, self.s_net.biases, self.ks, a.n_features_z, 1, S, mode)
            d.n_sigma_params, d.g_sigma = n_par(self.s_net), self.s_net.segment.data_ptr()
            largest = max(largest, d.n_sigma_params)
        if self.has_b:
            d.bias_net = mlp_mod._desc(self.b_net.weights, self.b_net.biases, self.ks, self.kb_bias, 0, S, mode)
            d.n_bias_params, d.g_bias_net = n_par(self.b_net), self.b_net.segment.data_ptr()
            largest = max(largest, d.n_bias_params)
        d.B, d.S, d.n_slices = B, S, n
        d.opt_T, d.has_lv, d.has_c, d.has_lvs, d.has_b = (int(x) for x in (self.opt_T, self.has_lv, self.has_c, self.has_lvs, self.has_b))
        d.n_features_z, d.ks, d.kb_bias, d.reg_type = a.n_features_z, self.ks, self.kb_bias, self.reg_type
        d.overlap_owner = int(self._overlap_owner) | (2 if self._adamw_in_owner else 0)
        d.delta, d.w_T = self.delta, self.w_T
        ptr = lambda t: None if t is None else t.data_ptr()
        d.axisangle, d.axisangle_init = ptr(m.axisangle), ptr(m.axisangle_init)
        d.psf_sigma, d.bounding_box = ptr(m.psf_sigma), ptr(m.inr.bounding_box)
        d.logit_coef = ptr(m.logit_coef) if self.has_c else None
        d.log_var_slice = ptr(m.log_var_slice) if self.has_lvs else None
        d.slice_embedding = ptr(m.slice_embedding.weight) if self.ks else None
        d.table = ptr(enc.params)
        d.g_axisangle = ptr(m.axisangle.grad) if self.opt_T else None
        d.g_logit_coef = ptr(m.logit_coef.grad) if self.has_c else None
        d.g_log_var_slice = ptr(m.log_var_slice.grad) if self.has_lvs else None
        d.g_slice_embedding = ptr(m.slice_embedding.weight.grad) if self.ks else None
        d.g_table = ptr(enc.params.grad)
        d.gw = self.gw.data_ptr()
        d.flat_param, d.flat_grad, d.flat_exp_avg, d.flat_exp_avg_sq = (t.data_ptr() for t in (f.param, f.grad, f.exp_avg, f.exp_avg_sq))
        d.flat_numel = f.numel
        d.small = new("small", n * 26 + 1 + 3 * mlp_mod.PREP_FLOATS)  # ... | max |dpe| | the networks' operand bounds and weight norms
        d.x, d.u, d.pe, d.z, d.dz, d.dpe = new("x", B, S, 3), new("u", N, 3), new("pe", E, N), new("z", zr, N), new("dz", zr, N), new("dpe", E, N)
        d.loss_pix, d.pix = new("loss_pix", B, 3), new("pix", 2, B)
        d.partial = new("partial", 3 * 256, largest)  # per-workgroup partial parameter gradients: one third per network
        def saved_buffers(net_desc, slots, tag, n_hidden):
            # saved activations of one network; compact save where the kernels allow it (mlp.saved_sizes: sign bits in slot 0)
            sizes = mlp_mod.saved_sizes(net_desc, N, n_hidden)
            net_desc.compact_save = int(n_hidden > 0 and sizes[0] != n_pad * 64)
            for i, n_el in enumerate(sizes):
                slots[i] = new(f"{tag}{i}", n_el)

        saved_buffers(d.density, d.saved_d, "saved_d", len(self.d_net.weights) - 1)
        if not self._fused_backward_takes_all(N):
            # some network's backward runs as a dX launch + a dW launch (samples per pixel or pixel features not in multiples of
            # 16, ...): one set of pre-activation gradients, shared by the networks' backwards one after the other
            for i in range(max(len(p.weights) - 1 for p in (self.d_net, self.s_net if self.has_lv else self.d_net, self.b_net if self.has_b else self.d_net))):
                d.dpre_scratch[i] = new(f"dpre{i}", n_pad * 64)
        if self.ks:
            d.se = new("se", B, self.ks)
        # pixel-feature gradients: one row per 16-sample group from the fused backward, one per sample from the launch pair
        # (csrc/step.hip decides per network by the same rule: divisibility and nesvor_mlp_backward_fused_ok at this N)
        group_div = N % 16 == 0 and S % 16 == 0 and self.ks % 16 == 0
        _, fused_s, fused_b = self._fused_backward_ok(N)
        rows_s = N // 16 if (group_div and fused_s) else N
        rows_b = N // 16 if (group_div and fused_b) else N
        if self.has_lv:
            d.log_var, d.dlv = new("log_var", N), new("dlv", N)
            saved_buffers(d.sigma, d.saved_s, "saved_s", len(self.s_net.weights) - 1)
            if self.ks:
                d.dxa = new("dxa", rows_s, self.ks)
        if self.has_b:
            d.log_bias, d.dlb, d.dpe_b = new("log_bias", N), new("dlb", N), new("dpe_b", self.kb_bias, N)
            d.lb_mean, d.mean_scratch = new("lb_mean", 1), new("mean_scratch", 256)
            saved_buffers(d.bias_net, d.saved_b, "saved_b", len(self.b_net.weights) - 1)
            if self.ks:
                d.dxa_b = new("dxa_b", rows_b, self.ks)
        if self.opt_T:
            d.dxl, d.du, d.dpix = new("dxl", B, S, 3), new("du", N, 3), new("dpix", B, 3, 4)
            d.trans_terms, d.g_trans = new("trans_terms", n), new("g_trans", n, 6)
        d.side_stream = self.side.cuda_stream
        handle = _lib.load().nesvor_step_create(ctypes.byref(d))
        if not handle:
            raise RuntimeError("nesvor_step_create failed")
        st = self._native[key] = {"desc": d, "handle": ctypes.c_void_p(handle), "buf": buf, "ws": None}
        if self._native_timing:
            _lib.check(_lib.load().nesvor_step_timing(st["handle"], 1), "step timing")
        return st

    # names of the spans nesvor_step_timing_read returns (NESVOR_STEP_SPAN_*), in order
    TIMED_SPANS = ("psf_transform_fwd", "hashgrid_fwd", "mlp_fwd_density", "mlp_fwd_sigma", "imaging_loss_bwd", "mlp_bwd_sigma",
                   "mlp_bwd_density", "hashgrid_bwd_aggregate", "hashgrid_bwd_owner", "psf_transform_bwd", "hashgrid_fwd_late",
                   "hashgrid_owner_fwd_union")

    def set_native_timing(self, on: bool) -> None:
        """HIP-event brackets around the launches of the one-call step, each on the stream its launch goes to (``nesvor_step_timing``:
        the product launches themselves are timed - the owner pass with its fused AdamW on the side stream)."""
        self._native_timing = bool(on)
        for st in self._native.values():
            _lib.check(_lib.load().nesvor_step_timing(st["handle"], int(on)), "step timing")

    def read_native_timing(self):
        """{span name: ms} of the LAST one-call step (waits for it); spans that did not occur are left out."""
        st = self._last_state
        if st is None:
            return {}
        out = (ctypes.c_float * len(self.TIMED_SPANS))()
        _lib.check(_lib.load().nesvor_step_timing_read(st["handle"], out), "step timing read")
        return {k: float(v) for k, v in zip(self.TIMED_SPANS, out) if v >= 0}

    def _run_native(self, xyz, v, slice_idx, adam=None, defer_table_join=False) -> Dict[str, torch.Tensor]:
        from .encoding import _workspace, queue_sizer

        m = self.model
        lib = _lib.load()
        dev = xyz.device
        B = xyz.shape[0]
        st = self._native_state(B)
        if self._owner_pending and self._pending_state is not None and self._pending_state is not st:
            # a table update the PREVIOUS step left on the side stream belongs to another context (other batch size or
            # operand mode): only that context's next run would wait for it - join here, before this one reads the table
            # and reuses the backward's workspace
            self.join_owner()
            _lib.check(lib.nesvor_step_join(self._pending_state["handle"], _lib.stream_ptr()), "step join")
        d = st["desc"]
        spec = m.inr.encoding.spec
        N = B * d.S
        sizer = queue_sizer(spec, N, dev)
        sizer.poll()  # grows the queues of levels that overflowed in an earlier backward (the workspace is then re-made)
        ws = _workspace(spec, N, dev, sizer)
        if ws is None:
            raise RuntimeError("hash grid outside the owner-computes backward's plan")
        if st["ws"] is not ws:
            st["ws"] = ws
            d.hg_workspace, d.queue_scale = ws.data_ptr(), ctypes.addressof(sizer.scale)
            _lib.check(lib.nesvor_step_update(st["handle"], ctypes.byref(d)), "step update")
        xyz, v, slice_idx = xyz.contiguous(), v.contiguous(), slice_idx.contiguous()
        seed = (torch.initial_seed() ^ self._noise_stream) & 0xFFFFFFFFFFFFFFFF
        offset = self._noise_calls
        self._noise_calls += 1
        vals = torch.empty(6, dtype=torch.float32, device=dev)
        a_ptr = None if adam is None else ctypes.byref(adam)
        stream = _lib.stream_ptr()
        args = (st["handle"], _lib.ptr(xyz), _lib.ptr(v), _lib.ptr(slice_idx), seed, offset, _lib.ptr(vals))
        with torch.cuda.device(dev):
            if self.split_level:
                from . import ddp

                _lib.check(lib.nesvor_step_run(*args, 1, self.split_level, None, stream), "training step (fine levels)")
                self._early = self._start_early()  # async: RCCL's stream, behind the launches above
                if self.early_update is not None and self.early_exchange is None and self._early is not None:
                    with torch.cuda.stream(self.side):
                        self.early_update(*self._early)
                _lib.check(lib.nesvor_step_run(*args, 2, self.split_level, None, stream), "training step (coarse levels)")
                # the coarse levels' owner pass runs on the side stream - and so did the early range's AdamW (early_update)
                self._owner_pending = bool(d.overlap_owner & 1) or (self.early_update is not None and self.early_exchange is None)
            else:
                # with its own AdamW the step has joined the owner pass - unless asked to leave the table's update (taken
                # inside the owner pass) on the side stream: the next run joins it right before its hash-grid forward, anyone
                # else through join_owner()
                defer = bool(defer_table_join) and adam is not None and (d.overlap_owner & 3) == 3
                _lib.check(lib.nesvor_step_run(*args, 0 | (_lib.STEP_DEFER_JOIN if defer else 0), 0, a_ptr, stream), "training step")
                self._owner_pending = bool(d.overlap_owner & 1) and (adam is None or defer)
        self._pending_state = st if self._owner_pending else None
        self._last_state = st
        sizer.snapshot(ws)
        losses = {D_LOSS: vals[0]}
        if self.has_var:
            losses[S_LOSS] = vals[1]
            losses[DS_LOSS] = vals[2]
        if self.opt_T:
            losses[T_REG] = vals[3]
        if self.has_b:
            losses[B_REG] = vals[5]
        losses[I_REG] = vals[4]
        return losses

    def __del__(self):
        try:
            lib = _lib.load()
            for st in self._native.values():
                lib.nesvor_step_destroy(st["handle"])
        except Exception:
            pass

    def join_owner(self) -> None:
        """Make the current stream wait for an owner pass still running on the side stream (the flat gradient of the hash
        table is complete only behind it).  ``run(defer_owner_join=True)`` leaves that to the caller - the fused trainer joins
        right before the optimizer / the gradient exchange, so the step's epilogue launch is not held back."""
        if self._owner_pending:
            torch.cuda.current_stream(self.flat.param.device).wait_stream(self.side)
            self._owner_pending = False

    @torch.no_grad()
    def run(self, xyz, v, slice_idx, noise=None, defer_owner_join: bool = False, adam=None, defer_table_join: bool = False) -> Dict[str, torch.Tensor]:
        """``adam`` (an ``_lib.AdamwT``): the one-call iteration also runs the optimizer (single process only); ignored - and
        left to the caller - on the Python issue path.  Returns the loss dict; ``self.ran_optimizer`` says who owns the step.
        ``defer_table_join`` (with ``adam``): the hash table's update may still be running on the side stream when this
        returns (losses and all other parameters are complete on the current stream); the next ``run`` waits for it where
        it first reads the table, any other reader of the table calls ``join_owner()`` first."""
        self.ran_optimizer = False
        if self.native_ready(noise):
            losses = self._run_native(xyz, v, slice_idx, adam, defer_table_join)
            self.ran_optimizer = adam is not None
            if not (defer_owner_join or (defer_table_join and adam is not None)):
                self.join_owner()
            return losses
        self.join_owner()  # (a table update a deferred native step left on the side stream)
        m, a = self.model, self.model.args
        lib = _lib.load()
        dev = xyz.device
        B, S = xyz.shape[0], a.n_samples
        N = B * S
        n = m.n_slices
        inr = m.inr
        enc = inr.encoding
        bb = inr.bounding_box
        xyz, v, slice_idx = xyz.contiguous(), v.contiguous(), slice_idx.contiguous()

        # pose regulariser (a long serial chain per slice, independent of the batch): on a side stream, joined
        # where its gradient is added
        main = torch.cuda.current_stream(dev)
        if self.opt_T:
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                per, g_t = trans_loss_raw(m.axisangle, m.axisangle_init)
                pose_reg_done = self.side.record_event()
            for t in (per, g_t):
                t.record_stream(main)

        # ---- forward ----------------------------------------------------------------------------------
        # PSF noise: explicit draws (replay mode), or drawn inside the sampler kernels from (seed, step counter) - the seed is
        # torch's current one, so torch.manual_seed() governs the run as before (NESVOR_PSF_NOISE=tensor: torch.randn)
        rng = None
        if noise is None:
            if self._kernel_noise:
                rng = (torch.initial_seed() ^ self._noise_stream, self._noise_calls)
                self._noise_calls += 1
            else:
                noise = torch.randn(B, S, 3, dtype=xyz.dtype, device=dev)
        # per-slice small tensors in one launch: c = n softmax(logit_coef), pose matrices, zeroed accumulators
        small = torch.empty(n * (1 + 12 + 13) + 1, dtype=torch.float32, device=dev)
        # last element: max |dpe|, raised by the density network's backward for the hash-grid backward (zeroed by the prologue)
        dpe_bound = None if self.has_b else small[26 * n :]
        c = small[:n] if self.has_c else None
        mat = small[n : 13 * n].view(n, 3, 4)
        acc = small[13 * n : 26 * n]  # [dc (n) | dmat (n,12)]
        with torch.cuda.device(dev):
            err = lib.nesvor_step_prologue(_lib.ptr(m.logit_coef if self.has_c else None), _lib.ptr(c), _lib.ptr(m.axisangle),
                                           _lib.ptr(mat), _lib.ptr(acc), 13 * n + 1, n, _lib.stream_ptr())
        _lib.check(err, "step prologue")
        x, u = sampler.forward_raw(mat, slice_idx, xyz, m.psf_sigma, noise, bb, rng, S)
        pe = hashgrid_forward(enc.spec, u, enc.params, _lib.LAYOUT_FEATURE_MAJOR, clustered=S >= 128)  # (E, N)
        dW, dB = self.d_net.weights, self.d_net.biases
        z, saved_d = mlp_mod.forward_raw(dW, dB, None, pe, 0, pe.shape[0], S, True, self.bf16)  # (1 + n_features_z, N)
        log_var = log_bias = lb_mean = None
        se = m.slice_embedding.weight[slice_idx] if self.ks else None
        if self.has_b:  # bias field: [slice embedding | the coarsest levels of pe] -> log bias (models.py:341-346)
            bW, bB = self.b_net.weights, self.b_net.biases
            log_bias, saved_b = mlp_mod.forward_raw(bW, bB, se, pe, 0, self.kb_bias, S, True, self.bf16)  # (1, N)
            lb_mean = log_bias.mean().reshape(1)
            if self.parallel:  # biasReg = (mean log_bias)^2 is not a mean of per-sample terms: use the GLOBAL mean, so that
                # the averaged gradients and the loss value are exactly those of the undivided batch (SURVEY.md 8e caveat 1)
                torch.distributed.all_reduce(lb_mean)
                lb_mean /= self.world
        if self.has_lv:
            sW, sB = self.s_net.weights, self.s_net.biases
            log_var, saved_s = mlp_mod.forward_raw(sW, sB, se, z, 1, a.n_features_z, S, True, self.bf16)  # (1, N)
        lvs = m.log_var_slice if self.has_lvs else None

        # ---- losses: values and gradients in one launch -------------------------------------------------
        dz = torch.empty_like(z)  # row 0 <- loss kernel, rows [1, 1+n_features_z) <- sigma_net backward
        written = 1 + (a.n_features_z if self.has_lv else 0)
        if written < z.shape[0]:
            dz[written:].zero_()
        dlv = torch.empty(N, dtype=torch.float32, device=dev) if self.has_lv else None
        dxl = torch.empty_like(x) if self.opt_T else None
        loss_pix = torch.empty((B, 3), dtype=torch.float32, device=dev)
        dc, dmat = acc[:n], acc[n:].view(n, 3, 4)  # zero-filled by the prologue
        pix = torch.empty((2, B), dtype=torch.float32, device=dev)
        dlb = torch.empty(N, dtype=torch.float32, device=dev) if self.has_b else None
        la = loss_mod._fill(z[0], log_var, log_bias, x, v, slice_idx, c, lvs, lb_mean, self.reg_type, self.delta)
        la.dlog_bias = None if dlb is None else dlb.data_ptr()
        la.gw, la.loss_pix, la.dz0 = self.gw.data_ptr(), loss_pix.data_ptr(), dz[0].data_ptr()
        la.dlog_var = None if dlv is None else dlv.data_ptr()
        la.dx = None if dxl is None else dxl.data_ptr()
        la.dc_pix = pix[0].data_ptr() if self.has_c else None
        la.dlvs_pix = pix[1].data_ptr() if self.has_lvs else None
        with torch.cuda.device(dev), _lib.kernel_timer.span("imaging_loss_bwd"):
            err = lib.nesvor_imaging_loss(ctypes.byref(la), _lib.stream_ptr())
        _lib.check(err, "imaging loss (value + gradient)")

        # ---- backward through the networks ----------------------------------------------------------------
        dxa = None
        if self.has_lv:
            dxa, partial_s = mlp_mod.backward_raw(sW, sB, se, z, dlv.view(1, N), saved_s, 1, a.n_features_z, S,
                                                  dz[1 : 1 + a.n_features_z], se is not None, self.bf16)
            self.s_net.store_grads(partial_s)
        dpe = torch.empty_like(pe)
        _, partial_d = mlp_mod.backward_raw(dW, dB, None, pe, dz, saved_d, 0, pe.shape[0], S, dpe, False, self.bf16, dxb_absmax=dpe_bound)
        self.d_net.store_grads(partial_d)
        dxa_b = None
        if self.has_b:
            dpe_b = torch.empty((self.kb_bias, N), dtype=torch.float32, device=dev)
            dxa_b, partial_b = mlp_mod.backward_raw(bW, bB, se, pe, dlb.view(1, N), saved_b, 0, self.kb_bias, S, dpe_b,
                                                    se is not None, self.bf16)
            self.b_net.store_grads(partial_b)
            dpe[: self.kb_bias] += dpe_b
            if dxa is None:
                dxa, dxa_b = dxa_b, None
        gt = enc.params.grad.view(-1)
        if self.split_level:
            from . import ddp

            L = enc.spec.n_levels
            _, du = hashgrid_backward(enc.spec, u, enc.params, dpe, gt, self.opt_T, _lib.LAYOUT_FEATURE_MAJOR,
                                      levels=(self.split_level, L), dy_bound=dpe_bound)
            self._early = self._start_early()  # async: RCCL's stream, behind the launch above
            _, du = hashgrid_backward(enc.spec, u, enc.params, dpe, gt, self.opt_T, _lib.LAYOUT_FEATURE_MAJOR,
                                      levels=(0, self.split_level), grad_u=du, first=False, dy_bound=dpe_bound)
        else:
            # (per-kernel event timing needs both launches on one stream; NESVOR_OWNER_OVERLAP=0: the same for a kernel trace)
            overlap_owner = not _lib.kernel_timer.enabled and self._overlap_owner
            _, du = hashgrid_backward(enc.spec, u, enc.params, dpe, gt, self.opt_T, _lib.LAYOUT_FEATURE_MAJOR,
                                      owner_stream=self.side if overlap_owner else None, dy_bound=dpe_bound)
            self._owner_pending = overlap_owner
        dpix = sampler.backward_raw(mat, slice_idx, xyz, m.psf_sigma, noise, bb, dxl, du, rng, S) if self.opt_T else None

        # ---- per-slice parameters ---------------------------------------------------------------------------
        g_se = m.slice_embedding.weight.grad if self.ks else None
        with torch.cuda.device(dev):
            err = lib.nesvor_slice_grads(
                _lib.ptr(slice_idx), la.dc_pix, la.dlvs_pix, _lib.ptr(dxa), _lib.ptr(dpix), _lib.ptr(dc),
                m.log_var_slice.grad.data_ptr() if self.has_lvs else None, _lib.ptr(g_se), _lib.ptr(dmat), B,
                (dxa.shape[0] // B) if dxa is not None else S, self.ks, _lib.stream_ptr())
        _lib.check(err, "slice_grads")
        if dxa_b is not None:  # second consumer of the slice embedding (sigma_net's share went in with the call above)
            with torch.cuda.device(dev):
                err = lib.nesvor_slice_grads(_lib.ptr(slice_idx), None, None, _lib.ptr(dxa_b), None, None, None, _lib.ptr(g_se),
                                             None, B, dxa_b.shape[0] // B, self.ks, _lib.stream_ptr())
            _lib.check(err, "slice_grads (bias field)")
        # d logit_coef, d axisangle (+ pose regulariser) and the loss values: one launch
        vals = torch.empty(5, dtype=torch.float32, device=dev)
        img_scale = (self.delta if self.reg_type == 0 else 1.0) / (B * S)
        img_off = -self.delta if self.reg_type == 0 else 0.0
        if self.opt_T:
            main.wait_event(pose_reg_done)  # the pose regulariser from the start of the step; NOT the owner pass behind it
        with torch.cuda.device(dev):
            err = lib.nesvor_step_epilogue(
                _lib.ptr(dc if self.has_c else None), _lib.ptr(c), _lib.ptr(m.logit_coef.grad if self.has_c else None),
                _lib.ptr(dmat if self.opt_T else None), _lib.ptr(m.axisangle), _lib.ptr(g_t if self.opt_T else None), self.w_T,
                _lib.ptr(m.axisangle.grad if self.opt_T else None), _lib.ptr(loss_pix), _lib.ptr(per if self.opt_T else None),
                _lib.ptr(vals), n, B, img_scale, img_off, _lib.stream_ptr())
        _lib.check(err, "step epilogue")
        losses = {D_LOSS: vals[0]}
        if self.has_var:
            losses[S_LOSS] = vals[1]
            losses[DS_LOSS] = vals[2]
        if self.opt_T:
            losses[T_REG] = vals[3]
        if self.has_b:
            losses[B_REG] = lb_mean[0] ** 2
        losses[I_REG] = vals[4]
        if not defer_owner_join:
            self.join_owner()
        return losses
