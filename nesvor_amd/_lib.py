"""ctypes binding of libnesvor_hip.so (the C ABI declared in include/nesvor_hip.h).

The library is the product's only compute backend for the native ops: there is
no CPU or eager fallback.  ``load()`` raises if the shared object is missing and
every op wrapper raises on non-device tensors.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_double, c_float, c_int, c_int32, c_int64, c_uint32, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# NESVOR_HIP_LIB: load another build of the same ABI (tools/ablate_hashgrid.py times variants of one kernel this way)
LIB_PATH = os.environ.get("NESVOR_HIP_LIB") or os.path.join(_HERE, "lib", "libnesvor_hip.so")
MAX_LEVELS = 32
ABI_VERSION = 35

LAYOUT_ROW_MAJOR = 0
LAYOUT_FEATURE_MAJOR = 1
STEP_DEFER_JOIN = 8  # nesvor_step_run: OR-ed into `phase` (NESVOR_STEP_DEFER_JOIN)
LAYOUT_CLUSTERED = 4  # forward hint, OR-ed into the layout: 256 consecutive points are one spatial cluster
LAYOUT_UNCLUSTERED = 8  # backward hint, OR-ed into the layout: consecutive points are not clustered - order them by cell first
LAYOUT_DY_SCRATCH = 16  # with LAYOUT_UNCLUSTERED | LAYOUT_FEATURE_MAJOR: the workspace (sized by ..._workspace_bytes_ex) has room for dpe as rows


class GridT(Structure):
    """Mirror of nesvor_grid_t."""

    _fields_ = [
        ("n_levels", c_int32),
        ("n_features", c_int32),
        ("scale", c_float * MAX_LEVELS),
        ("res", c_uint32 * MAX_LEVELS),
        ("size", c_uint32 * MAX_LEVELS),
        ("offset", c_uint32 * MAX_LEVELS),
        ("hashed", c_uint32 * MAX_LEVELS),
    ]


class MlpT(Structure):
    """Mirror of nesvor_mlp_t."""

    _fields_ = [
        ("width", c_int32), ("n_hidden", c_int32), ("out_dim", c_int32),
        ("k_a", c_int32), ("k_b", c_int32), ("b_row0", c_int32), ("samples_per_pixel", c_int32),
        ("dxa_group_sums", c_int32), ("bf16_operands", c_int32), ("compact_save", c_int32),
        ("weight", c_void_p * 4), ("bias", c_void_p * 4),
        ("prep", c_void_p), ("y_absmax", c_void_p), ("weight_images", c_void_p),
    ]


class MlpWideT(Structure):
    """Mirror of nesvor_mlp_wide_t (width <= 128, up to seven hidden layers: csrc/mlp_wide.hip)."""

    _fields_ = [
        ("width", c_int32), ("n_hidden", c_int32), ("out_dim", c_int32),
        ("k_a", c_int32), ("k_b", c_int32), ("b_row0", c_int32), ("samples_per_pixel", c_int32), ("reserved", c_int32),
        ("weight", c_void_p * 8), ("bias", c_void_p * 8),
    ]


class LossT(Structure):
    """Mirror of nesvor_loss_t."""

    _fields_ = [(n, c_void_p) for n in (
        "z0", "log_var", "log_bias", "x", "v", "slice_idx", "c", "log_var_slice", "log_bias_mean", "gw",
        "loss_pix", "dz0", "dlog_var", "dlog_bias", "dx", "dc_pix", "dlvs_pix")] + [
        ("B", c_int32), ("S", c_int32), ("reg_type", c_int32), ("delta", c_float),
        ("dz0_absmax", c_void_p), ("dlog_var_absmax", c_void_p), ("dlog_bias_absmax", c_void_p)]


class StepT(Structure):
    """Mirror of nesvor_step_t (one training iteration behind one entry point, csrc/step.hip)."""

    _fields_ = (
        [("grid", GridT), ("density", MlpT), ("sigma", MlpT), ("bias_net", MlpT)]
        + [(n, c_int32) for n in ("B", "S", "n_slices", "opt_T", "has_lv", "has_c", "has_lvs", "has_b", "n_features_z", "ks", "kb_bias",
                                  "reg_type", "overlap_owner")]
        + [("delta", c_float), ("w_T", c_float)]
        + [(n, c_void_p) for n in ("axisangle", "axisangle_init", "psf_sigma", "bounding_box", "logit_coef", "log_var_slice",
                                   "slice_embedding", "table")]
        + [(n, c_void_p) for n in ("g_axisangle", "g_logit_coef", "g_log_var_slice", "g_slice_embedding", "g_table", "g_density", "g_sigma",
                                   "g_bias_net")]
        + [(n, c_int32) for n in ("n_density_params", "n_sigma_params", "n_bias_params")]
        + [("gw", c_void_p)]
        + [(n, c_void_p) for n in ("flat_param", "flat_grad", "flat_exp_avg", "flat_exp_avg_sq")]
        + [("flat_numel", c_int64)]
        + [(n, c_void_p) for n in ("small", "x", "u", "pe", "z", "log_var", "log_bias", "se", "dz", "dlv", "dlb", "dxl", "loss_pix", "pix",
                                   "dpe", "dpe_b", "du", "dpix", "dxa", "dxa_b", "trans_terms", "g_trans", "lb_mean", "mean_scratch",
                                   "partial")]
        + [("saved_d", c_void_p * 4), ("saved_s", c_void_p * 4), ("saved_b", c_void_p * 4), ("dpre_scratch", c_void_p * 4)]
        + [("hg_workspace", c_void_p), ("queue_scale", c_void_p), ("side_stream", c_void_p)]
    )


class AdamwT(Structure):
    """Mirror of nesvor_adamw_t."""

    _fields_ = [(n, c_float) for n in ("lr", "beta1", "beta2", "eps", "weight_decay", "bias_correction1", "bias_correction2", "grad_scale")]


_lib = None

_P = c_void_p
_SIGNATURES = {
    "nesvor_hip_abi_version": ([], c_int),
    "nesvor_axisangle2mat_forward": ([_P, _P, c_int, _P], c_int),
    "nesvor_axisangle2mat_backward": ([_P, _P, _P, c_int, _P], c_int),
    "nesvor_mat2axisangle_forward": ([_P, _P, c_int, _P], c_int),
    "nesvor_mat2axisangle_backward": ([_P, _P, _P, c_int, _P], c_int),
    "nesvor_axisangle2mat_forward_f64": ([_P, _P, c_int, _P], c_int),
    "nesvor_axisangle2mat_backward_f64": ([_P, _P, _P, c_int, _P], c_int),
    "nesvor_mat2axisangle_forward_f64": ([_P, _P, c_int, _P], c_int),
    "nesvor_mat2axisangle_backward_f64": ([_P, _P, _P, c_int, _P], c_int),
    "nesvor_trans_loss": ([_P, _P, _P, _P, c_int, _P], c_int),
    "nesvor_slice_acq_forward": (
        [_P, _P, _P, _P, _P, _P, _P] + [c_int] * 9 + [c_float, c_int, _P],
        c_int,
    ),
    "nesvor_slice_acq_adjoint_forward": ([_P] * 8 + [c_int] * 9 + [c_float, c_int, _P], c_int),
    "nesvor_slice_acq_backward": ([_P] * 9 + [c_int] * 9 + [c_float, _P], c_int),
    "nesvor_slice_acq_adjoint_backward": ([_P] * 10 + [c_int] * 9 + [c_float, c_int, _P], c_int),
    "nesvor_slice_acq_forward_f64": ([_P, _P, _P, _P, _P, _P, _P] + [c_int] * 9 + [c_double, c_int, _P], c_int),
    "nesvor_slice_acq_adjoint_forward_f64": ([_P] * 8 + [c_int] * 9 + [c_double, c_int, _P], c_int),
    "nesvor_slice_acq_backward_f64": ([_P] * 9 + [c_int] * 9 + [c_double, _P], c_int),
    "nesvor_slice_acq_adjoint_backward_f64": ([_P] * 10 + [c_int] * 9 + [c_double, c_int, _P], c_int),
    # interp_psf = true (no scratch; outputs zero-filled by the caller)
    "nesvor_slice_acq_adjoint_forward_interp": ([_P] * 7 + [c_int] * 9 + [c_float, c_int, _P], c_int),
    "nesvor_slice_acq_backward_interp": ([_P] * 8 + [c_int] * 9 + [c_float, _P], c_int),
    "nesvor_slice_acq_adjoint_backward_interp": ([_P] * 10 + [c_int] * 9 + [c_float, c_int, _P], c_int),
    "nesvor_slice_acq_adjoint_forward_interp_f64": ([_P] * 7 + [c_int] * 9 + [c_double, c_int, _P], c_int),
    "nesvor_slice_acq_backward_interp_f64": ([_P] * 8 + [c_int] * 9 + [c_double, _P], c_int),
    "nesvor_slice_acq_adjoint_backward_interp_f64": ([_P] * 10 + [c_int] * 9 + [c_double, c_int, _P], c_int),
    "nesvor_hashgrid_forward": ([POINTER(GridT), _P, _P, _P, c_int64, c_int, _P], c_int),
    "nesvor_hashgrid_forward_bounded": ([POINTER(GridT), _P, _P, _P, c_int64, c_int, _P, _P], c_int),
    "nesvor_hashgrid_forward_workspace_bytes": ([POINTER(GridT), c_int64, c_int], c_int64),
    "nesvor_hashgrid_forward_unclustered": ([POINTER(GridT), _P, _P, _P, c_int64, c_int, _P, _P, c_int64, _P], c_int),
    "nesvor_hashgrid_backward_workspace_bytes": ([POINTER(GridT), c_int64, _P], c_int64),
    "nesvor_hashgrid_backward_workspace_bytes_ex": ([POINTER(GridT), c_int64, _P, c_int], c_int64),
    "nesvor_hashgrid_backward_workspace_zero_bytes": ([], c_int64),
    "nesvor_hashgrid_backward_overflow_offset": ([_P], c_int64),
    "nesvor_hashgrid_backward": ([POINTER(GridT), _P, _P, _P, _P, _P, c_int64, c_int, _P, c_int, _P, _P], c_int),
    "nesvor_hashgrid_backward_levels": ([POINTER(GridT), _P, _P, _P, _P, _P, c_int64, c_int, _P, c_int, c_int, c_int, _P, _P], c_int),
    "nesvor_hashgrid_backward_bounded": ([POINTER(GridT), _P, _P, _P, _P, _P, c_int64, c_int, _P, c_int, c_int, c_int, _P, _P, _P], c_int),
    "nesvor_hashgrid_backward_atomic": ([POINTER(GridT), _P, _P, _P, _P, _P, c_int64, c_int, _P], c_int),
    "nesvor_psf_transform_forward": ([_P] * 8 + [c_int, c_int, _P], c_int),
    "nesvor_psf_transform_backward": ([_P] * 9 + [c_int, c_int, _P], c_int),
    "nesvor_psf_transform_forward_rng_gather": ([_P] * 4 + [c_uint64, c_uint64] + [_P] * 3 + [c_int, c_int, _P, _P, c_int, _P], c_int),
    "nesvor_psf_transform_forward_rng": ([_P] * 4 + [c_uint64, c_uint64] + [_P] * 3 + [c_int, c_int, _P], c_int),
    "nesvor_psf_transform_backward_rng": ([_P] * 4 + [c_uint64, c_uint64] + [_P] * 4 + [c_int, c_int, _P], c_int),
    "nesvor_psf_transform_backward_rng_slices": ([_P] * 4 + [c_uint64, c_uint64] + [_P] * 5 + [c_int, c_int, _P], c_int),
    "nesvor_psf_noise": ([c_uint64, c_uint64, _P, c_int64, _P], c_int),
    "nesvor_mlp_compact_save_ok": ([POINTER(MlpT), c_int64], c_int),
    "nesvor_mlp_backward_fused_ok": ([POINTER(MlpT), c_int64], c_int),
    "nesvor_mlp_prepare": ([POINTER(MlpT), _P, _P, _P, c_int64, _P, c_int, _P], c_int),
    "nesvor_mlp_prepare_weights": ([_P, _P, c_int, _P, c_int64, _P, _P], c_int),
    "nesvor_mlp_prepare_weights_images": ([_P, _P, _P, c_int, _P, c_int64, _P, _P], c_int),
    "nesvor_mlp_weight_images_bytes": ([POINTER(MlpT)], c_int64),
    "nesvor_mlp_forward": ([POINTER(MlpT), _P, _P, _P, POINTER(c_void_p), c_int64, _P], c_int),
    "nesvor_mlp_backward": (
        [POINTER(MlpT), _P, _P, _P, POINTER(c_void_p), POINTER(c_void_p), _P, _P, _P, c_int, c_int64, _P],
        c_int,
    ),
    "nesvor_mlp_backward_bounded": (
        [POINTER(MlpT), _P, _P, _P, POINTER(c_void_p), POINTER(c_void_p), _P, _P, _P, c_int, c_int64, _P, _P],
        c_int,
    ),
    "nesvor_mlp_wide_saved_floats": ([POINTER(MlpWideT), c_int64], c_int64),
    "nesvor_mlp_wide_param_count": ([POINTER(MlpWideT)], c_int),
    "nesvor_mlp_wide_forward": ([POINTER(MlpWideT), _P, _P, _P, POINTER(c_void_p), c_int64, _P], c_int),
    "nesvor_mlp_wide_backward": (
        [POINTER(MlpWideT), _P, _P, _P, POINTER(c_void_p), POINTER(c_void_p), _P, _P, _P, c_int, c_int64, _P],
        c_int,
    ),
    "nesvor_mlp_wide_backward_bounded": (
        [POINTER(MlpWideT), _P, _P, _P, POINTER(c_void_p), POINTER(c_void_p), _P, _P, _P, c_int, c_int64, _P, _P],
        c_int,
    ),
    "nesvor_imaging_loss": ([POINTER(LossT), _P], c_int),
    "nesvor_slice_grads_by_slice": ([_P] * 9 + [c_int, c_int, c_int, c_int, _P], c_int),
    "nesvor_slice_grads": ([_P] * 9 + [c_int, c_int, c_int, _P], c_int),
    "nesvor_step_prologue": ([_P] * 5 + [c_int, c_int, _P], c_int),
    "nesvor_step_prologue_pose": ([_P] * 5 + [c_int, c_int, _P, _P, _P, _P], c_int),
    "nesvor_step_epilogue": ([_P] * 6 + [c_float, _P, _P, _P, _P, c_int, c_int, c_float, c_float, _P], c_int),
    "nesvor_hashgrid_backward_adamw": ([POINTER(GridT), _P, _P, _P, _P, _P, c_int64, c_int, _P, c_int, _P, _P, _P, _P, POINTER(AdamwT), _P], c_int),
    "nesvor_hashgrid_backward_adamw_levels": ([POINTER(GridT), _P, _P, _P, _P, _P, c_int64, c_int, _P, c_int, c_int, c_int, _P, _P, _P, _P, POINTER(AdamwT), _P], c_int),
    "nesvor_hashgrid_forward_levels": ([POINTER(GridT), _P, _P, _P, c_int64, c_int, _P, c_int, c_int, _P], c_int),
    "nesvor_adamw_step": (
        [_P, _P, _P, _P, c_int64] + [c_float] * 8 + [c_int, _P],
        c_int,
    ),
    "nesvor_sum_rows_multi": ([_P, _P, _P, _P, c_int, c_int, _P], c_int),
    "nesvor_sum_rows": ([_P, _P, c_int, c_int, c_int, _P], c_int),
    "nesvor_step_create": ([POINTER(StepT)], c_void_p),
    "nesvor_step_update": ([_P, POINTER(StepT)], c_int),
    "nesvor_step_destroy": ([_P], None),
    "nesvor_step_join": ([_P, _P], c_int),
    "nesvor_step_timing": ([_P, c_int], c_int),
    "nesvor_step_timing_read": ([_P, _P], c_int),
    "nesvor_step_run": ([_P, _P, _P, _P, c_uint64, c_uint64, _P, c_int, c_int, POINTER(AdamwT), _P], c_int),
    "nesvor_vvr_similarity": ([_P, c_int, c_int, c_int, _P, _P, _P, _P, c_int64, c_int, _P, _P, _P], c_int),
}


def exported_symbols():
    """Names declared in include/nesvor_hip.h (used by the CPU symbol test)."""
    return list(_SIGNATURES)


def load():
    """Load the shared library once; fail loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP backend has not been built. "
            "Run `python -m nesvor_amd.csrc.build` (or __graft_entry__.build()). "
            "There is no CPU fallback for the native ops."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (argtypes, restype) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = restype
    v = lib.nesvor_hip_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"libnesvor_hip ABI {v} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def stream_ptr(device=None):
    """hipStream_t of torch's current stream on `device` (so launches order with torch ops)."""
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())


def check(err, what):
    if err != 0:
        raise RuntimeError(f"{what}: HIP error {err}")


def require_device(*tensors, dtype=None, name="tensor"):
    """Reference semantics (CHECK_CUDA / CHECK_CONTIGUOUS): raise on host or strided input."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a device (HIP) tensor — the native ops have no CPU path")
        if not t.is_contiguous():
            raise RuntimeError(f"{name} must be contiguous")
        if dtype is not None and t.dtype != dtype:
            raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")


class KernelTimer:
    """Optional HIP-event timing of native launches on the launch stream (bench.py's roofline leg).
    Disabled by default: zero overhead on the product path."""

    def __init__(self):
        self.enabled = False
        self.records = {}

    def reset(self, enabled):
        self.enabled = enabled
        self.records = {}

    class _Span:
        def __init__(self, timer, name):
            self.timer, self.name = timer, name

        def __enter__(self):
            if self.timer.enabled:
                self.s = torch.cuda.Event(enable_timing=True)
                self.e = torch.cuda.Event(enable_timing=True)
                self.s.record()  # torch's current stream == the stream the C ABI launches on
            return self

        def __exit__(self, *exc):
            if self.timer.enabled:
                self.e.record()
                self.timer.records.setdefault(self.name, []).append((self.s, self.e))
            return False

    def span(self, name):
        return KernelTimer._Span(self, name)

    def summary(self):
        """name -> (count, mean ms); call after torch.cuda.synchronize()."""
        return {k: (len(v), sum(s.elapsed_time(e) for s, e in v) / len(v)) for k, v in self.records.items()}


kernel_timer = KernelTimer()
