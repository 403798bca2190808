"""PyTorch-ROCm custom operators of the NeSVoR path (dispatcher namespace ``nesvor``).

Every native kernel family of ``libnesvor_hip.so`` is registered with ``torch.library``:

* a schema (``torch.ops.nesvor.<op>``), so the ops are visible to the dispatcher, ``torch.compile`` and profilers;
* ONE backend kernel, under the ``CUDA`` dispatch key (PyTorch-ROCm files HIP tensors under that key).  No CPU kernel is
  registered: a host tensor fails in the dispatcher ("Could not run 'nesvor::...' with arguments from the 'CPU' backend") -
  the product has no CPU path, by construction;
* a fake (meta) kernel: output shapes / dtypes without touching data;
* an autograd formula (``register_autograd``) for the differentiable ops, built from the matching ``*_backward`` op.

The backend kernels are the C ABI of ``include/nesvor_hip.h`` (plain pointers + sizes + ``hipStream_t``) called through
``ctypes`` on torch's current stream; this module is the "thin torch-extension layer" of SURVEY.md 8(b): validate
device / dtype / contiguity, allocate outputs, launch.  The reference's two pybind modules keep their names
(``nesvor_amd.transform_convert_cuda``, ``nesvor_amd.slice_acq_cuda``) and forward here; the fused training step
(``nesvor_amd.direct``) calls the same C entry points back-to-back without going through the dispatcher.

Replaces: ``nesvor/transform/transform_convert_cuda.cpp:64-69`` and ``nesvor/slice_acquisition/slice_acq_cuda.cpp:156-161``
(PYBIND11_MODULE tables), ``tinycudann``'s autograd Functions (``nesvor/nesvor/models.py:25,31``).
"""
import ctypes
from typing import List, Optional

import torch
from torch.library import Library, register_autograd, register_fake

from . import _lib
from . import encoding as _enc
from . import loss as _loss
from . import mlp as _mlp
from . import sampler as _sampler
from .grid import HashGridSpec

NAMESPACE = "nesvor"
_LIBRARY = Library(NAMESPACE, "DEF")
SCHEMAS = {}


def _op(schema: str, kernel, fake=None, backward=None, setup_context=None):
    name = schema.split("(", 1)[0]
    _LIBRARY.define(schema)
    _LIBRARY.impl(name, kernel, "CUDA")
    if fake is not None:
        register_fake(f"{NAMESPACE}::{name}")(fake)
    if backward is not None:
        register_autograd(f"{NAMESPACE}::{name}", backward, setup_context=setup_context)
    SCHEMAS[name] = schema


def _empty(like: torch.Tensor) -> torch.Tensor:
    """Stand-in for an output that was not requested (the reference returns an undefined Tensor there)."""
    return torch.empty(0, dtype=like.dtype, device=like.device)


def _opt(t: torch.Tensor) -> Optional[torch.Tensor]:
    return t if t.numel() > 0 else None


# =====================================================================================================================
# rigid-transform conversions (reference module nesvor.transform_convert_cuda; float32 and float64)
# =====================================================================================================================
def _tc_suffix(t):
    if t.dtype == torch.float32:
        return ""
    if t.dtype == torch.float64:
        return "_f64"
    raise RuntimeError(f"transform_convert: unsupported dtype {t.dtype}")


def _tc_launch(symbol, what, n, *tensors):
    fn = getattr(_lib.load(), symbol)
    with torch.cuda.device(tensors[0].device):
        _lib.check(fn(*[_lib.ptr(t) for t in tensors], n, _lib.stream_ptr()), what)


def _ax2mat_fwd(axisangle):
    _lib.require_device(axisangle, name="axisangle")
    mat = torch.zeros((axisangle.shape[0], 3, 4), dtype=axisangle.dtype, device=axisangle.device)
    _tc_launch("nesvor_axisangle2mat_forward" + _tc_suffix(axisangle), "axisangle2mat_forward", axisangle.shape[0], axisangle, mat)
    return mat


def _ax2mat_bwd(grad_mat, axisangle):
    _lib.require_device(grad_mat, axisangle, name="grad_mat/axisangle")
    grad = torch.zeros((axisangle.shape[0], 6), dtype=axisangle.dtype, device=axisangle.device)
    _tc_launch("nesvor_axisangle2mat_backward" + _tc_suffix(axisangle), "axisangle2mat_backward", axisangle.shape[0], grad_mat, axisangle, grad)
    return grad


def _mat2ax_fwd(mat):
    _lib.require_device(mat, name="mat")
    ax = torch.zeros((mat.shape[0], 6), dtype=mat.dtype, device=mat.device)
    _tc_launch("nesvor_mat2axisangle_forward" + _tc_suffix(mat), "mat2axisangle_forward", mat.shape[0], mat, ax)
    return ax


def _mat2ax_bwd(mat, grad_axisangle):
    _lib.require_device(mat, grad_axisangle, name="mat/grad_axisangle")
    grad = torch.zeros((mat.shape[0], 3, 4), dtype=mat.dtype, device=mat.device)
    _tc_launch("nesvor_mat2axisangle_backward" + _tc_suffix(mat), "mat2axisangle_backward", mat.shape[0], mat, grad_axisangle, grad)
    return grad


_op("axisangle2mat_backward(Tensor grad_mat, Tensor axisangle) -> Tensor", _ax2mat_bwd,
    fake=lambda g, ax: ax.new_empty((ax.shape[0], 6)))
_op("axisangle2mat_forward(Tensor axisangle) -> Tensor", _ax2mat_fwd,
    fake=lambda ax: ax.new_empty((ax.shape[0], 3, 4)),
    backward=lambda ctx, g: torch.ops.nesvor.axisangle2mat_backward(g.contiguous(), ctx.saved_tensors[0]),
    setup_context=lambda ctx, inputs, output: ctx.save_for_backward(inputs[0]))
_op("mat2axisangle_backward(Tensor mat, Tensor grad_axisangle) -> Tensor", _mat2ax_bwd,
    fake=lambda mat, g: mat.new_empty((mat.shape[0], 3, 4)))
_op("mat2axisangle_forward(Tensor mat) -> Tensor", _mat2ax_fwd,
    fake=lambda mat: mat.new_empty((mat.shape[0], 6)),
    backward=lambda ctx, g: torch.ops.nesvor.mat2axisangle_backward(ctx.saved_tensors[0], g.contiguous()),
    setup_context=lambda ctx, inputs, output: ctx.save_for_backward(inputs[0]))


# pose regulariser: value and gradient in one launch (models.py:357-363)
def _trans_loss(axisangle, axisangle_init):
    from .transform import trans_loss_raw

    per, grad = trans_loss_raw(axisangle, axisangle_init)
    return per.sum(), grad


def _trans_loss_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])
    ctx.set_materialize_grads(False)


_op("trans_loss(Tensor axisangle, Tensor axisangle_init) -> (Tensor, Tensor)", _trans_loss,
    fake=lambda ax, ax0: (ax.new_empty(()), torch.empty_like(ax)),
    backward=lambda ctx, g, g_unused: (None if g is None else ctx.saved_tensors[0] * g, None),
    setup_context=_trans_loss_setup)


# =====================================================================================================================
# slice acquisition A / A^T (reference module nesvor.slice_acq_cuda); "None" masks are empty tensors, as in
# slice_acq.py:36-39; outputs that were not requested come back as empty tensors
# =====================================================================================================================
def _mask(m):
    m = m if (m is not None and m.numel() > 0) else None
    if m is not None:
        _lib.require_device(m, dtype=torch.bool, name="mask")
    return m


def _sa_dims(vol_shape, psf):
    return tuple(int(s) for s in vol_shape[-3:]) + tuple(int(s) for s in psf.shape)


def _sa_symbol(name, t, interp_psf=False):
    """float32 / float64 (the reference's AT_DISPATCH_FLOATING_TYPES, slice_acq_cuda_kernel.cu:970-1114).  The
    PSF-interpolating mode of the backward / adjoint operators (:229-279, :526-572, :754) has its own entry points
    (``*_interp``: per-pixel scatter as the reference formulates it; no caller in the reference tree uses the mode)."""
    if t.dtype not in (torch.float32, torch.float64):
        raise NotImplementedError(f"slice_acq: float32 or float64 expected, got {t.dtype}")
    return getattr(_lib.load(), name + ("_interp" if interp_psf else "") + ("_f64" if t.dtype == torch.float64 else ""))


def _sa_forward(transforms, vol, vol_mask, slices_mask, psf, slice_shape, res_slice, need_weight, interp_psf):
    _lib.require_device(transforms, vol, psf, dtype=vol.dtype, name="transforms/vol/psf")
    vm, sm = _mask(vol_mask), _mask(slices_mask)
    n, (h, w) = transforms.shape[0], (int(slice_shape[0]), int(slice_shape[1]))
    slices = torch.zeros((n, 1, h, w), dtype=vol.dtype, device=vol.device)
    weight = torch.zeros((n, 1, h, w), dtype=vol.dtype, device=vol.device) if need_weight else None
    with torch.cuda.device(vol.device):
        err = _sa_symbol("nesvor_slice_acq_forward", vol)(
            _lib.ptr(transforms), _lib.ptr(vol), _lib.ptr(vm), _lib.ptr(sm), _lib.ptr(psf), _lib.ptr(slices), _lib.ptr(weight),
            *_sa_dims(vol.shape, psf), n, h, w, float(res_slice), int(bool(interp_psf)), _lib.stream_ptr())
    _lib.check(err, "slice_acq forward")
    return [slices, weight] if need_weight else [slices]


def _sa_backward(transforms, vol, vol_mask, psf, grad_slices, slices_mask, res_slice, interp_psf, need_vol_grad, need_transforms_grad):
    _lib.require_device(transforms, vol, psf, grad_slices, dtype=vol.dtype, name="slice_acq backward input")
    vm, sm = _mask(vol_mask), _mask(slices_mask)
    n, h, w = grad_slices.shape[0], grad_slices.shape[-2], grad_slices.shape[-1]
    new = torch.zeros_like if interp_psf else torch.empty_like  # the interp kernels accumulate (atomics) into their outputs
    grad_vol = new(vol) if need_vol_grad else None
    grad_tf = new(transforms) if need_transforms_grad else None
    scratch = () if interp_psf else (_lib.ptr(torch.empty(n * h * w, dtype=vol.dtype, device=vol.device)),)
    with torch.cuda.device(vol.device):
        err = _sa_symbol("nesvor_slice_acq_backward", vol, interp_psf)(
            _lib.ptr(transforms), _lib.ptr(vol), _lib.ptr(vm), _lib.ptr(psf), _lib.ptr(grad_slices), _lib.ptr(sm),
            _lib.ptr(grad_vol), _lib.ptr(grad_tf), *scratch, *_sa_dims(vol.shape, psf), n, h, w, float(res_slice),
            _lib.stream_ptr())
    _lib.check(err, "slice_acq backward")
    return [grad_vol if need_vol_grad else _empty(vol), grad_tf if need_transforms_grad else _empty(vol)]


def _sa_adjoint_forward(transforms, psf, slices, slices_mask, vol_mask, vol_shape, res_slice, interp_psf, equalize):
    _lib.require_device(transforms, psf, slices, dtype=slices.dtype, name="slice_acq adjoint input")
    vm, sm = _mask(vol_mask), _mask(slices_mask)
    n, h, w = slices.shape[0], slices.shape[-2], slices.shape[-1]
    D, H, W = (int(s) for s in vol_shape)
    new = torch.zeros if interp_psf else torch.empty  # the interp kernel accumulates (atomics) into its outputs
    vol = new((1, 1, D, H, W), dtype=slices.dtype, device=slices.device)
    vol_weight = new((1, 1, D, H, W), dtype=slices.dtype, device=slices.device) if equalize else None
    scratch = () if interp_psf else (_lib.ptr(torch.empty(2 * n * h * w, dtype=slices.dtype, device=slices.device)),)
    with torch.cuda.device(slices.device):
        err = _sa_symbol("nesvor_slice_acq_adjoint_forward", slices, interp_psf)(
            _lib.ptr(transforms), _lib.ptr(psf), _lib.ptr(slices), _lib.ptr(sm), _lib.ptr(vm), _lib.ptr(vol), _lib.ptr(vol_weight),
            *scratch, *_sa_dims((D, H, W), psf), n, h, w, float(res_slice), int(bool(equalize)), _lib.stream_ptr())
    _lib.check(err, "slice_acq adjoint_forward")
    return [vol, vol_weight if equalize else _empty(slices)]


def _sa_adjoint_backward(transforms, grad_vol, vol_weight, vol_mask, psf, slices, slices_mask, vol, res_slice, interp_psf,
                         equalize, need_slices_grad, need_transforms_grad):
    _lib.require_device(transforms, grad_vol, psf, slices, dtype=slices.dtype, name="slice_acq adjoint_backward input")
    if equalize:
        _lib.require_device(vol_weight, vol, dtype=slices.dtype, name="slice_acq adjoint_backward vol/vol_weight")
    vm, sm = _mask(vol_mask), _mask(slices_mask)
    n, h, w = slices.shape[0], slices.shape[-2], slices.shape[-1]
    grad_slices = torch.zeros_like(slices) if need_slices_grad else None
    grad_tf = (torch.zeros_like if interp_psf else torch.empty_like)(transforms) if need_transforms_grad else None
    with torch.cuda.device(slices.device):
        err = _sa_symbol("nesvor_slice_acq_adjoint_backward", slices, interp_psf)(
            _lib.ptr(transforms), _lib.ptr(grad_vol), _lib.ptr(vol_weight if equalize else None), _lib.ptr(vm), _lib.ptr(psf),
            _lib.ptr(slices), _lib.ptr(sm), _lib.ptr(vol if equalize else None), _lib.ptr(grad_slices), _lib.ptr(grad_tf),
            *_sa_dims(grad_vol.shape, psf), n, h, w, float(res_slice), int(bool(equalize)), _lib.stream_ptr())
    _lib.check(err, "slice_acq adjoint_backward")
    return [grad_slices if need_slices_grad else _empty(slices), grad_tf if need_transforms_grad else _empty(slices)]


def _sa_forward_setup(ctx, inputs, output):
    transforms, vol, vol_mask, slices_mask, psf, _, res_slice, _, interp_psf = inputs
    # what the kernel actually read (the op requires contiguous inputs, so these are the caller's tensors)
    ctx.save_for_backward(transforms, vol, vol_mask, slices_mask, psf)
    ctx.res_slice, ctx.interp_psf = res_slice, interp_psf


def _sa_forward_backward(ctx, grads):
    transforms, vol, vol_mask, slices_mask, psf = ctx.saved_tensors
    gv, gt = torch.ops.nesvor.slice_acq_backward(
        transforms, vol, vol_mask, psf, grads[0].contiguous(), slices_mask, ctx.res_slice, ctx.interp_psf,
        ctx.needs_input_grad[1], ctx.needs_input_grad[0])
    return _opt(gt), _opt(gv), None, None, None, None, None, None, None


def _sa_adjoint_setup(ctx, inputs, output):
    transforms, psf, slices, slices_mask, vol_mask, _, res_slice, interp_psf, equalize = inputs
    ctx.save_for_backward(transforms, psf, slices, slices_mask, vol_mask, output[0], output[1])
    ctx.res_slice, ctx.interp_psf, ctx.equalize = res_slice, interp_psf, equalize


def _sa_adjoint_backward_formula(ctx, grads):
    transforms, psf, slices, slices_mask, vol_mask, vol, vol_weight = ctx.saved_tensors
    # the native op equalises grad_vol in place (as the reference does): work on a private contiguous copy
    grad_vol = grads[0].contiguous().clone() if ctx.equalize else grads[0].contiguous()
    gs, gt = torch.ops.nesvor.slice_acq_adjoint_backward(
        transforms, grad_vol, vol_weight, vol_mask, psf, slices, slices_mask, vol, ctx.res_slice, ctx.interp_psf, ctx.equalize,
        ctx.needs_input_grad[2], ctx.needs_input_grad[0])
    return _opt(gt), None, _opt(gs), None, None, None, None, None, None


def _fake_slices(transforms, vol, vol_mask, slices_mask, psf, slice_shape, res_slice, need_weight, interp_psf):
    out = vol.new_empty((transforms.shape[0], 1, int(slice_shape[0]), int(slice_shape[1])))
    return [out, torch.empty_like(out)] if need_weight else [out]


_op("slice_acq_backward(Tensor transforms, Tensor vol, Tensor vol_mask, Tensor psf, Tensor grad_slices, Tensor slices_mask, "
    "float res_slice, bool interp_psf, bool need_vol_grad, bool need_transforms_grad) -> Tensor[]", _sa_backward,
    fake=lambda tf, vol, vm, psf, gs, sm, r, i, nv, nt: [torch.empty_like(vol) if nv else vol.new_empty(0),
                                                         torch.empty_like(tf) if nt else vol.new_empty(0)])
_op("slice_acq_forward(Tensor transforms, Tensor vol, Tensor vol_mask, Tensor slices_mask, Tensor psf, int[] slice_shape, "
    "float res_slice, bool need_weight, bool interp_psf) -> Tensor[]", _sa_forward, fake=_fake_slices,
    backward=_sa_forward_backward, setup_context=_sa_forward_setup)
_op("slice_acq_adjoint_backward(Tensor transforms, Tensor(a!) grad_vol, Tensor vol_weight, Tensor vol_mask, Tensor psf, "
    "Tensor slices, Tensor slices_mask, Tensor vol, float res_slice, bool interp_psf, bool equalize, bool need_slices_grad, "
    "bool need_transforms_grad) -> Tensor[]", _sa_adjoint_backward,
    fake=lambda tf, gv, vw, vm, psf, s, sm, vol, r, i, e, ns, nt: [torch.empty_like(s) if ns else s.new_empty(0),
                                                                   torch.empty_like(tf) if nt else s.new_empty(0)])
_op("slice_acq_adjoint_forward(Tensor transforms, Tensor psf, Tensor slices, Tensor slices_mask, Tensor vol_mask, "
    "int[] vol_shape, float res_slice, bool interp_psf, bool equalize) -> Tensor[]", _sa_adjoint_forward,
    fake=lambda tf, psf, s, sm, vm, shape, r, i, e: [s.new_empty((1, 1) + tuple(int(v) for v in shape)),
                                                     s.new_empty((1, 1) + tuple(int(v) for v in shape)) if e else s.new_empty(0)],
    backward=_sa_adjoint_backward_formula, setup_context=_sa_adjoint_setup)


# =====================================================================================================================
# multi-resolution hash-grid encoding (tinycudann.Encoding "HashGrid"; the scalars are its encoding_config)
# =====================================================================================================================
_SPECS = {}


def grid_spec(n_levels, n_features_per_level, log2_hashmap_size, base_resolution, per_level_scale) -> HashGridSpec:
    key = (int(n_levels), int(n_features_per_level), int(log2_hashmap_size), int(base_resolution), float(per_level_scale))
    if key not in _SPECS:
        _SPECS[key] = HashGridSpec(*key)
    return _SPECS[key]


def _hg_encode(u, table, n_levels, n_features, log2_hashmap_size, base_resolution, per_level_scale, layout):
    return _enc.hashgrid_forward(grid_spec(n_levels, n_features, log2_hashmap_size, base_resolution, per_level_scale), u, table, layout)


def _hg_backward(u, table, dpe, n_levels, n_features, log2_hashmap_size, base_resolution, per_level_scale, layout, need_input_grad, method):
    spec = grid_spec(n_levels, n_features, log2_hashmap_size, base_resolution, per_level_scale)
    grad_table, grad_u = _enc.hashgrid_backward(spec, u, table, dpe, None, need_input_grad, layout, method)
    return grad_table, grad_u if need_input_grad else _empty(u)


def _hg_accumulate(u, table, dpe, grad_table, n_levels, n_features, log2_hashmap_size, base_resolution, per_level_scale, layout, need_input_grad):
    spec = grid_spec(n_levels, n_features, log2_hashmap_size, base_resolution, per_level_scale)
    _, grad_u = _enc.hashgrid_backward(spec, u, table, dpe, grad_table, need_input_grad, layout)
    return grad_u if need_input_grad else _empty(u)


def _hg_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])
    ctx.cfg = inputs[2:]


def _hg_backward_formula(ctx, dpe):
    u, table = ctx.saved_tensors
    gt, gu = torch.ops.nesvor.hashgrid_encode_backward(u, table, dpe.contiguous(), *ctx.cfg, ctx.needs_input_grad[0], "owner")
    return (_opt(gu) if ctx.needs_input_grad[0] else None, gt if ctx.needs_input_grad[1] else None) + (None,) * 6


def _hg_out_shape(u, n_levels, n_features, layout):
    e = n_levels * n_features
    return (u.shape[0], e) if layout == _lib.LAYOUT_ROW_MAJOR else (e, u.shape[0])


_op("hashgrid_encode_backward(Tensor u, Tensor table, Tensor dpe, int n_levels, int n_features_per_level, int log2_hashmap_size, "
    "int base_resolution, float per_level_scale, int layout, bool need_input_grad, str method) -> (Tensor, Tensor)", _hg_backward,
    fake=lambda u, t, d, L, F, T, b, s, lay, need, m: (torch.empty_like(t), torch.empty_like(u) if need else u.new_empty(0)))
_op("hashgrid_encode_backward_(Tensor u, Tensor table, Tensor dpe, Tensor(a!) grad_table, int n_levels, int n_features_per_level, "
    "int log2_hashmap_size, int base_resolution, float per_level_scale, int layout, bool need_input_grad) -> Tensor", _hg_accumulate,
    fake=lambda u, t, d, g, L, F, T, b, s, lay, need: torch.empty_like(u) if need else u.new_empty(0))
_op("hashgrid_encode(Tensor u, Tensor table, int n_levels, int n_features_per_level, int log2_hashmap_size, int base_resolution, "
    "float per_level_scale, int layout) -> Tensor", _hg_encode,
    fake=lambda u, t, L, F, T, b, s, lay: u.new_empty(_hg_out_shape(u, L, F, lay)),
    backward=_hg_backward_formula, setup_context=_hg_setup)


# =====================================================================================================================
# fused MLP (build_network's Linear/ReLU stacks and tinycudann.Network): feature-major input rows + per-pixel features
# =====================================================================================================================
def _mlp_forward(xa, xb, weights, biases, b_row0, k_b, samples_per_pixel, operands, save):
    y, saved = _mlp.forward_raw(list(weights), list(biases), xa, xb, b_row0, k_b, samples_per_pixel, save, _operands(operands))
    return y, saved


def _operands(code: int):
    """int of the schema -> the `bf16` argument of the raw calls: -1 = the fp32 default, else an explicit mode constant."""
    return False if code < 0 else (True if code == _mlp.BF16 else code)


def _mlp_backward(xa, xb, dy, weights, biases, saved, b_row0, k_b, samples_per_pixel, operands, need_dxa, need_dxb):
    dxb = torch.empty((k_b, xb.shape[1]), dtype=torch.float32, device=xb.device) if need_dxb else None
    dxa, partial = _mlp.backward_raw(list(weights), list(biases), xa, xb, dy, list(saved), b_row0, k_b, samples_per_pixel, dxb,
                                     need_dxa, _operands(operands))
    return (dxa if dxa is not None else _empty(xb), dxb if dxb is not None else _empty(xb), partial)


def _mlp_setup(ctx, inputs, output):
    xa, xb, weights, biases, b_row0, k_b, S, operands, save = inputs
    ctx.n_layers = len(weights)
    ctx.has_xa = xa is not None
    ctx.save_for_backward(*([xa] if xa is not None else []), xb, *weights, *biases, *output[1])
    ctx.cfg = (b_row0, k_b, S, operands)
    ctx.set_materialize_grads(False)


def _mlp_backward_formula(ctx, dy, d_saved):
    if dy is None:
        return (None,) * 9
    t = list(ctx.saved_tensors)
    xa = t.pop(0) if ctx.has_xa else None
    xb = t.pop(0)
    n = ctx.n_layers
    weights, biases, saved = t[:n], t[n : 2 * n], t[2 * n :]
    if len(saved) != n - 1:
        raise RuntimeError("fused_mlp was called with save=False but a gradient is requested: pass save=True")
    b_row0, k_b, S, operands = ctx.cfg
    need_xa, need_xb = ctx.has_xa and ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    dxa, dxb, partial = torch.ops.nesvor.fused_mlp_backward(xa, xb, dy.contiguous(), weights, biases, saved, b_row0, k_b, S, operands,
                                                            need_xa, need_xb)
    g_xb = None
    if need_xb:  # the kernel produced rows [b_row0, b_row0 + k_b) of the input gradient; the other rows get none
        g_xb = torch.zeros_like(xb)
        g_xb[b_row0 : b_row0 + k_b] = dxb
    g_xa = dxa.view(xa.shape[0], -1, dxa.shape[1]).sum(1) if need_xa else None
    flat = partial.sum(0)
    gw, gb, off = [], [], 0
    for w, b in zip(weights, biases):
        gw.append(flat[off : off + w.numel()].view_as(w))
        off += w.numel()
        gb.append(flat[off : off + b.numel()].view_as(b))
        off += b.numel()
    return g_xa, g_xb, gw, gb, None, None, None, None, None


def _mlp_fake(xa, xb, weights, biases, b_row0, k_b, S, operands, save):
    # the saved buffers' sizes follow from SHAPE fields alone (csrc/mlp.hip::compact_ok reads n_hidden, k_a, k_b, b_row0,
    # out_dim, samples_per_pixel, the operand mode and N through fill_args - no pointer of the descriptor), so this
    # pointer-less descriptor and forward_raw's real one always agree (tests/test_cabi.py::test_mlp_fake_sizes_match_forward)
    n = xb.shape[1]
    n_pad = (n + 15) // 16 * 16
    sdt = {_mlp.BF16: torch.bfloat16, _mlp.FP16: torch.float16}.get(operands, torch.float32)
    saved = []
    if save:
        d = _mlp.dims_desc(len(weights) - 1, weights[-1].shape[0], 0 if xa is None else xa.shape[1], k_b, b_row0, S, _operands(operands))
        saved = [xb.new_empty(m, dtype=sdt) for m in _mlp.saved_sizes(d, n, len(weights) - 1)]
    return xb.new_empty((weights[-1].shape[0], n)), saved


_op("fused_mlp_backward(Tensor? xa, Tensor xb, Tensor dy, Tensor[] weights, Tensor[] biases, Tensor[] saved, int b_row0, int k_b, "
    "int samples_per_pixel, int operands, bool need_dxa, bool need_dxb) -> (Tensor, Tensor, Tensor)", _mlp_backward)
_op("fused_mlp(Tensor? xa, Tensor xb, Tensor[] weights, Tensor[] biases, int b_row0, int k_b, int samples_per_pixel, int operands, "
    "bool save) -> (Tensor, Tensor[])", _mlp_forward, fake=_mlp_fake, backward=_mlp_backward_formula, setup_context=_mlp_setup)


# ---- the same networks at width <= 128 / up to seven hidden layers (csrc/mlp_wide.hip); `biases` empty = bias-free
def _wide_forward(xa, xb, weights, biases, b_row0, k_b, samples_per_pixel, save):
    return _mlp.wide_forward_raw(list(weights), list(biases), xa, xb, b_row0, k_b, samples_per_pixel, save)


def _wide_backward(xa, xb, dy, weights, biases, saved, b_row0, k_b, samples_per_pixel, need_dxa, need_dxb):
    dxb = torch.empty((k_b, xb.shape[1]), dtype=torch.float32, device=xb.device) if need_dxb else None
    dxa, partial = _mlp.wide_backward_raw(list(weights), list(biases), xa, xb, dy, list(saved), b_row0, k_b, samples_per_pixel, dxb, need_dxa)
    return (dxa if dxa is not None else _empty(xb), dxb if dxb is not None else _empty(xb), partial)


def _wide_setup(ctx, inputs, output):
    xa, xb, weights, biases, b_row0, k_b, S, save = inputs
    ctx.n_layers, ctx.n_bias = len(weights), len(biases)
    ctx.has_xa = xa is not None
    ctx.save_for_backward(*([xa] if xa is not None else []), xb, *weights, *biases, *output[1])
    ctx.cfg = (b_row0, k_b, S)
    ctx.set_materialize_grads(False)


def _wide_backward_formula(ctx, dy, d_saved):
    if dy is None:
        return (None,) * 8
    t = list(ctx.saved_tensors)
    xa = t.pop(0) if ctx.has_xa else None
    xb = t.pop(0)
    n, nb = ctx.n_layers, ctx.n_bias
    weights, biases, saved = t[:n], t[n : n + nb], t[n + nb :]
    if len(saved) != n - 1:
        raise RuntimeError("wide_mlp was called with save=False but a gradient is requested: pass save=True")
    b_row0, k_b, S = ctx.cfg
    need_xa, need_xb = ctx.has_xa and ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    dxa, dxb, partial = torch.ops.nesvor.wide_mlp_backward(xa, xb, dy.contiguous(), weights, biases, saved, b_row0, k_b, S, need_xa, need_xb)
    g_xb = None
    if need_xb:
        g_xb = torch.zeros_like(xb)
        g_xb[b_row0 : b_row0 + k_b] = dxb
    g_xa = dxa.view(xa.shape[0], -1, dxa.shape[1]).sum(1) if need_xa else None
    flat = partial.sum(0)
    gw, gb, off = [], [], 0
    for i, w in enumerate(weights):
        gw.append(flat[off : off + w.numel()].view_as(w))
        off += w.numel()
        if nb:
            gb.append(flat[off : off + biases[i].numel()].view_as(biases[i]))
            off += biases[i].numel()
    return g_xa, g_xb, gw, gb, None, None, None, None


def _wide_fake(xa, xb, weights, biases, b_row0, k_b, S, save):
    n = xb.shape[1]
    n_pad = (n + 15) // 16 * 16
    hb = 4 if weights[0].shape[0] <= 64 else 8
    saved = [xb.new_empty(n_pad * 16 * hb) for _ in range(len(weights) - 1)] if save else []
    return xb.new_empty((weights[-1].shape[0], n)), saved


_op("wide_mlp_backward(Tensor? xa, Tensor xb, Tensor dy, Tensor[] weights, Tensor[] biases, Tensor[] saved, int b_row0, int k_b, "
    "int samples_per_pixel, bool need_dxa, bool need_dxb) -> (Tensor, Tensor, Tensor)", _wide_backward)
_op("wide_mlp(Tensor? xa, Tensor xb, Tensor[] weights, Tensor[] biases, int b_row0, int k_b, int samples_per_pixel, bool save) -> "
    "(Tensor, Tensor[])", _wide_forward, fake=_wide_fake, backward=_wide_backward_formula, setup_context=_wide_setup)


# =====================================================================================================================
# PSF sampling + rigid transform + box normalisation (models.py:267-278, transform.py:259-280, models.py:143)
# =====================================================================================================================
def _psf_fwd(mat, slice_idx, xyz, psf_sigma, noise, bounding_box):
    return _sampler.forward_raw(mat, slice_idx, xyz, psf_sigma, noise, bounding_box)


def _psf_bwd(mat, slice_idx, xyz, psf_sigma, noise, bounding_box, dx, du):
    return _sampler.backward_raw(mat, slice_idx, xyz, psf_sigma, noise, bounding_box, dx, du)


def _psf_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)
    ctx.set_materialize_grads(False)


def _psf_backward_formula(ctx, dx, du):
    mat, slice_idx, xyz, psf_sigma, noise, bb = ctx.saved_tensors
    if not ctx.needs_input_grad[0] or (dx is None and du is None):
        return (None,) * 6
    dpix = torch.ops.nesvor.psf_transform_backward(mat, slice_idx, xyz, psf_sigma, noise, bb,
                                                   None if dx is None else dx.contiguous(), None if du is None else du.contiguous())
    return torch.zeros_like(mat).index_add_(0, slice_idx, dpix), None, None, None, None, None


_op("psf_transform_backward(Tensor mat, Tensor slice_idx, Tensor xyz, Tensor psf_sigma, Tensor noise, Tensor bounding_box, "
    "Tensor? dx, Tensor? du) -> Tensor", _psf_bwd,
    fake=lambda mat, idx, xyz, sig, noise, bb, dx, du: noise.new_empty((noise.shape[0], 3, 4)))
_op("psf_transform(Tensor mat, Tensor slice_idx, Tensor xyz, Tensor psf_sigma, Tensor noise, Tensor bounding_box) -> (Tensor, Tensor)",
    _psf_fwd, fake=lambda mat, idx, xyz, sig, noise, bb: (torch.empty_like(noise), noise.new_empty((noise.shape[0] * noise.shape[1], 3))),
    backward=_psf_backward_formula, setup_context=_psf_setup)


# =====================================================================================================================
# imaging model + losses (models.py:286-325, 366-384): values, and all input gradients in one launch
# =====================================================================================================================
def _loss_forward(z0, log_var, log_bias, x, v, slice_idx, c, log_var_slice, reg_type, delta):
    _lib.require_device(z0, log_var, log_bias, x, v, c, log_var_slice, dtype=torch.float32, name="imaging loss input")
    _lib.require_device(slice_idx, dtype=torch.int64, name="slice_idx")
    B, S = x.shape[0], x.shape[1]
    lb_mean = log_bias.mean().reshape(1) if log_bias is not None else torch.zeros(1, dtype=torch.float32, device=x.device)
    if log_bias is not None:
        from . import ddp

        if ddp.active():  # biasReg = (mean log_bias)^2 is not a mean of per-sample terms (models.py:322-323): take the GLOBAL
            # mean, so that value and averaged gradients are those of the undivided batch (as nesvor_amd.direct does)
            torch.distributed.all_reduce(lb_mean)
            lb_mean /= torch.distributed.get_world_size()
    loss_pix = torch.empty((B, 3), dtype=torch.float32, device=x.device)
    a = _loss._fill(z0, log_var, log_bias, x, v, slice_idx, c, log_var_slice, lb_mean if log_bias is not None else None, reg_type, delta)
    a.loss_pix = loss_pix.data_ptr()
    with torch.cuda.device(x.device), _lib.kernel_timer.span("imaging_loss_fwd"):
        err = _lib.load().nesvor_imaging_loss(ctypes.byref(a), _lib.stream_ptr())
    _lib.check(err, "imaging loss forward")
    sums = loss_pix.sum(0)
    mean_term = sums[2] / (B * S)
    ireg = delta * (mean_term - 1) if reg_type == 0 else mean_term
    return sums[0] / B, sums[1] / B, ireg, lb_mean[0] ** 2, lb_mean


def _loss_backward(z0, log_var, log_bias, x, v, slice_idx, c, log_var_slice, lb_mean, gw, reg_type, delta, need_dx):
    dev, B = x.device, x.shape[0]
    dz0 = torch.empty_like(z0)
    dlv = torch.empty_like(log_var) if log_var is not None else None
    dlb = torch.empty_like(log_bias) if log_bias is not None else None
    dx = torch.empty_like(x) if need_dx else None
    dc_pix = torch.empty(B, dtype=torch.float32, device=dev) if c is not None else None
    dlvs_pix = torch.empty(B, dtype=torch.float32, device=dev) if log_var_slice is not None else None
    a = _loss._fill(z0, log_var, log_bias, x, v, slice_idx, c, log_var_slice, lb_mean if log_bias is not None else None, reg_type, delta)
    a.gw = gw.data_ptr()
    for name, t in (("dz0", dz0), ("dlog_var", dlv), ("dlog_bias", dlb), ("dx", dx), ("dc_pix", dc_pix), ("dlvs_pix", dlvs_pix)):
        setattr(a, name, None if t is None else t.data_ptr())
    with torch.cuda.device(dev), _lib.kernel_timer.span("imaging_loss_bwd"):
        err = _lib.load().nesvor_imaging_loss(ctypes.byref(a), _lib.stream_ptr())
    _lib.check(err, "imaging loss backward")
    dc = torch.zeros_like(c).index_add_(0, slice_idx, dc_pix) if c is not None else _empty(x)
    dlvs = torch.zeros_like(log_var_slice).index_add_(0, slice_idx, dlvs_pix) if log_var_slice is not None else _empty(x)
    # (a fresh empty tensor per absent output: custom-op outputs must not alias each other)
    return (dz0, dlv if dlv is not None else _empty(x), dlb if dlb is not None else _empty(x),
            dx if dx is not None else _empty(x), dc, dlvs)


def _loss_setup(ctx, inputs, output):
    z0, log_var, log_bias, x, v, slice_idx, c, log_var_slice, reg_type, delta = inputs
    ctx.present = [t is not None for t in (log_var, log_bias, c, log_var_slice)]
    ctx.save_for_backward(z0, x, v, slice_idx, output[4], *[t for t in (log_var, log_bias, c, log_var_slice) if t is not None])
    ctx.cfg = (reg_type, delta)


def _loss_backward_formula(ctx, g_mse, g_logvar, g_ireg, g_breg, g_unused):
    z0, x, v, slice_idx, lb_mean, *rest = ctx.saved_tensors
    opt = [rest.pop(0) if p else None for p in ctx.present]
    log_var, log_bias, c, log_var_slice = opt
    gw = torch.stack([g_mse, g_logvar, g_ireg, g_breg]).to(torch.float32).contiguous()
    dz0, dlv, dlb, dx, dc, dlvs = torch.ops.nesvor.imaging_loss_backward(
        z0, log_var, log_bias, x, v, slice_idx, c, log_var_slice, lb_mean, gw, *ctx.cfg, ctx.needs_input_grad[3])
    return dz0, _opt(dlv), _opt(dlb), _opt(dx), None, None, _opt(dc), _opt(dlvs), None, None


_op("imaging_loss_backward(Tensor z0, Tensor? log_var, Tensor? log_bias, Tensor x, Tensor v, Tensor slice_idx, Tensor? c, "
    "Tensor? log_var_slice, Tensor log_bias_mean, Tensor gw, int reg_type, float delta, bool need_dx) -> "
    "(Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)", _loss_backward)
_op("imaging_loss(Tensor z0, Tensor? log_var, Tensor? log_bias, Tensor x, Tensor v, Tensor slice_idx, Tensor? c, "
    "Tensor? log_var_slice, int reg_type, float delta) -> (Tensor, Tensor, Tensor, Tensor, Tensor)", _loss_forward,
    fake=lambda z0, lv, lb, x, v, idx, c, lvs, r, d: tuple(x.new_empty(()) for _ in range(4)) + (x.new_empty(1),),
    backward=_loss_backward_formula, setup_context=_loss_setup)


# =====================================================================================================================
# AdamW + zero_grad over one flat buffer (train.py:144-152, 195-197)
# =====================================================================================================================
def _adamw_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale, zero_grad):
    _lib.require_device(param, grad, exp_avg, exp_avg_sq, dtype=torch.float32, name="adamw buffers")
    with torch.cuda.device(param.device):
        err = _lib.load().nesvor_adamw_step(
            _lib.ptr(param), _lib.ptr(grad), _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq), param.numel(), lr, beta1, beta2, eps,
            weight_decay, 1 - beta1**step, 1 - beta2**step, grad_scale, int(bool(zero_grad)), _lib.stream_ptr())
    _lib.check(err, "adamw step")


_op("adamw_step_(Tensor(a!) param, Tensor(b!) grad, Tensor(c!) exp_avg, Tensor(d!) exp_avg_sq, float lr, float beta1, float beta2, "
    "float eps, float weight_decay, int step, float grad_scale, bool zero_grad) -> ()", _adamw_step,
    fake=lambda *a: None)


def op_names() -> List[str]:
    return sorted(SCHEMAS)
