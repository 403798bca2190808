"""Fused small-MLP op (gfx950 fp32 matrix cores): autograd Function over nesvor_mlp_forward/backward.

The network is an ``nn.Sequential`` of Linear/ReLU exactly as ``build_network`` creates it in
single-precision mode (same parameters, same state_dict); this op only changes how it is
evaluated.  Input = [pixel features xa (P,k_a) broadcast over each pixel's S samples |
rows [b_row0, b_row0+k_b) of a feature-major matrix xb (rows,N)];  output (out_dim, N) feature-major.
"""
import ctypes
import os

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib

N_PARTIAL = 1024  # workgroups (= partial sums) of the dW kernel: 4 per CU so loads overlap MFMAs
N_PARTIAL_FUSED = 256  # fused backward: one persistent 8-wave workgroup per CU (wave-specialised kernel)
FUSED_BACKWARD = True  # False: separate dX and dW kernels (kept as cross-check and for depth 3)

# How an fp32 matrix product is evaluated (nesvor_mlp_t.bf16_operands):
#   MFMA_FP32 (0): v_mfma_f32_16x16x4_f32 - an fp32 FMA chain;
#   BF16 (1): operands rounded to bf16 - mixed precision, opt-in (args.mlp_bf16) / the half-precision model structure;
#   SPLIT (2): every fp32 operand written as two fp16 numbers of a scaled copy (x s = hi + lo, s a power of two per operand
#     tensor and launch from the bounds in ``prep``), three fp16 MFMAs per product, fp32 accumulation: error against fp64 at
#     or below the fp32 FMA chain's (profiles/r05_f16_split_probe.log: 2.1e-7 vs 3.6e-7 of the largest result), because the
#     fp32 matrix pipe of gfx950 is 16x slower than the 16-bit one.  (Rounds 2-4: three bf16 terms per operand, six MFMAs.)
# SPLIT is what "fp32" means by default; NESVOR_MLP_FP32=mfma (or FP32_OPERANDS = MFMA_FP32) selects the plain path.
#   FP16 (3): operands rounded to fp16 (round 6) - the arithmetic of the reference's DEFAULT mode (fp16 CutlassMLP,
#     nesvor/nesvor/models.py:28-41), same kernels as BF16 with fp16 MFMAs; its narrow exponent range is what the reference's
#     GradScaler exists for (train.py:161-164): opt-in with ``args.fp16_loss_scaling`` (nesvor_amd.fused.LossScaler).
#   FP16S (4): "scaled fp16" (round 6) - the SPLIT mode's machinery (power-of-two scales from ``prep``, the same operand images,
#     the bits-only save, the backward's per-sample scale) with the leading term of every split alone: operands rounded to fp16
#     AFTER scaling (11 bits, no overflow and no loss scaler), ONE MFMA per product.  What fp16 arithmetic buys on the kernels the
#     fp32 path runs on; opt-in for the fp32 model structure (``args.mlp_fp16``).  Kernels without that form evaluate the full
#     split or fp32 MFMAs - more accurate, always valid.
MFMA_FP32, BF16, SPLIT, FP16, FP16S = 0, 1, 2, 3, 4
SCALED_MODES = (SPLIT, FP16S)  # modes that need ``prep`` (operand bounds + weight norms)
FP32_OPERANDS = MFMA_FP32 if os.environ.get("NESVOR_MLP_FP32", "split").lower() == "mfma" else SPLIT


def operand_mode(bf16) -> int:
    """bf16: False -> the fp32 default (FP32_OPERANDS); True -> bf16-rounded operands; or an explicit mode constant
    passed as an int (MFMA_FP32 is spelled 0, which `False` must not be confused with: pass it via `operands=`)."""
    if bf16 is True or (bf16 is not False and int(bf16) == BF16):
        return BF16
    if bf16 is False:
        return FP32_OPERANDS
    return int(bf16)


def linear_layers(seq: nn.Sequential):
    layers = [m for m in seq if isinstance(m, nn.Linear)]
    others = [m for m in seq if not isinstance(m, (nn.Linear, nn.ReLU))]
    if others:
        raise ValueError(f"fused MLP supports Linear/ReLU stacks only, found {others}")
    return layers


def supported(seq) -> bool:
    from .tinycudann import Network

    if isinstance(seq, Network):  # the half-precision model structure: bias-free, output rows padded to 16
        sh = seq.shapes
        return (seq.activation == "ReLU" and seq.output_activation == "None" and 2 <= len(sh) <= 4 and sh[0][1] <= 64
                and all(o == 64 for o, _ in sh[:-1]) and seq.n_output_dims <= 16)
    if not isinstance(seq, nn.Sequential):
        return False
    try:
        layers = linear_layers(seq)
    except ValueError:
        return False
    if not 2 <= len(layers) <= 4:
        return False
    if any(not 1 <= l.out_features <= 64 for l in layers[:-1]) or any(a.out_features != b.in_features for a, b in zip(layers, layers[1:])):
        return False
    return layers[-1].out_features <= 16 and layers[0].in_features <= 64 and all(l.bias is not None for l in layers)


def native_width(seq) -> bool:
    """Hidden width exactly 64: the kernels run on the module's own parameters.  Narrower Linear/ReLU stacks
    (``--width`` < 64) are covered too, by zero padding - see ``kernel_params``."""
    from .tinycudann import Network

    if isinstance(seq, Network):
        return True
    return all(l.out_features == 64 for l in linear_layers(seq)[:-1])


def kernel_params(layers):
    """Per-layer (weights, biases) in the shape the kernels are built for (hidden width 64).  A narrower network is the
    same function as its zero-padded 64-wide twin - hidden units width..63 get zero weights and zero bias, stay at
    ReLU(0) = 0 and feed nothing forward - so it is evaluated EXACTLY by the same kernels (at the 64-wide cost).  The
    padding is a differentiable ``F.pad``: autograd hands the parameter gradients back as slices."""
    import torch.nn.functional as F

    ws, bs = [l.weight for l in layers], [l.bias for l in layers]
    if all(l.out_features == 64 for l in layers[:-1]):
        return ws, bs
    out_w, out_b = [], []
    for i, (w, b) in enumerate(zip(ws, bs)):
        last = i == len(ws) - 1
        pad_in = 0 if i == 0 else 64 - w.shape[1]
        pad_out = 0 if last else 64 - w.shape[0]
        out_w.append(F.pad(w, (0, pad_in, 0, pad_out)))
        out_b.append(b if last else F.pad(b, (0, pad_out)))
    return out_w, out_b


def n_hidden_layers(net) -> int:
    from .tinycudann import Network

    return (len(net.shapes) if isinstance(net, Network) else len(linear_layers(net))) - 1


class NetParams:
    """One network in the kernels' terms (per-layer weight / bias tensors) for both structures ``build_network``
    creates (nesvor/nesvor/models.py:28-69):

    * single precision: ``nn.Sequential`` of Linear/ReLU - the layers' own Parameters;
    * half precision: bias-free ``tinycudann.Network`` with one flat parameter vector - per-layer views of it (of the
      padded last layer only the first ``n_output_dims`` rows are evaluated; the padding rows never reach an output and
      keep a zero gradient, as in tinycudann) and one shared all-zero bias vector.

    ``store_grads`` reduces the backward kernel's per-workgroup partial sums (columns W0,b0,W1,b1,...) into the
    parameters' ``.grad`` - which the fused trainer has re-homed into its flat gradient buffer.
    """

    def __init__(self, net):
        from .tinycudann import Network

        self.net = net
        self.flat_params = isinstance(net, Network)
        if self.flat_params:
            p = net.params.data
            self.weights, self.w_off, off = [], [], 0
            for li, (o, i) in enumerate(net.shapes):
                rows = net.n_output_dims if li == len(net.shapes) - 1 else o
                self.weights.append(p[off : off + rows * i].view(rows, i))
                self.w_off.append(off)
                off += o * i
            zero = torch.zeros(64, dtype=torch.float32, device=p.device)
            self.biases = [zero[: w.shape[0]] for w in self.weights]
            self.segment = None
        else:
            layers = linear_layers(net)
            if not native_width(net):
                # narrower than 64: zero-padded copies (inference; the autograd-free training step takes 64-wide networks only)
                with torch.no_grad():
                    self.weights, self.biases = (list(t) for t in kernel_params(layers))
                self.segment = None
                return
            self.weights = [l.weight for l in layers]
            self.biases = [l.bias for l in layers]
            # one contiguous gradient segment in the partial sums' column order? (the flat layout of fused.FlatParams)
            ps = [t for l in layers for t in (l.weight, l.bias)]
            g0 = ps[0].grad
            contiguous = g0 is not None and all(
                q.grad is not None and q.grad.data_ptr() == p_.grad.data_ptr() + 4 * p_.numel() for p_, q in zip(ps, ps[1:]))
            n = sum(t.numel() for t in ps)
            self.segment = torch.as_strided(g0, (n,), (1,), g0.storage_offset()) if contiguous else None

    def n_hidden(self):
        return len(self.weights) - 1

    def _sum_rows(self, src_ptr, dst, rows, cols, ld):
        with torch.cuda.device(dst.device):
            err = _lib.load().nesvor_sum_rows(src_ptr, _lib.ptr(dst), rows, cols, ld, _lib.stream_ptr())
        _lib.check(err, "sum_rows")

    def store_grads(self, partial):
        rows, ld = partial.shape
        if self.flat_params:
            g = self.net.params.grad
            col = 0
            for w, off in zip(self.weights, self.w_off):
                self._sum_rows(partial.data_ptr() + 4 * col, g[off : off + w.numel()], rows, w.numel(), ld)
                col += w.numel() + w.shape[0]
            return
        if self.segment is not None and self.segment.numel() == ld:
            self._sum_rows(partial.data_ptr(), self.segment, rows, ld, ld)
            return
        flat = partial.sum(0)
        off = 0
        for w, b in zip(self.weights, self.biases):
            for p in (w, b):
                p.grad.copy_(flat[off : off + p.numel()].view_as(p))
                off += p.numel()


ABSMAX_FLOATS = 16 * 64  # NESVOR_ABSMAX_FLOATS: an operand bound is the maximum over 16 slots, one 256-byte line apart
PREP_FLOATS = 3 * ABSMAX_FLOATS + 4 * 4  # NESVOR_MLP_PREP_FLOATS
PREP_INPUT, PREP_DY, PREP_WEIGHTS = 1, 2, 4  # `what` of nesvor_mlp_prepare


def _desc(weights, biases, k_a, k_b, b_row0, S, bf16=False, prep=None):
    d = _lib.MlpT()
    d.bf16_operands = operand_mode(bf16)
    d.width, d.n_hidden, d.out_dim = 64, len(weights) - 1, weights[-1].shape[0]
    d.k_a, d.k_b, d.b_row0, d.samples_per_pixel = k_a, k_b, b_row0, S
    for i, (w, b) in enumerate(zip(weights, biases)):
        d.weight[i], d.bias[i] = w.data_ptr(), b.data_ptr()
    if prep is not None:
        d.prep = prep.data_ptr()
    return d


def prepare(d, xa, xb, dy, N, what, prep=None):
    """The split mode's operand bounds and weight norms (``nesvor_mlp_t.prep``) by ``nesvor_mlp_prepare``: a reduction over the
    operands named by ``what``.  The training step does not come here - its producing kernels publish the bounds
    (nesvor_amd/direct.py) - this is for standalone calls of the op."""
    if prep is None:
        prep = torch.zeros(PREP_FLOATS, dtype=torch.float32, device=xb.device)
    with torch.cuda.device(xb.device):
        err = _lib.load().nesvor_mlp_prepare(ctypes.byref(d), _lib.ptr(xa), _lib.ptr(xb), _lib.ptr(dy), N, _lib.ptr(prep), what,
                                             _lib.stream_ptr())
    _lib.check(err, "mlp prepare")
    d.prep = prep.data_ptr()
    return prep


def build_weight_images(d, prep):
    """The network's split operand images for its CURRENT weights (``nesvor_mlp_t.weight_images``): one launch that also rewrites the
    weight norms of ``prep`` (same values), so that a launch copies its LDS images instead of building them.  The training step does
    this once per iteration inside ``nesvor_step_run``; standalone calls normally leave the field NULL (in-kernel builds) - this helper
    exists for tests and tools.  Returns the buffer (keep it alive while ``d`` is in use)."""
    lib = _lib.load()
    n = int(lib.nesvor_mlp_weight_images_bytes(ctypes.byref(d)))
    if n <= 0:
        raise RuntimeError("no prebuilt operand images for this shape")
    buf = torch.empty(n, dtype=torch.uint8, device=prep.device)
    nets, preps, imgs = (ctypes.c_void_p * 1)(ctypes.addressof(d)), (ctypes.c_void_p * 1)(prep.data_ptr()), (ctypes.c_void_p * 1)(buf.data_ptr())
    with torch.cuda.device(prep.device):
        _lib.check(lib.nesvor_mlp_prepare_weights_images(nets, preps, imgs, 1, None, 0, None, _lib.stream_ptr()), "mlp weight images")
    d.weight_images = buf.data_ptr()
    return buf


def compact_save(d, N: int) -> bool:
    """Whether a training forward of descriptor ``d`` over N samples saves compactly (``nesvor_mlp_t.compact_save``): one
    sign bit per hidden unit and sample (16 N bytes in ``saved[0]``) and nothing else; the backward recomputes the hidden
    layers.  Host-side logic of the library (no device work)."""
    return FUSED_BACKWARD and bool(_lib.load().nesvor_mlp_compact_save_ok(ctypes.byref(d), N))


def dims_desc(n_hidden: int, out_dim: int, k_a: int, k_b: int, b_row0: int, S: int, bf16=False):
    """A descriptor without parameter pointers (shape queries: ``saved_sizes``)."""
    d = _lib.MlpT()
    d.bf16_operands = operand_mode(bf16)
    d.width, d.n_hidden, d.out_dim = 64, n_hidden, out_dim
    d.k_a, d.k_b, d.b_row0, d.samples_per_pixel = k_a, k_b, b_row0, S
    return d


def saved_sizes(d, N: int, n_hidden: int):
    """Element counts of the ``saved`` buffers of one training forward (fp32 elements; bf16 elements in the bf16 mode)."""
    n_pad = (N + 15) // 16 * 16
    sizes = [n_pad * 64] * n_hidden
    if n_hidden and compact_save(d, N):
        # one uint32 per (16-sample group, lane): the gate bits of every hidden layer; the backward recomputes the values
        # (the other slots are placeholders: the kernels do not touch them)
        sizes = [n_pad * 4] + [16] * (n_hidden - 1)
    return sizes


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * 4)()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def forward_raw(weights, biases, xa, xb, b_row0, k_b, S, need_saved, bf16=False, prep=None, y_absmax=None, weight_images=False):
    """One forward launch, no autograd: -> (y (out_dim, N), saved hidden activations [] when not need_saved).
    bf16: False = fp32 (FP32_OPERANDS picks split-fp16 or fp32-MFMA evaluation of the products), True = matrix operands
    rounded to bf16 with fp32 accumulation (opt-in mixed precision); an int selects a mode constant directly.
    prep: the split mode's bounds (``nesvor_mlp_t.prep``; None: computed here by a pass over the inputs and the weights);
    y_absmax: zero-filled tensor of ABSMAX_FLOATS floats whose maximum is raised to max |y|."""
    _lib.require_device(xb, *weights, *biases, dtype=torch.float32, name="fused MLP input/params")
    N = xb.shape[1]
    k_a = 0 if xa is None else xa.shape[1]
    if xa is not None:
        _lib.require_device(xa, dtype=torch.float32, name="fused MLP pixel features")
        if xa.shape[0] * S != N:
            raise RuntimeError("pixel features: P * samples_per_pixel must equal N")
    if weights[0].shape[1] != k_a + k_b:
        raise RuntimeError("first layer width does not match k_a + k_b")
    d = _desc(weights, biases, k_a, k_b, b_row0, S, bf16, prep)
    if d.bf16_operands in SCALED_MODES and prep is None:
        prep = prepare(d, xa, xb, None, N, PREP_INPUT | PREP_WEIGHTS)
    images = build_weight_images(d, prep) if (weight_images and d.bf16_operands in SCALED_MODES) else None  # noqa: F841 (kept alive until the launch is enqueued; the stream orders the free)
    if y_absmax is not None:
        d.y_absmax = y_absmax.data_ptr()
    # saved hidden activations, MFMA fragment layout; bf16 mode stores them as bf16 (they are only used as bf16 operands);
    # compact save (split-operand mode, whole tiles): saved[0] holds sign bits only
    sdt = {BF16: torch.bfloat16, FP16: torch.float16}.get(operand_mode(bf16), torch.float32)
    saved = []
    if need_saved:
        sizes = saved_sizes(d, N, len(weights) - 1)
        d.compact_save = int(sizes[0] != (N + 15) // 16 * 16 * 64)
        saved = [torch.empty(n, dtype=sdt, device=xb.device) for n in sizes]
    y = torch.empty((d.out_dim, N), dtype=torch.float32, device=xb.device)
    with torch.cuda.device(xb.device), _lib.kernel_timer.span("mlp_fwd"):
        err = _lib.load().nesvor_mlp_forward(
            ctypes.byref(d), _lib.ptr(xa), _lib.ptr(xb), _lib.ptr(y), _ptr_array(saved) if need_saved else None,
            N, _lib.stream_ptr())
    _lib.check(err, "mlp forward")
    return y, saved


def backward_raw(weights, biases, xa, xb, dy, saved, b_row0, k_b, S, dxb, need_dxa, bf16=False, dxb_absmax=None, prep=None,
                 weight_images=False):
    """One backward pass, no autograd.  dxb: (k_b, N) contiguous tensor the input gradient is written to (or
    None); -> (dxa (N, k_a) per-sample or (N/16, k_a) per 16-sample group | None - sum it over each pixel's rows -,
    partial (n_partial, n_params) to be summed over dim 0).  dxb_absmax: zero-filled 1-element tensor raised to max |dxb|
    (handed to the hash-grid backward as its ``dy_bound``)."""
    n_layers = len(weights)
    N = xb.shape[1]
    k_a = 0 if xa is None else xa.shape[1]
    d = _desc(weights, biases, k_a, k_b, b_row0, S, bf16, prep)
    if d.bf16_operands in SCALED_MODES and prep is None:  # (the same input bounds and weight norms as the forward's: same data)
        prep = prepare(d, xa, xb, dy, N, PREP_INPUT | PREP_DY | PREP_WEIGHTS)
    images = build_weight_images(d, prep) if (weight_images and d.bf16_operands in SCALED_MODES) else None  # noqa: F841
    d.compact_save = int(saved[0].numel() == (N + 15) // 16 * 16 * 4)  # (the forward that wrote `saved` decided)
    dev = xb.device
    # the wave-specialised fused kernel (dX + dW + db in one launch) needs no dpre scratch (signalled by NULL entries); shapes it
    # does not take (ragged N, S or k_a not multiples of 16, three hidden layers ...) run as a dX launch + a dW launch
    fused = FUSED_BACKWARD and bool(_lib.load().nesvor_mlp_backward_fused_ok(ctypes.byref(d), N))
    dpre = [] if fused else [torch.empty_like(s) for s in saved]
    # pixel-feature gradient: one row per 16-sample group (summed in the kernel) when a group lies inside a pixel
    group_sums = fused and N % 16 == 0 and S % 16 == 0 and k_a % 16 == 0
    d.dxa_group_sums = 1 if group_sums else 0
    rows = N // 16 if group_sums else N
    dxa = torch.empty((rows, k_a), dtype=torch.float32, device=dev) if (xa is not None and need_dxa) else None
    total = sum(w.numel() + b.numel() for w, b in zip(weights, biases))
    n_partial = N_PARTIAL_FUSED if fused else N_PARTIAL
    partial = torch.empty((n_partial, total), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), _lib.kernel_timer.span("mlp_bwd"):
        err = _lib.load().nesvor_mlp_backward_bounded(
            ctypes.byref(d), _lib.ptr(xa), _lib.ptr(xb), _lib.ptr(dy), _ptr_array(saved), _ptr_array(dpre),
            _lib.ptr(dxa), _lib.ptr(dxb), _lib.ptr(partial), n_partial, N, _lib.ptr(dxb_absmax), _lib.stream_ptr())
    _lib.check(err, "mlp backward")
    return dxa, partial


# ------------------------------------------------------------------------------------------------ wider / deeper networks
WIDE_MAX_WIDTH, WIDE_MAX_HIDDEN = 128, 7  # nesvor_mlp_wide_t (csrc/mlp_wide.hip)
N_PARTIAL_WIDE = 1024


def wide_supported(seq) -> bool:
    """Linear/ReLU stacks (with or without biases) outside the 64-wide fused kernels but inside the hand-written wide kernels
    (round 6): one hidden width <= 128, 1-7 hidden layers, <= 64 inputs, <= 16 outputs; and the bias-free ``tinycudann.Network``
    of such shapes."""
    from .tinycudann import Network

    if isinstance(seq, Network):
        sh = seq.shapes
        return (seq.activation == "ReLU" and seq.output_activation == "None" and 2 <= len(sh) <= WIDE_MAX_HIDDEN + 1 and sh[0][1] <= 64
                and len({o for o, _ in sh[:-1]}) == 1 and sh[0][0] <= WIDE_MAX_WIDTH and seq.n_output_dims <= 16)
    if not isinstance(seq, nn.Sequential):
        return False
    try:
        layers = linear_layers(seq)
    except ValueError:
        return False
    if not 2 <= len(layers) <= WIDE_MAX_HIDDEN + 1:
        return False
    widths = {l.out_features for l in layers[:-1]}
    if len(widths) != 1 or not 1 <= next(iter(widths)) <= WIDE_MAX_WIDTH or any(a.out_features != b.in_features for a, b in zip(layers, layers[1:])):
        return False
    has_bias = {l.bias is not None for l in layers}
    return layers[-1].out_features <= 16 and layers[0].in_features <= 64 and len(has_bias) == 1


def _wide_desc(weights, biases, k_a, k_b, b_row0, S):
    d = _lib.MlpWideT()
    d.width, d.n_hidden, d.out_dim = weights[0].shape[0], len(weights) - 1, weights[-1].shape[0]
    d.k_a, d.k_b, d.b_row0, d.samples_per_pixel = k_a, k_b, b_row0, S
    for i, w in enumerate(weights):
        d.weight[i] = w.data_ptr()
        d.bias[i] = biases[i].data_ptr() if biases else None
    return d


def _ptr_array8(tensors):
    arr = (ctypes.c_void_p * 8)()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def wide_forward_raw(weights, biases, xa, xb, b_row0, k_b, S, need_saved):
    """One forward launch of the wide kernels, no autograd: -> (y (out_dim, N), saved hidden activations).  ``biases``: one per
    layer, or an empty list (bias-free network)."""
    _lib.require_device(xb, *weights, *biases, dtype=torch.float32, name="wide MLP input/params")
    N = xb.shape[1]
    k_a = 0 if xa is None else xa.shape[1]
    if xa is not None:
        _lib.require_device(xa, dtype=torch.float32, name="wide MLP pixel features")
        if xa.shape[0] * S != N:
            raise RuntimeError("pixel features: P * samples_per_pixel must equal N")
    if weights[0].shape[1] != k_a + k_b:
        raise RuntimeError("first layer width does not match k_a + k_b")
    d = _wide_desc(weights, biases, k_a, k_b, b_row0, S)
    lib = _lib.load()
    saved = []
    if need_saved:
        n_el = lib.nesvor_mlp_wide_saved_floats(ctypes.byref(d), N)
        saved = [torch.empty(n_el, dtype=torch.float32, device=xb.device) for _ in range(d.n_hidden)]
    y = torch.empty((d.out_dim, N), dtype=torch.float32, device=xb.device)
    with torch.cuda.device(xb.device), _lib.kernel_timer.span("mlp_fwd"):
        err = lib.nesvor_mlp_wide_forward(ctypes.byref(d), _lib.ptr(xa), _lib.ptr(xb), _lib.ptr(y), _ptr_array8(saved) if need_saved else None,
                                          N, _lib.stream_ptr())
    _lib.check(err, "wide mlp forward")
    return y, saved


def wide_backward_raw(weights, biases, xa, xb, dy, saved, b_row0, k_b, S, dxb, need_dxa):
    """-> (dxa (N, k_a) per sample | None, partial (N_PARTIAL_WIDE, n_params): sum over dim 0 = W0, b0, W1, b1, ... gradients)."""
    N = xb.shape[1]
    k_a = 0 if xa is None else xa.shape[1]
    d = _wide_desc(weights, biases, k_a, k_b, b_row0, S)
    lib = _lib.load()
    dpre = [torch.empty_like(s) for s in saved]
    dxa = torch.empty((N, k_a), dtype=torch.float32, device=xb.device) if (xa is not None and need_dxa) else None
    total = lib.nesvor_mlp_wide_param_count(ctypes.byref(d))
    partial = torch.empty((N_PARTIAL_WIDE, total), dtype=torch.float32, device=xb.device)
    with torch.cuda.device(xb.device), _lib.kernel_timer.span("mlp_bwd"):
        err = lib.nesvor_mlp_wide_backward(ctypes.byref(d), _lib.ptr(xa), _lib.ptr(xb), _lib.ptr(dy), _ptr_array8(saved), _ptr_array8(dpre),
                                           _lib.ptr(dxa), _lib.ptr(dxb), _lib.ptr(partial), N_PARTIAL_WIDE, N, _lib.stream_ptr())
    _lib.check(err, "wide mlp backward")
    return dxa, partial


def wide_mlp(seq: nn.Sequential, xa, xb, b_row0: int, k_b: int, samples_per_pixel: int):
    """``seq`` (a shape ``wide_supported`` accepts) on [xa broadcast | xb rows] -> (out_dim, N) feature-major, differentiable in
    xa, xb and the parameters: the dispatcher op ``torch.ops.nesvor.wide_mlp``."""
    layers = linear_layers(seq)
    need = torch.is_grad_enabled() and (xb.requires_grad or (xa is not None and xa.requires_grad) or any(
        p.requires_grad for l in layers for p in l.parameters()))
    weights = [l.weight for l in layers]
    biases = [l.bias for l in layers] if layers[0].bias is not None else []
    y, _ = torch.ops.nesvor.wide_mlp(xa, xb.contiguous(), weights, biases, b_row0, k_b, samples_per_pixel, need)
    return y


def fused_mlp(seq: nn.Sequential, xa, xb, b_row0: int, k_b: int, samples_per_pixel: int):
    """Evaluate `seq` on [xa broadcast | xb rows] -> (out_dim, N) feature-major; differentiable in xa, xb and the
    parameters: the dispatcher op ``torch.ops.nesvor.fused_mlp`` (``nesvor_amd.ops``)."""
    layers = linear_layers(seq)
    need = torch.is_grad_enabled() and (xb.requires_grad or (xa is not None and xa.requires_grad) or any(
        p.requires_grad for l in layers for p in (l.weight, l.bias)))
    weights, biases = kernel_params(layers)  # the layers' own parameters at width 64, zero-padded twins below
    y, _ = torch.ops.nesvor.fused_mlp(xa, xb.contiguous(), weights, biases, b_row0, k_b, samples_per_pixel, -1, need)
    return y


_warned_library = set()


def library_mlp(seq: nn.Sequential, xa, xb, b_row0: int, k_b: int, samples_per_pixel: int):
    """A Linear/activation stack of ANY width / depth / activation on library GEMMs (rocBLAS through ``torch.matmul``),
    evaluated feature-major like the fused kernels: h = W x + b with x (k, N).  The per-pixel features enter as
    ``W[:, :k_a] xa^T`` computed once per pixel and repeated over the pixel's samples, so neither the expanded slice
    embedding nor a concatenated input matrix exists (models.py:339-353 builds both).  Differentiable by autograd."""
    mods = list(seq)
    first = mods[0]
    if not isinstance(first, nn.Linear):
        raise ValueError("network must start with a Linear layer")
    k_a = 0 if xa is None else xa.shape[1]
    w0 = first.weight
    h = w0[:, k_a:] @ xb[b_row0 : b_row0 + k_b]
    if xa is not None:
        h = h + (w0[:, :k_a] @ xa.t()).repeat_interleave(samples_per_pixel, dim=1)
    if first.bias is not None:
        h = h + first.bias[:, None]
    for m in mods[1:]:
        if isinstance(m, nn.Linear):
            h = m.weight @ h
            if m.bias is not None:
                h = h + m.bias[:, None]
        else:
            h = m(h)  # elementwise activation: layout-agnostic
    return h


def apply_net(net, xa, xb, b_row0: int, k_b: int, samples_per_pixel: int):
    """One network of the model on [xa (P, k_a) broadcast over each pixel's samples | rows [b_row0, b_row0 + k_b) of the
    feature-major xb] -> (out_dim, N) feature-major, differentiable.  Picks the evaluation:

    * Linear/ReLU stacks inside the fused kernels' shapes: ``fused_mlp`` (the dispatcher op);
    * Linear/ReLU stacks up to width 128 / seven hidden layers (``--width`` > 64, ``--depth`` > 3: the reference accepts any,
      cli/main.py:68-73): ``wide_mlp`` - hand-written fp32-MFMA kernels with one layer's weights in LDS at a time (round 6);
    * anything beyond (other activations, width > 128): ``library_mlp`` - correct, but on library GEMMs;
    * the half-precision structure (``tinycudann.Network``): its row-major module interface."""
    from .tinycudann import Network

    if isinstance(net, Network):
        x = xb[b_row0 : b_row0 + k_b].t()
        if xa is not None:
            x = torch.cat([xa.repeat_interleave(samples_per_pixel, dim=0), x], 1)
        return net(x).t()
    if supported(net):
        return fused_mlp(net, xa, xb, b_row0, k_b, samples_per_pixel)
    if wide_supported(net):  # width <= 128, up to seven hidden layers: the hand-written wide kernels (csrc/mlp_wide.hip)
        return wide_mlp(net, xa, xb, b_row0, k_b, samples_per_pixel)
    key = tuple((type(m).__name__, getattr(m, "in_features", 0), getattr(m, "out_features", 0)) for m in net)
    if key not in _warned_library:
        _warned_library.add(key)
        import logging

        logging.warning("MLP %s is outside the hand-written HIP kernels (ReLU; width <= 128, 1-7 hidden layers of one width, <= 64 "
                        "inputs, <= 16 outputs): its products run on library GEMMs - expect a several times slower iteration",
                        [k[1:] for k in key if k[0] == "Linear"])
    return library_mlp(net, xa, xb, b_row0, k_b, samples_per_pixel)


class FlatNetworkFunction(Function):
    """``tinycudann.Network`` (one flat bias-free parameter vector) on the fused kernels, 16-bit matrix operands (``HALF_OPERANDS``:
    bf16 by default, fp16 under ``args.fp16_loss_scaling``):
    x (N, k) row-major -> y (N, n_output_dims).  The kernels read a feature-major input and write a feature-major
    output; the two transposes are the price of tinycudann's row-major module interface (the training step proper
    never pays it: nesvor_amd.direct feeds the kernels feature-major tensors)."""

    @staticmethod
    def forward(ctx, x, params, net):
        p = NetParams(net)
        n = x.shape[0]
        n_pad = (n + 15) // 16 * 16  # the bf16 backward exists for whole 16-sample groups only: pad with zero rows
        xb = torch.zeros((x.shape[1], n_pad), dtype=torch.float32, device=x.device)
        xb[:, :n] = x.detach().t()
        need = any(ctx.needs_input_grad)
        if need and (p.n_hidden() > 2 or (p.n_hidden() == 2 and x.shape[1] > 32)):
            raise NotImplementedError("half-precision Network backward: built for 1-2 hidden layers (<= 32 inputs with 2)")
        y, saved = forward_raw(p.weights, p.biases, None, xb, 0, xb.shape[0], 16, need, HALF_OPERANDS[0])
        ctx.net, ctx.n, ctx.mode = net, n, HALF_OPERANDS[0]
        ctx.save_for_backward(xb, *saved)
        return y[:, :n].t()

    @staticmethod
    def backward(ctx, dy):
        xb, *saved = ctx.saved_tensors
        net, n = ctx.net, ctx.n
        p = NetParams(net)
        dyb = torch.zeros((dy.shape[1], xb.shape[1]), dtype=torch.float32, device=xb.device)
        dyb[:, :n] = dy.t()
        dxb = torch.empty_like(xb) if ctx.needs_input_grad[0] else None
        _, partial = backward_raw(p.weights, p.biases, None, xb, dyb, saved, 0, xb.shape[0], 16, dxb, False, ctx.mode)
        g = None
        if ctx.needs_input_grad[1]:
            g = torch.zeros_like(net.params)
            col, flat = 0, partial.sum(0)
            for w, off in zip(p.weights, p.w_off):  # padding rows of the last layer keep a zero gradient
                g[off : off + w.numel()] = flat[col : col + w.numel()]
                col += w.numel() + w.shape[0]
        return (None if dxb is None else dxb[:, :n].t()), g, None


HALF_OPERANDS = [True]  # the 16-bit operand mode of the half-precision structure's module path: True (bf16) or FP16 (set by train())


def flat_network(net, x):
    _lib.require_device(x, name="tinycudann.Network input")
    return FlatNetworkFunction.apply(x, net.params, net)


def inference_operands(inr, args):
    """How the density network's products are evaluated at inference (the `bf16` argument of ``forward_raw``) -
    the same choice the training step makes (nesvor_amd.direct): bf16 operands for the half-precision model structure
    and for ``args.mlp_bf16``, otherwise fp32 (split-fp16 evaluation, or the fp32 MFMAs with ``args.mlp_fp32_mfma``)."""
    from .tinycudann import Network

    net = inr.density_net
    if not supported(net):
        return None  # library GEMMs (apply_net)
    if isinstance(net, Network):
        return FP16 if getattr(args, "fp16_loss_scaling", False) else True
    if getattr(args, "mlp_bf16", False):
        return True
    if getattr(args, "mlp_fp16", False):
        return FP16S
    return MFMA_FP32 if getattr(args, "mlp_fp32_mfma", False) else False
