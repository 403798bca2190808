"""Hash-grid encoding: raw launches of the HIP kernels (used by the fused training step and by the dispatcher ops of
``nesvor_amd.ops``) and the differentiable ``hashgrid_encode``."""
import ctypes

import torch
from torch.autograd import Function

from . import _lib
from .grid import HashGridSpec


_FWD_WS = {}
UNCLUSTERED_FWD_MIN_POINTS = 1 << 15  # below this the launches of the ordered forward cost more than they save
_FWD_MODE = __import__("os").environ.get("NESVOR_HASHGRID_FWD", "")  # cloud | level | gather: force one kernel (csrc/hashgrid.hip); sorted: force the ordered forward


def hashgrid_forward(spec: HashGridSpec, u: torch.Tensor, table: torch.Tensor, layout=_lib.LAYOUT_ROW_MAJOR, clustered=False):
    """clustered=True: the caller's promise that every 256 consecutive points are spatially clustered (the PSF samples of a
    slice pixel are contiguous; consecutive voxels of a raster-ordered lattice): the one-workgroup-per-256-points kernel on the
    points as given.  clustered=False (the default: any tcnn-style caller, nesvor/nesvor/models.py:25; points in arbitrary
    order, sample.py:29) - what was measured at N = 2^20 uniform points (tools/bench_hg_fwd_unclustered.py,
    profiles/r06_hashgrid_fwd_unclustered.log) decides:
      * feature-major output: one block per (256 points, level) on the points as given - every level's table slice in turn is
        what the L2s hold, and the x-pairs of corners are fetched with one request where their entries are neighbours;
      * row-major output (tinycudann's layout) from ``UNCLUSTERED_FWD_MIN_POINTS`` points on: the points are first put into the
        order of a coarse lattice's cells and the per-cloud kernel runs on workgroups of neighbouring points, every thread
        writing its point's whole row (``nesvor_hashgrid_forward_unclustered``; 0.36 against 0.49 ms - the per-level kernel
        writes 8 bytes per (point, level) into rows of 128).
    A performance hint only: the encoded values are the same."""
    _lib.require_device(u, table, dtype=torch.float32, name="hashgrid input/table")
    N = u.shape[0]
    E = spec.n_output_dims
    shape = (N, E) if layout == _lib.LAYOUT_ROW_MAJOR else (E, N)
    pe = torch.empty(shape, dtype=torch.float32, device=u.device)
    lib = _lib.load()
    ordered = ((not clustered and layout == _lib.LAYOUT_ROW_MAJOR and N >= UNCLUSTERED_FWD_MIN_POINTS and _FWD_MODE == "")
               or (_FWD_MODE == "sorted" and N > 0))
    with torch.cuda.device(u.device), _lib.kernel_timer.span("hashgrid_fwd"):
        if ordered:
            lay = layout | _lib.LAYOUT_UNCLUSTERED
            nbytes = lib.nesvor_hashgrid_forward_workspace_bytes(ctypes.byref(spec.c_struct), N, lay)
            ws = _FWD_WS.get(u.device)
            if ws is None or ws.numel() < nbytes:
                ws = _FWD_WS[u.device] = torch.empty(nbytes, dtype=torch.uint8, device=u.device)
            err = lib.nesvor_hashgrid_forward_unclustered(ctypes.byref(spec.c_struct), _lib.ptr(u), _lib.ptr(table), _lib.ptr(pe), N, lay,
                                                          None, _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
        else:
            err = lib.nesvor_hashgrid_forward(
                ctypes.byref(spec.c_struct), _lib.ptr(u), _lib.ptr(table), _lib.ptr(pe), N,
                layout | (_lib.LAYOUT_CLUSTERED if clustered else 0), _lib.stream_ptr()
            )
    _lib.check(err, "hashgrid forward")
    return pe


def points_are_ordered(u: torch.Tensor, box_fraction: float = 1.0 / 16) -> bool:
    """Whether consecutive points of ``u`` (N, 3, normalised to the unit cube) are spatial neighbours - a raster-ordered lattice
    (``sample_volume``'s voxel list), PSF clouds - so that 256 of them span a small lattice box: the median step between
    neighbours of a 4096-point prefix is below ``box_fraction`` of the cube.  One small reduction and one host read per CALL of an
    inference entry point (not per chunk): picks the forward kernel, never the result."""
    n = min(int(u.shape[0]), 4096)
    if n < 2:
        return False
    step = (u[1:n] - u[: n - 1]).abs().amax(-1)
    return bool(step.median() < box_fraction)


_WORKSPACES = {}


class QueueSizer:
    """Capacity of the backward's record queues, per level, as a fraction of the worst case (8 records per point and
    level = 1.8 GB at N = 2^20, L = 16).  A PSF-cloud batch fills a few percent of that at most levels, uniform points
    fill the fine levels completely: the sizer starts every level at ``START`` of the worst case and grows (x4, up to 1)
    the levels whose overflow counter the kernels raised.  An overflowing record is added with a global atomic, so the
    result is exact at any capacity; only the first iterations of a run - until the capacities have settled, two growth
    steps at most - pay for the atomics.  The counters are read without synchronising: an asynchronous copy into pinned
    memory after a backward, looked at before a later one (every call for the first 64 calls, every 4th afterwards, so
    that a change of the point distribution is noticed within a few calls).

    ``NESVOR_HASHGRID_QUEUE=worst`` (or ``QueueSizer.policy = "worst"``) allocates the worst case once and never looks.
    ``NESVOR_HASHGRID_QUEUE=load:<file>`` starts from the capacities an earlier process settled on and wrote with
    ``save_queue_scales(<file>)`` (a resumed run, or a profiled run that must not contain the settling launches: bench.py
    writes the file when ``NESVOR_HASHGRID_QUEUE_SAVE`` names one); growth continues from there as in the adaptive policy."""

    START = 1.0 / 16
    policy = __import__("os").environ.get("NESVOR_HASHGRID_QUEUE", "adaptive")

    def __init__(self, n_levels: int, key=None) -> None:
        start = 1.0 if QueueSizer.policy == "worst" else QueueSizer.START
        self.scale = (ctypes.c_float * _lib.MAX_LEVELS)(*([start] * _lib.MAX_LEVELS))
        if QueueSizer.policy.startswith("load:") and key is not None:
            import json, os

            path = QueueSizer.policy[5:]
            saved = json.load(open(path)).get(_key_name(key)) if os.path.exists(path) else None
            if saved is not None:
                for l, v in enumerate(saved[: _lib.MAX_LEVELS]):
                    self.scale[l] = min(1.0, max(QueueSizer.START, float(v)))
        self.n_levels = n_levels
        self.calls = 0
        self.pending = None  # event of the counter copy in flight
        self.host = None  # pinned int32[32] the counters are copied into

    def poll(self) -> bool:
        """Look at a finished counter copy; True if a level was grown (the workspace must then be re-made)."""
        if self.pending is None or not self.pending.query():
            return False
        counts = self.host.tolist()
        self.pending = None
        grown = False
        for l in range(self.n_levels):
            if counts[l] > 0 and self.scale[l] < 1.0:
                self.scale[l] = min(1.0, self.scale[l] * 4)
                grown = True
        return grown

    def snapshot(self, ws: torch.Tensor) -> None:
        """Queue an asynchronous copy of the latest backward's overflow counters, on a stream of its own behind the
        current one: nothing but ``poll`` ever waits for it (on the training step's streams it sat in front of the
        optimizer)."""
        self.calls += 1
        if QueueSizer.policy == "worst" or self.pending is not None or not (self.calls <= 64 or self.calls % 4 == 0):
            return
        if all(self.scale[l] >= 1.0 for l in range(self.n_levels)):
            return
        off = _lib.load().nesvor_hashgrid_backward_overflow_offset(_lib.ptr(ws))
        if self.host is None:
            self.host = torch.empty(_lib.MAX_LEVELS, dtype=torch.int32, pin_memory=True)
            self.stream = torch.cuda.Stream(device=ws.device)
        self.stream.wait_stream(torch.cuda.current_stream(ws.device))
        with torch.cuda.stream(self.stream):
            self.host.copy_(ws[off : off + 4 * _lib.MAX_LEVELS].view(torch.int32), non_blocking=True)
            self.pending = torch.cuda.Event()
            self.pending.record()


_SIZERS = {}


def _key_name(key) -> str:
    return "/".join(str(k) for k in key[1:])  # (without the device: the capacities follow from the points, not the card)


def save_queue_scales(path: str) -> None:
    """Write the capacities every sizer of this process has settled on (``NESVOR_HASHGRID_QUEUE=load:<path>`` reads them)."""
    import json

    with open(path, "w") as f:
        json.dump({_key_name(k): [float(v) for v in sz.scale] for k, sz in _SIZERS.items()}, f)


def queue_sizer(spec, N, device, clustered=True) -> QueueSizer:
    """One sizer per (device, batch size, grid, clustered hint): an unclustered batch fills the fine levels' queues completely, a
    PSF-cloud batch a few percent - a capacity grown for the one would make the other's owner pass walk empty slices."""
    key = (device, N, spec.n_levels, spec.n_features, spec.log2_hashmap_size, spec.base_resolution, spec.per_level_scale, bool(clustered))
    sz = _SIZERS.get(key)
    if sz is None:
        sz = _SIZERS[key] = QueueSizer(spec.n_levels, key)
    return sz


def _workspace(spec, N, device, sizer=None, layout=0):
    """Scratch for the owner-computes backward (queues of (entry, grad) records), cached per (device, size).  ``layout``: the one
    the backward will be called with, hints included (an unclustered feature-major backward keeps a re-ordered copy of dpe here)."""
    sizer = sizer or queue_sizer(spec, N, device)
    nbytes = _lib.load().nesvor_hashgrid_backward_workspace_bytes_ex(ctypes.byref(spec.c_struct), N, sizer.scale, layout)
    if nbytes < 0:
        return None
    key = (device, nbytes)
    ws = _WORKSPACES.get(key)
    if ws is None:
        stale = [k for k in _WORKSPACES if k[0] == device]
        if stale:
            # the old workspace may still be read on streams the caching allocator knows nothing about (an owner pass on
            # the caller's side stream, the sizer's counter copy): let the device drain before its memory is recycled.
            # Happens a handful of times per run (first iterations, while the queue capacities settle)
            torch.cuda.synchronize(device)
        for k in stale:
            del _WORKSPACES[k]  # one live workspace per device
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        ws[: _lib.load().nesvor_hashgrid_backward_workspace_zero_bytes()].zero_()  # queue tails: zero once, kept by the kernels
        _WORKSPACES[key] = ws
    return ws


def hashgrid_backward(spec, u, table, dpe, grad_table=None, need_input_grad=True, layout=_lib.LAYOUT_ROW_MAJOR,
                      method="owner", levels=None, grad_u=None, first=True, owner_stream=None, dy_bound=None, clustered=True):
    """Accumulates into grad_table (allocated zero-filled if None); returns (grad_table, grad_u|None).
    method: "owner" (LDS aggregation + per-chunk owners, the MI355X path) or "atomic" (per-corner atomics).
    levels = (begin, end): only these levels ("owner" method) - a data-parallel step splits the backward in two so
    that the all-reduce of the first part overlaps the second; later parts pass the first part's ``grad_u`` and
    ``first=False`` (queue tails are not reset, the input gradient is added).
    owner_stream: launch the owner pass (which only finishes ``grad_table``) on this stream, behind the aggregation pass;
    ``grad_u`` is complete on the current stream, the caller joins ``owner_stream`` before it reads ``grad_table``.
    dy_bound: 1-element device tensor >= max |dpe| ("owner" method): the aggregation pass then skips its own pass over dpe
    (the training step gets the bound from the MLP backward that produced dpe, csrc/step.hip).
    clustered=False: consecutive points are not spatially clustered (uniform points, a shuffled batch) - the "owner" method
    then orders them by coarse lattice cell first (``NESVOR_LAYOUT_UNCLUSTERED``); the default is what the training step
    produces, the S PSF samples of a pixel next to each other.  The gradients do not depend on the hint."""
    _lib.require_device(u, table, dpe, dtype=torch.float32, name="hashgrid backward input")
    if not clustered:
        layout = layout | _lib.LAYOUT_UNCLUSTERED | _lib.LAYOUT_DY_SCRATCH
    N = u.shape[0]
    if grad_table is None:
        grad_table = torch.zeros_like(table)
    if grad_u is None:
        if not first and need_input_grad:
            raise RuntimeError("a later part of a split backward needs the first part's grad_u")
        grad_u = torch.empty_like(u) if need_input_grad else None
    lib = _lib.load()
    sizer = queue_sizer(spec, N, u.device, clustered) if method == "owner" else None
    if sizer is not None and first:
        sizer.poll()  # grows the queues of levels that overflowed in an earlier backward (the workspace is then re-made)
    ws = _workspace(spec, N, u.device, sizer, layout) if method == "owner" else None
    if levels is not None and ws is None:
        raise RuntimeError("level ranges exist for the owner method only")
    with torch.cuda.device(u.device):
        if ws is not None:
            args = (ctypes.byref(spec.c_struct), _lib.ptr(u), _lib.ptr(table), _lib.ptr(dpe), _lib.ptr(grad_table),
                    _lib.ptr(grad_u), N, layout, _lib.ptr(ws))
            l0, l1 = (0, spec.n_levels) if levels is None else levels
            extra = 0 if first else (4 | 8)  # keep the queue tails, add to grad_u
            call = lambda stage: lib.nesvor_hashgrid_backward_bounded(*args, stage | extra, l0, l1, sizer.scale, _lib.ptr(dy_bound),
                                                                      _lib.stream_ptr())
            if _lib.kernel_timer.enabled:  # bracket each of the two launches with its own events
                with _lib.kernel_timer.span("hashgrid_bwd_aggregate"):
                    err = call(1)
                if err == 0:
                    with _lib.kernel_timer.span("hashgrid_bwd_owner"):
                        err = call(2)
            elif owner_stream is not None:
                err = call(1)
                if err == 0:
                    owner_stream.wait_stream(torch.cuda.current_stream(u.device))
                    err = lib.nesvor_hashgrid_backward_bounded(*args, 2 | extra, l0, l1, sizer.scale, _lib.ptr(dy_bound),
                                                               ctypes.c_void_p(owner_stream.cuda_stream))
            else:
                err = call(3)
        else:
            err = lib.nesvor_hashgrid_backward_atomic(
                ctypes.byref(spec.c_struct), _lib.ptr(u), _lib.ptr(table), _lib.ptr(dpe), _lib.ptr(grad_table),
                _lib.ptr(grad_u), N, layout, _lib.stream_ptr(),
            )
    _lib.check(err, "hashgrid backward")
    if sizer is not None and ws is not None:
        sizer.snapshot(ws)  # (behind the aggregation pass, which wrote the counters; on the sizer's own stream)
    return grad_table, grad_u


def hashgrid_backward_adamw(spec, u, table, dpe, grad_table, exp_avg, exp_avg_sq, adam, need_input_grad=True,
                            layout=_lib.LAYOUT_ROW_MAJOR, dy_bound=None):
    """``hashgrid_backward`` (owner method, all levels) whose owner pass also takes the AdamW step on the table
    (``nesvor_hashgrid_backward_adamw``): equals ``hashgrid_backward`` into ``grad_table`` followed by
    ``nesvor_adamw_step(table, grad_table, exp_avg, exp_avg_sq, ..., zero_grad=1)`` - ``table`` and the moments are updated
    in place, ``grad_table`` is zero afterwards - without the table gradient passing through HBM.  ``adam``: ``_lib.AdamwT``.
    Returns grad_u (or None).  The training step (csrc/step.hip) makes the same call when one call covers gradient and
    update."""
    _lib.require_device(u, table, dpe, grad_table, exp_avg, exp_avg_sq, dtype=torch.float32, name="hashgrid backward + AdamW input")
    N = u.shape[0]
    grad_u = torch.empty_like(u) if need_input_grad else None
    lib = _lib.load()
    sizer = queue_sizer(spec, N, u.device)
    sizer.poll()
    ws = _workspace(spec, N, u.device, sizer)
    if ws is None:
        raise RuntimeError("the grid does not fit the owner method's plan")
    with torch.cuda.device(u.device):
        err = lib.nesvor_hashgrid_backward_adamw(ctypes.byref(spec.c_struct), _lib.ptr(u), _lib.ptr(table), _lib.ptr(dpe), _lib.ptr(grad_table),
                                                 _lib.ptr(grad_u), N, layout, _lib.ptr(ws), 3, sizer.scale, _lib.ptr(dy_bound),
                                                 _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq), ctypes.byref(adam), _lib.stream_ptr())
    _lib.check(err, "hashgrid backward + AdamW")
    sizer.snapshot(ws)
    return grad_u


def hashgrid_encode(u, table, spec, layout=_lib.LAYOUT_ROW_MAJOR, grad_accum=None):
    """pe = encode(u, table), differentiable in both: the dispatcher op ``torch.ops.nesvor.hashgrid_encode``
    (``nesvor_amd.ops``; row-major (N, L*F) output like tinycudann, or feature-major).  ``grad_accum``: a tensor the
    table gradient is accumulated into by the backward (the fused trainer's flat gradient buffer) instead of a fresh
    30 MB tensor that autograd then adds."""
    cfg = (spec.n_levels, spec.n_features, spec.log2_hashmap_size, spec.base_resolution, spec.per_level_scale, layout)
    if grad_accum is None:
        return torch.ops.nesvor.hashgrid_encode(u.contiguous(), table, *cfg)
    return _AccumulatingEncode.apply(u.contiguous(), table, cfg, grad_accum)


class _AccumulatingEncode(Function):
    """The same two ops with the in-place backward variant (``hashgrid_encode_backward_``): used where the caller owns
    the gradient buffer."""

    @staticmethod
    def forward(ctx, u, table, cfg, grad_accum):
        ctx.save_for_backward(u, table)
        ctx.cfg, ctx.grad_accum = cfg, grad_accum
        with torch.no_grad():
            return torch.ops.nesvor.hashgrid_encode(u, table, *cfg)

    @staticmethod
    def backward(ctx, dpe):
        u, table = ctx.saved_tensors
        gu = torch.ops.nesvor.hashgrid_encode_backward_(u, table, dpe.contiguous(), ctx.grad_accum, *ctx.cfg, ctx.needs_input_grad[0])
        return (gu if ctx.needs_input_grad[0] else None), None, None, None
