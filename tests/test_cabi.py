"""CPU: the C-ABI library loads and exports every symbol include/nesvor_hip.h declares
(no compute calls — there is no GPU here), and the product refuses host tensors."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "nesvor_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t|void\s*\*?)\s*(nesvor_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    from nesvor_amd import _lib

    ge.build()
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 10
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_lib.exported_symbols()) == declared, "ctypes signature table out of sync with the header"
    assert lib.nesvor_hip_abi_version() == _lib.ABI_VERSION


def test_no_cpu_fallback():
    """Host tensors are refused: by the dispatcher for the registered ops (only a CUDA-key = HIP kernel exists), by the
    raw launch helpers the fused step uses."""
    from nesvor_amd import slice_acq_cuda, transform_convert_cuda
    from nesvor_amd.encoding import hashgrid_encode, hashgrid_forward
    from nesvor_amd.grid import HashGridSpec

    no_cpu_kernel = "Could not run 'nesvor::"
    with pytest.raises(NotImplementedError, match=no_cpu_kernel):
        transform_convert_cuda.axisangle2mat_forward(torch.zeros(2, 6))
    with pytest.raises(NotImplementedError, match=no_cpu_kernel):
        slice_acq_cuda.forward(torch.zeros(1, 3, 4), torch.zeros(1, 1, 4, 4, 4), torch.empty(0), torch.empty(0),
                               torch.ones(1, 1, 1), (2, 2), 1.0, False, False)
    spec = HashGridSpec(2, 2, 8, 4, 1.5)
    with pytest.raises(RuntimeError, match="device"):
        hashgrid_forward(spec, torch.rand(4, 3), torch.zeros(spec.n_params))
    with pytest.raises(NotImplementedError, match=no_cpu_kernel):
        hashgrid_encode(torch.rand(4, 3), torch.zeros(spec.n_params), spec)
    with pytest.raises(NotImplementedError, match=no_cpu_kernel):
        slice_acq_cuda.adjoint_forward(torch.zeros(1, 3, 4), torch.ones(1, 1, 1), torch.zeros(1, 1, 2, 2), torch.empty(0),
                                       torch.empty(0), (4, 4, 4), 1.0, False, False)


def test_custom_ops_are_registered_with_schemas_and_meta_kernels():
    """north_star: "exposed as PyTorch-ROCm custom ops".  Every op of nesvor_amd.ops is a dispatcher op in the ``nesvor``
    namespace with a schema; only the CUDA (= HIP) key has a kernel; the fake kernels give shapes on meta tensors."""
    import nesvor_amd.ops as ops

    names = ops.op_names()
    for required in ("axisangle2mat_forward", "axisangle2mat_backward", "mat2axisangle_forward", "mat2axisangle_backward",
                     "slice_acq_forward", "slice_acq_backward", "slice_acq_adjoint_forward", "slice_acq_adjoint_backward",
                     "hashgrid_encode", "hashgrid_encode_backward", "fused_mlp", "fused_mlp_backward", "psf_transform",
                     "psf_transform_backward", "imaging_loss", "imaging_loss_backward", "trans_loss", "adamw_step_"):
        assert required in names, required
        op = getattr(torch.ops.nesvor, required).default
        assert str(op._schema).startswith(f"nesvor::{required}(")
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f"nesvor::{required}", "CUDA")
        assert not torch._C._dispatch_has_kernel_for_dispatch_key(f"nesvor::{required}", "CPU")
    m = lambda *s, dtype=torch.float32: torch.empty(*s, dtype=dtype, device="meta")
    assert torch.ops.nesvor.axisangle2mat_forward(m(5, 6)).shape == (5, 3, 4)
    assert torch.ops.nesvor.mat2axisangle_forward(m(5, 3, 4, dtype=torch.float64)).dtype == torch.float64
    out = torch.ops.nesvor.slice_acq_forward(m(7, 3, 4), m(1, 1, 8, 9, 10), m(0), m(0), m(3, 3, 3), [11, 12], 1.5, True, False)
    assert [tuple(t.shape) for t in out] == [(7, 1, 11, 12)] * 2
    vol, wgt = torch.ops.nesvor.slice_acq_adjoint_forward(m(7, 3, 4), m(3, 3, 3), m(7, 1, 11, 12), m(0), m(0), [8, 9, 10], 1.5, False, True)
    assert tuple(vol.shape) == tuple(wgt.shape) == (1, 1, 8, 9, 10)
    assert torch.ops.nesvor.hashgrid_encode(m(100, 3), m(1000), 4, 2, 8, 4, 1.5, 0).shape == (100, 8)
    assert torch.ops.nesvor.hashgrid_encode(m(100, 3), m(1000), 4, 2, 8, 4, 1.5, 1).shape == (8, 100)
    x, u = torch.ops.nesvor.psf_transform(m(3, 3, 4), m(10, dtype=torch.int64), m(10, 3), m(3, 3), m(10, 16, 3), m(2, 3))
    assert x.shape == (10, 16, 3) and u.shape == (160, 3)
    y, saved = torch.ops.nesvor.fused_mlp(None, m(32, 160), [m(64, 32), m(64, 64), m(16, 64)], [m(64), m(64), m(16)], 0, 32, 16, -1, True)
    assert y.shape == (16, 160) and len(saved) == 2


def test_mlp_fake_sizes_match_forward():
    """The fake kernel of ``nesvor::fused_mlp`` sizes the saved-activation buffers from a descriptor WITHOUT pointers; the
    real forward decides the compact save on its full descriptor.  Both must agree for every shape (the predicate,
    csrc/mlp.hip::compact_ok, reads shape fields only) - host logic of the library, no device work."""
    from nesvor_amd import mlp

    for n_hidden, out_dim, k_a, k_b, b_row0, S, N in [(2, 16, 0, 32, 0, 256, 1 << 20), (2, 1, 16, 15, 1, 256, 1 << 20),
                                                       (1, 16, 0, 32, 0, 64, 1 << 16), (2, 16, 0, 32, 0, 24, 24 * 100),
                                                       (2, 1, 16, 8, 0, 256, 1 << 18), (3, 16, 0, 32, 0, 256, 1 << 18),
                                                       (2, 16, 0, 48, 0, 256, 1 << 18), (2, 16, 0, 32, 0, 256, (1 << 18) + 16)]:
        dims = [k_a + k_b] + [64] * n_hidden + [out_dim]
        ws = [torch.zeros(o, i) for i, o in zip(dims[:-1], dims[1:])]
        bs = [torch.zeros(o) for o in dims[1:]]
        for mode in (False, True, mlp.MFMA_FP32):
            real = mlp._desc(ws, bs, k_a, k_b, b_row0, S, mode)
            bare = mlp.dims_desc(n_hidden, out_dim, k_a, k_b, b_row0, S, mode)
            assert mlp.saved_sizes(real, N, n_hidden) == mlp.saved_sizes(bare, N, n_hidden), (n_hidden, out_dim, k_a, k_b, S, N, mode)
    # the headline networks do save compactly (the default evaluation of the fp32 products)
    assert mlp.saved_sizes(mlp.dims_desc(2, 16, 0, 32, 0, 256), 1 << 20, 2)[0] == (1 << 20) * 4


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under nesvor_amd/ may reference it."""
    pkg = os.path.join(ROOT, "nesvor_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "from .. import oracle" not in src and "from oracle" not in src, f


def test_backward_workspace_size_queries_follow_the_layout_hints():
    """Host-side arithmetic only (no GPU): ``nesvor_hashgrid_backward_workspace_bytes_ex`` asks for the re-ordered copy of d pe
    exactly when the backward will be called feature-major, unclustered AND with the scratch hint; every other combination is the
    plain size, which already holds the point order (perm, per-cell strips, spill list)."""
    import ctypes

    from nesvor_amd import _lib
    from nesvor_amd.grid import HashGridSpec

    lib = _lib.load()
    spec = HashGridSpec(16, 2, 19, 9, 1.26)
    N = (1 << 20) - 77
    n_pad = (N + 255) // 256 * 256
    base = lib.nesvor_hashgrid_backward_workspace_bytes(ctypes.byref(spec.c_struct), N, None)
    assert base > 0
    ex = lambda layout: lib.nesvor_hashgrid_backward_workspace_bytes_ex(ctypes.byref(spec.c_struct), N, None, layout)
    U, S = _lib.LAYOUT_UNCLUSTERED, _lib.LAYOUT_DY_SCRATCH
    assert ex(0 | U | S) == ex(1 | U) == ex(0 | U) == base
    assert ex(1 | U | S) == base + 256 + n_pad * 32 * 4
    # the order's scratch - 4 N (perm) + counters + 4 N (spill) + 8 N (tickets) + strips (4096 cells x 512 slots at this N) - is
    # part of the plain query (whose callers may pass any hint later) and of every query WITH the unclustered hint; a caller that
    # names a layout without it - the training step - does not pay for it (round-5 advisor: 26 MB at N = 2^20)
    sort_bytes = (n_pad * 2 + 2 * ((1 << 18) + 2)) * 4 + n_pad * 8 + 4096 * 512 * 4
    sort_bytes = (sort_bytes + 255) // 256 * 256
    assert ex(0) == ex(1) == ex(1 | S) == ex(1 | _lib.LAYOUT_CLUSTERED) == base - sort_bytes
    # ... on top of the worst-case record queues (8 records of 12 B per point and level, with slack)
    assert base > 128 * 12 * N and base - 128 * 12 * N * 1.2 < 64 * N
    # the unclustered FORWARD's scratch: the order, plus the encoded rows in that order for feature-major output
    fw = lambda layout: lib.nesvor_hashgrid_forward_workspace_bytes(ctypes.byref(spec.c_struct), N, layout)
    assert fw(0) == fw(1) == fw(1 | _lib.LAYOUT_CLUSTERED) == 0
    assert fw(0 | U) == 256 + sort_bytes and fw(1 | U) == 256 + sort_bytes + n_pad * 32 * 4
    small = lib.nesvor_hashgrid_backward_workspace_bytes(ctypes.byref(spec.c_struct), 1000, None)
    assert 0 < small < base
