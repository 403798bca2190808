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
    return sorted(set(re.findall(r"\b(?:int|int64_t)\s+(nesvor_\w+)\s*\(", text)))


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
    from nesvor_amd import slice_acq_cuda, transform_convert_cuda
    from nesvor_amd.encoding import hashgrid_forward
    from nesvor_amd.grid import HashGridSpec

    with pytest.raises(RuntimeError, match="device"):
        transform_convert_cuda.axisangle2mat_forward(torch.zeros(2, 6))
    with pytest.raises(RuntimeError, match="device"):
        slice_acq_cuda.forward(torch.zeros(1, 3, 4), torch.zeros(1, 1, 4, 4, 4), torch.empty(0), torch.empty(0),
                               torch.ones(1, 1, 1), (2, 2), 1.0, False, False)
    spec = HashGridSpec(2, 2, 8, 4, 1.5)
    with pytest.raises(RuntimeError, match="device"):
        hashgrid_forward(spec, torch.rand(4, 3), torch.zeros(spec.n_params))
    with pytest.raises(NotImplementedError):  # the PSF-interpolation mode is not built: it raises, it never falls back
        slice_acq_cuda.adjoint_backward(None, None, None, None, None, None, None, None, 1.0, True, False, True, True)
    with pytest.raises(RuntimeError, match="device"):
        slice_acq_cuda.adjoint_forward(torch.zeros(1, 3, 4), torch.ones(1, 1, 1), torch.zeros(1, 1, 2, 2), torch.empty(0),
                                       torch.empty(0), (4, 4, 4), 1.0, False, False)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under nesvor_amd/ may reference it."""
    pkg = os.path.join(ROOT, "nesvor_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "from .. import oracle" not in src and "from oracle" not in src, f
