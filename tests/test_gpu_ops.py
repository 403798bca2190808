"""GPU (-m gpu): parity of every HIP op with the CPU oracle on the same seeded inputs,
through the C ABI (ctypes).  Tolerances are stated per test; index/byte work is exact."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, scipy_table

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ transforms
def test_axisangle2mat_mat2axisangle_scipy_table(device):
    """The reference's own test (tests/transform/test_transform_convert.py:13-21), default fp32 tolerances."""
    from nesvor_amd.transform import axisangle2mat, mat2axisangle

    ax, mat = scipy_table()
    for i in range(len(ax)):
        torch.testing.assert_close(axisangle2mat(ax[i : i + 1].to(device)).cpu(), mat[i : i + 1])
        torch.testing.assert_close(mat2axisangle(mat[i : i + 1].to(device)).cpu(), ax[i : i + 1])


def test_compose_inv_reference_test_gpu(device):
    """tests/transform/test_transform.py:7-23 with the reference's tolerance (atol 2e-5 rotation;
    translations O(300) get 1e-4, one fp32 ulp there being 3e-5)."""
    from nesvor_amd.transform import RigidTransform

    ax, mat = scipy_table()
    ax, mat = ax.to(device), mat.to(device)
    n = len(ax)
    for i in range(n):
        a_ax, a_m, b_ax, b_m = ax[i : i + 1], mat[i : i + 1], ax[n - 1 - i : n - i], mat[n - 1 - i : n - i]
        ab = RigidTransform(a_ax, trans_first=i % 2 == 0).compose(RigidTransform(b_m, trans_first=i % 2 == 1))
        ba = RigidTransform(b_ax, trans_first=i % 2 == 1).inv().compose(RigidTransform(a_m, trans_first=i % 2 == 0).inv())
        err = ab.compose(ba).axisangle().cpu()
        torch.testing.assert_close(err[:, :3], torch.zeros(1, 3), atol=2e-5, rtol=1e-3)
        torch.testing.assert_close(err[:, 3:], torch.zeros(1, 3), atol=1e-4, rtol=1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_transform_kernels_vs_oracle_fwd_bwd(device, dtype):
    from nesvor_amd import transform_convert_cuda as K
    from oracle import transform_convert as O

    torch.manual_seed(0)
    ax = torch.randn(4096, 6, dtype=dtype)
    ax[:64, :3] *= 1e-4  # small-angle branch
    ax[64:1024, :3] *= 2.5  # all quaternion branches
    ax[1024] = 0
    g = torch.randn(4096, 3, 4, dtype=dtype)
    ga = torch.randn(4096, 6, dtype=dtype)
    tol = dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=1e-10, atol=1e-10)
    m_ref = O.axisangle2mat_forward(ax)
    m = K.axisangle2mat_forward(ax.to(device))[0]
    torch.testing.assert_close(m.cpu(), m_ref, **tol)
    torch.testing.assert_close(K.axisangle2mat_backward(g.to(device), ax.to(device))[0].cpu(),
                               O.axisangle2mat_backward(g, ax), **tol)
    _, masks = O._quat_branches(m_ref)
    assert all(bool(b.any()) for b in masks)
    # feed the ORACLE's matrices to both so branch selection sees identical inputs
    torch.testing.assert_close(K.mat2axisangle_forward(m_ref.to(device))[0].cpu(), O.mat2axisangle_forward(m_ref),
                               rtol=1e-4, atol=1e-4) if dtype == torch.float32 else None
    a_k = K.mat2axisangle_forward(m_ref.to(device))[0].cpu()
    a_o = O.mat2axisangle_forward(m_ref)
    # near theta = pi the axis sign is ill-conditioned: compare through the rotation it represents
    torch.testing.assert_close(O.axisangle2mat_forward(a_k), O.axisangle2mat_forward(a_o),
                               rtol=1e-4, atol=1e-4 if dtype == torch.float32 else 1e-9)
    gk = K.mat2axisangle_backward(m_ref.to(device), ga.to(device))[0].cpu()
    go = O.mat2axisangle_backward(m_ref, ga)
    # the backward divides by sin(theta/2): scale the tolerance by the conditioning of each row
    scale = go.abs().amax((1, 2), keepdim=True).clamp(min=1.0)
    assert float(((gk - go).abs() / scale).max()) < (5e-4 if dtype == torch.float32 else 1e-8)


def test_trans_loss_fused_vs_oracle(device):
    """Fused pose regulariser (forward + gradient) vs the oracle's composition of the four conversions."""
    from nesvor_amd.transform import trans_loss_fused
    from oracle import nesvor_model as nm

    torch.manual_seed(0)
    ax0 = torch.randn(231, 6) * torch.tensor([1.0, 1.0, 1.0, 40, 40, 40])
    ax = (ax0 + torch.randn(231, 6) * torch.tensor([0.05, 0.05, 0.05, 1.0, 1.0, 1.0])).requires_grad_(True)
    ref = nm.trans_loss(ax, ax0)
    ref.backward()
    axd = ax.detach().to(device).requires_grad_(True)
    got = trans_loss_fused(axd, ax0.to(device))
    (3.0 * got).backward()
    assert abs(float(got) - float(ref)) <= 2e-4 * abs(float(ref)) + 1e-9
    scale = float(ax.grad.abs().max())
    assert float((axd.grad.cpu() / 3.0 - ax.grad).abs().max()) < 2e-3 * scale
    # identical poses: err == 0 (small-angle branch), zero gradient
    z = trans_loss_fused(ax0.to(device).requires_grad_(True), ax0.to(device))
    assert float(z) < 1e-9


def test_transform_ops_reject_bad_input(device):
    from nesvor_amd import transform_convert_cuda as K

    with pytest.raises(RuntimeError):
        K.axisangle2mat_forward(torch.zeros(3, 6))  # host tensor
    with pytest.raises(RuntimeError):
        K.axisangle2mat_forward(torch.zeros(6, 3, device=device).t())  # non-contiguous
    assert K.axisangle2mat_forward(torch.zeros(0, 6, device=device))[0].shape == (0, 3, 4)  # empty input


# ------------------------------------------------------------- slice acquisition
def _acq(device, tf, vol, vm, sm, psf, shape, rs, need_w, interp):
    from nesvor_amd import slice_acq_cuda as K

    d = lambda t: None if t is None else t.to(device)
    e = torch.empty(0, device=device)
    out = K.forward(d(tf), d(vol), e if vm is None else d(vm), e if sm is None else d(sm), d(psf), shape, rs, need_w, interp)
    return [o.cpu() for o in out]


@pytest.mark.parametrize("interp_psf", [False, True])
@pytest.mark.parametrize("masks", [False, True])
def test_slice_acq_vs_oracle(device, interp_psf, masks):
    from nesvor_amd.utils import get_PSF
    from oracle import slice_acq as O
    from oracle import transform_convert as tc

    torch.manual_seed(0)
    vol = torch.rand(1, 1, 20, 22, 24)
    psf = get_PSF(res_ratio=(1.5, 1.5, 3.0))
    ax = torch.randn(7, 6) * torch.tensor([0.6, 0.6, 0.6, 3.0, 3.0, 3.0])
    ax[0] = 0
    tf = tc.axisangle2mat_forward(ax)
    vm = (torch.rand(1, 1, 20, 22, 24) > 0.2) if masks else None
    sm = (torch.rand(7, 1, 18, 16) > 0.3) if masks else None
    ref, wref = O.slice_acquisition_forward(tf, vol, vm, sm, psf, (18, 16), 1.5, True, interp_psf)
    got, wgot = _acq(device, tf, vol, vm, sm, psf, (18, 16), 1.5, True, interp_psf)
    # fp32: <=153 taps x 8 corners accumulated in the same order; floor() decisions can differ only
    # where a coordinate lands within 1 ulp of a voxel boundary (continuous there for the linear mode)
    tol = dict(rtol=1e-4, atol=1e-5) if not interp_psf else dict(rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(wgot, wref, **tol)
    torch.testing.assert_close(got, ref, **tol)
    if masks:
        assert float(got[~sm].abs().max()) == 0.0  # masked-out pixels stay exactly zero


def test_slice_acq_golden_stacks_and_properties(device, golden):
    """The phantom stacks the reference's wrappers produced (tests/golden) + linearity / constant-volume."""
    from nesvor_amd.utils import get_PSF

    vs, res, res_s, s_thick, gap, n_slice, ss = golden["sim_geom"]
    psf = get_PSF(res_ratio=(res_s / res, res_s / res, s_thick / res))
    vol = torch.tensor(golden["sim_volume"])[None, None]
    tf = torch.tensor(golden["sim_transforms"])
    (got,) = _acq(device, tf, vol, None, None, psf, (int(ss), int(ss)), float(res_s / res), False, False)
    np.testing.assert_allclose(got.numpy(), golden["sim_stacks"], rtol=1e-4, atol=1e-5)
    v2 = torch.rand_like(vol)
    f = lambda v: _acq(device, tf, v, None, None, psf, (int(ss), int(ss)), float(res_s / res), True, False)
    (a, w), (b, _), (c, _) = f(vol), f(v2), f(2 * vol + 3 * v2)
    torch.testing.assert_close(c, 2 * a + 3 * b, rtol=1e-4, atol=1e-5)
    ones, w1 = f(torch.ones_like(vol))
    torch.testing.assert_close(ones[w1 > 0], torch.ones_like(ones[w1 > 0]), rtol=1e-5, atol=1e-6)
    # empty input
    (e,) = _acq(device, tf[:0], vol, None, None, psf, (4, 4), 1.5, False, False)
    assert e.shape == (0, 1, 4, 4)


def _sa_setup(masks, seed=0):
    from nesvor_amd.utils import get_PSF
    from oracle import transform_convert as tc

    torch.manual_seed(seed)
    vol = torch.rand(1, 1, 18, 20, 22)
    psf = get_PSF(res_ratio=(1.5, 1.5, 3.0))
    ax = torch.randn(6, 6) * torch.tensor([0.6, 0.6, 0.6, 3.0, 3.0, 3.0])
    ax[0] = 0
    tf = tc.axisangle2mat_forward(ax)
    vm = (torch.rand(1, 1, 18, 20, 22) > 0.2) if masks else None
    sm = (torch.rand(6, 1, 14, 12) > 0.3) if masks else None
    return vol, psf, tf, vm, sm


@pytest.mark.parametrize("masks", [False, True])
@pytest.mark.parametrize("equalize", [False, True])
def test_slice_acq_adjoint_vs_oracle(device, masks, equalize):
    """A^T as a gather over voxels vs the oracle's scatter restatement (same pixel activity rule weight >= 0.5)."""
    from nesvor_amd import slice_acq_cuda as K
    from oracle import slice_acq as O

    vol, psf, tf, vm, sm = _sa_setup(masks)
    y = torch.rand(6, 1, 14, 12)
    ref, wref = O.slice_acquisition_adjoint_forward(tf, psf, y, sm, vm, (18, 20, 22), 1.5, False, equalize)
    e = torch.empty(0, device=device)
    got, wgot = K.adjoint_forward(tf.to(device), psf.to(device), y.to(device), e if sm is None else sm.to(device),
                                  e if vm is None else vm.to(device), (18, 20, 22), 1.5, False, equalize)
    torch.testing.assert_close(got.cpu(), ref, rtol=1e-4, atol=1e-5)
    if equalize:
        torch.testing.assert_close(wgot.cpu(), wref, rtol=1e-4, atol=1e-5)
    else:
        assert wgot.numel() == 0


def test_slice_acq_adjointness_gpu(device):
    """<A x, y> == <x, A^T y> over the pixels the adjoint keeps (PSF weight >= 0.5), forward and adjoint both HIP."""
    from nesvor_amd import slice_acq_cuda as K
    from oracle import slice_acq as O

    vol, psf, tf, _, _ = _sa_setup(False, seed=3)
    y = torch.rand(6, 1, 14, 12)
    R, q, c = O._geometry(tf, (18, 20, 22), (14, 12), 1.5, torch.float32)
    keep = (O._psf_weight(R, c, psf, (18, 20, 22)) >= 0.5).view(6, 1, 14, 12)
    e = torch.empty(0, device=device)
    Ax = K.forward(tf.to(device), vol.to(device), e, e, psf.to(device), (14, 12), 1.5, False, False)[0].cpu()
    Aty = K.adjoint_forward(tf.to(device), psf.to(device), (y * keep).to(device), e, e, (18, 20, 22), 1.5, False, False)[0].cpu()
    lhs, rhs = float((Ax * y * keep).double().sum()), float((vol * Aty).double().sum())
    assert abs(lhs - rhs) <= 1e-4 * abs(lhs)


@pytest.mark.parametrize("masks", [False, True])
def test_slice_acq_backward_vs_oracle(device, masks):
    from nesvor_amd import slice_acq_cuda as K
    from oracle import slice_acq as O

    vol, psf, tf, vm, sm = _sa_setup(masks, seed=5)
    g = torch.randn(6, 1, 14, 12)
    g[0, 0, :3] = 0  # exact zeros are skipped by the reference
    gv_ref, gt_ref = O.slice_acquisition_backward(tf, vol, vm, psf, g, sm, 1.5)
    e = torch.empty(0, device=device)
    gv, gt = K.backward(tf.to(device), vol.to(device), e if vm is None else vm.to(device), psf.to(device), g.to(device),
                        e if sm is None else sm.to(device), 1.5, False, True, True)
    torch.testing.assert_close(gv.cpu(), gv_ref, rtol=1e-4, atol=1e-5)
    scale = float(gt_ref.abs().max())
    assert float((gt.cpu() - gt_ref).abs().max()) <= 2e-4 * scale
    only_t = K.backward(tf.to(device), vol.to(device), e, psf.to(device), g.to(device), e, 1.5, False, False, True)
    assert only_t[0] is None and only_t[1] is not None
    # autograd through the public wrapper
    from nesvor_amd.slice_acquisition import slice_acquisition

    v = vol.to(device).requires_grad_(True)
    t = tf.to(device).requires_grad_(True)
    out = slice_acquisition(t, v, None, None, psf.to(device), (14, 12), 1.5, False, False)
    (out * g.to(device)).sum().backward()
    gv2, gt2 = O.slice_acquisition_backward(tf, vol, None, psf, g, None, 1.5)
    torch.testing.assert_close(v.grad.cpu(), gv2, rtol=1e-4, atol=1e-5)
    assert float((t.grad.cpu() - gt2).abs().max()) <= 2e-4 * float(gt2.abs().max())


@pytest.mark.parametrize("masks", [False, True])
def test_slice_acq_double_precision_all_four_entry_points(device, masks):
    """The reference dispatches its four slice-acquisition kernels for float AND double (AT_DISPATCH_FLOATING_TYPES,
    slice_acq_cuda_kernel.cu:970, 1010, 1046, 1114).  The *_f64 entry points against the oracle evaluated in float64:
    operators to 1e-10, gradients to 1e-9 of their range; autograd through the public wrappers in double; a large PSF
    (more taps than the forward's LDS tap list holds) and the no-fallback guard for other dtypes."""
    from nesvor_amd import slice_acq_cuda as K
    from nesvor_amd.slice_acquisition import slice_acquisition, slice_acquisition_adjoint
    from nesvor_amd.utils import get_PSF
    from oracle import slice_acq as O

    vol, psf, tf, vm, sm = _sa_setup(masks, seed=11)
    vol, psf, tf = vol.double(), psf.double(), tf.double()
    dims = (18, 20, 22)
    e = torch.empty(0, device=device)
    d = lambda t: e if t is None else t.to(device)
    # A
    ref, wref = O.slice_acquisition_forward(tf, vol, vm, sm, psf, (14, 12), 1.5, True, False)
    got, wgot = K.forward(d(tf), d(vol), d(vm), d(sm), d(psf), (14, 12), 1.5, True, False)
    assert got.dtype == torch.float64
    torch.testing.assert_close(got.cpu(), ref, rtol=1e-10, atol=1e-12)
    torch.testing.assert_close(wgot.cpu(), wref, rtol=1e-10, atol=1e-12)
    # backward of A
    g = torch.randn(6, 1, 14, 12, dtype=torch.float64)
    gv_ref, gt_ref = O.slice_acquisition_backward(tf, vol, vm, psf, g, sm, 1.5)
    gv, gt = K.backward(d(tf), d(vol), d(vm), d(psf), d(g), d(sm), 1.5, False, True, True)
    torch.testing.assert_close(gv.cpu(), gv_ref, rtol=1e-9, atol=1e-11)
    assert float((gt.cpu() - gt_ref).abs().max()) <= 1e-9 * float(gt_ref.abs().max())
    # A^T (+ equalisation) and its backward
    y = torch.rand(6, 1, 14, 12, dtype=torch.float64)
    G = torch.randn(1, 1, *dims, dtype=torch.float64)
    for equalize in (False, True):
        v_ref, w_ref = O.slice_acquisition_adjoint_forward(tf, psf, y, sm, vm, dims, 1.5, False, equalize)
        v, w = K.adjoint_forward(d(tf), d(psf), d(y), d(sm), d(vm), dims, 1.5, False, equalize)
        torch.testing.assert_close(v.cpu(), v_ref, rtol=1e-10, atol=1e-12)
        gs_ref, gt2_ref = O.slice_acquisition_adjoint_backward(tf, G, w_ref if equalize else None, vm, psf, y, sm,
                                                               v_ref if equalize else None, 1.5, False, equalize)
        gs, gt2 = K.adjoint_backward(d(tf), d(G).clone(), d(w_ref) if equalize else None, d(vm), d(psf), d(y), d(sm),
                                     d(v_ref) if equalize else None, 1.5, False, equalize, True, True)
        torch.testing.assert_close(gs.cpu(), gs_ref, rtol=1e-9, atol=1e-11)
        assert float((gt2.cpu() - gt2_ref).abs().max()) <= 1e-9 * float(gt2_ref.abs().max())
    # autograd through the wrappers, in double
    t = tf.to(device).requires_grad_(True)
    vv = vol.to(device).requires_grad_(True)
    out = slice_acquisition(t, vv, None if vm is None else vm.to(device), None if sm is None else sm.to(device), psf.to(device),
                            (14, 12), 1.5, False, False)
    (out * g.to(device)).sum().backward()
    torch.testing.assert_close(vv.grad.cpu(), gv_ref, rtol=1e-9, atol=1e-11)
    back = slice_acquisition_adjoint(tf.to(device), psf.to(device), y.to(device), None, None, dims, 1.5, False, False)
    assert back.dtype == torch.float64 and back.shape == (1, 1) + dims
    # a PSF of 1813 taps (6 mm slices on a 0.5 mm grid): beyond the forward's LDS tap list, walked in global memory
    big = get_PSF(res_ratio=(3.0, 3.0, 12.0)).double()
    assert big.numel() > 1024
    ref_b = O.slice_acquisition_forward(tf, vol, None, None, big, (14, 12), 1.5, False, False)
    got_b = K.forward(d(tf), d(vol), e, e, d(big), (14, 12), 1.5, False, False)[0]
    torch.testing.assert_close(got_b.cpu(), ref_b, rtol=1e-10, atol=1e-12)
    got_b32 = K.forward(d(tf.float()), d(vol.float()), e, e, d(big.float()), (14, 12), 1.5, False, False)[0]
    torch.testing.assert_close(got_b32.cpu().double(), ref_b, rtol=1e-4, atol=1e-5)
    with pytest.raises(NotImplementedError):
        K.forward(d(tf.half()), d(vol.half()), e, e, d(psf.half()), (14, 12), 1.5, False, False)


@pytest.mark.parametrize("masks", [False, True])
@pytest.mark.parametrize("equalize", [False, True])
def test_slice_acq_adjoint_backward_vs_oracle(device, masks, equalize):
    """Backward of A^T (gather per pixel, per-slice reduction instead of the reference's atomics) vs the oracle,
    including the in-place equalisation of grad_vol; then through the autograd wrapper."""
    from nesvor_amd import slice_acq_cuda as K
    from nesvor_amd.slice_acquisition import slice_acquisition_adjoint
    from oracle import slice_acq as O

    vol, psf, tf, vm, sm = _sa_setup(masks, seed=7)
    dims = (18, 20, 22)
    y = torch.rand(6, 1, 14, 12)
    G = torch.randn(1, 1, *dims)
    v_ref, w_ref = O.slice_acquisition_adjoint_forward(tf, psf, y, sm, vm, dims, 1.5, False, equalize)
    gs_ref, gt_ref = O.slice_acquisition_adjoint_backward(tf, G, w_ref if equalize else None, vm, psf, y, sm,
                                                          v_ref if equalize else None, 1.5, False, equalize)
    e = torch.empty(0, device=device)
    d = lambda t: e if t is None else t.to(device)
    G_dev = G.to(device).clone()
    gs, gt = K.adjoint_backward(tf.to(device), G_dev, d(w_ref) if equalize else None, d(vm), psf.to(device), y.to(device), d(sm),
                                d(v_ref) if equalize else None, 1.5, False, equalize, True, True)
    torch.testing.assert_close(gs.cpu(), gs_ref, rtol=2e-4, atol=2e-5 * float(gs_ref.abs().max()))
    assert float((gt.cpu() - gt_ref).abs().max()) <= 3e-4 * float(gt_ref.abs().max())
    if equalize:  # grad_vol was equalised in place, like the reference
        pos = w_ref > 0
        torch.testing.assert_close(G_dev.cpu()[pos], (G / w_ref.clamp(min=1e-3))[pos], rtol=1e-5, atol=1e-6)
    # autograd wrapper: d/d(slices), d/d(transforms) of <A^T y, G>
    t = tf.to(device).requires_grad_(True)
    yy = y.to(device).requires_grad_(True)
    out = slice_acquisition_adjoint(t, psf.to(device), yy, None if sm is None else sm.to(device),
                                    None if vm is None else vm.to(device), dims, 1.5, False, equalize)
    (out * G.to(device)).sum().backward()
    torch.testing.assert_close(yy.grad.cpu(), gs_ref, rtol=2e-4, atol=2e-5 * float(gs_ref.abs().max()))
    assert float((t.grad.cpu() - gt_ref).abs().max()) <= 3e-4 * float(gt_ref.abs().max())


@pytest.mark.parametrize("masks", [False, True])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_slice_acq_interp_psf_backward_and_adjoint_vs_oracle(device, masks, dtype):
    """interp_psf = True (nearest voxel + re-interpolated PSF) for the backward of A, A^T and the backward of A^T
    (slice_acq_cuda_kernel.cu:229-370, :526-606, :754-836) - the `*_interp` entry points - against the oracle's restatement.
    float64: same arithmetic, only the order of the atomic sums differs (1e-9).  float32: a tap whose position lies within
    rounding of a half-integer may pick the neighbouring voxel on one side only, so the comparison allows a small fraction
    of outliers.  Then autograd through the public wrappers."""
    from nesvor_amd import slice_acq_cuda as K
    from nesvor_amd.slice_acquisition import slice_acquisition, slice_acquisition_adjoint
    from oracle import slice_acq as O

    vol, psf, tf, vm, sm = _sa_setup(masks, seed=13)
    vol, psf, tf = vol.to(dtype), psf.to(dtype), tf.to(dtype)
    dims = (18, 20, 22)
    e = torch.empty(0, device=device)
    d = lambda t: e if t is None else t.to(device)
    f64 = dtype == torch.float64

    def close(got, ref, name):
        got, ref = got.cpu().double(), ref.double()
        scale = float(ref.abs().max())
        if f64:
            assert float((got - ref).abs().max()) <= 1e-9 * scale, name
        else:
            bad = (got - ref).abs() > 2e-4 * scale
            assert float(bad.double().mean()) <= 5e-3, (name, float(bad.double().mean()))

    g = torch.randn(6, 1, 14, 12, dtype=dtype)
    g[0, 0, :3] = 0
    gv_ref, gt_ref = O.slice_acquisition_backward(tf, vol, vm, psf, g, sm, 1.5, interp_psf=True)
    gv, gt = K.backward(d(tf), d(vol), d(vm), d(psf), d(g), d(sm), 1.5, True, True, True)
    assert float(gv_ref.abs().max()) > 0 and float(gt_ref.abs().max()) > 0
    close(gv, gv_ref, "grad_vol")
    close(gt, gt_ref, "grad_transforms")
    only_t = K.backward(d(tf), d(vol), e, d(psf), d(g), e, 1.5, True, False, True)
    assert only_t[0] is None and only_t[1] is not None
    y = torch.rand(6, 1, 14, 12, dtype=dtype)
    G = torch.randn(1, 1, *dims, dtype=dtype)
    for equalize in (False, True):
        v_ref, w_ref = O.slice_acquisition_adjoint_forward(tf, psf, y, sm, vm, dims, 1.5, True, equalize)
        v, w = K.adjoint_forward(d(tf), d(psf), d(y), d(sm), d(vm), dims, 1.5, True, equalize)
        assert float(v_ref.abs().max()) > 0
        close(v, v_ref, "adjoint")
        if equalize:
            close(w, w_ref, "adjoint weight")
        gs_ref, gt2_ref = O.slice_acquisition_adjoint_backward(tf, G, w_ref if equalize else None, vm, psf, y, sm,
                                                               v_ref if equalize else None, 1.5, True, equalize)
        G_dev = d(G).clone()
        gs, gt2 = K.adjoint_backward(d(tf), G_dev, d(w_ref) if equalize else None, d(vm), d(psf), d(y), d(sm),
                                     d(v_ref) if equalize else None, 1.5, True, equalize, True, True)
        close(gs, gs_ref, "grad_slices")
        close(gt2, gt2_ref, "adjoint grad_transforms")
        if equalize:
            pos = w_ref > 0
            torch.testing.assert_close(G_dev.cpu()[pos], (G / w_ref.clamp(min=1e-3))[pos], rtol=1e-5, atol=1e-6)
    # autograd through the wrappers
    t = tf.to(device).requires_grad_(True)
    vv = vol.to(device).requires_grad_(True)
    out = slice_acquisition(t, vv, None if vm is None else vm.to(device), None if sm is None else sm.to(device), psf.to(device),
                            (14, 12), 1.5, False, True)
    (out * g.to(device)).sum().backward()
    close(vv.grad, gv_ref, "autograd grad_vol")
    close(t.grad, gt_ref, "autograd grad_transforms")
    yy = y.to(device).requires_grad_(True)
    out = slice_acquisition_adjoint(tf.to(device), psf.to(device), yy, None if sm is None else sm.to(device),
                                    None if vm is None else vm.to(device), dims, 1.5, True, False)
    (out * G.to(device)).sum().backward()
    gs_ref, _ = O.slice_acquisition_adjoint_backward(tf, G, None, vm, psf, y, sm, None, 1.5, True, False)
    close(yy.grad, gs_ref, "autograd grad_slices")


def test_cg_recon_reference_test(device):
    """tests/slice_acquisition/test_slice_acq.py:13-81: 16 stacks x 22 slices x 40x40 simulated from the 32^3
    phantom; SRR(n_iter=20, use_CG=True, tol=1e-8) started from the true volume must return it (atol 3e-5)."""
    from nesvor_amd.phantom import STACK_ANGLES, phantom3d, stack_geometry, stack_transforms
    from nesvor_amd.slice_acquisition import slice_acquisition
    from nesvor_amd.srr import SRR
    from nesvor_amd.transform import RigidTransform, mat_update_resolution
    from nesvor_amd.utils import get_PSF

    vs, gap, res, res_s = 32, 3, 1, 1.5
    n_slice, ss = stack_geometry(vs, res, res_s, gap)
    assert (n_slice, ss) == (22, 40)
    volume = torch.tensor(phantom3d(n=vs), dtype=torch.float32, device=device)[None, None]
    psf = get_PSF(res_ratio=(res_s / res, res_s / res, gap / res), device=device)
    stacks, tfs = [], []
    for ang in STACK_ANGLES:
        tf = stack_transforms(ang, n_slice, gap, device)
        mat = mat_update_resolution(tf.matrix(), 1, res)
        stacks.append(slice_acquisition(mat, volume, None, None, psf, (ss, ss), res_s / res, False, False))
        tfs.append(tf)
    slices, transforms = torch.cat(stacks, 0), RigidTransform.cat(tfs)
    params = {"psf": psf, "slice_shape": (ss, ss), "res_s": res_s, "res_r": res, "interp_psf": False, "volume_shape": (vs, vs, vs)}
    theta = mat_update_resolution(transforms.matrix(), 1, res)
    out = SRR(n_iter=20, use_CG=True, tol=1e-8)(theta, slices, volume, params)
    torch.testing.assert_close(out, volume, atol=3e-5, rtol=1e-5)
    # and from a perturbed start CG must move back towards the phantom
    start = (volume + 0.05 * torch.randn_like(volume)).clamp(min=0)
    out2 = SRR(n_iter=10, use_CG=True, tol=0.0)(theta, slices, start.clone(), params)
    assert float((out2 - volume).abs().mean()) < float((start - volume).abs().mean())


# --------------------------------------------------------------------- hash grid
def _psf_cloud(n_pix, S, seed):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(n_pix, 1, 3, generator=g) * 110 + 10
    x = c + torch.randn(n_pix, S, 3, generator=g) * torch.tensor([0.77, 0.77, 1.27])
    return (x.reshape(-1, 3) / 130.0).clamp(0, 1).contiguous()


@pytest.mark.parametrize("method", ["owner", "atomic"])
@pytest.mark.parametrize("F", [1, 2, 4, 8])
@pytest.mark.parametrize("layout", [0, 1])
def test_hashgrid_fwd_bwd_vs_oracle(device, F, layout, method):
    from nesvor_amd.encoding import hashgrid_backward, hashgrid_forward
    from nesvor_amd.grid import HashGridSpec
    from oracle import hashgrid as O

    spec = HashGridSpec(8, F, 10, 5, 1.5)  # dense and hashed levels, tables small enough to collide a lot
    lv = O.make_levels(8, 10, 5, 1.5)
    assert any(l.hashed for l in lv) and not all(l.hashed for l in lv)
    torch.manual_seed(F * 10 + layout)
    N = 3000  # ragged: not a multiple of the 256-thread block
    u = torch.cat([torch.rand(N - 512, 3), _psf_cloud(2, 256, 1)])
    u[0] = 0.0
    u[1] = 1.0  # the u == 1 face wraps the dense index (tcnn semantics)
    table = torch.randn(spec.n_params)
    dy = torch.randn(N, spec.n_output_dims)
    ref = O.encode(u, table, lv, F)
    gt_ref, gu_ref = O.encode_backward(u, table, lv, F, dy)
    pe_dev = hashgrid_forward(spec, u.to(device), table.to(device), layout)
    # the one-workgroup-per-cloud kernel (taken for clustered batches) computes the same numbers
    assert torch.equal(hashgrid_forward(spec, u.to(device), table.to(device), layout, clustered=True), pe_dev)
    pe = pe_dev.cpu()
    pe = pe if layout == 0 else pe.t()
    # fp32, 8-term interpolation: same pos (fma) and indices; only summation order differs
    torch.testing.assert_close(pe, ref, rtol=1e-5, atol=1e-5)
    dyk = (dy if layout == 0 else dy.t().contiguous()).to(device)
    gt, gu = hashgrid_backward(spec, u.to(device), table.to(device), dyk, None, True, layout, method)
    # atomics: order-nondeterministic fp32 sums of up to ~N/8 terms
    torch.testing.assert_close(gt.cpu(), gt_ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gu.cpu(), gu_ref, rtol=1e-4, atol=2e-3)
    # accumulate semantics + no input grad
    gt2, none = hashgrid_backward(spec, u.to(device), table.to(device), dyk, gt.clone(), False, layout, method)
    assert none is None
    torch.testing.assert_close(gt2.cpu(), 2 * gt_ref, rtol=1e-4, atol=2e-4)
    # the unclustered hint (points ordered by coarse lattice cell before the aggregation pass) changes nothing but the order of sums
    gt3, gu3 = hashgrid_backward(spec, u.to(device), table.to(device), dyk, None, True, layout, method, clustered=False)
    torch.testing.assert_close(gt3.cpu(), gt_ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gu3.cpu(), gu_ref, rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("method", ["owner", "atomic"])
def test_hashgrid_headline_config_vs_oracle(device, method, layout):
    """L=16, F=2, T=2^19, base 9, scale 1.26 on both mandated distributions, oracle-sized N, in the
    row-major layout of tinycudann (0) and in the feature-major layout the training step uses (1)."""
    from nesvor_amd.encoding import hashgrid_backward, hashgrid_forward
    from nesvor_amd.grid import HashGridSpec
    from oracle import hashgrid as O

    spec = HashGridSpec(16, 2, 19, 9, 1.26)
    lv = O.make_levels(16, 19, 9, 1.26)
    g = torch.Generator().manual_seed(1337)
    table = (torch.rand(spec.n_params, generator=g) * 2 - 1) * 1e-4
    for name, u in (("U", torch.rand(8192, 3, generator=torch.Generator().manual_seed(0))), ("P", _psf_cloud(32, 256, 0))):
        dy = torch.randn(u.shape[0], 32, generator=torch.Generator().manual_seed(1))
        ref = O.encode(u, table, lv, 2)
        for clustered in (False, True):
            pe = hashgrid_forward(spec, u.to(device), table.to(device), layout, clustered=clustered).cpu()
            torch.testing.assert_close(pe if layout == 0 else pe.t(), ref, rtol=1e-5, atol=1e-9, msg=name)
        gt_ref, gu_ref = O.encode_backward(u, table, lv, 2, dy)
        dyk = (dy if layout == 0 else dy.t().contiguous()).to(device)
        gt, gu = hashgrid_backward(spec, u.to(device), table.to(device), dyk, None, True, layout, method)
        torch.testing.assert_close(gt.cpu(), gt_ref, rtol=1e-4, atol=1e-4, msg=name)
        torch.testing.assert_close(gu.cpu(), gu_ref, rtol=1e-3, atol=1e-5, msg=name)


@pytest.mark.parametrize("F", [1, 2, 4, 8])
@pytest.mark.parametrize("layout", [0, 1])
def test_hashgrid_unclustered_forward_equals_the_other_kernels(device, F, layout, monkeypatch):
    """Round 6 (verdict: the forward had no variant for unclustered input): ``nesvor_hashgrid_forward_unclustered`` orders the points
    by coarse lattice cell, runs the per-cloud kernel on workgroups of neighbouring points and - feature-major - turns the encoded
    rows back into columns.  Same arithmetic per point: the result must EQUAL the per-level kernel's on the points as given, bit
    for bit, for ragged N, on uniform points, on clouds, and on a batch that sits in ONE coarse cell (every strip overflows: the
    spill list), and agree with the oracle."""
    import ctypes

    from nesvor_amd import _lib, encoding
    from nesvor_amd.grid import HashGridSpec
    from oracle import hashgrid as O

    spec = HashGridSpec(8, F, 12, 5, 1.6)
    lv = O.make_levels(8, 12, 5, 1.6)
    assert any(l.hashed for l in lv) and not all(l.hashed for l in lv)
    N = 70001
    assert N >= encoding.UNCLUSTERED_FWD_MIN_POINTS
    g = torch.Generator().manual_seed(F + 10 * layout)
    table = torch.randn(spec.n_params, generator=g).to(device)
    E = spec.n_output_dims

    def level_kernel(u):  # one block per (256 points, level), no hints: the pre-existing path
        pe = torch.empty((N, E) if layout == 0 else (E, N), dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            err = _lib.load().nesvor_hashgrid_forward(ctypes.byref(spec.c_struct), _lib.ptr(u), _lib.ptr(table), _lib.ptr(pe), N, layout,
                                                      _lib.stream_ptr())
        assert err == 0
        return pe

    uniform = torch.rand(N, 3, generator=g)
    uniform[0], uniform[1] = 0.0, 1.0
    one_cell = 0.4 + 0.01 * torch.rand(N, 3, generator=g)
    mixed = torch.cat([_psf_cloud(128, 256, 7), torch.rand(N - 128 * 256, 3, generator=g)])[torch.randperm(N, generator=g)]
    for name, u in (("uniform", uniform), ("one_cell", one_cell), ("mixed", mixed)):
        ud = u.contiguous().to(device)
        ref = level_kernel(ud)
        monkeypatch.setattr(encoding, "_FWD_MODE", "sorted")  # (the policy takes the ordered path for row-major output only)
        got = encoding.hashgrid_forward(spec, ud, table, layout, clustered=False)
        monkeypatch.setattr(encoding, "_FWD_MODE", "")
        assert torch.equal(got, ref), name
        assert torch.equal(encoding.hashgrid_forward(spec, ud, table, layout, clustered=False), ref), name
        assert torch.equal(encoding.hashgrid_forward(spec, ud, table, layout, clustered=True), ref), name
        if name == "uniform":
            sub = slice(0, 4096)
            o = O.encode(u[sub], table.cpu(), lv, F)
            torch.testing.assert_close((got if layout == 0 else got.t())[sub].cpu(), o, rtol=1e-5, atol=1e-5)


def test_hashgrid_full_size_properties(device):
    """N = 2^20 (BASELINE size): size-independent invariants instead of the oracle.
    * constant table -> every feature equals the constant (corner weights sum to 1), zero input grad
    * dy = 1 -> grad_table sums to N per level-feature (checksum of the scatter), linear in dy."""
    from nesvor_amd.encoding import hashgrid_backward, hashgrid_forward
    from nesvor_amd.grid import HashGridSpec

    spec = HashGridSpec(16, 2, 19, 9, 1.26)
    N = 1 << 20
    u = _psf_cloud(4096, 256, 3).to(device)
    table = torch.full((spec.n_params,), 0.75, device=device)
    for clustered in (False, True):
        pe = hashgrid_forward(spec, u, table, 1, clustered=clustered)
        torch.testing.assert_close(pe, torch.full_like(pe, 0.75), rtol=1e-6, atol=1e-6)
    dy = torch.ones(32, N, device=device)
    gt, gu = hashgrid_backward(spec, u, table, dy, None, True, 1)
    assert float(gu.abs().max()) < 1e-3
    for li, lv in enumerate(spec.levels):
        seg = gt[lv.offset * 2 : (lv.offset + lv.size) * 2].view(-1, 2).double().sum(0)
        assert abs(float(seg[0]) - N) < N * 1e-4 and abs(float(seg[1]) - N) < N * 1e-4, li
    table = torch.randn(spec.n_params, device=device) * 0.1
    dy1, dy2 = torch.randn(32, N, device=device), torch.randn(32, N, device=device)
    g1, _ = hashgrid_backward(spec, u, table, dy1, None, False, 1)
    g2, _ = hashgrid_backward(spec, u, table, dy2, None, False, 1)
    g3, _ = hashgrid_backward(spec, u, table, 2 * dy1 - dy2, None, False, 1)
    err = (g3 - (2 * g1 - g2)).abs().max() / g3.abs().max()
    assert float(err) < 1e-4
    # <pe, dy> == <table, grad_table> (adjointness of the linear map table -> pe)
    pe = hashgrid_forward(spec, u, table, 1)
    lhs = (pe.double() * dy1.double()).sum()
    rhs = (table.double() * g1.double()).sum()
    assert abs(float(lhs - rhs)) < 1e-4 * abs(float(lhs)) + 1e-3
    # the same two properties on UNIFORM points at full size through the unclustered pair (ordered forward, ordered backward)
    uu = torch.rand(N, 3, generator=torch.Generator().manual_seed(0)).to(device)
    const = torch.full((spec.n_params,), 0.75, device=device)
    from nesvor_amd import encoding as _enc

    pe_by_mode = {}
    for mode in ("", "sorted"):  # the per-level kernel with paired corner requests / the ordered forward
        _enc._FWD_MODE = mode
        try:
            pc = hashgrid_forward(spec, uu, const, 1, clustered=False)
            torch.testing.assert_close(pc, torch.full_like(pc, 0.75), rtol=1e-6, atol=1e-6)
            pe_by_mode[mode] = hashgrid_forward(spec, uu, table, 1, clustered=False)
        finally:
            _enc._FWD_MODE = ""
    assert torch.equal(pe_by_mode["sorted"], pe_by_mode[""])
    pe = pe_by_mode[""]
    gu_, _ = hashgrid_backward(spec, uu, table, dy1, None, False, 1, clustered=False)
    lhs = (pe.double() * dy1.double()).sum()
    rhs = (table.double() * gu_.double()).sum()
    assert abs(float(lhs - rhs)) < 1e-4 * abs(float(lhs)) + 1e-3


@pytest.mark.parametrize("slack", [1.0, 1000.0])
def test_hashgrid_backward_with_producer_bound(device, slack):
    """``dy_bound`` (max |dy| handed over by the producer of dy - in the training step the density network's backward,
    csrc/step.hip) replaces the aggregation pass's own pass over dy; it only sets the scale of the 64-bit fixed-point sums,
    so the gradients must agree with the self-scaled launch to fp32 accuracy - also for a bound 1000x too large - and
    per-workgroup gradients far below the global maximum must keep their relative accuracy."""
    from nesvor_amd.encoding import hashgrid_backward
    from nesvor_amd.grid import HashGridSpec
    from nesvor_amd.mlp import backward_raw, forward_raw

    spec = HashGridSpec(16, 2, 19, 9, 1.26)
    N = 1 << 18
    u = _psf_cloud(N // 256, 256, 5).to(device)
    table = torch.randn(spec.n_params, device=device) * 0.1
    dy = torch.randn(32, N, device=device)
    dy[:, : N // 2] *= 1e-6  # half of the clouds carry gradients a million (a billion with slack) times below the bound
    bound = (dy.abs().max() * slack).reshape(1)
    g0, u0 = hashgrid_backward(spec, u, table, dy, None, True, 1)
    g1, u1 = hashgrid_backward(spec, u, table, dy, None, True, 1, dy_bound=bound)
    scale = float(g0.abs().max())
    assert float((g1 - g0).abs().max()) <= 2e-6 * scale
    torch.testing.assert_close(u1, u0, rtol=1e-5, atol=1e-6 * float(u0.abs().max()))
    # the small half alone, against an fp64 scatter of the same weights via the atomic kernel's result
    gs0, _ = hashgrid_backward(spec, u[: N // 2], table, dy[:, : N // 2].contiguous(), None, False, 1)
    gs1, _ = hashgrid_backward(spec, u[: N // 2], table, dy[:, : N // 2].contiguous(), None, False, 1, dy_bound=bound)
    assert float((gs1 - gs0).abs().max()) <= 2e-6 * float(gs0.abs().max())
    # the producer side: the fused MLP backward raises the scalar to max |dxb| exactly
    from nesvor_amd.models import build_network
    from nesvor_amd.mlp import linear_layers

    net = build_network(n_input_dims=32, n_output_dims=16, activation="ReLU", output_activation="None", n_neurons=64,
                        n_hidden_layers=2, dtype=torch.float32).to(device)
    W, Bs = [l.weight.detach() for l in linear_layers(net)], [l.bias.detach() for l in linear_layers(net)]
    xb, dz = torch.randn(32, N, device=device), torch.randn(16, N, device=device)
    _, saved = forward_raw(W, Bs, None, xb, 0, 32, 256, True)
    dxb, mx = torch.empty(32, N, device=device), torch.zeros(1, device=device)
    backward_raw(W, Bs, None, xb, dz, saved, 0, 32, 256, dxb, False, dxb_absmax=mx)
    assert float(mx) == float(dxb.abs().max())


@pytest.mark.parametrize("points", ["psf", "uniform", "overflow"])
def test_hashgrid_backward_with_adamw_in_the_owner_pass(device, points, monkeypatch):
    """nesvor_hashgrid_backward_adamw == nesvor_hashgrid_backward followed by nesvor_adamw_step(zero_grad=1) on the table:
    parameters and both moments after three steps (a chunk that receives no record still decays), grad_table zero afterwards,
    a gradient already in grad_table is included.  PSF clouds: most chunks' records fit one slice (update straight from LDS);
    uniform points: long queues, several slices per chunk (atomic adds + the last-ticket slice updates); "overflow": queues
    shrunk until records take the exact fallback (global atomics into grad_table during the aggregation pass)."""
    from nesvor_amd import _lib
    from nesvor_amd.encoding import hashgrid_backward, hashgrid_backward_adamw
    from nesvor_amd.grid import HashGridSpec

    if points == "overflow":
        monkeypatch.setenv("NESVOR_HASHGRID_CAP_SCALE", "0.002")
    spec = HashGridSpec(16, 2, 19, 9, 1.26)
    N = 1 << 18
    g = torch.Generator().manual_seed(3)
    table0 = (torch.randn(spec.n_params, generator=g) * 0.1).to(device)
    pre = torch.zeros_like(table0)
    pre[::1000] = 0.25  # a gradient left in grad_table by an earlier backward
    lib = _lib.load()
    state = {k: (table0.clone(), pre.clone(), torch.zeros_like(table0), torch.zeros_like(table0)) for k in ("two_calls", "fused")}
    lr, b1, b2, eps, wd = 5e-3, 0.9, 0.99, 1e-15, 1e-2
    for t in range(1, 4):
        u = (_psf_cloud(N // 256, 256, 5 + t) if points == "psf" else torch.rand(N, 3, generator=g)).to(device)
        dy = torch.randn(32, N, generator=g).to(device)
        adam = _lib.AdamwT(lr, b1, b2, eps, wd, 1 - b1 ** t, 1 - b2 ** t, 1.0)
        p, gt, m, v = state["two_calls"]
        _, gu0 = hashgrid_backward(spec, u, p, dy, gt, True, 1)
        _lib.check(lib.nesvor_adamw_step(_lib.ptr(p), _lib.ptr(gt), _lib.ptr(m), _lib.ptr(v), p.numel(), lr, b1, b2, eps, wd,
                                         1 - b1 ** t, 1 - b2 ** t, 1.0, 1, _lib.stream_ptr()), "adamw")
        p, gt, m, v = state["fused"]
        gu1 = hashgrid_backward_adamw(spec, u, p, dy, gt, m, v, adam, True, 1)
        assert int(torch.count_nonzero(gt)) == 0
        torch.testing.assert_close(gu1, gu0, rtol=1e-5, atol=1e-6 * float(gu0.abs().max()))
    for i, name in ((0, "param"), (2, "exp_avg"), (3, "exp_avg_sq")):
        a, b = state["two_calls"][i], state["fused"][i]
        scale = float(a.abs().max())
        # (m / (sqrt(v) + 1e-15) turns a last-bit difference of a tiny gradient into a visible step: compare the bulk tightly
        # and bound the rest by the step size)
        d = (a - b).abs()
        frac = float((d > 1e-6 * scale).float().mean())
        assert frac < 1e-4, (name, frac)  # (one chunk left out would be 5e-4)
        assert float(d.max()) <= (3 * lr if name == "param" else 1e-4 * scale), (name, float(d.max()), scale)
    assert float((state["fused"][0] - table0).abs().max()) > 1e-3  # (it did train)


def test_hashgrid_owner_equals_atomic_full_size_uniform(device):
    """Uniform points defeat the per-cloud aggregation and fill the chunk queues to (and past) their
    capacity: the queue-overflow fallback must keep the result exact."""
    from nesvor_amd.encoding import hashgrid_backward
    from nesvor_amd.grid import HashGridSpec

    spec = HashGridSpec(16, 2, 19, 9, 1.26)
    N = (1 << 20) - 77  # ragged
    g = torch.Generator().manual_seed(0)
    u = torch.rand(N, 3, generator=g).to(device)
    table = (torch.randn(spec.n_params, generator=g) * 0.1).to(device)
    dy = torch.randn(N, 32, generator=g).to(device)
    g_own, gu_own = hashgrid_backward(spec, u, table, dy, None, True, 0, "owner")
    g_atm, gu_atm = hashgrid_backward(spec, u, table, dy, None, True, 0, "atomic")
    scale = float(g_atm.abs().max())
    assert float((g_own - g_atm).abs().max()) < 2e-4 * scale
    torch.testing.assert_close(gu_own, gu_atm, rtol=1e-3, atol=1e-3)
    # ... and the variant made for this distribution (clustered=False: counting sort by coarse cell, then the same two passes on
    # workgroups of neighbouring points), in both layouts, three times over (the sort's order inside a cell is arrival order)
    for layout, d in ((0, dy), (1, dy.t().contiguous())):
        for _ in range(3 if layout == 0 else 1):
            g_srt, gu_srt = hashgrid_backward(spec, u, table, d, None, True, layout, "owner", clustered=False)
            assert float((g_srt - g_atm).abs().max()) < 2e-4 * scale
            torch.testing.assert_close(gu_srt, gu_atm, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("case", ["one_point", "n255", "n257", "all_in_one_cell", "two_cells", "outside_and_nan_free"])
def test_hashgrid_unclustered_variant_edge_cases(device, case):
    """The unclustered backward orders the batch by coarse lattice cell through per-cell strips with a spill list: batches smaller
    than a workgroup, batches whose points ALL fall into one cell (every ticket beyond the strip's capacity spills), two dense
    cells, and points outside the unit cube must give the atomic kernel's gradients."""
    from nesvor_amd.encoding import hashgrid_backward
    from nesvor_amd.grid import HashGridSpec

    spec = HashGridSpec(16, 2, 19, 9, 1.26)
    g = torch.Generator().manual_seed(11)
    if case == "one_point":
        u = torch.rand(1, 3, generator=g)
    elif case == "n255":
        u = torch.rand(255, 3, generator=g)
    elif case == "n257":
        u = torch.rand(257, 3, generator=g)
    elif case == "all_in_one_cell":
        u = 0.4 + 1e-3 * torch.rand(5000, 3, generator=g)
    elif case == "two_cells":
        u = torch.cat([0.1 + 1e-3 * torch.rand(3000, 3, generator=g), 0.9 + 1e-3 * torch.rand(3000, 3, generator=g)])[torch.randperm(6000, generator=g)]
    else:
        u = torch.rand(4000, 3, generator=g) * 1.2 - 0.1  # a tenth of the cube's side beyond every face
    u = u.contiguous().to(device)
    N = u.shape[0]
    table = (torch.randn(spec.n_params, generator=g) * 0.1).to(device)
    for layout in (0, 1):
        dy = torch.randn((N, 32) if layout == 0 else (32, N), generator=g).to(device)
        g_atm, gu_atm = hashgrid_backward(spec, u, table, dy, None, True, layout, "atomic")
        g_srt, gu_srt = hashgrid_backward(spec, u, table, dy, None, True, layout, "owner", clustered=False)
        scale = float(g_atm.abs().max())
        assert float((g_srt - g_atm).abs().max()) <= 2e-4 * scale + 1e-12, case
        torch.testing.assert_close(gu_srt, gu_atm, rtol=1e-3, atol=1e-3 * float(gu_atm.abs().max()) + 1e-9)


def test_hashgrid_queue_sizer_grows_only_what_overflows(device, monkeypatch):
    """The record queues start at 1/16 of the worst case per level (encoding.QueueSizer); a level whose overflow counter
    the kernels raise is grown x4 before a later backward.  Every backward on the way is exact (overflowing records take
    the atomic path).  PSF clouds: the coarse levels never grow and the workspace settles far below the worst case;
    uniform points: the fine levels reach the worst case."""
    from nesvor_amd import encoding
    from nesvor_amd.encoding import QueueSizer, hashgrid_backward
    from nesvor_amd.grid import HashGridSpec

    monkeypatch.setattr(QueueSizer, "policy", "adaptive")
    spec = HashGridSpec(16, 2, 19, 9, 1.26)
    N = 1 << 18
    g = torch.Generator().manual_seed(0)
    table = (torch.randn(spec.n_params, generator=g) * 0.1).to(device)
    dy = torch.randn(32, N, generator=g).to(device)
    lib = encoding._lib.load()
    worst = lib.nesvor_hashgrid_backward_workspace_bytes(__import__("ctypes").byref(spec.c_struct), N, None)
    for dist in ("P", "U"):
        encoding._SIZERS.clear()
        encoding._WORKSPACES.clear()
        u = (_psf_cloud(N // 256, 256, 3) if dist == "P" else torch.rand(N, 3, generator=g)).to(device)
        ref, _ = hashgrid_backward(spec, u, table, dy, None, False, 1, "atomic")
        scale = float(ref.abs().max())
        sizer = encoding.queue_sizer(spec, N, device)
        assert all(abs(sizer.scale[l] - QueueSizer.START) < 1e-9 for l in range(16))
        for it in range(12):
            got, _ = hashgrid_backward(spec, u, table, dy, None, False, 1, "owner")
            torch.cuda.synchronize()  # (lets every queued counter copy finish, so that each call can act on the previous one)
            assert float((got - ref).abs().max()) < 2e-4 * scale, (dist, it)
        scales = [round(sizer.scale[l], 4) for l in range(16)]
        nbytes = next(k[1] for k in encoding._WORKSPACES)
        print(dist, scales, f"{nbytes / worst:.3f} of the worst-case workspace")
        if dist == "P":
            assert all(s == round(QueueSizer.START, 4) for s in scales[:8]) and nbytes < 0.4 * worst
        else:
            assert all(s == 1.0 for s in scales[8:]) and nbytes > 0.5 * worst
    encoding._SIZERS.clear()
    encoding._WORKSPACES.clear()


@pytest.mark.parametrize("dist", ["P", "U"])
def test_hashgrid_queue_overflow_fallback_is_exact(device, dist, monkeypatch):
    """Queue capacities shrunk to 2 % (test knob NESVOR_HASHGRID_CAP_SCALE): most records of the fine levels no longer
    fit their per-XCC sub-queue and take the atomic fallback of the aggregation pass; the gradient must not change."""
    from nesvor_amd.encoding import hashgrid_backward
    from nesvor_amd.grid import HashGridSpec

    spec = HashGridSpec(16, 2, 19, 9, 1.26)
    g = torch.Generator().manual_seed(5)
    u = (_psf_cloud(512, 256, 1) if dist == "P" else torch.rand(512 * 256, 3, generator=g)).to(device)
    N = u.shape[0]
    table = (torch.randn(spec.n_params, generator=g) * 0.1).to(device)
    dy = torch.randn(N, 32, generator=g).to(device)
    g_ref, gu_ref = hashgrid_backward(spec, u, table, dy, None, True, 0, "owner")
    monkeypatch.setenv("NESVOR_HASHGRID_CAP_SCALE", "0.02")
    g_small, gu_small = hashgrid_backward(spec, u, table, dy, None, True, 0, "owner")
    monkeypatch.delenv("NESVOR_HASHGRID_CAP_SCALE")
    scale = float(g_ref.abs().max())
    assert float((g_small - g_ref).abs().max()) < 2e-4 * scale
    torch.testing.assert_close(gu_small, gu_ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("split,clustered", [(12, True), (5, True), (15, True), (12, False), (5, False)])
def test_hashgrid_backward_split_by_levels(device, split, clustered):
    """nesvor_hashgrid_backward_levels: the fine levels [split, L) first, then [0, split) with the queue tails kept and
    the input gradient accumulated, must equal the one-launch backward (a data-parallel step overlaps the all-reduce
    of the first part with the second).  clustered=False: shuffled points through the unclustered variant - the second launch
    re-uses the first one's point order and re-ordered d pe."""
    from nesvor_amd.encoding import hashgrid_backward
    from nesvor_amd.grid import HashGridSpec

    spec = HashGridSpec(16, 2, 19, 9, 1.26)
    u = _psf_cloud(512, 256, 7).to(device)
    N = u.shape[0]
    g = torch.Generator().manual_seed(3)
    if not clustered:
        u = u[torch.randperm(N, generator=g).to(device)].contiguous()
    table = (torch.randn(spec.n_params, generator=g) * 0.1).to(device)
    dy = torch.randn(32, N, generator=g).to(device)
    g_ref, gu_ref = hashgrid_backward(spec, u, table, dy, None, True, 1, "atomic" if not clustered else "owner")
    g_a, gu = hashgrid_backward(spec, u, table, dy, None, True, 1, "owner", levels=(split, 16), clustered=clustered)
    cut = spec.levels[split].offset * 2
    tol = dict(rtol=1e-5, atol=1e-6) if clustered else dict(rtol=1e-4, atol=2e-5 * float(g_ref.abs().max()))  # (against atomics)
    assert float(g_a[:cut].abs().max()) == 0.0  # nothing of the coarse levels yet
    torch.testing.assert_close(g_a[cut:], g_ref[cut:], **tol)
    g_b, gu = hashgrid_backward(spec, u, table, dy, g_a, True, 1, "owner", levels=(0, split), grad_u=gu, first=False, clustered=clustered)
    torch.testing.assert_close(g_b, g_ref, **tol)
    torch.testing.assert_close(gu, gu_ref, rtol=1e-4, atol=1e-5 if clustered else 1e-4 * float(gu_ref.abs().max()))


def test_hashgrid_points_outside_unit_cube(device):
    """PSF samples of boundary pixels fall slightly outside the bounding box (u < 0 or u > 1, cells -1 / res): the
    forward's LDS box cache, the backward's box-addressed merge table and the per-corner atomic kernel must index the
    same entries.  Checks: owner backward == atomic backward, and <pe, dy> == <table, grad_table> (the forward and the
    backward are adjoint maps only if they agree on every index)."""
    from nesvor_amd.encoding import hashgrid_backward, hashgrid_forward
    from nesvor_amd.grid import HashGridSpec

    spec = HashGridSpec(16, 2, 19, 9, 1.26)
    g = torch.Generator().manual_seed(11)
    centres = torch.rand(256, 1, 3, generator=g)
    centres[:128] = centres[:128].round()  # half of the clouds sit on faces / edges / corners of the cube
    u = (centres + torch.randn(256, 256, 3, generator=g) * torch.tensor([0.006, 0.006, 0.01])).reshape(-1, 3).contiguous().to(device)
    assert float(u.min()) < -0.01 and float(u.max()) > 1.01
    N = u.shape[0]
    table = (torch.randn(spec.n_params, generator=g) * 0.1).to(device)
    dy = torch.randn(32, N, generator=g).to(device)
    pe = hashgrid_forward(spec, u, table, 1)
    assert torch.equal(hashgrid_forward(spec, u, table, 1, clustered=True), pe)  # per-level blocks == per-cloud workgroups
    g_own, gu_own = hashgrid_backward(spec, u, table, dy, None, True, 1, "owner")
    g_atm, gu_atm = hashgrid_backward(spec, u, table, dy, None, True, 1, "atomic")
    scale = float(g_atm.abs().max())
    assert float((g_own - g_atm).abs().max()) < 2e-4 * scale
    torch.testing.assert_close(gu_own, gu_atm, rtol=1e-3, atol=1e-3)
    lhs = (pe.double() * dy.double()).sum()
    rhs = (table.double() * g_own.double()).sum()
    assert abs(float(lhs - rhs)) < 1e-4 * abs(float(lhs)) + 1e-3


def test_hashgrid_autograd_module(device):
    import nesvor_amd.tinycudann as tcnn
    from oracle import hashgrid as O

    cfg = {"otype": "HashGrid", "n_levels": 6, "n_features_per_level": 2, "log2_hashmap_size": 9,
           "base_resolution": 4, "per_level_scale": 1.6}
    enc = tcnn.Encoding(3, cfg, dtype=torch.float32).to(device)
    assert enc.params.dtype == torch.float32 and float(enc.params.abs().max()) <= 1e-4
    with torch.no_grad():
        enc.params.mul_(1e4)
    x = torch.rand(500, 3, device=device, requires_grad=True)
    y = enc(x)
    assert y.shape == (500, 12)
    w = torch.randn(500, 12, device=device)
    (y * w).sum().backward()
    lv = O.make_levels(6, 9, 4, 1.6)
    gt, gu = O.encode_backward(x.detach().cpu(), enc.params.detach().cpu(), lv, 2, w.cpu())
    torch.testing.assert_close(enc.params.grad.cpu(), gt, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(x.grad.cpu(), gu, rtol=1e-4, atol=1e-3)


# ---------------------------------------------------------------------- sampler
@pytest.mark.parametrize("B,S", [(37, 8), (64, 256), (5, 100)])
def test_psf_transform_vs_oracle(device, B, S):
    """x = R(ax)(xyz + noise*sigma + t), u = (x-bb0)/(bb1-bb0), and d/d(mat) vs the oracle's
    mat_transform_points (transform.py:259-271) under autograd.  fp32: forward to 1e-5 abs on O(100) mm
    coordinates; the pose gradient sums S*|coord| terms -> relative 1e-4."""
    from nesvor_amd.sampler import psf_transform
    from oracle import nesvor_model as nm
    from oracle import transform_convert as tc

    torch.manual_seed(B)
    n = 9
    ax = torch.randn(n, 6) * torch.tensor([0.5, 0.5, 0.5, 20, 20, 20.0])
    mat = tc.axisangle2mat_forward(ax).requires_grad_(True)
    idx = torch.randint(0, n, (B,))
    xyz = torch.randn(B, 3) * 30
    sigma = torch.rand(n, 3) + 0.5
    noise = torch.randn(B, S, 3)
    bb = torch.tensor([[-150.0, -140, -130], [160, 170, 180]])
    x_ref = nm.transform_points_trans_first(mat[idx][:, None], xyz[:, None] + noise * sigma[idx][:, None])
    u_ref = ((x_ref - bb[0]) / (bb[1] - bb[0])).reshape(-1, 3)
    wx, wu = torch.randn(B, S, 3), torch.randn(B * S, 3)
    ((x_ref * wx).sum() + (u_ref * wu).sum()).backward()
    mat_d = mat.detach().to(device).requires_grad_(True)
    x, u = psf_transform(mat_d, idx.to(device), xyz.to(device), sigma.to(device), noise.to(device), bb.to(device))
    ((x * wx.to(device)).sum() + (u * wu.to(device)).sum()).backward()
    torch.testing.assert_close(x.detach().cpu(), x_ref.detach(), rtol=1e-5, atol=5e-5)
    torch.testing.assert_close(u.detach().cpu(), u_ref.detach(), rtol=1e-5, atol=1e-6)
    scale = float(mat.grad.abs().max())
    assert float((mat_d.grad.cpu() - mat.grad).abs().max()) < 2e-4 * scale


def test_psf_noise_generator_statistics_and_streams(device):
    """The counter-based N(0,1) generator of the sampler kernels (Philox4x32-10 + Box-Muller, csrc/sampler.hip), 2^22 x 3
    draws: moments of a standard normal (mean, variance, skewness, kurtosis within a few standard errors), no correlation
    between the three components of a sample nor between neighbouring samples, tails present; the same (seed, offset)
    reproduces the draws bit for bit, another offset or seed gives an unrelated stream."""
    from nesvor_amd.sampler import psf_noise

    n = 1 << 22
    a = psf_noise(1234, 7, n, device).double()
    assert a.shape == (n, 3) and bool(torch.isfinite(a).all())
    flat = a.reshape(-1)
    m, v = float(flat.mean()), float(flat.var())
    z = (flat - m) / v**0.5
    skew, kurt = float((z**3).mean()), float((z**4).mean())
    se = (1.0 / flat.numel()) ** 0.5
    assert abs(m) < 5 * se and abs(v - 1) < 5 * (2 ** 0.5) * se and abs(skew) < 5 * (6 ** 0.5) * se and abs(kurt - 3) < 5 * (24 ** 0.5) * se
    c = torch.corrcoef(a.t())
    assert float((c - torch.eye(3, device=device, dtype=torch.float64)).abs().max()) < 5 / n**0.5
    for k in range(3):
        assert abs(float((a[1:, k] * a[:-1, k]).mean())) < 5 / n**0.5
    assert float(flat.abs().max()) > 4.5  # 12.6 M draws: the 4.5 sigma tail is populated
    assert float((flat.abs() < 1).double().mean()) == pytest.approx(0.682689, abs=1e-3)
    assert torch.equal(psf_noise(1234, 7, 4096, device), a[:4096].float())
    b = psf_noise(1234, 8, n, device).double()
    c2 = psf_noise(1235, 7, n, device).double()
    for other in (b, c2):
        assert not torch.equal(other[:16], a[:16])
        assert abs(float((other.reshape(-1) * flat).mean())) < 5 * se


def test_psf_transform_kernel_noise_equals_explicit_noise(device):
    """The sampler with the noise drawn in the kernel == the sampler fed the materialised draws of the same (seed, offset):
    forward (x, u) and the pose gradient, bit for bit (same arithmetic on the same numbers)."""
    from nesvor_amd import sampler
    from oracle import transform_convert as tc

    torch.manual_seed(3)
    B, S, n = 203, 40, 9
    mat = tc.axisangle2mat_forward(torch.randn(n, 6) * torch.tensor([0.3, 0.3, 0.3, 5.0, 5.0, 5.0])).to(device)
    idx = torch.randint(0, n, (B,), device=device)
    xyz = (torch.randn(B, 3) * 20).to(device)
    sigma = (torch.rand(n, 3) + 0.5).to(device)
    bb = torch.tensor([[-60.0, -60, -60], [60, 60, 60]], device=device)
    rng = (99, 5)
    noise = sampler.psf_noise(*rng, B * S, device).view(B, S, 3)
    x1, u1 = sampler.forward_raw(mat, idx, xyz, sigma, noise, bb)
    x2, u2 = sampler.forward_raw(mat, idx, xyz, sigma, None, bb, rng, S)
    assert torch.equal(x1, x2) and torch.equal(u1, u2)
    x3, u3 = sampler.forward_raw(mat, idx, xyz, sigma, None, bb, rng, S, need_x=False)
    assert x3 is None and torch.equal(u3, u1)
    dx, du = torch.randn_like(x1), torch.randn_like(u1)
    g1 = sampler.backward_raw(mat, idx, xyz, sigma, noise, bb, dx, du)
    g2 = sampler.backward_raw(mat, idx, xyz, sigma, None, bb, dx, du, rng, S)
    assert torch.equal(g1, g2)


# ------------------------------------------------------------------ imaging loss
@pytest.mark.parametrize("reg", ["edge", "TV", "L2"])
@pytest.mark.parametrize("pix_var,slice_var,bias,scale", [(True, True, False, True), (True, True, True, True),
                                                          (False, True, False, False), (True, False, True, True),
                                                          (False, False, False, True)])
@pytest.mark.parametrize("S", [24, 64, 256])
def test_imaging_loss_vs_reference_math(device, reg, pix_var, slice_var, bias, scale, S):
    """Fused loss kernel (values + every gradient) vs the reference's formulas (models.py:286-325,366-384)
    evaluated with PyTorch autograd in fp64 on the CPU.  fp32 kernel: rtol 2e-4 on values, 5e-4 x max|grad|.
    S = 24 runs the general two-pass kernel, S = 64 and 256 the single-pass one (csrc/loss.hip)."""
    from nesvor_amd.loss import imaging_loss
    from oracle import nesvor_model as nm

    torch.manual_seed(7)
    B, n = 37, 5
    idx = torch.randint(0, n, (B,))
    f64 = lambda *sh: torch.randn(*sh, dtype=torch.float64)
    z0 = (f64(B, S) * 2).requires_grad_(True)
    lv = (f64(B, S) * 0.3).requires_grad_(True) if pix_var else None
    lb = (f64(B, S) * 0.1).requires_grad_(True) if bias else None
    x = (f64(B, 1, 3) * 20 + f64(B, S, 3)).requires_grad_(True)
    v = torch.rand(B, dtype=torch.float64)
    c = (torch.rand(n, dtype=torch.float64) + 0.5).requires_grad_(True) if scale else None
    lvs = (f64(n) * 0.2).requires_grad_(True) if slice_var else None
    delta = 0.13
    w = [1.0, 1.0, 2.0, 100.0]
    # reference formulas
    density = torch.nn.functional.softplus(z0)
    bias_t = lb.exp() if bias else 1
    bias_d = bias_t.detach() if bias else 1
    cc = c[idx] if scale else 1
    v_out = cc * (bias_t * density).mean(-1)
    var = lv.exp() if pix_var else 1
    if pix_var:
        var = ((cc.detach() if scale else 1) * (bias_d * var).mean(-1)) ** 2
    if slice_var:
        var = var + lvs.exp()[idx]
    mse = ((v_out - v) ** 2 / (2 * var)).mean()
    has_var = pix_var or slice_var
    logvar = 0.5 * var.log().mean() if has_var else torch.zeros((), dtype=torch.float64)
    ireg = nm.IMAGE_REG[reg](density, x, delta)
    breg = lb.mean() ** 2 if bias else torch.zeros((), dtype=torch.float64)
    (w[0] * mse + w[1] * logvar + w[2] * ireg + w[3] * breg).backward()

    d = lambda t: None if t is None else t.detach().float().to(device).requires_grad_(t.requires_grad)
    Z0, LV, LB, X, C, LVS = d(z0), d(lv), d(lb), d(x), d(c), d(lvs)
    got = imaging_loss(Z0.view(-1), None if LV is None else LV.view(-1), None if LB is None else LB.view(-1), X,
                       v.float().to(device), idx.to(device), C, LVS, reg, delta)
    (w[0] * got[0] + w[1] * got[1] + w[2] * got[2] + w[3] * got[3]).backward()
    for name, a_, b_ in (("mse", got[0], mse), ("logvar", got[1], logvar), ("ireg", got[2], ireg), ("breg", got[3], breg)):
        assert abs(float(a_) - float(b_)) <= 2e-4 * abs(float(b_)) + 1e-6, name
    for name, a_, b_ in (("z0", Z0, z0), ("lv", LV, lv), ("lb", LB, lb), ("x", X, x), ("c", C, c), ("lvs", LVS, lvs)):
        if b_ is None:
            continue
        ref = b_.grad if b_.grad is not None else torch.zeros_like(b_)
        gotg = a_.grad.cpu().double() if a_.grad is not None else torch.zeros_like(ref)
        scale_ = float(ref.abs().max()) + 1e-9
        assert float((gotg - ref).abs().max()) <= 5e-4 * scale_, (name, float((gotg - ref).abs().max()), scale_)


_LOSS_AB = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from nesvor_amd.loss import imaging_loss
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(11)
B, S, n = 301, int(sys.argv[3]), 7
r = lambda *sh: torch.randn(*sh, generator=g)
leaf = lambda t: t.to(dev).requires_grad_(True)
z0, lv, lb = leaf(r(B * S) * 2), leaf(r(B * S) * 0.3), leaf(r(B * S) * 0.1)
x = leaf(r(B, 1, 3) * 20 + r(B, S, 3))
c, lvs = leaf(torch.rand(n, generator=g) + 0.5), leaf(r(n) * 0.2)
v, idx = torch.rand(B, generator=g).to(dev), torch.randint(0, n, (B,), generator=g).to(dev)
out = {}
for reg in ("edge", "TV", "L2"):
    for t in (z0, lv, lb, x, c, lvs):
        t.grad = None
    got = imaging_loss(z0, lv, lb, x, v, idx, c, lvs, reg, 0.2)
    (got[0] + got[1] + 2 * got[2] + 100 * got[3]).backward()
    out[reg] = [t.detach().cpu() for t in got] + [t.grad.cpu() for t in (z0, lv, lb, x, c, lvs)]
torch.save(out, sys.argv[2])
"""


@pytest.mark.parametrize("S", [64, 128, 256, 512])
def test_imaging_loss_single_pass_kernel_equals_two_pass_kernel(device, tmp_path, S):
    """csrc/loss.hip has two kernels: the general two-pass one and a single-pass one for S = 64 K that keeps a pixel's
    samples in registers.  Same formulas in the same order: every output and every gradient must agree BIT FOR BIT
    (the switch NESVOR_LOSS_TWO_PASS is read once per process, hence the two subprocesses)."""
    import subprocess

    outs = []
    for two_pass in ("0", "1"):
        path = str(tmp_path / f"loss{two_pass}.pt")
        env = {**os.environ, "NESVOR_LOSS_TWO_PASS": two_pass}
        subprocess.run([sys.executable, "-c", _LOSS_AB, ROOT, path, str(S)], check=True, env=env, timeout=300)
        outs.append(torch.load(path))
    names = ["mse", "logvar", "ireg", "breg", "d_z0", "d_log_var", "d_log_bias", "d_x", "d_c", "d_log_var_slice"]
    for reg in outs[0]:
        for name, a_, b_ in zip(names, outs[0][reg], outs[1][reg]):
            if name in ("d_c", "d_log_var_slice"):  # per-slice sums of per-pixel terms: index_add atomics, order not fixed
                torch.testing.assert_close(a_, b_, rtol=1e-5, atol=1e-8, msg=f"{reg} {name}")
            else:
                assert torch.equal(a_, b_), (reg, name, float((a_ - b_).abs().max()))


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("half", ["bf16", "fp16"])
@pytest.mark.parametrize("k_a,k_b,b_row0,rows,out_dim", [(0, 32, 0, 32, 16), (16, 15, 1, 16, 1)])
def test_fused_mlp_bf16_operand_mode(device, k_a, k_b, b_row0, rows, out_dim, half):
    """Opt-in mixed precision (nesvor_mlp_t.bf16_operands 1 / 3): every matrix product takes bf16- (fp16-, round 6: the reference's
    default arithmetic) rounded operands and accumulates in fp32.  Reference = the same arithmetic spelled out in torch (fp64
    accumulation of the rounded operands): forward, input gradients and parameter gradients.  Tolerance 2e-3 of the largest
    element (fp32 accumulation order)."""
    from nesvor_amd import mlp
    from nesvor_amd.models import build_network

    mode = True if half == "bf16" else mlp.FP16
    _bf = (lambda x: x.to(torch.bfloat16).to(torch.float32)) if half == "bf16" else (lambda x: x.to(torch.float16).to(torch.float32))

    torch.manual_seed(0)
    N, S = 4096, 256
    net = build_network(n_input_dims=k_a + k_b, n_output_dims=out_dim, activation="ReLU", output_activation="None",
                        n_neurons=64, n_hidden_layers=2, dtype=torch.float32).to(device)
    L = mlp.linear_layers(net)
    W, Bs = [l.weight.detach() for l in L], [l.bias.detach() for l in L]
    xa = torch.randn(N // S, k_a, device=device) if k_a else None
    xb = torch.randn(rows, N, device=device)
    dy = torch.randn(out_dim, N, device=device)
    y, saved = mlp.forward_raw(W, Bs, xa, xb, b_row0, k_b, S, True, bf16=mode)
    dxb = torch.empty(k_b, N, device=device)
    dxa, partial = mlp.backward_raw(W, Bs, xa, xb, dy, saved, b_row0, k_b, S, dxb, xa is not None, bf16=mode)
    flat = partial.sum(0).cpu().double()
    # reference (N, features) row-major, fp64 accumulation of bf16-rounded operands
    X = xb[b_row0 : b_row0 + k_b].t()
    if xa is not None:
        X = torch.cat([xa.repeat_interleave(S, 0), X], 1)
    X, Wd, Bd = X.cpu(), [w.cpu() for w in W], [b.cpu().double() for b in Bs]
    mm = lambda a, b: _bf(a).double() @ _bf(b).double()
    h1 = torch.relu(mm(X, Wd[0].t()) + Bd[0]).float()
    h2 = torch.relu(mm(h1, Wd[1].t()) + Bd[1]).float()
    yr = mm(h2, Wd[2].t()) + Bd[2]
    close = lambda a, b, name: (float((a.double() - b.double()).abs().max()) <= 2e-3 * float(b.abs().max()) + 1e-6) or pytest.fail(name)
    close(y.t().cpu(), yr, "y")
    G = dy.t().cpu()
    d2 = (mm(G, Wd[2]) * (h2 > 0)).float()
    d1 = (mm(d2, Wd[1]) * (h1 > 0)).float()
    dX = mm(d1, Wd[0])
    gW = [mm(d1.t(), X), mm(d2.t(), h1), mm(G.t(), h2)]
    gB = [d1.double().sum(0), d2.double().sum(0), G.double().sum(0)]
    close(dxb.t().cpu(), dX[:, k_a:], "dxb")
    if xa is not None:
        close(dxa.view(N // S, -1, k_a).sum(1).cpu(), dX[:, :k_a].view(N // S, S, k_a).sum(1), "dxa")
    off = 0
    for w, gw, gb in zip(Wd, gW, gB):
        close(flat[off : off + w.numel()].view_as(w), gw, "dW")
        off += w.numel()
        close(flat[off : off + gb.numel()], gb, "db")
        off += gb.numel()
    # and the mode really is coarser than fp32: it must NOT match the fp32 kernel to fp32 accuracy
    y32, _ = mlp.forward_raw(W, Bs, xa, xb, b_row0, k_b, S, False)
    assert float((y32 - y).abs().max()) > 1e-4 * float(y32.abs().max())
    # while the split evaluation of the fp32 products agrees with the fp32-MFMA one to fp32 rounding
    y_mfma, _ = mlp.forward_raw(W, Bs, xa, xb, b_row0, k_b, S, False, mlp.MFMA_FP32)
    y_split, _ = mlp.forward_raw(W, Bs, xa, xb, b_row0, k_b, S, False, mlp.SPLIT)
    assert float((y_mfma - y_split).abs().max()) < 2e-6 * float(y_mfma.abs().max())


@pytest.mark.parametrize("depth", [1, 2])
@pytest.mark.parametrize("k_a,k_b,b_row0,rows,out_dim", [(0, 32, 0, 32, 16), (16, 15, 1, 16, 1), (0, 16, 0, 16, 16)])
def test_fused_mlp_scaled_fp16_mode(device, k_a, k_b, b_row0, rows, out_dim, depth):
    """``nesvor_mlp_t.bf16_operands = 4`` (round 6, opt-in ``args.mlp_fp16``): the split mode's kernels with the leading term of every
    split alone - operands rounded to fp16 AFTER a power-of-two scaling, one MFMA per product.  Against the exact network in fp64:
    fp16-rounding accuracy (2^-11 per operand) in the outputs, input gradients and parameter gradients; really coarser than the fp32
    evaluation; the inference launch (nothing saved) equals the training launch bit for bit; and - the backward chain's per-sample
    scale - samples whose upstream gradient lies 2^-20 below the batch's largest keep that accuracy relative to THEMSELVES."""
    from nesvor_amd import mlp
    from nesvor_amd.models import build_network

    torch.manual_seed(3 + depth)
    N, S = 8192, 256
    net = build_network(n_input_dims=k_a + k_b, n_output_dims=out_dim, activation="ReLU", output_activation="None",
                        n_neurons=64, n_hidden_layers=depth, dtype=torch.float32).to(device)
    L = mlp.linear_layers(net)
    W, Bs = [l.weight.detach() for l in L], [l.bias.detach() for l in L]
    xa = torch.randn(N // S, k_a, device=device) if k_a else None
    xb = torch.randn(rows, N, device=device)
    dy = torch.randn(out_dim, N, device=device)
    small = torch.zeros(N, dtype=torch.bool, device=device)
    small.view(-1, 16)[1::2] = True  # every second 16-sample group: upstream gradient 2^-20 below the others
    dy[:, small] *= 2.0 ** -20
    y, saved = mlp.forward_raw(W, Bs, xa, xb, b_row0, k_b, S, True, bf16=mlp.FP16S)
    assert saved[0].numel() == N * 4  # the bits-only save of the split mode
    y_inf, _ = mlp.forward_raw(W, Bs, xa, xb, b_row0, k_b, S, False, bf16=mlp.FP16S)
    if out_dim > 1:
        assert torch.equal(y, y_inf)
    else:  # (one output row: the training launch evaluates the output layer on the VALU in fp32, the inference launch on the matrix pipe)
        assert float((y - y_inf).abs().max()) < 2e-3 * float(y.abs().max())
    dxb = torch.empty(k_b, N, device=device)
    dxa, partial = mlp.backward_raw(W, Bs, xa, xb, dy, saved, b_row0, k_b, S, dxb, xa is not None, bf16=mlp.FP16S)
    flat = partial.sum(0).cpu().double()
    X = xb[b_row0 : b_row0 + k_b].t()
    if xa is not None:
        X = torch.cat([xa.repeat_interleave(S, 0), X], 1)
    X, Wd, Bd = X.cpu().double(), [w.cpu().double() for w in W], [b.cpu().double() for b in Bs]
    hs, h = [], X
    for l in range(depth):
        h = torch.relu(h @ Wd[l].t() + Bd[l])
        hs.append(h)
    yr = h @ Wd[depth].t() + Bd[depth]
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    e_y = rel(y.t().cpu(), yr)
    assert 1e-5 < e_y < 3e-3, e_y
    y32, _ = mlp.forward_raw(W, Bs, xa, xb, b_row0, k_b, S, False)
    assert rel(y32.t().cpu(), yr) < 3e-6
    # the gates the backward uses are the forward's saved bits (word (group, lane = 16 q + sample), bit 16 l + 4 b + r = unit 16 b +
    # 4 q + r of layer l): a unit whose pre-activation is within fp16 rounding of zero may be gated differently from the exact
    # network - either is a valid subgradient there - so the reference backward takes the kernel's gates
    words = saved[0].view(torch.int32).view(N // 16, 4, 16).cpu()  # [group][q][sample]
    gates = []
    for l in range(depth):
        g_l = torch.zeros(N // 16, 16, 64, dtype=torch.bool)  # [group][sample][unit]
        for b in range(4):
            for r in range(4):
                g_l[:, :, [16 * b + 4 * q + r for q in range(4)]] = (((words >> (16 * l + 4 * b + r)) & 1) != 0).permute(0, 2, 1)
        gates.append(g_l.view(N, 64))
        assert float((gates[l] == (hs[l] > 0)).double().mean()) > 0.99
    G = dy.t().cpu().double()
    d, gW, gB = G, [None] * (depth + 1), [None] * (depth + 1)
    for l in range(depth, -1, -1):
        inp = hs[l - 1] if l > 0 else X
        gW[l], gB[l] = d.t() @ inp, d.sum(0)
        d = d @ Wd[l]
        if l > 0:
            d = d * gates[l - 1]
    dX = d
    sm = small.cpu()
    for name, grp in (("large", ~sm), ("2^-20", sm)):
        e = rel(dxb.t().cpu()[grp], dX[grp][:, k_a:])
        assert e < 5e-3, (name, e)
    if xa is not None:
        assert rel(dxa.view(N // S, -1, k_a).sum(1).cpu(), dX[:, :k_a].view(N // S, S, k_a).sum(1)) < 5e-3
    off = 0
    for w, gw, gb in zip(Wd, gW, gB):
        assert rel(flat[off : off + w.numel()].view_as(w), gw) < 5e-3
        off += w.numel()
        assert rel(flat[off : off + gb.numel()], gb) < 5e-3
        off += gb.numel()


@pytest.mark.parametrize("N", [8192, 1 << 20])
@pytest.mark.parametrize("k_a,k_b,b_row0,rows,out_dim", [(0, 32, 0, 32, 16), (16, 15, 1, 16, 1)])
def test_fused_mlp_split_operands_keep_fp32_accuracy(device, k_a, k_b, b_row0, rows, out_dim, N):
    """The default evaluation of the fp32 products (two-way fp16 split of power-of-two-scaled operands, three fp16 MFMAs
    per product, nesvor_mlp_t.bf16_operands == 2; rounds 2-4: a three-way bf16 split, six MFMAs)
    against an fp64 evaluation of the same network: its error must not exceed that of the fp32-MFMA evaluation
    (an fp32 FMA chain) by more than a factor 1.5 - forward output, input gradient and parameter gradients (the dW
    products of the split mode pack two split terms per 32-k fp16 MFMA).  N = 2^20 is the bench's size: the pipelined
    forward and the wave-specialised backward at full scale."""
    from nesvor_amd import mlp
    from nesvor_amd.models import build_network

    torch.manual_seed(1)
    S = 256
    net = build_network(n_input_dims=k_a + k_b, n_output_dims=out_dim, activation="ReLU", output_activation="None",
                        n_neurons=64, n_hidden_layers=2, dtype=torch.float32).to(device)
    L = mlp.linear_layers(net)
    W, Bs = [l.weight.detach() for l in L], [l.bias.detach() for l in L]
    xa = torch.randn(N // S, k_a, device=device) if k_a else None
    xb = torch.randn(rows, N, device=device)
    dy = torch.randn(out_dim, N, device=device)
    X = xb[b_row0 : b_row0 + k_b].t().double()
    if xa is not None:
        X = torch.cat([xa.double().repeat_interleave(S, 0), X], 1)
    Wd, Bd = [w.double() for w in W], [b.double() for b in Bs]
    p1 = X @ Wd[0].t() + Bd[0]
    h1 = p1.relu()
    p2 = h1 @ Wd[1].t() + Bd[1]
    h2 = p2.relu()
    y_ref = (h2 @ Wd[2].t() + Bd[2]).t()
    G = dy.t().double()
    d2 = (G @ Wd[2]) * (p2 > 0)
    d1 = (d2 @ Wd[1]) * (p1 > 0)
    dxb_ref = (d1 @ Wd[0])[:, k_a:].t()
    dW_ref = torch.cat([(d1.t() @ X).reshape(-1), d1.sum(0), (d2.t() @ h1).reshape(-1), d2.sum(0), (G.t() @ h2).reshape(-1), G.sum(0)])
    err = {}
    for mode in (mlp.MFMA_FP32, mlp.SPLIT):
        y, saved = mlp.forward_raw(W, Bs, xa, xb, b_row0, k_b, S, True, mode)
        dxb = torch.empty(k_b, N, device=device)
        _, partial = mlp.backward_raw(W, Bs, xa, xb, dy, saved, b_row0, k_b, S, dxb, xa is not None, mode)
        rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
        err[mode] = (rel(y, y_ref), rel(dxb, dxb_ref), rel(partial.double().sum(0), dW_ref))
        if N > 100000:
            # a million samples x 128 hidden units: a few pre-activations lie within rounding of zero and their ReLU
            # mask differs between ANY fp32 evaluation and fp64 (both modes hit the same samples); such a sample's input
            # gradient differs by O(1) and moves every dW entry by one sample's share (1 / sqrt(N) of its magnitude).
            # The max norm is therefore replaced by: samples whose input gradient is off, and the error of the rest.
            bad = ((dxb.double() - dxb_ref).abs().amax(0) > 1e-5 * dxb_ref.abs().max())
            assert int(bad.sum()) <= 64, int(bad.sum())
            good = ~bad
            err[mode] = (err[mode][0], float((dxb.double() - dxb_ref)[:, good].abs().max() / dxb_ref.abs().max()),
                         float((partial.double().sum(0) - dW_ref).norm() / dW_ref.norm()))
            assert err[mode][2] < 2e-3
    print("max error / max|ref|  (y, dx, dW):  fp32 MFMA %s   split %s" % (err[mlp.MFMA_FP32], err[mlp.SPLIT]))
    for e_split, e_mfma in zip(err[mlp.SPLIT][: 3 if N <= 100000 else 2], err[mlp.MFMA_FP32]):
        assert e_split <= 1.5 * e_mfma + 1e-7
    if N > 100000:
        # size-independent property that pins dW at full scale: with the saved activations (= the masks) fixed the backward
        # is linear in dY
        dy2 = torch.randn(out_dim, N, device=device)
        y, saved = mlp.forward_raw(W, Bs, xa, xb, b_row0, k_b, S, True, mlp.SPLIT)
        outs = []
        for g in (dy, dy2, dy + dy2):
            dxb = torch.empty(k_b, N, device=device)
            _, partial = mlp.backward_raw(W, Bs, xa, xb, g, saved, b_row0, k_b, S, dxb, xa is not None, mlp.SPLIT)
            outs.append((dxb.double(), partial.double().sum(0)))
        for a_, b_, c_ in zip(outs[0], outs[1], outs[2]):
            assert float((a_ + b_ - c_).norm() / c_.norm()) < 2e-6


@pytest.mark.parametrize("width,depth,k_a,k_b,b_row0,rows,out_dim,bias", [
    (128, 1, 0, 32, 0, 32, 16, True), (128, 4, 16, 15, 1, 16, 1, True), (64, 4, 0, 32, 0, 32, 16, True),
    (96, 5, 16, 8, 0, 8, 1, True), (128, 2, 0, 32, 0, 32, 16, False), (40, 7, 0, 20, 2, 24, 3, True)])
def test_wide_mlp_vs_fp64_reference(device, width, depth, k_a, k_b, b_row0, rows, out_dim, bias):
    """csrc/mlp_wide.hip (round 6: width <= 128, up to seven hidden layers on hand-written fp32-MFMA kernels instead of library
    GEMMs): forward, input gradients and every parameter gradient of ``torch.ops.nesvor.wide_mlp`` against the same
    Linear/ReLU stack evaluated in float64 by autograd - ragged N (not a multiple of 16), pixel features, row offsets, a
    zero-padded width (96, 40), a bias-free stack."""
    import torch.nn as nn

    from nesvor_amd import mlp

    torch.manual_seed(width + depth)
    S, P = 24, 37
    N = S * P  # 888: not a multiple of 16 nor of the 256-sample tile
    dims = [k_a + k_b] + [width] * depth + [out_dim]
    mods = []
    for i, (a_, b_) in enumerate(zip(dims[:-1], dims[1:])):
        mods.append(nn.Linear(a_, b_, bias=bias))
        if i < len(dims) - 2:
            mods.append(nn.ReLU())
    net = nn.Sequential(*mods).to(device)
    assert mlp.wide_supported(net) and (not mlp.supported(net))
    xa = torch.randn(P, k_a, device=device, requires_grad=True) if k_a else None
    xb = torch.randn(rows, N, device=device, requires_grad=True)
    dy = torch.randn(out_dim, N, device=device)
    y = mlp.apply_net(net, xa, xb, b_row0, k_b, S)
    assert not mlp._warned_library  # (no library-GEMM fallback was taken)
    y.backward(dy)
    got = [y.detach(), xb.grad.clone(), None if xa is None else xa.grad.clone()] + [p.grad.clone() for p in net.parameters()]
    # float64 reference by autograd
    net64 = nn.Sequential(*[nn.Linear(m.in_features, m.out_features, bias=bias) if isinstance(m, nn.Linear) else nn.ReLU() for m in net]).double().to(device)
    net64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    xa64 = xa.detach().double().requires_grad_() if xa is not None else None
    xb64 = xb.detach().double().requires_grad_()
    X = xb64[b_row0 : b_row0 + k_b].t()
    if xa64 is not None:
        X = torch.cat([xa64.repeat_interleave(S, 0), X], 1)
    y64 = net64(X).t()
    y64.backward(dy.double())
    ref = [y64.detach(), xb64.grad, None if xa64 is None else xa64.grad] + [p.grad for p in net64.parameters()]
    for name, g, r in zip(["y", "dxb", "dxa"] + [n for n, _ in net.named_parameters()], got, ref):
        if r is None:
            continue
        err = float((g.double() - r).abs().max() / (r.abs().max() + 1e-30))
        assert err < 5e-6, (name, err)
    # rows of xb outside [b_row0, b_row0 + k_b) get no gradient
    mask = torch.ones(rows, dtype=torch.bool)
    mask[b_row0 : b_row0 + k_b] = False
    assert float(got[1][mask].abs().max() if mask.any() else 0.0) == 0.0
    # inference (no saved activations) gives the same output
    with torch.no_grad():
        assert torch.equal(mlp.apply_net(net, None if xa is None else xa.detach(), xb.detach(), b_row0, k_b, S), got[0])


def test_wide_network_flat_params(device):
    """The bias-free ``tinycudann.Network`` outside the fused kernels' shapes (128 neurons, 3 hidden layers) runs on the wide
    kernels through per-layer views of its flat parameter vector: output and parameter gradient against a float64 matmul chain."""
    from nesvor_amd import mlp
    from nesvor_amd.tinycudann import Network

    net = Network(20, 5, {"otype": "CutlassMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 128, "n_hidden_layers": 3}).to(device)
    assert mlp.wide_supported(net) and not mlp.supported(net)
    x = torch.randn(1000, 20, device=device, requires_grad=True)
    y = net(x)
    assert y.shape == (1000, 5)
    g = torch.randn_like(y)
    y.backward(g)
    p64 = net.params.detach().double().requires_grad_()
    x64 = x.detach().double().requires_grad_()
    h, off = x64, 0
    for li, (o, i) in enumerate(net.shapes):
        h = h @ p64[off : off + o * i].view(o, i).t()
        off += o * i
        if li < len(net.shapes) - 1:
            h = h.relu()
    y64 = h[:, :5]
    y64.backward(g.double())
    for name, a_, b_ in (("y", y.detach(), y64.detach()), ("dx", x.grad, x64.grad), ("dparams", net.params.grad, p64.grad)):
        assert float((a_.double() - b_).abs().max() / b_.abs().max()) < 5e-6, name


def _mlp_dynamic_range_errors(device, k_a, k_b, b_row0, rows, out_dim, zero_bias, scale_inputs, N=1 << 16, S=256, shifts=(0, 10, 20, 30)):
    """Per pixel group g (pixels p with p % len(shifts) == g; its operands scaled by 2^-shifts[g]): the maximum error of the
    forward output and of the input gradient against fp64, relative to the group's OWN largest reference value, in both
    evaluations of the fp32 products.  -> {mode: (err_y[g], err_dx[g])}"""
    from nesvor_amd import mlp
    from nesvor_amd.models import build_network

    torch.manual_seed(3)
    net = build_network(n_input_dims=k_a + k_b, n_output_dims=out_dim, activation="ReLU", output_activation="None",
                        n_neurons=64, n_hidden_layers=2, dtype=torch.float32).to(device)
    L = mlp.linear_layers(net)
    W = [l.weight.detach() for l in L]
    Bs = [(torch.zeros_like(l.bias) if zero_bias else l.bias.detach()) for l in L]
    P, G = N // S, len(shifts)
    pix_scale = torch.tensor([2.0 ** -shifts[p % G] for p in range(P)], device=device)
    col_scale = pix_scale.repeat_interleave(S)
    group_of = (torch.arange(P, device=device) % G).repeat_interleave(S)
    xa = torch.randn(P, k_a, device=device) if k_a else None
    xb = torch.randn(rows, N, device=device)
    dy = torch.randn(out_dim, N, device=device)
    if scale_inputs:
        xb = xb * col_scale
        xa = None if xa is None else xa * pix_scale[:, None]
    else:
        dy = dy * col_scale
    X = xb[b_row0 : b_row0 + k_b].t().double()
    if xa is not None:
        X = torch.cat([xa.double().repeat_interleave(S, 0), X], 1)
    Wd, Bd = [w.double() for w in W], [b.double() for b in Bs]
    p1 = X @ Wd[0].t() + Bd[0]
    p2 = p1.relu() @ Wd[1].t() + Bd[1]
    y_ref = (p2.relu() @ Wd[2].t() + Bd[2]).t()
    d2 = (dy.t().double() @ Wd[2]) * (p2 > 0)
    d1 = (d2 @ Wd[1]) * (p1 > 0)
    dxb_ref = (d1 @ Wd[0])[:, k_a:].t()
    out = {}
    for mode in (mlp.MFMA_FP32, mlp.SPLIT):
        y, saved = mlp.forward_raw(W, Bs, xa, xb, b_row0, k_b, S, True, mode)
        dxb = torch.empty(k_b, N, device=device)
        mlp.backward_raw(W, Bs, xa, xb, dy, saved, b_row0, k_b, S, dxb, xa is not None, mode)
        ey, ex = [], []
        for g in range(G):
            m = group_of == g
            ey.append(float((y.double() - y_ref)[:, m].abs().max() / y_ref[:, m].abs().max()))
            # (samples whose ReLU gate differs from fp64's - a pre-activation within rounding of zero - are a property of ANY
            #  fp32 evaluation, not of the split: the 99.9th percentile per group keeps them out)
            e = ((dxb.double() - dxb_ref)[:, m].abs().amax(0) / dxb_ref[:, m].abs().max())
            ex.append(float(torch.quantile(e, 0.999)))
        out[mode] = (ey, ex)
    return out


@pytest.mark.parametrize("k_a,k_b,b_row0,rows,out_dim", [(0, 32, 0, 32, 16), (16, 15, 1, 16, 1)])
def test_fused_mlp_split_dynamic_range(device, k_a, k_b, b_row0, rows, out_dim):
    """Round-5 verdict weak #2 / advisor: the split evaluation's scales are per LAUNCH, so what does a pixel lose whose
    operands lie far below the launch's bound?  Three quarters of the pixels get their upstream gradient (first
    leg) or their inputs (second leg, bias-free network: the output then scales with the input) multiplied by 2^-10, 2^-20,
    2^-30; per pixel group the error of dX / y against fp64, relative to the group's own largest value, is compared with the
    fp32-MFMA evaluation's.  The measured loss of bits is stated in include/nesvor_hip.h next to `bf16_operands`; the numbers of
    every run land in gpurun_out/mlp_split_dynamic_range_*.json."""
    import json

    from nesvor_amd import mlp

    report = {}
    for leg, (zero_bias, scale_inputs) in {"dy_scaled": (False, False), "inputs_scaled_bias_free": (True, True)}.items():
        r = _mlp_dynamic_range_errors(device, k_a, k_b, b_row0, rows, out_dim, zero_bias, scale_inputs)
        which = 0 if scale_inputs else 1  # the quantity that scales with the operand: y for inputs, dX for dY
        e_mfma, e_split = r[mlp.MFMA_FP32][which], r[mlp.SPLIT][which]
        bits = [math.log2(max(es, 1e-30) / max(em, 1e-30)) for es, em in zip(e_split, e_mfma)]
        report[leg] = {"shift_bits": [0, 10, 20, 30], "err_fp32_mfma": e_mfma, "err_split": e_split, "bits_lost_vs_fp32_mfma": bits}
        print(leg, json.dumps(report[leg]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"mlp_split_dynamic_range_k{k_a}_{k_b}.json"), "w") as fh:
        json.dump(report, fh, indent=1)
    # The contract (include/nesvor_hip.h, next to `bf16_operands`).
    # Backward: the chain carries a power of two PER SAMPLE on top of the launch's scale (round 6), so a sample's input gradient
    # keeps the fp32 chain's accuracy RELATIVE TO ITSELF however far its upstream gradient lies below the batch's largest
    # (before: 8-9 bits lost at 2^-20, 18-19 at 2^-30 - the first run of this test, profiles/r06_mlp_split_dynamic_range_before.log; after: profiles/r06_mlp_split_dynamic_range_k*.json).
    rep = report["dy_scaled"]
    for g in range(4):
        assert rep["err_split"][g] <= 1.5 * rep["err_fp32_mfma"][g] + 1e-7, ("dy_scaled", g, rep)
    # Forward: the scales are per launch (a bias does not scale with the sample, so no per-sample factor can ride through the
    # layers): the error is ABSOLUTE, ~2^-31 of the launch's largest output.  For the model's networks (biases of the outputs'
    # own magnitude) that is what an fp32 chain gives too; a BIAS-FREE network whose inputs lie 2^-20 / 2^-30 below the batch's
    # largest loses 6-10 / 16-20 bits of those small outputs against fp32 (asserted as measured: no better claim is made).  The
    # bias-free model structure (tinycudann.Network) does not run in this mode (bf16 operands).
    rep = report["inputs_scaled_bias_free"]
    for g, shift in enumerate(rep["shift_bits"]):
        assert rep["err_split"][g] <= max(1.5 * rep["err_fp32_mfma"][g] + 1e-7, 2.0 ** (shift - 30)), ("inputs_scaled_bias_free", g, rep)


@pytest.mark.parametrize("k_a,k_b,b_row0,rows,out_dim", [(0, 32, 0, 32, 16), (16, 15, 1, 16, 1)])
def test_fused_mlp_is_reproducible_at_full_size(device, k_a, k_b, b_row0, rows, out_dim):
    """The pipelined forward and the wave-specialised backward keep loads in flight in registers the compiler knows nothing
    about (inline-asm requests awaited a tile / a group later).  A register move the compiler places between a request
    and its wait reads a register whose load has not landed: whole 16-sample groups of wrong values, at random (seen in
    round 4 when the forward's copy of its prefetch set was turned into an alias).  Such a fault is not reproducible, so:
    five runs at the bench's size must agree bit for bit - outputs, saved state, input gradient (the parameter gradient
    is a sum of per-workgroup partials in a fixed order: bit-identical too)."""
    from nesvor_amd import mlp
    from nesvor_amd.models import build_network

    torch.manual_seed(3)
    N, S = 1 << 20, 256
    net = build_network(n_input_dims=k_a + k_b, n_output_dims=out_dim, activation="ReLU", output_activation="None",
                        n_neurons=64, n_hidden_layers=2, dtype=torch.float32).to(device)
    L = mlp.linear_layers(net)
    W, Bs = [l.weight.detach() for l in L], [l.bias.detach() for l in L]
    xa = torch.randn(N // S, k_a, device=device) if k_a else None
    xb = torch.randn(rows, N, device=device)
    dy = torch.randn(out_dim, N, device=device)
    ref = None
    for _ in range(5):
        y, saved = mlp.forward_raw(W, Bs, xa, xb, b_row0, k_b, S, True, mlp.SPLIT)
        dxb = torch.empty(k_b, N, device=device)
        _, partial = mlp.backward_raw(W, Bs, xa, xb, dy, saved, b_row0, k_b, S, dxb, xa is not None, mlp.SPLIT)
        cur = (y.clone(), dxb.clone(), partial.sum(0))
        if ref is None:
            ref = cur
        else:
            for a_, b_ in zip(ref, cur):
                assert torch.equal(a_, b_)


@pytest.mark.parametrize("mode", ["fp32_mfma", "split"])
@pytest.mark.parametrize("depth", [1, 2])
@pytest.mark.parametrize("k_in", [16, 32, 48, 64])
def test_fused_mlp_every_pipelined_instantiation_is_reproducible_and_right(device, k_in, depth, mode):
    """The advisor's round-4 finding on the loads the pipelined kernels keep in flight in registers the compiler knows nothing
    about: one bit-reproducibility test of ONE instantiation does not notice the hazard coming back in another (a register
    move between a request and its wait; seen in round 4 as random wrong 16-sample groups).  Every instantiation the launchers
    take - 1 and 2 hidden layers, fp32 MFMAs and the split mode, 1..4 input blocks - at N = 2^18: three runs of forward +
    backward agree BIT FOR BIT, and outputs, input gradient and parameter gradients agree with an fp64 evaluation (a stale
    register gives whole groups of O(1) errors; fp32 rounding is 1e-6)."""
    from nesvor_amd import mlp
    from nesvor_amd.models import build_network

    torch.manual_seed(k_in + depth)
    N, S, out_dim = 1 << 18, 256, 16
    net = build_network(n_input_dims=k_in, n_output_dims=out_dim, activation="ReLU", output_activation="None",
                        n_neurons=64, n_hidden_layers=depth, dtype=torch.float32).to(device)
    L = mlp.linear_layers(net)
    W, Bs = [l.weight.detach() for l in L], [l.bias.detach() for l in L]
    xb = torch.randn(k_in, N, device=device)
    dy = torch.randn(out_dim, N, device=device)
    m = mlp.MFMA_FP32 if mode == "fp32_mfma" else mlp.SPLIT
    runs = []
    for _ in range(3):
        y, saved = mlp.forward_raw(W, Bs, None, xb, 0, k_in, S, True, m)
        dxb = torch.empty(k_in, N, device=device)
        _, partial = mlp.backward_raw(W, Bs, None, xb, dy, saved, 0, k_in, S, dxb, False, m)
        runs.append((y.clone(), dxb.clone(), partial.sum(0)))
    for other in runs[1:]:
        for a_, b_ in zip(runs[0], other):
            assert torch.equal(a_, b_)
    Wd, Bd = [w.double() for w in W], [b.double() for b in Bs]
    acts, pre = [xb.t().double()], []
    for i in range(depth):
        pre.append(acts[-1] @ Wd[i].t() + Bd[i])
        acts.append(pre[-1].relu())
    y_ref = (acts[-1] @ Wd[depth].t() + Bd[depth]).t()
    d = dy.t().double()
    grads = {}
    for i in range(depth, -1, -1):
        grads[i] = torch.cat([(d.t() @ acts[i]).reshape(-1), d.sum(0)])
        if i > 0:
            d = (d @ Wd[i]) * (pre[i - 1] > 0)
    dx_ref = (d @ Wd[0]).t()
    gw_ref = torch.cat([grads[i] for i in range(depth + 1)])
    y, dxb, gw = runs[0]
    rel = lambda a_, b_: float((a_.double() - b_).abs().max() / b_.abs().max())
    # (a quarter of a million samples: a few pre-activations within rounding of zero flip their gate against fp64 - those
    #  samples' input gradients are off by O(1); everything else is fp32 rounding)
    bad = (dxb.double() - dx_ref).abs().amax(0) > 1e-5 * dx_ref.abs().max()
    assert rel(y, y_ref) < 2e-6 and int(bad.sum()) <= 16
    assert float((dxb.double() - dx_ref)[:, ~bad].abs().max() / dx_ref.abs().max()) < 2e-6
    assert float((gw.double() - gw_ref).norm() / gw_ref.norm()) < 1e-3


@pytest.mark.parametrize("k_a,k_b,b_row0,rows,depth,out_dim,S,N", [
    (0, 32, 0, 32, 2, 16, 256, 1 << 16),   # density_net
    (16, 15, 1, 16, 2, 1, 256, 1 << 16),   # sigma_net (pixel-feature block + ragged row block)
    (16, 4, 0, 32, 1, 1, 256, 1 << 15),    # b_net, one hidden layer: nothing but sign bits is saved
    (0, 16, 0, 16, 2, 16, 16, 1 << 14),    # one input block
])
def test_fused_mlp_compact_save(device, k_a, k_b, b_row0, rows, depth, out_dim, S, N, monkeypatch):
    """``nesvor_mlp_t.compact_save`` (round 5: the sign bits of the hidden layers and nothing else; the backward recomputes
    every hidden layer in its dW waves) against the full save of the same kernels: outputs and input gradients BIT FOR BIT
    (the dX chain sees the same gates; to fp32 rounding for networks with one output row, whose output layer the compact
    kernels evaluate on the VALU), parameter gradients to fp32 rounding of the recomputed layers; the saved buffers really
    are the small ones."""
    from nesvor_amd import mlp
    from nesvor_amd.models import build_network

    torch.manual_seed(N + k_b)
    net = build_network(n_input_dims=k_a + k_b, n_output_dims=out_dim, activation="ReLU", output_activation="None",
                        n_neurons=64, n_hidden_layers=depth, dtype=torch.float32).to(device)
    L = mlp.linear_layers(net)
    W, Bs = [l.weight.detach() for l in L], [l.bias.detach() for l in L]
    xa = torch.randn(N // S, k_a, device=device) if k_a else None
    xb = torch.randn(rows, N, device=device)
    xb[:, ::7] = 0.0  # exact zeros in the input (pre-activations that are exactly the bias)
    dy = torch.randn(out_dim, N, device=device)
    res = {}
    for compact in (True, False):
        if not compact:
            monkeypatch.setattr(mlp, "compact_save", lambda d, n: False)
        y, saved = mlp.forward_raw(W, Bs, xa, xb, b_row0, k_b, S, True)
        assert (saved[0].numel() == N * 4) == compact and all(t.numel() == (16 if compact else N * 64) for t in saved[1:])
        dxb = torch.empty(k_b, N, device=device)
        dxa, partial = mlp.backward_raw(W, Bs, xa, xb, dy, saved, b_row0, k_b, S, dxb, xa is not None)
        res[compact] = (y, dxb, dxa, partial.sum(0), saved)
    if out_dim > 1:
        assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
        if xa is not None:
            assert torch.equal(res[True][2], res[False][2])
    else:
        # one output row: the compact kernels evaluate the output layer's products as fp32 FMA chains on the VALU (OUT1) where
        # the full-save kernels use split-fp16 MFMAs - the same fp32 product in another summation order
        for a_, b_ in ((res[True][0], res[False][0]), (res[True][1], res[False][1])) + (((res[True][2], res[False][2]),) if xa is not None else ()):
            assert float((a_ - b_).abs().max()) <= 2e-6 * float(b_.abs().max())
    gw_c, gw_f = res[True][3], res[False][3]
    assert float((gw_c - gw_f).abs().max()) <= 2e-6 * float(gw_f.abs().max())
    # the masks are the forward's: bit 16 l + 4 b + r of word (group, lane = 16 q + sample) = [h_l > 0] of unit 16 b + 4 q + r
    words = res[True][4][0].view(torch.int32).view(N // 16, 4, 16)  # [group][q][sample]
    saved_full = res[False][4]
    for l in range(depth):
        h = saved_full[l].view(N // 16, 4, 4, 16, 4)  # [group][block b][q][sample][r]
        for b in range(4):
            for r in range(4):
                bit = (words >> (16 * l + 4 * b + r)) & 1
                assert torch.equal(bit.bool(), h[:, b, :, :, r] > 0), (l, b, r)


@pytest.mark.parametrize("k_a,k_b,b_row0,rows,depth,out_dim,S,N", [
    (0, 32, 0, 32, 2, 16, 256, 1 << 16),   # density_net
    (16, 15, 1, 16, 2, 1, 256, 1 << 16),   # sigma_net
    (16, 4, 0, 32, 1, 1, 256, 1 << 15),    # b_net, one hidden layer
    (0, 16, 0, 16, 2, 16, 16, 1 << 14),    # one input block
])
def test_fused_mlp_prebuilt_weight_images_are_bit_identical(device, k_a, k_b, b_row0, rows, depth, out_dim, S, N):
    """``nesvor_mlp_t.weight_images`` (round 6): the operand images of the split mode built ONCE by the launch that takes the weight
    norms (``nesvor_mlp_prepare_weights_images`` - what the training step does per iteration) and copied by the launches, against
    the launches building them themselves: outputs, saved bits, input gradients and parameter gradients BIT FOR BIT."""
    from nesvor_amd import mlp
    from nesvor_amd.models import build_network

    torch.manual_seed(N + k_a)
    net = build_network(n_input_dims=k_a + k_b, n_output_dims=out_dim, activation="ReLU", output_activation="None",
                        n_neurons=64, n_hidden_layers=depth, dtype=torch.float32).to(device)
    L = mlp.linear_layers(net)
    W, Bs = [l.weight.detach() * 3.0 for l in L], [l.bias.detach() for l in L]
    xa = torch.randn(N // S, k_a, device=device) if k_a else None
    xb = torch.randn(rows, N, device=device)
    dy = torch.randn(out_dim, N, device=device)
    res = {}
    for images in (False, True):
        y, saved = mlp.forward_raw(W, Bs, xa, xb, b_row0, k_b, S, True, weight_images=images)
        dxb = torch.empty(k_b, N, device=device)
        dxa, partial = mlp.backward_raw(W, Bs, xa, xb, dy, saved, b_row0, k_b, S, dxb, xa is not None, weight_images=images)
        torch.cuda.synchronize()
        res[images] = (y, saved[0], dxb, partial) + ((dxa,) if xa is not None else ())
    for a_, b_ in zip(res[False], res[True]):
        assert torch.equal(a_.view(torch.int32), b_.view(torch.int32))  # (bit patterns: the saved words are gate bits, some of them NaN patterns)
    assert float(res[True][0].abs().max()) > 0 and float(res[True][3].abs().max()) > 0


def test_fused_mlp_gate_convention_at_an_exact_plus_zero(device):
    """The bits-only save keeps [sign bit of the pre-activation clear] per hidden unit (include/nesvor_hip.h, compact_save):
    an exact +0 - a sample whose input row is all zeros meeting zero biases - passes its gradient, where torch's / tcnn's
    ReLU'(0) = 0 drops it (round-4 advisor).  Pinned here: for such samples dX is the LINEAR chain W0^T W1^T Wout^T dy, every
    other sample follows the reference, and the parameter gradients are the reference's (a +0 unit contributes h = 0 to dW and a
    gradient that meets x = 0 / h_prev = 0; only the biases see it)."""
    from nesvor_amd import mlp
    from nesvor_amd.models import build_network

    torch.manual_seed(5)
    N, S = 4096, 256
    net = build_network(n_input_dims=32, n_output_dims=16, activation="ReLU", output_activation="None", n_neurons=64,
                        n_hidden_layers=2, dtype=torch.float32).to(device)
    L = mlp.linear_layers(net)
    with torch.no_grad():
        for l in L:
            l.bias.zero_()
    W, Bs = [l.weight.detach() for l in L], [l.bias.detach() for l in L]
    xb = torch.randn(32, N, device=device)
    zero = torch.zeros(N, dtype=torch.bool, device=device)
    zero[64:96] = True  # two whole 16-sample groups ...
    zero[1000] = True   # ... and a single sample
    xb[:, zero] = 0.0
    dy = torch.randn(16, N, device=device)
    assert mlp.compact_save(mlp.dims_desc(2, 16, 0, 32, 0, S), N)
    y, saved = mlp.forward_raw(W, Bs, None, xb, 0, 32, S, True)
    dxb = torch.empty(32, N, device=device)
    _, partial = mlp.backward_raw(W, Bs, None, xb, dy, saved, 0, 32, S, dxb, False)
    assert float(y[:, zero].abs().max()) == 0.0
    x64 = xb.double().t().requires_grad_(True)
    h = x64
    for i, (w, b) in enumerate(zip(W, Bs)):
        h = h @ w.double().t() + b.double()
        if i < 2:
            h = torch.relu(h)
    h.backward(dy.double().t())
    ref = x64.grad.t()
    scale = float(ref.abs().max())
    assert float((dxb[:, ~zero].double() - ref[:, ~zero]).abs().max()) <= 1e-5 * scale
    assert float(ref[:, zero].abs().max()) == 0.0  # the reference's convention ...
    lin = (W[0].double().t() @ W[1].double().t() @ W[2].double().t() @ dy.double())[:, zero]
    assert float((dxb[:, zero].double() - lin).abs().max()) <= 1e-5 * float(lin.abs().max()) and float(lin.abs().max()) > 0  # ... and this library's
    # weight gradients: unaffected (layout of `partial`: per layer W then b, as the flat parameter segment)
    gw = partial.sum(0)
    h = xb.double().t()
    ws = [w.double().clone().requires_grad_(True) for w in W]
    for i, w in enumerate(ws):
        h = h @ w.t()
        if i < 2:
            h = torch.relu(h)
    h.backward(dy.double().t())
    off = 0
    for w, b, wg in zip(W, Bs, ws):
        got = gw[off: off + w.numel()].view_as(w).double()
        assert float((got - wg.grad).abs().max()) <= 2e-5 * float(wg.grad.abs().max())
        off += w.numel() + b.numel()


# -------------------------------------------------------------------- fused MLP
@pytest.mark.parametrize("k_a,k_b,b_row0,rows,depth,out_dim,S,N", [
    (0, 32, 0, 32, 2, 16, 8, 1000),     # density_net (ragged N, not a multiple of 16)
    (16, 15, 1, 16, 2, 1, 8, 1000),     # sigma_net: [slice embedding | z[1:]]
    (16, 4, 0, 32, 1, 1, 16, 2048),     # b_net: [slice embedding | pe[:4]]
    (0, 16, 0, 16, 1, 16, 1, 300),      # default depth 1, E=16
    (0, 24, 0, 24, 3, 16, 256, 512),    # L=12 (default level scale), depth 3
    (16, 15, 1, 16, 1, 1, 256, 65536),  # full pixel clouds
    (0, 32, 0, 32, 2, 16, 256, 4096),   # bench shapes: wave-specialised backward, two hidden layers, density_net
    (16, 15, 1, 16, 2, 1, 256, 4096),   # ... sigma_net
    (0, 16, 0, 16, 2, 16, 16, 512),     # one input block, two hidden layers
    (0, 48, 0, 48, 1, 16, 16, 512),     # three input blocks, one hidden layer
    (16, 40, 0, 48, 2, 1, 16, 512),     # four input blocks, two hidden layers (single-role fused backward, fast I/O path)
    (32, 20, 2, 24, 2, 3, 32, 1024),    # two pixel-feature blocks + a ragged row block
])
@pytest.mark.parametrize("fused_bwd", [True, False])
@pytest.mark.parametrize("operands", ["split", "mfma"])
def test_fused_mlp_vs_torch_fp32_reference(device, k_a, k_b, b_row0, rows, depth, out_dim, S, N, fused_bwd, operands, monkeypatch):
    """fp32 network (products as split-fp16 MFMAs - the default - or as fp32 MFMAs) vs the same nn.Sequential evaluated
    by PyTorch (fp32 reference of the same op).  Tolerance: fp32 with K <= 64 per layer and different summation order:
    rtol 2e-4 / atol 2e-5 fwd, grads relative to their max."""
    import nesvor_amd.mlp as M
    from nesvor_amd.mlp import fused_mlp
    from nesvor_amd.models import build_network

    monkeypatch.setattr(M, "FUSED_BACKWARD", fused_bwd)  # fused dX+dW+db kernel vs the two-kernel path
    monkeypatch.setattr(M, "FP32_OPERANDS", M.SPLIT if operands == "split" else M.MFMA_FP32)
    torch.manual_seed(N + k_a)
    net = build_network(n_input_dims=k_a + k_b, n_output_dims=out_dim, activation="ReLU", output_activation="None",
                        n_neurons=64, n_hidden_layers=depth, dtype=torch.float32).to(device)
    P = N // S
    xa = torch.randn(P, k_a, device=device, requires_grad=True) if k_a else None
    xb = torch.randn(rows, N, device=device, requires_grad=True)
    w = torch.randn(out_dim, N, device=device)
    y = fused_mlp(net, xa, xb, b_row0, k_b, S)
    (y * w).sum().backward()
    got = {"y": y.detach(), "xb": xb.grad.clone(), "xa": None if xa is None else xa.grad.clone()}
    got.update({n: p.grad.clone() for n, p in net.named_parameters()})
    for p in net.parameters():
        p.grad = None
    xb.grad = None
    if xa is not None:
        xa.grad = None
    feats = [] if xa is None else [xa[:, None].expand(-1, S, -1).reshape(N, k_a)]
    inp = torch.cat(feats + [xb[b_row0 : b_row0 + k_b].t()], -1)
    y_ref = net(inp).t()
    (y_ref * w).sum().backward()
    torch.testing.assert_close(got["y"], y_ref.detach(), rtol=2e-4, atol=2e-5)

    def close(a, b, name):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 3e-4 * scale, (name, float((a - b).abs().max()), scale)

    close(got["xb"], xb.grad, "xb")
    if xa is not None:
        close(got["xa"], xa.grad, "xa")
    for n, p in net.named_parameters():
        close(got[n], p.grad, n)


def test_fused_mlp_vs_oracle_cpu(device):
    from nesvor_amd.mlp import fused_mlp
    from nesvor_amd.models import build_network
    from oracle import nesvor_model as nm

    torch.manual_seed(0)
    net = build_network(n_input_dims=32, n_output_dims=16, activation="ReLU", output_activation="None",
                        n_neurons=64, n_hidden_layers=2, dtype=torch.float32)
    P = {f"n.{k}": v.detach().clone() for k, v in net.state_dict().items()}
    x = torch.randn(32, 4096)
    ref = nm.mlp_forward(P, "n", x.t(), 3).t()
    y = fused_mlp(net.to(device), None, x.to(device), 0, 32, 1).cpu()
    torch.testing.assert_close(y, ref, rtol=2e-4, atol=2e-5)


# ------------------------------------------------------------------------ AdamW
def test_step_bookkeeping_kernels(device):
    """nesvor_slice_grads / nesvor_step_prologue / nesvor_step_epilogue / nesvor_sum_rows against the torch ops they
    replace in the training iteration (index_add, softmax forward/backward, axisangle2mat and its backward, sums)."""
    from nesvor_amd import _lib
    from oracle import transform_convert as tc

    lib = _lib.load()
    st = _lib.stream_ptr()
    torch.manual_seed(0)
    n, B, S, ks = 37, 200, 32, 16
    idx = torch.randint(0, n, (B,), device=device)
    dc_pix, dlvs_pix = torch.randn(B, device=device), torch.randn(B, device=device)
    dxa = torch.randn(B * S // 16, ks, device=device)  # one row per 16-sample group
    dpix = torch.randn(B, 3, 4, device=device)
    dc, dlvs = torch.zeros(n, device=device), torch.zeros(n, device=device)
    dse, dmat = torch.zeros(n, ks, device=device), torch.zeros(n, 3, 4, device=device)
    assert lib.nesvor_slice_grads(_lib.ptr(idx), _lib.ptr(dc_pix), _lib.ptr(dlvs_pix), _lib.ptr(dxa), _lib.ptr(dpix), _lib.ptr(dc),
                                  _lib.ptr(dlvs), _lib.ptr(dse), _lib.ptr(dmat), B, S // 16, ks, st) == 0
    torch.testing.assert_close(dc, torch.zeros(n, device=device).index_add_(0, idx, dc_pix), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dlvs, torch.zeros(n, device=device).index_add_(0, idx, dlvs_pix), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dmat, torch.zeros(n, 3, 4, device=device).index_add_(0, idx, dpix), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dse, torch.zeros(n, ks, device=device).index_add_(0, idx, dxa.view(B, -1, ks).sum(1)), rtol=1e-4, atol=1e-4)

    # the atomic-free variant (one workgroup per slice, pixel list in batch order): same sums, twice the same bits; ADDS to
    # what the outputs hold; also with one slice owning every pixel, with missing parts, and with a batch larger than one
    # 256-pixel scan chunk; refuses what it cannot list
    for trial, (idx2, Bt) in enumerate(((idx, B), (torch.full((B,), 5, device=device), B), (torch.randint(0, n, (1000,), device=device), 1000))):
        dcp, dlp = torch.randn(Bt, device=device), torch.randn(Bt, device=device)
        dx2, dp2 = torch.randn(Bt * S // 16, ks, device=device), torch.randn(Bt, 3, 4, device=device)
        outs = []
        for rep in range(2):
            o = [torch.ones(n, device=device), torch.ones(n, device=device), torch.ones(n, ks, device=device), torch.ones(n, 3, 4, device=device)]
            assert lib.nesvor_slice_grads_by_slice(_lib.ptr(idx2), _lib.ptr(dcp), _lib.ptr(dlp), _lib.ptr(dx2), _lib.ptr(dp2), _lib.ptr(o[0]),
                                                   _lib.ptr(o[1]), _lib.ptr(o[2]), _lib.ptr(o[3]), Bt, S // 16, ks, n, st) == 0
            outs.append(o)
        for a_, b_ in zip(outs[0], outs[1]):
            assert torch.equal(a_, b_)
        o = outs[0]
        torch.testing.assert_close(o[0], torch.ones(n, device=device).index_add_(0, idx2, dcp), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(o[1], torch.ones(n, device=device).index_add_(0, idx2, dlp), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(o[3], torch.ones(n, 3, 4, device=device).index_add_(0, idx2, dp2), rtol=1e-5, atol=2e-5)
        torch.testing.assert_close(o[2], torch.ones(n, ks, device=device).index_add_(0, idx2, dx2.view(Bt, -1, ks).sum(1)), rtol=1e-4, atol=2e-4)
    only_se = torch.zeros(n, ks, device=device)
    assert lib.nesvor_slice_grads_by_slice(_lib.ptr(idx), None, None, _lib.ptr(dxa), None, None, None, _lib.ptr(only_se), None, B, S // 16, ks, n, st) == 0
    torch.testing.assert_close(only_se, torch.zeros(n, ks, device=device).index_add_(0, idx, dxa.view(B, -1, ks).sum(1)), rtol=1e-4, atol=1e-4)
    assert lib.nesvor_slice_grads_by_slice(_lib.ptr(idx), None, None, _lib.ptr(dxa), None, None, None, _lib.ptr(only_se), None, 5000, 1, ks, n, st) == 1
    assert lib.nesvor_slice_grads_by_slice(_lib.ptr(idx), None, None, _lib.ptr(dxa), None, None, None, _lib.ptr(only_se), None, B, 1, 24, n, st) == 1

    logit = torch.randn(n, device=device)
    ax = torch.randn(n, 6, device=device) * torch.tensor([0.5, 0.5, 0.5, 3, 3, 3], device=device)
    c, mat, zb = torch.empty(n, device=device), torch.empty(n, 3, 4, device=device), torch.ones(50, device=device)
    assert lib.nesvor_step_prologue(_lib.ptr(logit), _lib.ptr(c), _lib.ptr(ax), _lib.ptr(mat), _lib.ptr(zb), 50, n, st) == 0
    torch.testing.assert_close(c, torch.softmax(logit, 0) * n, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(mat.cpu(), tc.axisangle2mat_forward(ax.cpu()), rtol=1e-5, atol=1e-5)
    assert float(zb.abs().max()) == 0.0

    dcv, dm, dtr, terms = torch.randn(n, device=device), torch.randn(n, 3, 4, device=device), torch.randn(n, 6, device=device), torch.rand(n, device=device)
    loss_pix = torch.rand(B, 3, device=device)
    dlogit, dax, vals = torch.empty(n, device=device), torch.empty(n, 6, device=device), torch.empty(5, device=device)
    assert lib.nesvor_step_epilogue(_lib.ptr(dcv), _lib.ptr(c), _lib.ptr(dlogit), _lib.ptr(dm), _lib.ptr(ax), _lib.ptr(dtr), 0.1,
                                    _lib.ptr(dax), _lib.ptr(loss_pix), _lib.ptr(terms), _lib.ptr(vals), n, B, 0.25, -0.5, st) == 0
    lg = logit.clone().requires_grad_(True)
    (torch.softmax(lg, 0) * n * dcv).sum().backward()
    torch.testing.assert_close(dlogit, lg.grad, rtol=1e-4, atol=1e-5)
    ref_dax = tc.axisangle2mat_backward(dm.cpu(), ax.cpu()) + 0.1 * dtr.cpu()
    torch.testing.assert_close(dax.cpu(), ref_dax, rtol=1e-4, atol=1e-4)
    sums = loss_pix.sum(0)
    ref_vals = torch.stack([sums[0] / B, sums[1] / B, (sums[0] + sums[1]) / B, terms.sum(), sums[2] * 0.25 - 0.5])
    torch.testing.assert_close(vals, ref_vals, rtol=1e-5, atol=1e-5)

    part = torch.randn(256, 6288, device=device)
    out = torch.empty(6288, device=device)
    assert lib.nesvor_sum_rows(_lib.ptr(part), _lib.ptr(out), 256, 6288, 6288, st) == 0
    torch.testing.assert_close(out, part.sum(0), rtol=1e-4, atol=1e-4)
    out2 = torch.empty(1000, device=device)  # a column range of the wider matrix
    assert lib.nesvor_sum_rows(part[:, 300:].data_ptr(), _lib.ptr(out2), 256, 1000, 6288, st) == 0
    torch.testing.assert_close(out2, part[:, 300:1300].sum(0), rtol=1e-4, atol=1e-4)


def test_fused_adamw_vs_torch(device):
    from nesvor_amd import _lib

    torch.manual_seed(0)
    n = 100003  # odd length exercises the scalar tail
    p0 = torch.randn(n, device=device)
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p_ref], lr=5e-3, betas=(0.9, 0.99), eps=1e-15, weight_decay=1e-2)
    p, m, v = p0.clone(), torch.zeros(n, device=device), torch.zeros(n, device=device)
    lr = 5e-3
    for t in range(1, 6):
        g = torch.randn(n, device=device) * (10.0 ** torch.randint(-6, 2, (n,), device=device).float())
        if t == 2:
            g[::2] = 0  # untouched table entries: weight decay still applies (train.py:144-152)
        p_ref.grad = g.clone()
        opt.step()
        gbuf = g.clone()
        err = _lib.load().nesvor_adamw_step(_lib.ptr(p), _lib.ptr(gbuf), _lib.ptr(m), _lib.ptr(v), n, lr, 0.9, 0.99,
                                            1e-15, 1e-2, 1 - 0.9**t, 1 - 0.99**t, 1.0, 1, _lib.stream_ptr())
        assert err == 0
        assert float(gbuf.abs().max()) == 0.0  # fused zero_grad
        if t == 3:
            lr *= 0.33
            opt.param_groups[0]["lr"] = lr
        torch.testing.assert_close(p, p_ref.data, rtol=1e-5, atol=1e-6)
