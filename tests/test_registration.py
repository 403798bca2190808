"""Stack registration: `resample` / `ncc_loss` properties on the CPU, and on the GPU the
reference's own VVR test (tests/svort/test_vvr.py) plus the --registration stack pipeline."""
import math

import pytest
import torch


def test_resample_sizes_and_linear_ramp():
    from nesvor_amd.registration import resample

    x = torch.arange(5 * 7 * 9, dtype=torch.float32).reshape(1, 1, 5, 7, 9)
    same = resample(x, (1.0, 1.0, 1.0), (1.0, 1.0, 1.0))
    torch.testing.assert_close(same, x)
    y = resample(x, (1.5, 1.0, 3.0), (1.0, 1.0, 1.5))  # x: 9 -> int(13.5) = 13, y: 7, z: 5 -> 10
    assert y.shape == (1, 1, 10, 7, 13)
    # a linear ramp stays the same linear ramp of the physical coordinate
    ramp = torch.linspace(0, 8, 9).view(1, 1, 1, 9).expand(1, 1, 4, 9).contiguous()  # value = x index, voxel 2 mm
    r = resample(ramp, (2.0, 2.0), (1.0, 1.0))
    assert r.shape == (1, 1, 8, 18)
    centre_old, centre_new = (9 - 1) / 2 * 2.0, (18 - 1) / 2 * 1.0
    expect = ((torch.arange(18) * 1.0 - centre_new) + centre_old) / 2.0
    inside = (expect >= 0) & (expect <= 8)  # the outermost new samples lie half a voxel outside the old grid (zero padding)
    torch.testing.assert_close(r[0, 0, 3][inside], expect[inside], rtol=1e-5, atol=1e-5)


def test_ncc_loss_properties():
    from nesvor_amd.utils import ncc_loss

    g = torch.Generator().manual_seed(0)
    I = torch.rand(2, 1, 6, 8, 8, generator=g)
    J = 3.0 * I + 0.5
    torch.testing.assert_close(ncc_loss(I, J, win=None), -torch.ones(2, 1), rtol=1e-3, atol=1e-3)  # affine intensity map
    K = torch.rand(2, 1, 6, 8, 8, generator=g)
    a, b = I.flatten(1), K.flatten(1)
    cov = (a * b).mean(1) - a.mean(1) * b.mean(1)
    ref = -(cov**2) / (a.var(1, unbiased=False) * b.var(1, unbiased=False) + 1e-6)
    torch.testing.assert_close(ncc_loss(I, K, win=None).view(-1), ref, rtol=1e-5, atol=1e-6)
    m = (torch.rand(2, 1, 6, 8, 8, generator=g) > 0.3).float()
    n = m.flatten(1).sum(1) + 1e-6
    am, bm = (I * m).flatten(1), (K * m).flatten(1)
    mean = lambda t: t.sum(1) / n
    cov = mean(am * bm) - mean(am) * mean(bm)
    ref = -(cov**2) / ((mean(am * am) - mean(am) ** 2) * (mean(bm * bm) - mean(bm) ** 2) + 1e-6)
    torch.testing.assert_close(ncc_loss(I, K, mask=m, win=None).view(-1), ref, rtol=1e-5, atol=1e-6)
    assert ncc_loss(I, K, win=5).shape == I.shape and ncc_loss(I, K, win=9, level=1).shape == I.shape
    assert float(ncc_loss(I, J, win=None, reduction="mean")) == pytest.approx(-1.0, abs=1e-3)


@pytest.mark.gpu
def test_device_filters_without_a_convolution_library_match_the_cpu_convolutions(device):
    """On a HIP device `gaussian_blur` and the box window of `ncc_loss` run as shifted multiply-adds (the first MIOpen
    convolution of a process costs seconds of kernel search); on the CPU they are the reference's depthwise convolutions.
    Same numbers up to fp32 summation order, in 2-D and 3-D, zero padding at the borders included."""
    from nesvor_amd.utils import gaussian_blur, ncc_loss

    g = torch.Generator().manual_seed(5)
    x3, y3 = torch.rand(2, 1, 11, 13, 17, generator=g), torch.rand(2, 1, 11, 13, 17, generator=g)
    x2 = torch.rand(3, 2, 19, 23, generator=g)
    for x, sig, tr in ((x3, 2.0, 3), (x3, [0.7, 1.3, 2.9], 4.0), (x2, 1.5, 3)):
        torch.testing.assert_close(gaussian_blur(x.to(device), sig, tr).cpu(), gaussian_blur(x, sig, tr), rtol=1e-5, atol=1e-6)
    for win, level in ((5, 0), (9, 1), (9, 0)):
        torch.testing.assert_close(ncc_loss(x3.to(device), y3.to(device), win=win, level=level).cpu(),
                                   ncc_loss(x3, y3, win=win, level=level), rtol=1e-3, atol=1e-4)


@pytest.mark.gpu
def test_vvr_reference_test(device):
    """tests/svort/test_vvr.py:16-44: the 128^3 phantom registered to itself from a known offset (3 levels, 8 rounds,
    finite-difference gradient, momentum 0.1, global NCC) must come back to the target pose: atol 1e-5, rtol 1e-3."""
    from nesvor_amd.phantom import phantom3d
    from nesvor_amd.registration import VVR
    from nesvor_amd.transform import RigidTransform
    from nesvor_amd.utils import ncc_loss

    volume = torch.tensor(phantom3d(n=128), dtype=torch.float32, device=device)[None, None]
    vvr = VVR(num_levels=3, num_steps=8, step_size=2, max_iter=20, optimizer={"name": "gd", "momentum": 0.1},
              loss=lambda s, x, y: ncc_loss(x[None], y[None], win=None, reduction="none"), auto_grad=False)
    trans_first = False
    ax = torch.tensor([[0.4, 0.1, -0.6, 20, -50, 100]], dtype=torch.float32, device=device)
    t_target = RigidTransform(torch.tensor([[0.4 + 0.05, 0.1 - 0.05, -0.6 + 0.1, 20 + 3, -50 - 2, 100 + 1.5]],
                                           dtype=torch.float32, device=device), trans_first=trans_first)
    ax_out, loss = vvr(ax, volume, volume, {"res_s": 1, "s_thick": 1.5}, t_target, trans_first)
    torch.testing.assert_close(ax_out, t_target.axisangle(trans_first=trans_first), atol=1e-5, rtol=1e-3)
    assert float(loss) < -0.99


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["ncc", "mse"])
def test_vvr_fused_similarity_equals_grid_sample_path(device, kind):
    """nesvor_vvr_similarity (one launch, 13 poses, fp64 moment sums) against F.grid_sample + the loss for the same poses;
    and the reference's VVR test through the fused path (loss given by name) reaches the same pose."""
    from nesvor_amd.phantom import phantom3d
    from nesvor_amd.registration import VVR, _Level
    from nesvor_amd.transform import RigidTransform

    volume = torch.tensor(phantom3d(n=64), dtype=torch.float32, device=device)[None, None]
    loss = {"name": "ncc", "win": None} if kind == "ncc" else {"name": "mse"}
    vvr = VVR(num_levels=3, num_steps=8, step_size=2, max_iter=20, optimizer={"name": "gd", "momentum": 0.1}, loss=loss, auto_grad=False)
    vvr.theta_t = RigidTransform(torch.tensor([[0.2, -0.1, 0.3, 4.0, -3.0, 2.0]], device=device), trans_first=False)
    vvr.trans_first = False
    vvr.res, vvr.relative_res = 1.0, [1.5, 1.0, 1.0]
    lv = vvr._level(1, volume, volume)
    assert isinstance(lv, _Level) and lv.values.numel() > 1000
    g = torch.Generator().manual_seed(0)
    thetas = torch.tensor([[11.0, -6.0, 17.0, 4.5, -2.0, 2.5]]) + torch.randn(13, 6, generator=g) * torch.tensor([2.0, 2.0, 2.0, 1.5, 1.5, 1.5])
    thetas = thetas.to(device)
    fused = vvr._objective_fused(thetas, lv)
    if kind == "ncc":
        from nesvor_amd.utils import ncc_loss
        ref = torch.cat([ncc_loss(vvr._warp(t[None], lv)[None], lv.values.view(1, 1, -1), win=None).view(-1) for t in thetas])
    else:
        ref = torch.cat([((vvr._warp(t[None], lv) - lv.values[None]) ** 2).mean(1) for t in thetas])
    torch.testing.assert_close(fused, ref, rtol=2e-4, atol=1e-6)
    if kind == "ncc":
        ax = torch.tensor([[0.4, 0.1, -0.6, 20, -50, 100]], dtype=torch.float32, device=device)
        t_target = RigidTransform(torch.tensor([[0.45, 0.05, -0.5, 23, -52, 101.5]], dtype=torch.float32, device=device), trans_first=False)
        big = torch.tensor(phantom3d(n=128), dtype=torch.float32, device=device)[None, None]
        out, l = vvr(ax, big, big, {"res_s": 1, "s_thick": 1.5}, t_target, False)
        torch.testing.assert_close(out, t_target.axisangle(trans_first=False), atol=1e-5, rtol=1e-3)


@pytest.mark.gpu
def test_vvr_autograd_gradient_variant(device):
    """auto_grad=True (gradient through grid_sample) from a smaller offset, MSE loss given as a dict."""
    from nesvor_amd.phantom import phantom3d
    from nesvor_amd.registration import VVR
    from nesvor_amd.transform import RigidTransform

    volume = torch.tensor(phantom3d(n=64), dtype=torch.float32, device=device)[None, None]
    vvr = VVR(num_levels=2, num_steps=6, step_size=1, max_iter=20, optimizer={"name": "gd", "momentum": 0.0},
              loss={"name": "mse"}, auto_grad=True)
    truth = torch.tensor([[0.1, -0.2, 0.05, 3.0, -2.0, 1.0]], dtype=torch.float32, device=device)
    start = truth + torch.tensor([[0.02, -0.02, 0.03, 1.0, -1.0, 0.5]], device=device)
    out, _ = vvr(start, volume, volume, {"res_s": 1, "s_thick": 1}, RigidTransform(truth, trans_first=True), True)
    assert float((out - truth)[:, :3].abs().max()) < 2e-3 and float((out - truth)[:, 3:].abs().max()) < 0.05


@pytest.mark.gpu
def test_register_stacks_removes_a_rigid_offset(device):
    """--registration stack: two stacks with the same content; the second one's poses are off by 4 deg / 3 mm.
    After registration both stacks must sit at the same place again (poses of corresponding slices agree)."""
    from nesvor_amd.image import Stack
    from nesvor_amd.phantom import phantom3d
    from nesvor_amd.registration import register_stacks
    from nesvor_amd.transform import RigidTransform

    vol = torch.tensor(phantom3d(n=64), dtype=torch.float32, device=device)
    n = 32
    slices = vol[::2][:, None].contiguous()  # (32,1,64,64): 1 mm in-plane, 2 mm through-plane
    def poses(offset):
        t = torch.zeros((n, 6), dtype=torch.float32, device=device)
        t[:, -1] = (torch.arange(n, dtype=torch.float32, device=device) - (n - 1) / 2) * 2.0
        return offset.compose(RigidTransform(t))
    ident = RigidTransform(torch.zeros((1, 6), device=device))
    off = RigidTransform(torch.tensor([[math.radians(4.0), 0.0, math.radians(-2.0), 3.0, -2.0, 1.0]], device=device))
    stacks = [Stack(slices.clone(), slices > 0, poses(ident), resolution_x=1.0, resolution_y=1.0, thickness=2.0, gap=2.0),
              Stack(slices.clone(), slices > 0, poses(off), resolution_x=1.0, resolution_y=1.0, thickness=2.0, gap=2.0)]
    register_stacks(stacks)
    a, b = stacks[0].transformation.matrix(), stacks[1].transformation.matrix()
    # the same content at the same place: rotation within 0.3 deg, translation within 0.3 mm
    assert float((a[:, :, :3] - b[:, :, :3]).abs().max()) < 6e-3
    assert float((a[:, :, 3] - b[:, :, 3]).abs().max()) < 0.3
