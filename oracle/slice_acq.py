"""CPU oracle: slice acquisition forward operator A (volume -> PSF-blurred slices).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates, vectorised over slice
pixels with a Python loop over PSF taps,
``slice_acquisition_forward_cuda_kernel``
(nesvor/slice_acquisition/slice_acq_cuda_kernel.cu:17-171; host :954-991):

* pixel centre in voxel units: R (p + T) + (dims-1)/2              (.cu:42-56)
* tap loop bounds -d/2 .. (d+1)/2, zero taps skipped               (.cu:61-65)
* tap position = centre + R (ix_p, iy_p, iz_p); dropped unless
  0 <= pos < dim-1                                                 (.cu:66-69)
* default: trilinear sampling of the volume with PSF weight        (.cu:110-160)
* ``interp_psf``: nearest voxel + trilinear re-interpolation of
  the PSF at the voxel's offset from the centre                    (.cu:71-109)
* out = sum(w v) / sum(w) where sum(w) > 0                         (.cu:167-170)
"""
import torch


def slice_acquisition_forward(
    transforms: torch.Tensor,  # (n,3,4), translation in voxel units
    vol: torch.Tensor,  # (1,1,D,H,W)
    vol_mask,  # bool (1,1,D,H,W) or None
    slices_mask,  # bool (n,1,h,w) or None
    psf: torch.Tensor,  # (d_p,h_p,w_p)
    slice_shape,
    res_slice: float,
    need_weight: bool = False,
    interp_psf: bool = False,
):
    dt = vol.dtype
    n = transforms.shape[0]
    h, w = int(slice_shape[0]), int(slice_shape[1])
    D, H, W = vol.shape[-3:]
    d_p, h_p, w_p = psf.shape
    volf = vol.reshape(-1)
    vmask = None if vol_mask is None or vol_mask.numel() == 0 else vol_mask.reshape(-1)
    R = transforms[:, :, :3]  # (n,3,3)
    T = transforms[:, :, 3]  # (n,3)
    # the reference evaluates (i - (w-1)/2.) * res_slice + t in double and rounds once (.cu:46-47)
    rs = float(torch.tensor(res_slice, dtype=dt))
    px = (torch.arange(w, dtype=torch.float64) - (w - 1) / 2.0) * rs  # (w,)
    py = (torch.arange(h, dtype=torch.float64) - (h - 1) / 2.0) * rs  # (h,)
    _x = (px[None, None, :] + T[:, 0, None, None].double()).to(dt)  # (n,1,w)
    _y = (py[None, :, None] + T[:, 1, None, None].double()).to(dt)  # (n,h,1)
    _z = T[:, 2, None, None]
    _x, _y, _z = torch.broadcast_tensors(_x, _y, _z)
    _x, _y, _z = _x.expand(n, h, w), _y.expand(n, h, w), _z.expand(n, h, w)

    def rot(row, a, b, c):
        return (
            R[:, row, 0, None, None] * a
            + R[:, row, 1, None, None] * b
            + R[:, row, 2, None, None] * c
        )

    xc = rot(0, _x, _y, _z) + (W - 1) / 2.0
    yc = rot(1, _x, _y, _z) + (H - 1) / 2.0
    zc = rot(2, _x, _y, _z) + (D - 1) / 2.0
    val = torch.zeros(n, h, w, dtype=dt)
    wsum = torch.zeros(n, h, w, dtype=dt)
    active = torch.ones(n, h, w, dtype=torch.bool)
    if slices_mask is not None and slices_mask.numel() > 0:
        active = slices_mask.reshape(n, h, w).clone()
    Sy, Sz = W, H * W
    psff = psf.reshape(-1)
    i_p = -1
    for iz_p in range(-(d_p // 2), (d_p + 1) // 2):
        for iy_p in range(-(h_p // 2), (h_p + 1) // 2):
            for ix_p in range(-(w_p // 2), (w_p + 1) // 2):
                i_p += 1
                pv = psff[i_p]
                if float(pv) == 0.0:
                    continue
                # left-to-right like the reference: ((centre + r1 ix) + r2 iy) + r3 iz   (.cu:66-68)
                x = xc + R[:, 0, 0, None, None] * ix_p + R[:, 0, 1, None, None] * iy_p + R[:, 0, 2, None, None] * iz_p
                y = yc + R[:, 1, 0, None, None] * ix_p + R[:, 1, 1, None, None] * iy_p + R[:, 1, 2, None, None] * iz_p
                z = zc + R[:, 2, 0, None, None] * ix_p + R[:, 2, 1, None, None] * iy_p + R[:, 2, 2, None, None] * iz_p
                ok = active & (x >= 0) & (y >= 0) & (z >= 0) & (x < W - 1) & (y < H - 1) & (z < D - 1)
                if not bool(ok.any()):
                    continue
                xs = torch.where(ok, x, torch.zeros_like(x))
                ys = torch.where(ok, y, torch.zeros_like(y))
                zs = torch.where(ok, z, torch.zeros_like(z))
                if interp_psf:
                    xr = torch.floor(xs + 0.5)
                    yr = torch.floor(ys + 0.5)
                    zr = torch.floor(zs + 0.5)
                    iv = (zr * Sz + yr * Sy + xr).long()
                    if vmask is not None:
                        ok = ok & vmask[iv]
                    v_ = volf[iv]
                    dx_, dy_, dz_ = xr - xc, yr - yc, zr - zc
                    # R^T (voxel - centre) + PSF half extent
                    xp = R[:, 0, 0, None, None] * dx_ + R[:, 1, 0, None, None] * dy_ + R[:, 2, 0, None, None] * dz_ + (w_p - 1) / 2.0
                    yp = R[:, 0, 1, None, None] * dx_ + R[:, 1, 1, None, None] * dy_ + R[:, 2, 1, None, None] * dz_ + (h_p - 1) / 2.0
                    zp = R[:, 0, 2, None, None] * dx_ + R[:, 1, 2, None, None] * dy_ + R[:, 2, 2, None, None] * dz_ + (d_p - 1) / 2.0
                    ok = ok & (xp >= 0) & (yp >= 0) & (zp >= 0) & (xp < w_p - 1) & (yp < h_p - 1) & (zp < d_p - 1)
                    xp = torch.where(ok, xp, torch.zeros_like(xp))
                    yp = torch.where(ok, yp, torch.zeros_like(yp))
                    zp = torch.where(ok, zp, torch.zeros_like(zp))
                    xf, yf, zf = torch.floor(xp), torch.floor(yp), torch.floor(zp)
                    wx, wy, wz = xp - xf, yp - yf, zp - zf
                    ip = (zf * (w_p * h_p) + yf * w_p + xf).long()
                    pw = torch.zeros_like(xp)
                    for cz in (0, 1):
                        for cy in (0, 1):
                            for cx in (0, 1):
                                wgt = (wx if cx else 1 - wx) * (wy if cy else 1 - wy) * (wz if cz else 1 - wz)
                                pw = pw + wgt * psff[ip + cx + cy * w_p + cz * w_p * h_p]
                    pw = torch.where(ok, pw, torch.zeros_like(pw))
                    val = val + pw * v_
                    wsum = wsum + pw
                else:
                    xf, yf, zf = torch.floor(xs), torch.floor(ys), torch.floor(zs)
                    wx, wy, wz = xs - xf, ys - yf, zs - zf
                    iv = (zf * Sz + yf * Sy + xf).long()
                    for cz in (0, 1):
                        for cy in (0, 1):
                            for cx in (0, 1):
                                wgt = (wx if cx else 1 - wx) * (wy if cy else 1 - wy) * (wz if cz else 1 - wz) * pv
                                ic = iv + cx + cy * Sy + cz * Sz
                                okc = ok if vmask is None else (ok & vmask[ic])
                                wgt = torch.where(okc, wgt, torch.zeros_like(wgt))
                                val = val + wgt * volf[ic]
                                wsum = wsum + wgt
    pos = wsum > 0
    out = torch.where(pos, val / torch.where(pos, wsum, torch.ones_like(wsum)), torch.zeros_like(val))
    out = out.view(n, 1, h, w)
    if need_weight:
        return out, torch.where(pos, wsum, torch.zeros_like(wsum)).view(n, 1, h, w)
    return out


# --------------------------------------------------------------------------- adjoint and backward
# SURVEY.md §8(f) rank 1.  Both interpolation modes; `interp_psf=True` is never used by the reference's own callers
# (svort/srr.py, svort/models.py, svort/inference.py all pass False) and is restated from the kernels alone.
def _geometry(transforms, shape_dhw, slice_shape, res_slice, dt):
    """Pixel centres in voxel units (n,h,w) x 3 and q = pixel + t in the slice frame."""
    n = transforms.shape[0]
    h, w = int(slice_shape[0]), int(slice_shape[1])
    D, H, W = shape_dhw
    R = transforms[:, :, :3]
    T = transforms[:, :, 3]
    rs = float(torch.tensor(res_slice, dtype=dt))
    px = (torch.arange(w, dtype=torch.float64) - (w - 1) / 2.0) * rs
    py = (torch.arange(h, dtype=torch.float64) - (h - 1) / 2.0) * rs
    qx = (px[None, None, :] + T[:, 0, None, None].double()).to(dt).expand(n, h, w)
    qy = (py[None, :, None] + T[:, 1, None, None].double()).to(dt).expand(n, h, w)
    qz = T[:, 2, None, None].expand(n, h, w)

    def rot(row, a, b, c):
        return R[:, row, 0, None, None] * a + R[:, row, 1, None, None] * b + R[:, row, 2, None, None] * c

    xc = rot(0, qx, qy, qz) + (W - 1) / 2.0
    yc = rot(1, qx, qy, qz) + (H - 1) / 2.0
    zc = rot(2, qx, qy, qz) + (D - 1) / 2.0
    return R, (qx, qy, qz), (xc, yc, zc)


def _taps(psf):
    d_p, h_p, w_p = psf.shape
    psff = psf.reshape(-1)
    i_p = -1
    for iz in range(-(d_p // 2), (d_p + 1) // 2):
        for iy in range(-(h_p // 2), (h_p + 1) // 2):
            for ix in range(-(w_p // 2), (w_p + 1) // 2):
                i_p += 1
                if float(psff[i_p]) != 0.0:
                    yield ix, iy, iz, psff[i_p]


def _tap_pos(R, centre, ix, iy, iz):
    xc, yc, zc = centre
    x = xc + R[:, 0, 0, None, None] * ix + R[:, 0, 1, None, None] * iy + R[:, 0, 2, None, None] * iz
    y = yc + R[:, 1, 0, None, None] * ix + R[:, 1, 1, None, None] * iy + R[:, 1, 2, None, None] * iz
    z = zc + R[:, 2, 0, None, None] * ix + R[:, 2, 1, None, None] * iy + R[:, 2, 2, None, None] * iz
    return x, y, z


def _psf_weight(R, centre, psf, dims):
    """sum of the PSF taps that fall inside the volume (pass 1 of the backward / adjoint kernels,
    .cu:220-261, :510-558): note it ignores vol_mask and the trilinear split."""
    D, H, W = dims
    wsum = torch.zeros_like(centre[0])
    for ix, iy, iz, pv in _taps(psf):
        x, y, z = _tap_pos(R, centre, ix, iy, iz)
        ok = (x >= 0) & (y >= 0) & (z >= 0) & (x < W - 1) & (y < H - 1) & (z < D - 1)
        wsum = wsum + torch.where(ok, pv.expand_as(x), torch.zeros_like(x))
    return wsum


def _interp_tap(R, centre, psf, x, y, z, ok, dims, need_grad=False):
    """The interp_psf branch shared by the kernels (.cu:229-257, :279-308, :553-600, :754-783): the tap position is rounded
    to the nearest voxel and the PSF is re-interpolated (trilinear) at that voxel's offset from the pixel centre, taken back
    to the slice frame with R^T.  -> (ok, voxel index, pw, (xr, yr, zr), (d pw/d x_psf, d/d y_psf, d/d z_psf) | None);
    pw and the derivatives are 0 where not ok."""
    D, H, W = dims
    d_p, h_p, w_p = psf.shape
    psff = psf.reshape(-1)
    xc, yc, zc = centre
    zero = torch.zeros_like(x)
    xs, ys, zs = (torch.where(ok, t, zero) for t in (x, y, z))
    xr, yr, zr = torch.floor(xs + 0.5), torch.floor(ys + 0.5), torch.floor(zs + 0.5)
    iv = (zr * (H * W) + yr * W + xr).long()
    dx_, dy_, dz_ = xr - xc, yr - yc, zr - zc
    r = lambda a, b: R[:, a, b, None, None]
    xp = r(0, 0) * dx_ + r(1, 0) * dy_ + r(2, 0) * dz_ + (w_p - 1) / 2.0
    yp = r(0, 1) * dx_ + r(1, 1) * dy_ + r(2, 1) * dz_ + (h_p - 1) / 2.0
    zp = r(0, 2) * dx_ + r(1, 2) * dy_ + r(2, 2) * dz_ + (d_p - 1) / 2.0
    ok = ok & (xp >= 0) & (yp >= 0) & (zp >= 0) & (xp < w_p - 1) & (yp < h_p - 1) & (zp < d_p - 1)
    xp, yp, zp = (torch.where(ok, t, zero) for t in (xp, yp, zp))
    xf, yf, zf = torch.floor(xp), torch.floor(yp), torch.floor(zp)
    wx, wy, wz = xp - xf, yp - yf, zp - zf
    ip = (zf * (w_p * h_p) + yf * w_p + xf).long()
    pw, gx, gy, gz = zero.clone(), zero.clone(), zero.clone(), zero.clone()
    for cx, cy, cz in ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1)):  # .cu order
        ax, ay, az = (wx if cx else 1 - wx), (wy if cy else 1 - wy), (wz if cz else 1 - wz)
        c = psff[ip + cx + cy * w_p + cz * w_p * h_p]
        pw = pw + ax * ay * az * c
        if need_grad:
            gx = gx + (1.0 if cx else -1.0) * ay * az * c
            gy = gy + (1.0 if cy else -1.0) * ax * az * c
            gz = gz + (1.0 if cz else -1.0) * ax * ay * c
    pw = torch.where(ok, pw, zero)
    grads = tuple(torch.where(ok, t, zero) for t in (gx, gy, gz)) if need_grad else None
    return ok, iv, pw, (xr, yr, zr), grads


def _psf_weight_interp(R, centre, psf, dims):
    """pass 1 of the backward / adjoint kernels in interp_psf mode (.cu:229-257, :526-558): sum of the re-interpolated
    PSF values of the taps inside the volume; vol_mask is ignored."""
    D, H, W = dims
    wsum = torch.zeros_like(centre[0])
    for ix, iy, iz, _ in _taps(psf):
        x, y, z = _tap_pos(R, centre, ix, iy, iz)
        ok = (x >= 0) & (y >= 0) & (z >= 0) & (x < W - 1) & (y < H - 1) & (z < D - 1)
        wsum = wsum + _interp_tap(R, centre, psf, x, y, z, ok, dims)[2]
    return wsum


def _pose_terms_interp(dx, dy, dz, voxel, dims):
    """(..., 12) pose-gradient terms of one tap in interp_psf mode (.cu:354-361, :822-829), row-major (3, 4)."""
    D, H, W = dims
    xr, yr, zr = voxel
    ox, oy, oz = xr - (W - 1) / 2.0, yr - (H - 1) / 2.0, zr - (D - 1) / 2.0
    return torch.stack([dx * ox, dy * ox, dz * ox, -dx, dx * oy, dy * oy, dz * oy, -dy, dx * oz, dy * oz, dz * oz, -dz], -1)


def slice_acquisition_adjoint_forward(transforms, psf, slices, slices_mask, vol_mask, vol_shape, res_slice,
                                      interp_psf=False, equalize=False):
    """A^T: slices (n,1,h,w) -> (vol (1,1,D,H,W), vol_weight | None).
    Restates slice_acquisition_adjoint_forward_cuda_kernel (.cu:472-670) + equalize (.cu:672-693):
    every pixel with PSF weight >= 0.5 scatters s * psf/weight * trilinear into the volume."""
    dt = slices.dtype
    D, H, W = (int(s) for s in vol_shape)
    n, _, h, w = slices.shape
    R, _, centre = _geometry(transforms, (D, H, W), (h, w), res_slice, dt)
    weight = (_psf_weight_interp if interp_psf else _psf_weight)(R, centre, psf, (D, H, W))
    active = weight >= 0.5
    if slices_mask is not None and slices_mask.numel() > 0:
        active = active & slices_mask.reshape(n, h, w)
    s = slices.reshape(n, h, w)
    vol = torch.zeros(D * H * W, dtype=dt)
    vw = torch.zeros(D * H * W, dtype=dt)
    vm = None if vol_mask is None or vol_mask.numel() == 0 else vol_mask.reshape(-1)
    wsafe = torch.where(active, weight, torch.ones_like(weight))
    for ix, iy, iz, pv in _taps(psf):
        x, y, z = _tap_pos(R, centre, ix, iy, iz)
        ok = active & (x >= 0) & (y >= 0) & (z >= 0) & (x < W - 1) & (y < H - 1) & (z < D - 1)
        if interp_psf:  # .cu:582-606: one voxel per tap
            ok, iv, pw, _, _ = _interp_tap(R, centre, psf, x, y, z, ok, (D, H, W))
            okm = ok if vm is None else (ok & vm[iv])
            wgt = torch.where(okm, pw / wsafe, torch.zeros_like(pw))
            vol.index_add_(0, iv.reshape(-1), (wgt * s).reshape(-1))
            vw.index_add_(0, iv.reshape(-1), wgt.reshape(-1))
            continue
        xs, ys, zs = (torch.where(ok, t, torch.zeros_like(t)) for t in (x, y, z))
        xf, yf, zf = torch.floor(xs), torch.floor(ys), torch.floor(zs)
        wx, wy, wz = xs - xf, ys - yf, zs - zf
        iv = (zf * (H * W) + yf * W + xf).long()
        pn = pv / wsafe
        for cz in (0, 1):
            for cy in (0, 1):
                for cx in (0, 1):
                    wgt = (wx if cx else 1 - wx) * (wy if cy else 1 - wy) * (wz if cz else 1 - wz) * pn
                    ic = iv + cx + cy * W + cz * H * W
                    okc = ok if vm is None else (ok & vm[ic])
                    wgt = torch.where(okc, wgt, torch.zeros_like(wgt))
                    vol.index_add_(0, ic.reshape(-1), (wgt * s).reshape(-1))
                    vw.index_add_(0, ic.reshape(-1), wgt.reshape(-1))
    if equalize:
        pos = vw > 0
        vol = torch.where(pos, vol / torch.where(pos, vw, torch.ones_like(vw)), vol)
    return vol.view(1, 1, D, H, W), (vw.view(1, 1, D, H, W) if equalize else None)


def slice_acquisition_backward(transforms, vol, vol_mask, psf, grad_slices, slices_mask, res_slice,
                               interp_psf=False, need_vol_grad=True, need_transforms_grad=True):
    """Backward of A as the reference computes it (.cu:173-470): gs = grad / sum(psf in bounds) is
    distributed with psf * trilinear weights; the normalising weight is treated as constant."""
    dt = vol.dtype
    D, H, W = vol.shape[-3:]
    n, _, h, w = grad_slices.shape
    R, q, centre = _geometry(transforms, (D, H, W), (h, w), res_slice, dt)
    weight = (_psf_weight_interp if interp_psf else _psf_weight)(R, centre, psf, (D, H, W))
    g = grad_slices.reshape(n, h, w)
    active = (weight != 0) & (g != 0)
    if slices_mask is not None and slices_mask.numel() > 0:
        active = active & slices_mask.reshape(n, h, w)
    gs = torch.where(active, g / torch.where(active, weight, torch.ones_like(weight)), torch.zeros_like(g))
    volf = vol.reshape(-1)
    vm = None if vol_mask is None or vol_mask.numel() == 0 else vol_mask.reshape(-1)
    gvol = torch.zeros(D * H * W, dtype=dt)
    gR = torch.zeros(n, 3, 3, dtype=dt)
    gT = torch.zeros(n, 3, dtype=dt)
    qx, qy, qz = q
    for ix, iy, iz, pv in _taps(psf):
        x, y, z = _tap_pos(R, centre, ix, iy, iz)
        ok = active & (x >= 0) & (y >= 0) & (z >= 0) & (x < W - 1) & (y < H - 1) & (z < D - 1)
        if interp_psf:  # .cu:279-370
            ok, iv, pw, voxel, (gx, gy, gz) = _interp_tap(R, centre, psf, x, y, z, ok, (D, H, W), need_grad=True)
            okm = ok if vm is None else (ok & vm[iv])
            zero = torch.zeros_like(pw)
            if need_vol_grad:
                gvol.index_add_(0, iv.reshape(-1), torch.where(okm, pw * gs, zero).reshape(-1))
            if need_transforms_grad:
                tmp = torch.where(okm, gs * volf[iv], zero)
                terms = _pose_terms_interp(gx * tmp, gy * tmp, gz * tmp, voxel, (D, H, W)).sum((1, 2)).view(n, 3, 4)
                gR += terms[:, :, :3]
                gT += terms[:, :, 3]
            continue
        xs, ys, zs = (torch.where(ok, t, torch.zeros_like(t)) for t in (x, y, z))
        xf, yf, zf = torch.floor(xs), torch.floor(ys), torch.floor(zs)
        wx, wy, wz = xs - xf, ys - yf, zs - zf
        iv = (zf * (H * W) + yf * W + xf).long()
        pg = pv * gs
        dx = torch.zeros_like(x)
        dy = torch.zeros_like(x)
        dz = torch.zeros_like(x)
        for cz in (0, 1):
            for cy in (0, 1):
                for cx in (0, 1):
                    ax_, ay_, az_ = (wx if cx else 1 - wx), (wy if cy else 1 - wy), (wz if cz else 1 - wz)
                    ic = iv + cx + cy * W + cz * H * W
                    okc = ok if vm is None else (ok & vm[ic])
                    if need_vol_grad:
                        gvol.index_add_(0, ic.reshape(-1), torch.where(okc, ax_ * ay_ * az_ * pg, torch.zeros_like(pg)).reshape(-1))
                    if need_transforms_grad:
                        val = torch.where(okc, pg * volf[ic], torch.zeros_like(pg))
                        dx = dx + (1.0 if cx else -1.0) * ay_ * az_ * val
                        dy = dy + (1.0 if cy else -1.0) * ax_ * az_ * val
                        dz = dz + (1.0 if cz else -1.0) * ax_ * ay_ * val
        if need_transforms_grad:
            ox, oy, oz = qx + ix, qy + iy, qz + iz
            for row, dd in enumerate((dx, dy, dz)):
                gR[:, row, 0] += (dd * ox).sum((1, 2))
                gR[:, row, 1] += (dd * oy).sum((1, 2))
                gR[:, row, 2] += (dd * oz).sum((1, 2))
            for c in range(3):
                gT[:, c] += (dx * R[:, 0, c, None, None] + dy * R[:, 1, c, None, None] + dz * R[:, 2, c, None, None]).sum((1, 2))
    grad_vol = gvol.view(vol.shape) if need_vol_grad else None
    grad_tf = torch.cat([gR, gT[:, :, None]], -1) if need_transforms_grad else None
    return grad_vol, grad_tf


def slice_acquisition_adjoint_backward(transforms, grad_vol, vol_weight, vol_mask, psf, slices, slices_mask, vol, res_slice,
                                       interp_psf=False, equalize=False):
    """Backward of A^T as the reference computes it (.cu:695-950 + equalize_cuda_kernel with is_grad, .cu:672-693):
    -> (grad_slices (n,1,h,w), grad_transforms (n,3,4)).  grad_vol is NOT modified here (the native op does it in place).
      g = grad_vol / max(vol_weight, 1e-3) where vol_weight > 0            [equalize only]
      grad_slices[p] = sum_taps psf * trilinear(g) / sum_taps psf          (taps inside the volume, masked corners dropped)
      grad_T: the same expression differentiated w.r.t. the tap position, corner values weighted by
              (slices[p] - vol[corner]) when equalising, by slices[p] otherwise."""
    dt = slices.dtype
    D, H, W = grad_vol.shape[-3:]
    n, h, w = slices.shape[0], slices.shape[-2], slices.shape[-1]
    g = grad_vol.reshape(-1).clone()
    if equalize:
        wv = vol_weight.reshape(-1)
        pos = wv > 0
        g[pos] = g[pos] / wv[pos].clamp(min=1e-3)
    volf = vol.reshape(-1) if equalize else None
    vm = vol_mask.reshape(-1) if vol_mask is not None else None
    R, q, centre = _geometry(transforms, (D, H, W), (h, w), res_slice, dt)
    sv = slices.reshape(n, h, w)
    val = torch.zeros((n, h, w), dtype=dt)
    weight = torch.zeros((n, h, w), dtype=dt)
    gT = torch.zeros((n, h, w, 12), dtype=dt)
    Sy, Sz = W, H * W
    for ix, iy, iz, pv in _taps(psf):
        x, y, z = _tap_pos(R, centre, ix, iy, iz)
        inb = (x >= 0) & (y >= 0) & (z >= 0) & (x < W - 1) & (y < H - 1) & (z < D - 1)
        if interp_psf:  # .cu:754-836
            ok, iv, pw, voxel, (gx, gy, gz) = _interp_tap(R, centre, psf, x, y, z, inb, (D, H, W), need_grad=True)
            okm = ok if vm is None else (ok & vm[iv])
            zero = torch.zeros_like(pw)
            gv = torch.where(okm, g[iv], zero)
            sfac = torch.where(okm, (sv - volf[iv] if equalize else sv) * g[iv], zero)
            gT = gT + _pose_terms_interp(gx * sfac, gy * sfac, gz * sfac, voxel, (D, H, W))
            val = val + torch.where(okm, pw, zero) * gv
            weight = weight + torch.where(okm, pw, zero)
            continue
        xf, yf, zf = x.floor().clamp(0, W - 2), y.floor().clamp(0, H - 2), z.floor().clamp(0, D - 2)
        wx, wy, wz = x - xf, y - yf, z - zf
        i0 = (zf * Sz + yf * Sy + xf).long()
        v_ = torch.zeros_like(x)
        dx, dy, dz = torch.zeros_like(x), torch.zeros_like(x), torch.zeros_like(x)
        for c in range(8):
            cx, cy, cz = c & 1, (c >> 1) & 1, c >> 2
            ic = i0 + cx + cy * Sy + cz * Sz
            ok = inb if vm is None else inb & vm[ic]
            gv = torch.where(ok, g[ic], torch.zeros_like(x))
            ax, ay, az = (wx if cx else 1 - wx), (wy if cy else 1 - wy), (wz if cz else 1 - wz)
            v_ = v_ + ax * ay * az * gv
            s = (sv - volf[ic] if equalize else sv) * gv
            dx = dx + (s if cx else -s) * ay * az
            dy = dy + (s if cy else -s) * ax * az
            dz = dz + (s if cz else -s) * ax * ay
        inbf = inb.to(dt)
        val = val + pv * v_ * inbf
        weight = weight + pv * inbf
        dx, dy, dz = dx * pv * inbf, dy * pv * inbf, dz * pv * inbf
        ox, oy, oz = q[0] + ix, q[1] + iy, q[2] + iz
        r = lambda a, b: R[:, a, b].view(n, 1, 1)
        terms = [dx * ox, dx * oy, dx * oz, dx * r(0, 0) + dy * r(1, 0) + dz * r(2, 0),
                 dy * ox, dy * oy, dy * oz, dx * r(0, 1) + dy * r(1, 1) + dz * r(2, 1),
                 dz * ox, dz * oy, dz * oz, dx * r(0, 2) + dy * r(1, 2) + dz * r(2, 2)]
        gT = gT + torch.stack(terms, -1)
    live = weight > 0
    if slices_mask is not None:
        live = live & slices_mask.reshape(n, h, w)
    safe = torch.where(live, weight, torch.ones_like(weight))
    grad_slices = torch.where(live, val / safe, torch.zeros_like(val)).view(n, 1, h, w)
    grad_transforms = (torch.where(live[..., None], gT / safe[..., None], torch.zeros_like(gT))).sum((1, 2)).view(n, 3, 4)
    return grad_slices, grad_transforms
