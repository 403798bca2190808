"""CPU oracle: axis-angle <-> 3x4 rigid matrix, forward and analytic backward.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates, vectorised over n,
the four kernels of the reference's ``transform_convert_cuda`` extension:

* ``axisangle2mat_forward``   nesvor/transform/transform_convert_cuda_kernel.cu:14-65
* ``axisangle2mat_backward``  ...:68-188
* ``mat2axisangle_forward``   ...:190-264
* ``mat2axisangle_backward``  ...:266-440

Branch thresholds: ``TRANSFORM_EPS = 1e-6`` (.cu:10); quaternion branch masks
``r22 < EPS``, ``r00 > r11``, ``r00 < -r11`` (.cu:211-213).

The reference evaluates sqrt/sin/cos/atan2 in single precision even when the
tensors are double (``sqrtf`` etc.); this oracle evaluates them in the tensor's
dtype, which is identical for the fp32 parity target and makes fp64 gradcheck
meaningful.
"""
import torch

EPS = 1e-6


def axisangle2mat_forward(ax: torch.Tensor) -> torch.Tensor:
    """(n,6) [rotvec | t] -> (n,3,4) [R | t]   (.cu:14-65)."""
    a = ax[:, :3]
    theta2 = (a * a).sum(-1)
    big = theta2 > EPS
    theta = torch.sqrt(torch.where(big, theta2, torch.ones_like(theta2)))
    k = a / theta[:, None]
    s, c = torch.sin(theta), torch.cos(theta)
    oc = 1 - c
    kx, ky, kz = k.unbind(-1)
    Rb = torch.stack(
        [
            c + kx * kx * oc, kx * ky * oc - kz * s, ky * s + kx * kz * oc,
            kz * s + kx * ky * oc, c + ky * ky * oc, -kx * s + ky * kz * oc,
            -ky * s + kx * kz * oc, kx * s + ky * kz * oc, c + kz * kz * oc,
        ],
        -1,
    )
    ax_, ay_, az_ = a.unbind(-1)
    one = torch.ones_like(ax_)
    Rs = torch.stack([one, -az_, ay_, az_, one, -ax_, -ay_, ax_, one], -1)
    R = torch.where(big[:, None], Rb, Rs).view(-1, 3, 3)
    return torch.cat([R, ax[:, 3:, None]], -1)


def axisangle2mat_backward(grad_mat: torch.Tensor, ax: torch.Tensor) -> torch.Tensor:
    """grad (n,3,4), ax (n,6) -> grad_ax (n,6)   (.cu:68-188)."""
    a = ax[:, :3]
    G = grad_mat[:, :, :3]
    theta2 = (a * a).sum(-1)
    big = theta2 > EPS
    theta = torch.sqrt(torch.where(big, theta2, torch.ones_like(theta2)))
    k = a / theta[:, None]
    s, c = torch.sin(theta), torch.cos(theta)
    oc = 1 - c
    # R = c I + oc k k^T + s [k]x, with k treated as independent of theta first
    eye = torch.eye(3, dtype=ax.dtype, device=ax.device)
    kk = k[:, :, None] * k[:, None, :]
    # skew-part of G: vee(G - G^T)
    skew = torch.stack(
        [G[:, 2, 1] - G[:, 1, 2], G[:, 0, 2] - G[:, 2, 0], G[:, 1, 0] - G[:, 0, 1]], -1
    )
    dc = (G * (eye - kk)).sum((1, 2))  # d/dc of (c I + (1-c) kk^T)
    ds = (skew * k).sum(-1)
    # d/dk: oc * (G + G^T) k + s * skew
    dk = oc[:, None] * torch.einsum("nij,nj->ni", G + G.transpose(1, 2), k) + s[:, None] * skew
    # chain through theta = |a|, k = a/theta
    dtheta = c * ds - s * dc
    proj = dk - (dk * k).sum(-1, keepdim=True) * k
    g_big = dtheta[:, None] * k + proj / theta[:, None]
    g_small = skew  # first-order branch R = I + [a]x (.cu:176-182)
    g_rot = torch.where(big[:, None], g_big, g_small)
    return torch.cat([g_rot, grad_mat[:, :, 3]], -1)


def _quat_branches(mat: torch.Tensor):
    r = [[mat[:, i, j] for j in range(3)] for i in range(3)]
    m_d2 = r[2][2] < EPS
    m_d0_d1 = r[0][0] > r[1][1]
    m_d0_nd1 = r[0][0] < -r[1][1]
    b0 = (~m_d2) & (~m_d0_nd1)
    b1 = m_d2 & m_d0_d1
    b2 = m_d2 & (~m_d0_d1)
    b3 = ~(b0 | b1 | b2)
    return r, (b0, b1, b2, b3)


def _select(masks, vals):
    out = torch.zeros_like(vals[0])
    for m, v in zip(masks, vals):
        out = torch.where(m, v, out)
    return out


def _quaternion(mat: torch.Tensor):
    """Shepperd-style 4-branch rotation -> (w,x,y,z) before the w>=0 sign fix."""
    r, masks = _quat_branches(mat)
    tr = [
        r[0][0] + r[1][1] + r[2][2] + 1,
        r[0][0] - r[1][1] - r[2][2] + 1,
        r[1][1] - r[0][0] - r[2][2] + 1,
        r[2][2] - r[0][0] - r[1][1] + 1,
    ]
    # guard sqrt of the non-selected branches (their value is discarded)
    ss = [2 * torch.sqrt(torch.where(m, t, torch.ones_like(t))) for m, t in zip(masks, tr)]
    a = r[2][1] - r[1][2]
    b = r[0][2] - r[2][0]
    cc = r[1][0] - r[0][1]
    p01 = r[0][1] + r[1][0]
    p02 = r[0][2] + r[2][0]
    p12 = r[1][2] + r[2][1]
    w = _select(masks, [0.25 * ss[0], a / ss[1], b / ss[2], cc / ss[3]])
    x = _select(masks, [a / ss[0], 0.25 * ss[1], p01 / ss[2], p02 / ss[3]])
    y = _select(masks, [b / ss[0], p01 / ss[1], 0.25 * ss[2], p12 / ss[3]])
    z = _select(masks, [cc / ss[0], p02 / ss[1], p12 / ss[2], 0.25 * ss[3]])
    s = _select(masks, ss)
    return (w, x, y, z), s, masks


def mat2axisangle_forward(mat: torch.Tensor) -> torch.Tensor:
    """(n,3,4) -> (n,6)   (.cu:190-264)."""
    (w, x, y, z), _, _ = _quaternion(mat)
    neg = w < 0
    sign = torch.where(neg, -torch.ones_like(w), torch.ones_like(w))
    w, x, y, z = w * sign, x * sign, y * sign, z * sign
    tmp = x * x + y * y + z * z
    si = torch.sqrt(tmp)
    theta = 2 * torch.atan2(si, w)
    big = tmp > EPS
    fac = torch.where(big, theta / torch.where(big, si, torch.ones_like(si)), 2.0 / w)
    rot = torch.stack([x * fac, y * fac, z * fac], -1)
    return torch.cat([rot, mat[:, :, 3]], -1)


def mat2axisangle_backward(mat: torch.Tensor, grad_ax: torch.Tensor) -> torch.Tensor:
    """mat (n,3,4), grad_ax (n,6) -> grad_mat (n,3,4)   (.cu:266-440).

    Mirrors the reference's formula, including its small-angle regularisation
    (``si + EPS`` in the denominators when ``tmp <= EPS``).
    """
    (w, x, y, z), s, masks = _quaternion(mat)
    neg = w < 0
    sign = torch.where(neg, -torch.ones_like(w), torch.ones_like(w))
    w, x, y, z = w * sign, x * sign, y * sign, z * sign
    g0, g1, g2 = grad_ax[:, 0], grad_ax[:, 1], grad_ax[:, 2]
    tmp = x * x + y * y + z * z
    si = torch.sqrt(tmp)
    theta = 2 * torch.atan2(si, w)
    big = tmp > EPS
    dot = x * g0 + y * g1 + z * g2
    si_d = torch.where(big, si, si + EPS)
    si_safe = torch.where(big, si, torch.ones_like(si))
    fac = torch.where(big, theta / si_safe, 2.0 / w)
    inv = 2 / (w * w + si * si)
    dw = -dot * inv
    t2 = (w * inv - fac) / si_d
    dx = dot * t2 * (x / si_d) + fac * g0
    dy = dot * t2 * (y / si_d) + fac * g1
    dz = dot * t2 * (z / si_d) + fac * g2
    # undo sign fix
    w, x, y, z = w * sign, x * sign, y * sign, z * sign
    dw, dx, dy, dz = dw * sign, dx * sign, dy * sign, dz * sign
    q = (w, x, y, z)
    dq = (dw, dx, dy, dz)
    n = mat.shape[0]
    gm = torch.zeros(n, 3, 4, dtype=mat.dtype, device=mat.device)
    # per-branch: which quaternion component is the "0.25*s" one (index p), and the
    # (component -> (matrix entry pair, sign of second entry)) wiring.
    # entries: a = r21 - r12 ; b = r02 - r20 ; c = r10 - r01 ; p01, p02, p12 symmetric sums
    wiring = [
        # branch 0: w major; x<-a, y<-b, z<-c
        (0, {1: ("a",), 2: ("b",), 3: ("c",)}, (1, 1, 1)),
        # branch 1: x major; w<-a, y<-p01, z<-p02
        (1, {0: ("a",), 2: ("p01",), 3: ("p02",)}, (1, -1, -1)),
        # branch 2: y major; w<-b, x<-p01, z<-p12
        (2, {0: ("b",), 1: ("p01",), 3: ("p12",)}, (-1, 1, -1)),
        # branch 3: z major; w<-c, x<-p02, y<-p12
        (3, {0: ("c",), 1: ("p02",), 2: ("p12",)}, (-1, -1, 1)),
    ]
    ent = {
        "a": ((2, 1), (1, 2), -1.0),
        "b": ((0, 2), (2, 0), -1.0),
        "c": ((1, 0), (0, 1), -1.0),
        "p01": ((0, 1), (1, 0), 1.0),
        "p02": ((0, 2), (2, 0), 1.0),
        "p12": ((1, 2), (2, 1), 1.0),
    }
    for (p, wires, diag_sign), m in zip(wiring, masks):
        gb = torch.zeros_like(gm)
        acc = torch.zeros_like(w)
        for comp, (name,) in wires.items():
            (i0, j0), (i1, j1), sg = ent[name]
            gb[:, i0, j0] = dq[comp] / s
            gb[:, i1, j1] = sg * dq[comp] / s
            acc = acc + q[comp] * dq[comp]
        ds = (-acc / s + 0.25 * dq[p]) * (2 / s)
        for d in range(3):
            gb[:, d, d] = diag_sign[d] * ds
        gm = torch.where(m[:, None, None], gb, gm)
    gm[:, :, 3] = grad_ax[:, 3:]
    return gm
