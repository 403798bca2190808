"""Randomised consistency soak of the hash-grid kernels: random sizes / distributions / feature counts / level splits,
owner-computes backward vs the per-corner atomic kernel, forward (LDS box cache) vs adjointness.  Exits non-zero on the
first mismatch.    python tools/stress_hashgrid.py [n_cases]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nesvor_amd.encoding import hashgrid_backward, hashgrid_forward
from nesvor_amd.grid import HashGridSpec
dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = torch.Generator().manual_seed(1234)
ri = lambda a, b: int(torch.randint(a, b + 1, (1,), generator=g))
for case in range(n_cases):
    F = [1, 2, 2, 2, 4, 8][ri(0, 5)]
    L = ri(2, 16)
    spec = HashGridSpec(L, F, ri(10, 19), ri(4, 12), [1.26, 1.3819, 1.5, 2.0][ri(0, 3)])
    kind = ri(0, 3)
    if kind == 0:
        N = ri(1, 200000); u = torch.rand(N, 3, generator=g)
    else:
        P = ri(1, 700); S = [16, 100, 256, 512][ri(0, 3)]
        c = torch.rand(P, 1, 3, generator=g)
        if kind == 2: c = c.round() * 0.98 + 0.01
        sig = torch.tensor([0.006, 0.006, 0.01]) * [0.3, 1.0, 3.0][ri(0, 2)]
        u = (c + torch.randn(P, S, 3, generator=g) * sig).reshape(-1, 3)
        if kind != 3: u = u.clamp(0, 1)
        N = u.shape[0]
    u = u.contiguous().to(dev)
    layout = ri(0, 1)
    table = (torch.randn(spec.n_params, generator=g) * 0.1).to(dev)
    E = spec.n_output_dims
    dy = torch.randn((N, E) if layout == 0 else (E, N), generator=g).to(dev)
    pe = hashgrid_forward(spec, u, table, layout)
    assert torch.equal(hashgrid_forward(spec, u, table, layout, clustered=True), pe), "cloud forward != level forward"
    g_atm, gu_atm = hashgrid_backward(spec, u, table, dy, None, True, layout, "atomic")
    from nesvor_amd.encoding import _workspace
    cl = bool(ri(0, 1))  # the clustered hint is free to be wrong: the unclustered variant orders any batch by cell first
    if ri(0, 3) == 0:
        perm = torch.randperm(N, generator=g).to(dev)  # a shuffled batch
        u = u[perm].contiguous(); dy = (dy[perm] if layout == 0 else dy[:, perm]).contiguous(); pe = (pe[perm] if layout == 0 else pe[:, perm]).contiguous()
        g_atm, gu_atm = hashgrid_backward(spec, u, table, dy, None, True, layout, "atomic")
    if L > 1 and ri(0, 1) and _workspace(spec, N, dev) is not None:  # (tables beyond the plan's 256 chunks per level use the atomic kernel)
        split = ri(1, L - 1)
        g_own, gu = hashgrid_backward(spec, u, table, dy, None, True, layout, "owner", levels=(split, L), clustered=cl)
        g_own, gu_own = hashgrid_backward(spec, u, table, dy, g_own, True, layout, "owner", levels=(0, split), grad_u=gu, first=False, clustered=cl)
    else:
        g_own, gu_own = hashgrid_backward(spec, u, table, dy, None, True, layout, "owner", clustered=cl)
    scale = float(g_atm.abs().max()) + 1e-12
    e1 = float((g_own - g_atm).abs().max()) / scale
    e2 = float((gu_own - gu_atm).abs().max()) / (float(gu_atm.abs().max()) + 1e-12)
    lhs = float((pe.double() * dy.double()).sum()); rhs = float((table.double() * g_own.double()).sum())
    e3 = abs(lhs - rhs) / (abs(lhs) + 1e-3 * N**0.5 + 1e-9)
    ok = e1 < 5e-4 and e2 < 5e-3 and e3 < 5e-4
    print(f"case {case:3d} F={F} L={L:2d} T=2^{spec.levels[-1].size.bit_length()-1 if spec.levels[-1].hashed else 0:2d} kind={kind} N={N:7d} layout={layout} {'clustered' if cl else 'unclust. '}: dW {e1:.1e} du {e2:.1e} adj {e3:.1e} {'ok' if ok else 'FAIL'}", flush=True)
    if not ok: sys.exit(1)
print("all ok")
