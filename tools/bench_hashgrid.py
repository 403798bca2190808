"""Micro-benchmark of the hash-grid kernels at BASELINE size (N = 2^20, L=16, F=2, T=2^19)
on the two mandated point distributions: U = uniform, P = PSF clouds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
import torch
from nesvor_amd.encoding import hashgrid_backward, hashgrid_forward
from nesvor_amd.grid import HashGridSpec

dev = torch.device("cuda:0")
spec = HashGridSpec(16, 2, 19, 9, 1.26)
N = 1 << 20
L = 16
g = torch.Generator().manual_seed(0)
uU = torch.rand(N, 3, generator=g)
c = torch.rand(4096, 1, 3, generator=g) * 110 + 10
uP = ((c + torch.randn(4096, 256, 3, generator=g) * torch.tensor([0.77, 0.77, 1.27])).reshape(-1, 3) / 130.0).clamp(0, 1)
table = ((torch.rand(spec.n_params, generator=torch.Generator().manual_seed(1337)) * 2 - 1) * 1e-4).to(dev)
FWD_B = N * (12 + 64 * L + 8 * L)
BWD_B = N * (12 + 8 * L + 64 * L)
BWDI_B = BWD_B + N * (64 * L + 12)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for name, u in (("U", uU), ("P", uP)):
    u = u.contiguous().to(dev)
    for layout in (0, 1):
        dy = torch.randn((N, 32) if layout == 0 else (32, N), device=dev)
        gt = torch.zeros_like(table)
        tf = timeit(lambda: hashgrid_forward(spec, u, table, layout, clustered=name == "P"))
        for method in ("owner", "owner, points as given", "atomic"):
            if method == "owner, points as given" and name == "P":
                continue
            n = 3 if method == "atomic" else 20
            # "owner": with the hint a caller of this distribution gives (U: clustered=False, the backward orders the points by cell)
            cl = name == "P" or method == "owner, points as given"
            m = "atomic" if method == "atomic" else "owner"
            for _ in range(8):  # (queue capacities settle)
                hashgrid_backward(spec, u, table, dy, gt, True, layout, m, clustered=cl)
                torch.cuda.synchronize()
            tb = timeit(lambda: hashgrid_backward(spec, u, table, dy, gt, False, layout, m, clustered=cl), n)
            tbi = timeit(lambda: hashgrid_backward(spec, u, table, dy, gt, True, layout, m, clustered=cl), n)
            print(f"{name} layout={layout} {method}: fwd {tf:.3f} ms ({FWD_B/tf/1e6:.0f} GB/s alg)  bwd {tb:.3f} ms ({BWD_B/tb/1e6:.0f} GB/s)"
                  f"  bwd+input {tbi:.3f} ms ({BWDI_B/tbi/1e6:.0f} GB/s)  fwd+bwd frac of 8TB/s: {(FWD_B+BWD_B)/((tf+tb)*1e-3)/8e12:.3f}", flush=True)
