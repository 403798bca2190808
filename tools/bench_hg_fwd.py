"""Forward hash-grid kernel variants (NESVOR_HASHGRID_FWD = cloud | level | gather) on both distributions, N = 2^20 and 2^24.
    for m in cloud level gather; do NESVOR_HASHGRID_FWD=$m python tools/bench_hg_fwd.py; done"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nesvor_amd.encoding import hashgrid_forward
from nesvor_amd.grid import HashGridSpec
dev = torch.device("cuda:0")
spec = HashGridSpec(16, 2, 19, 9, 1.26)
table = ((torch.rand(spec.n_params, generator=torch.Generator().manual_seed(1337)) * 2 - 1) * 1e-4).to(dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
out = []
for logn in (20, 24):
    N = 1 << logn
    g = torch.Generator().manual_seed(0)
    uU = torch.rand(N, 3, generator=g).to(dev)
    P = N // 256
    c = torch.rand(P, 1, 3, generator=g) * 110 + 10
    uP = ((c + torch.randn(P, 256, 3, generator=g) * torch.tensor([0.77, 0.77, 1.27])).reshape(-1, 3) / 130.0).clamp(0, 1).contiguous().to(dev)
    for name, u in (("U", uU), ("P", uP)):
        for layout in (1, 0):
            out.append(f"N=2^{logn} {name} layout={layout}: {timeit(lambda: hashgrid_forward(spec, u, table, layout), 10):.3f} ms")
print(os.environ.get("NESVOR_HASHGRID_FWD", "cloud"), " | ".join(out))
