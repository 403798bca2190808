"""A few passes of the hash-grid owner-backward at N=2^20 (for rocprofv3 --pmc / --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NESVOR_HASHGRID_QUEUE", "worst")  # timing tool: worst-case queues from the first call
import torch
from nesvor_amd.encoding import hashgrid_backward, hashgrid_forward
from nesvor_amd.grid import HashGridSpec
dev = torch.device("cuda:0")
spec = HashGridSpec(16, 2, 19, 9, 1.26)
N = 1 << 20
g = torch.Generator().manual_seed(0)
dist = sys.argv[1] if len(sys.argv) > 1 else "P"
if dist == "U":
    u = torch.rand(N, 3, generator=g)
else:
    c = torch.rand(4096, 1, 3, generator=g) * 110 + 10
    u = ((c + torch.randn(4096, 256, 3, generator=g) * torch.tensor([0.77, 0.77, 1.27])).reshape(-1, 3) / 130.0).clamp(0, 1)
u = u.contiguous().to(dev)
table = ((torch.rand(spec.n_params, generator=torch.Generator().manual_seed(1337)) * 2 - 1) * 1e-4).to(dev)
dy = torch.randn(32, N, device=dev)
gt = torch.zeros_like(table)
for _ in range(3):
    hashgrid_forward(spec, u, table, 1, clustered=(dist != "U"))  # the kernel the training step launches: per-cloud for PSF clouds
    hashgrid_backward(spec, u, table, dy, gt, True, 1, "owner")
torch.cuda.synchronize()
