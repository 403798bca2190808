"""Uniform points through the unclustered hash-grid backward (counting sort + aggregation + owner), for rocprofv3:
    rocprofv3 --kernel-trace --stats -d out -- python tools/prof_hg_uniform.py [layout]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nesvor_amd.encoding import hashgrid_backward
from nesvor_amd.grid import HashGridSpec

layout = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
spec = HashGridSpec(16, 2, 19, 9, 1.26)
N = 1 << 20
u = torch.rand(N, 3, generator=torch.Generator().manual_seed(0)).to(dev)
table = ((torch.rand(spec.n_params, generator=torch.Generator().manual_seed(1337)) * 2 - 1) * 1e-4).to(dev)
dy = torch.randn((N, 32) if layout == 0 else (32, N), device=dev)
gt = torch.zeros_like(table)
for _ in range(10):
    hashgrid_backward(spec, u, table, dy, gt, True, layout, clustered=False)
    torch.cuda.synchronize()
s, e = torch.cuda.Event(True), torch.cuda.Event(True)
s.record()
for _ in range(20):
    hashgrid_backward(spec, u, table, dy, gt, True, layout, clustered=False)
e.record()
torch.cuda.synchronize()
print(f"uniform, layout {layout}: backward with input gradient {s.elapsed_time(e) / 20:.3f} ms")
