"""Per-level record counts of the hash-grid backward queues (phase 1 only) on the PSF-cloud distribution."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NESVOR_HASHGRID_QUEUE", "worst")  # timing tool: worst-case queues from the first call
import torch
from nesvor_amd import _lib
from nesvor_amd.encoding import _workspace
from nesvor_amd.grid import HashGridSpec
dev = torch.device("cuda:0")
spec = HashGridSpec(16, 2, 19, 9, 1.26)
N = 1 << 20
g = torch.Generator().manual_seed(0)
c = torch.rand(4096, 1, 3, generator=g) * 110 + 10
u = ((c + torch.randn(4096, 256, 3, generator=g) * torch.tensor([0.77, 0.77, 1.27])).reshape(-1, 3) / 130.0).clamp(0, 1).contiguous().to(dev)
table = torch.zeros(spec.n_params, device=dev); dy = torch.randn(32, N, device=dev); gt = torch.zeros_like(table)
ws = _workspace(spec, N, dev)
_SCALE = __import__('nesvor_amd.encoding', fromlist=['queue_sizer']).queue_sizer(spec, N, dev).scale
lib = _lib.load()
err = lib.nesvor_hashgrid_backward(ctypes.byref(spec.c_struct), _lib.ptr(u), _lib.ptr(table), _lib.ptr(dy), _lib.ptr(gt), None, N, 1, _lib.ptr(ws), 1, _SCALE, _lib.stream_ptr())
torch.cuda.synchronize()
tails = ws[:8 * 4096 * 4].view(torch.int32).cpu()
b = 0
tot = 0
for li, lv in enumerate(spec.levels):
    nc = (lv.size + 8191) // 8192
    t = tails[b:b + nc]
    print(f"level {li:2d} res {lv.res:3d} chunks {nc:3d} records {int(t.sum()):9d} max/bucket {int(t.max()):7d} per-pixel {int(t.sum())/4096:7.1f}")
    tot += int(t.sum()); b += nc
print("total records", tot, "=", tot / 4096, "per pixel;", tot * 12 / 1e6, "MB")
