"""Per-level record counts of the hash-grid backward's queues after one aggregation pass (N = 2^20, headline grid):
    python tools/queue_stats.py          PSF clouds
    python tools/queue_stats.py U [0|1]  uniform points through the unclustered variant (layout 0 / 1)"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NESVOR_HASHGRID_QUEUE", "worst")
import torch
from nesvor_amd import _lib
from nesvor_amd.encoding import _workspace, queue_sizer
from nesvor_amd.grid import HashGridSpec
dev = torch.device("cuda:0")
spec = HashGridSpec(16, 2, 19, 9, 1.26)
N = 1 << 20
UNIFORM = len(sys.argv) > 1 and sys.argv[1] == "U"
LAYOUT = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = torch.Generator().manual_seed(0)
c = torch.rand(4096, 1, 3, generator=g) * 110 + 10
u = ((c + torch.randn(4096, 256, 3, generator=g) * torch.tensor([0.77, 0.77, 1.27])).reshape(-1, 3) / 130.0).clamp(0, 1).contiguous().to(dev)
if UNIFORM:
    u = torch.rand(N, 3, generator=g).to(dev)
hints = (_lib.LAYOUT_UNCLUSTERED | _lib.LAYOUT_DY_SCRATCH) if UNIFORM else 0
table = torch.zeros(spec.n_params, device=dev); dy = torch.randn((32, N) if LAYOUT == 1 else (N, 32), device=dev); gt = torch.zeros_like(table)
sizer = queue_sizer(spec, N, dev, not UNIFORM)
ws = _workspace(spec, N, dev, sizer, LAYOUT | hints)
lib = _lib.load()
err = lib.nesvor_hashgrid_backward(ctypes.byref(spec.c_struct), _lib.ptr(u), _lib.ptr(table), _lib.ptr(dy), _lib.ptr(gt), None, N, LAYOUT | hints, _lib.ptr(ws), 1, sizer.scale, _lib.stream_ptr())
torch.cuda.synchronize()
assert err == 0
STRIDE, SUBS = 4096, 8
off = lib.nesvor_hashgrid_backward_overflow_offset(_lib.ptr(ws)) - (STRIDE - 32) * 4  # start of the tail region this backward used
tails = ws[off: off + SUBS * STRIDE * 4].view(torch.int32).cpu().reshape(SUBS, STRIDE)
b = tot = 0
for li, lv in enumerate(spec.levels):
    nc = (lv.size + 4095) // 4096
    t = tails[:, b:b + nc].sum(0)
    print(f"level {li:2d} res {lv.res:3d} chunks {nc:3d} records {int(t.sum()):9d}  per point {int(t.sum()) / N:6.3f}  overflowed {int(tails[0, STRIDE - 32 + li])}")
    tot += int(t.sum()); b += nc
print("total records", tot, "=", round(tot / N, 2), "per point;", tot * 12 / 1e6, "MB")
