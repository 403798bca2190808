"""Aggregation-pass time per level (P distribution, N = 2^20; `U [layout]`: uniform points through the unclustered variant, which
includes the counting sort in every launch): launches restricted to levels [0, l) via nesvor_hashgrid_backward_levels, differences
of consecutive prefixes."""
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
UNIFORM = len(sys.argv) > 1 and sys.argv[1] == "U"
LAYOUT = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if UNIFORM:
    u = torch.rand(N, 3, generator=g).to(dev)
table = ((torch.rand(spec.n_params, generator=torch.Generator().manual_seed(1337)) * 2 - 1) * 1e-4).to(dev)
dy = torch.randn((32, N) if LAYOUT == 1 else (N, 32), device=dev); gt = torch.zeros_like(table); gu = torch.empty(N, 3, device=dev)
HINTS = (_lib.LAYOUT_UNCLUSTERED | _lib.LAYOUT_DY_SCRATCH) if UNIFORM else 0
ws = _workspace(spec, N, dev, None, LAYOUT | HINTS)
_SCALE = __import__('nesvor_amd.encoding', fromlist=['queue_sizer']).queue_sizer(spec, N, dev).scale
lib = _lib.load()
def run(l0, l1, stage=1):
    return lib.nesvor_hashgrid_backward_levels(ctypes.byref(spec.c_struct), _lib.ptr(u), _lib.ptr(table), _lib.ptr(dy), _lib.ptr(gt), _lib.ptr(gu), N, LAYOUT | HINTS, _lib.ptr(ws), stage, l0, l1, _SCALE, _lib.stream_ptr())
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
prev = 0.0
for l in range(1, 17):
    t = timeit(lambda: run(0, l))
    lv = spec.levels[l - 1]
    print(f"levels [0,{l:2d}): {t:.3f} ms   level {l-1:2d} ({'hashed' if lv.hashed else 'dense '}, res {lv.res:4d}): +{t - prev:.3f}", flush=True)
    prev = t
