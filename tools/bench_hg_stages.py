"""Aggregate / owner launch times of the hash-grid backward (PSF-cloud and uniform points), N = 2^20."""
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
uP = ((c + torch.randn(4096, 256, 3, generator=g) * torch.tensor([0.77, 0.77, 1.27])).reshape(-1, 3) / 130.0).clamp(0, 1).contiguous().to(dev)
uU = torch.rand(N, 3, generator=g).to(dev)
table = ((torch.rand(spec.n_params, generator=torch.Generator().manual_seed(1337)) * 2 - 1) * 1e-4).to(dev)
dy = torch.randn(32, N, device=dev); gt = torch.zeros_like(table); gu = torch.empty(N, 3, device=dev)
ws = _workspace(spec, N, dev)
_SCALE = __import__('nesvor_amd.encoding', fromlist=['queue_sizer']).queue_sizer(spec, N, dev).scale
lib = _lib.load()
def run(u, stage, inp=True):
    return lib.nesvor_hashgrid_backward(ctypes.byref(spec.c_struct), _lib.ptr(u), _lib.ptr(table), _lib.ptr(dy), _lib.ptr(gt), _lib.ptr(gu) if inp else None, N, 1, _lib.ptr(ws), stage, _SCALE, _lib.stream_ptr())
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name, u in (("P", uP), ("U", uU)):
    run(u, 1); torch.cuda.synchronize()
    nb = 0
    tails = ws[:8 * 4096 * 4].view(torch.int32).cpu()  # kSubQueues x kTailStride counters
    tot = int(tails.sum())
    tn = timeit(lambda: run(u, 1, False))
    print(f"{name}: aggregate without input grad {tn:.3f} ms")
    ta = timeit(lambda: run(u, 1)); to = timeit(lambda: run(u, 2)); t3 = timeit(lambda: run(u, 3))
    print(f"{name}: records {tot/1e6:.2f} M  aggregate {ta:.3f} ms  owner {to:.3f} ms  both {t3:.3f} ms", flush=True)
