"""Ablation timing of the hash-grid backward aggregation kernel (debug entry)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
table = torch.zeros(spec.n_params, device=dev)
dy = torch.randn(32, N, device=dev)
gt = torch.zeros_like(table)
ws = _workspace(spec, N, dev)
lib = _lib.load()
fn = lib.nesvor_hashgrid_backward_debug
fn.restype = ctypes.c_int
names = {0: "adaptive add + DPP merge", 32: "always CAS + DPP merge", 1: "no-merge", 2: "no-insert(keep merge)", 3: "no-merge,no-insert", 4: "no-flush", 6: "no-insert,no-flush", 7: "nothing (locate+dy only)", 8: "ds_add_f32 + DPP", 16: "CAS-add + bpermute", 24: "ds_add_f32 + bpermute (old)", 9: "ds_add_f32, no merge"}
for nm, u in (("P", uP), ("U", uU)):
    for var in (0, 8, 32, 2, 3, 4):
        for owner in ((0, 1, 2) if var == 0 else (0,)):
            def run():
                e = fn(ctypes.byref(spec.c_struct), _lib.ptr(u), _lib.ptr(table), _lib.ptr(dy), _lib.ptr(gt), ctypes.c_int64(N), _lib.ptr(ws), var, owner, _lib.stream_ptr())
                assert e == 0, e
            for _ in range(2): run()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(True), torch.cuda.Event(True)
            s.record()
            for _ in range(5): run()
            e.record(); torch.cuda.synchronize()
            print(f"{nm} var={var} ({names[var]}) owner={owner}: {s.elapsed_time(e)/5:.3f} ms", flush=True)
