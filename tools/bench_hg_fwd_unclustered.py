"""Round 6: the forward on input that is NOT a batch of PSF clouds, N = 2^20, headline grid (L = 16, F = 2, T = 2^19):
uniform points (SURVEY 8d's second distribution) and a raster-ordered voxel lattice (one point per voxel: the reference's
--no-output-psf inference, nesvor/nesvor/sample.py:29) through every forward path:
    level   one block per (256 points, level) on the points as given (rounds 1-5)
    cloud   the per-cloud kernel on the points as given
    sorted  points ordered by coarse lattice cell first (nesvor_hashgrid_forward_unclustered, round 6)
python tools/bench_hg_fwd_unclustered.py [--stages]   (--stages: rocprof-free breakdown by timing prefixes is not possible; use
rocprofv3 --kernel-trace --stats on this script for the per-kernel times)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nesvor_amd import _lib, encoding
from nesvor_amd.grid import HashGridSpec

dev = torch.device("cuda:0")
spec = HashGridSpec(16, 2, 19, 9, 1.26)
table = ((torch.rand(spec.n_params, generator=torch.Generator().manual_seed(1337)) * 2 - 1) * 1e-4).to(dev)
N = 1 << 20


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


g = torch.Generator().manual_seed(0)
uU = torch.rand(N, 3, generator=g).to(dev)
# 128 x 128 x 64 voxels at 0.8 mm inside a 130 mm box, x fastest (raster order), as sample_volume hands them over
zz, yy, xx = torch.meshgrid(torch.arange(64), torch.arange(128), torch.arange(128), indexing="ij")
uL = ((torch.stack([xx, yy, zz], -1).reshape(-1, 3).float() * 0.8 + 10.0) / 130.0).contiguous().to(dev)
assert uL.shape[0] == N
perm = torch.randperm(N, generator=g).to(dev)
uLs = uL[perm].contiguous()  # the same lattice in shuffled order (a masked, shuffled point list)
lib = _lib.load()


def raw(u, layout, hint):
    pe = torch.empty((N, 32) if layout == 0 else (32, N), dtype=torch.float32, device=dev)
    def f():
        err = lib.nesvor_hashgrid_forward(ctypes.byref(spec.c_struct), _lib.ptr(u), _lib.ptr(table), _lib.ptr(pe), N, layout | hint, _lib.stream_ptr())
        assert err == 0
    return f


for name, u in (("uniform", uU), ("lattice_raster", uL), ("lattice_shuffled", uLs)):
    for layout in (1, 0):
        t_level = timeit(raw(u, layout, 0))
        t_cloud = timeit(raw(u, layout, _lib.LAYOUT_CLUSTERED))
        encoding._FWD_MODE = "sorted"
        t_sorted = timeit(lambda: encoding.hashgrid_forward(spec, u, table, layout, clustered=False))
        encoding._FWD_MODE = ""
        print(f"N=2^20 {name:17s} layout={'feature-major' if layout else 'row-major    '}: level {t_level:.3f} ms | cloud as given {t_cloud:.3f} ms | sorted {t_sorted:.3f} ms", flush=True)
