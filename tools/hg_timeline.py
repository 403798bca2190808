"""Where a workgroup of the hash-grid aggregation pass spends its time, phase by phase, without a profiler: builds
csrc/hashgrid.hip with -DNESVOR_HG_TIMELINE=1 (thread 0 of the first 64 workgroups stamps s_memtime - 100 MHz - at the pass's
barriers), runs the pass on PSF clouds at N = 2^20 and N = 2^17 and prints the mean time between consecutive marks.

    python tools/hg_timeline.py            # on a gfx950 box
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SITES = {0: "entry", 1: "samples sorted", 2: "bounding box / max|dy| published", 3: "lattice boxes + round schedule", 4: "first round's table copy in place",
         5: "first round's inserts issued", 6: "box of the samples + scale (all threads)", 7: "per-level boxes, window bits, parameters (lane = level)",
         8: "round schedule (wave 0)", 0x0F: "last records written"}
def site_name(s):
    if s in SITES: return SITES[s]
    hi, lv = s & 0xF0, s & 0x0F
    if s & 0xE0 in (0x20, 0x40, 0x60, 0x80, 0xA0):
        hi, lv = s & 0xE0, s & 0x1F
        return {0x20: "inserts complete (barrier)", 0x40: "prev. records written + round drained", 0x60: "drain barrier", 0x80: "reservation + next round's inserts issued",
                0xA0: "reservation returned"}[hi] + f", round from level {lv}"
    return {0xC0: "hashed level: slots claimed + adds issued", 0xD0: "hashed level: insert barrier", 0xE0: "single-level round: drained / ranked",
            0xF0: "single-level round: reservation returned + next level prepared"}.get(hi, hex(hi)) + f", level {lv}"
if len(sys.argv) == 1:
    out = "/tmp/nesvor_tl"; os.makedirs(out, exist_ok=True)
    libdir = os.path.join(ROOT, "nesvor_amd", "lib")
    others = [os.path.join(libdir, f) for f in os.listdir(libdir) if f.endswith(".o") and f != "hashgrid.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=off", "-w",
                           "-DNESVOR_HG_TIMELINE=1", "-I", os.path.join(ROOT, "include"), "-c", os.path.join(ROOT, "nesvor_amd", "csrc", "hashgrid.hip"), "-o", f"{out}/hg.o"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", f"{out}/hg.o", *others, "-o", f"{out}/libtl.so"])
    for n_clouds in (4096, 512):
        subprocess.check_call([sys.executable, __file__, "run", str(n_clouds)], env={**os.environ, "NESVOR_HIP_LIB": f"{out}/libtl.so", "NESVOR_HASHGRID_QUEUE": "worst"})
else:
    sys.path.insert(0, ROOT)
    import ctypes
    import numpy as np
    import torch
    from nesvor_amd import _lib
    from nesvor_amd.encoding import _workspace, queue_sizer
    from nesvor_amd.grid import HashGridSpec
    n_clouds = int(sys.argv[2]); N = n_clouds * 256
    dev = torch.device("cuda:0")
    spec = HashGridSpec(16, 2, 19, 9, 1.26)
    g = torch.Generator().manual_seed(0)
    c = torch.rand(n_clouds, 1, 3, generator=g) * 110 + 10
    u = ((c + torch.randn(n_clouds, 256, 3, generator=g) * torch.tensor([0.77, 0.77, 1.27])).reshape(-1, 3) / 130.0).clamp(0, 1).contiguous().to(dev)
    table = ((torch.rand(spec.n_params, generator=torch.Generator().manual_seed(1337)) * 2 - 1) * 1e-4).to(dev)
    dy = torch.randn(32, N, device=dev); gt = torch.zeros_like(table); gu = torch.empty(N, 3, device=dev)
    bound = dy.abs().max().reshape(1)
    sizer = queue_sizer(spec, N, dev); ws = _workspace(spec, N, dev, sizer)
    lib = _lib.load()
    run = lambda: lib.nesvor_hashgrid_backward_bounded(ctypes.byref(spec.c_struct), _lib.ptr(u), _lib.ptr(table), _lib.ptr(dy), _lib.ptr(gt), _lib.ptr(gu), N, 1,
                                                       _lib.ptr(ws), 1, 0, 16, sizer.scale, _lib.ptr(bound), _lib.stream_ptr())
    for _ in range(3): assert run() == 0
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record(); run(); e.record(); torch.cuda.synchronize()
    buf = np.zeros((64, 96), dtype=np.uint64)
    fn = lib.nesvor_debug_hg_timeline; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p]
    assert fn(buf.ctypes.data) == 0
    site, t = (buf >> np.uint64(56)).astype(np.int64), (buf & np.uint64((1 << 56) - 1)).astype(np.int64)
    n_marks = int((t[0] != 0).sum())
    print(f"\n== N = {N} points ({n_clouds} clouds): the launch took {s.elapsed_time(e) * 1e3:.1f} us; workgroups 0..63 (the first wave of workgroups), {n_marks} marks each; us at 100 MHz")
    life = (t[:, :n_marks].max(1) - t[:, 0]) / 100.0
    print(f"   workgroup lifetime: mean {life.mean():.1f} us, min {life.min():.1f}, max {life.max():.1f}")
    same = all((site[w, :n_marks] == site[0, :n_marks]).all() for w in range(64) if (t[w] != 0).sum() == n_marks)
    if not same: print("   (the workgroups took different paths: sites of workgroup 0 shown, deltas averaged over the workgroups with the same path)")
    ok = [w for w in range(64) if (t[w] != 0).sum() == n_marks and (site[w, :n_marks] == site[0, :n_marks]).all()]
    d = (t[ok, 1:n_marks] - t[ok, : n_marks - 1]) / 100.0
    acc = 0.0
    for k in range(n_marks - 1):
        acc += d[:, k].mean()
        print(f"   {d[:, k].mean():6.2f} us  (to {acc:6.1f})  -> {site_name(int(site[0, k + 1]))}")
