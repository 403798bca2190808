"""Aggregation pass of the hash-grid backward with pieces compiled out (NESVOR_ABLATE bits in csrc/hashgrid.hip):
builds one library per variant on the GPU box and times the pass on the PSF-cloud distribution (N = 2^20).
Results of the ablated variants are wrong by construction; this is a timing tool.

    python tools/ablate_hashgrid.py            # on a gfx950 box
"""
import os, subprocess, sys
os.environ.setdefault("NESVOR_HASHGRID_QUEUE", "worst")  # timing tool: worst-case queues from the first call
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = [(0, "full kernel"), (1, "Morton sort done twice (=> cost of one sort)"), (2, "without the DPP run scan"),
            (4, "without table insertion (=> empty drain, no records)"), (8, "without record writes"),
            (16, "without the queue-reservation atomics"), (24, "without reservations and record writes")]
if len(sys.argv) == 1:
    src = os.path.join(ROOT, "nesvor_amd", "csrc")
    out = "/tmp/nesvor_ablate"
    os.makedirs(out, exist_ok=True)
    others = [os.path.join(ROOT, "nesvor_amd", "lib", f) for f in os.listdir(os.path.join(ROOT, "nesvor_amd", "lib")) if f.endswith(".o") and f != "hashgrid.o"]
    procs = []
    for bits, _ in VARIANTS:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=off",
               f"-DNESVOR_ABLATE={bits}", "-I", os.path.join(ROOT, "include"), "-c", os.path.join(src, "hashgrid.hip"), "-o", f"{out}/hg{bits}.o"]
        procs.append(subprocess.Popen(cmd, stderr=subprocess.DEVNULL))
    for p in procs:
        assert p.wait() == 0
    for bits, _ in VARIANTS:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", f"{out}/hg{bits}.o", *others, "-o", f"{out}/lib{bits}.so"])
    for bits, name in VARIANTS:
        r = subprocess.run([sys.executable, __file__, "run"], env={**os.environ, "NESVOR_HIP_LIB": f"{out}/lib{bits}.so"}, capture_output=True, text=True)
        print(f"{name:58s} {r.stdout.strip()}", flush=True)
else:
    sys.path.insert(0, ROOT)
    import ctypes
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
    table = ((torch.rand(spec.n_params, generator=torch.Generator().manual_seed(1337)) * 2 - 1) * 1e-4).to(dev)
    dy = torch.randn(32, N, device=dev); gt = torch.zeros_like(table); gu = torch.empty(N, 3, device=dev)
    ws = _workspace(spec, N, dev)
    _SCALE = __import__('nesvor_amd.encoding', fromlist=['queue_sizer']).queue_sizer(spec, N, dev).scale
    lib = _lib.load()
    run = lambda: lib.nesvor_hashgrid_backward(ctypes.byref(spec.c_struct), _lib.ptr(u), _lib.ptr(table), _lib.ptr(dy), _lib.ptr(gt), _lib.ptr(gu), N, 1, _lib.ptr(ws), 1, _SCALE, _lib.stream_ptr())
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(20): run()
    e.record(); torch.cuda.synchronize()
    print(f"{s.elapsed_time(e) / 20:.3f} ms")
