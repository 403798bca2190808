"""A/B builds of csrc/hashgrid.hip on the GPU box: every variant = extra -D flags; for each one the aggregation pass,
the owner pass and the forward are timed on the PSF-cloud distribution (N = 2^20, feature-major) and the table gradient
is checked against the atomic kernel of the same build.

    python tools/hg_variants.py base: fixed32:-DNESVOR_FIXED32=1
"""
import os, subprocess, sys
os.environ.setdefault("NESVOR_HASHGRID_QUEUE", "worst")  # timing tool: worst-case queues from the first call
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "run":
    variants = [a.split(":", 1) for a in sys.argv[1:]]
    src = os.path.join(ROOT, "nesvor_amd", "csrc")
    out = "/tmp/nesvor_variants"
    os.makedirs(out, exist_ok=True)
    libdir = os.path.join(ROOT, "nesvor_amd", "lib")
    others = [os.path.join(libdir, f) for f in os.listdir(libdir) if f.endswith(".o") and f != "hashgrid.o"]
    procs = []
    for name, flags in variants:
        source = os.path.join(src, "hashgrid.hip")
        flist = [f for f in flags.split(",") if f]
        for f in list(flist):
            if f.startswith("src="):  # another version of the source file (e.g. an older kernel written out of the history: git show <rev>:nesvor_amd/csrc/hashgrid.hip > /tmp/old.hip)
                flist.remove(f)
                source = f"{out}/{name}.hip"
                text = open(os.path.join(ROOT, f[4:])).read().replace('"../../include/nesvor_hip.h"', '"nesvor_hip.h"').replace('"common.h"', f'"{src}/common.h"')
                open(source, "w").write(text)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=off",
               *flist, "-I", os.path.join(ROOT, "include"), "-c", source, "-o", f"{out}/{name}.o"]
        procs.append(subprocess.Popen(cmd, stderr=subprocess.DEVNULL))
    for p in procs:
        assert p.wait() == 0
    for name, _ in variants:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", f"{out}/{name}.o", *others, "-o", f"{out}/lib{name}.so"])
    for name, flags in variants:
        r = subprocess.run([sys.executable, __file__, "run"], env={**os.environ, "NESVOR_HIP_LIB": f"{out}/lib{name}.so"}, capture_output=True, text=True)
        print(f"{name:14s} {flags:34s} {r.stdout.strip()} {r.stderr.strip()[-300:]}", flush=True)
else:
    sys.path.insert(0, ROOT)
    import ctypes
    import torch
    from nesvor_amd import _lib
    from nesvor_amd.encoding import _workspace, hashgrid_backward, hashgrid_forward
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
    def run(stage, gin=True):
        return lib.nesvor_hashgrid_backward(ctypes.byref(spec.c_struct), _lib.ptr(u), _lib.ptr(table), _lib.ptr(dy), _lib.ptr(gt), _lib.ptr(gu if gin else None), N, 1, _lib.ptr(ws), stage, _SCALE, _lib.stream_ptr())
    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n
    t_agg = timeit(lambda: run(1)); t_agg_noin = timeit(lambda: run(1, False)); t_own = timeit(lambda: run(2))
    t_fwd = timeit(lambda: hashgrid_forward(spec, u, table, 1, clustered=True))
    def truth64():
        """fp64 scatter of the fp32 corner weights (the kernels' own fmaf / floor / fp32 weights, products and sums in fp64)."""
        import numpy as np
        gt = torch.zeros(spec.n_entries, 2, dtype=torch.float64, device=dev)
        P1, P2, M = 2654435761, 805459861, 0xFFFFFFFF
        for li, lv in enumerate(spec.levels):
            pos = (u.double() * float(np.float32(lv.scale)) + 0.5).float()
            cell = torch.floor(pos)
            w = (pos - cell)
            g = cell.long()
            d = dy[2 * li : 2 * li + 2].t().double()
            for c in range(8):
                cx, cy, cz = c & 1, (c >> 1) & 1, c >> 2
                wk = ((w[:, 0] if cx else 1 - w[:, 0]) * (w[:, 1] if cy else 1 - w[:, 1])).float() * (w[:, 2] if cz else 1 - w[:, 2])
                x, y, z = g[:, 0] + cx, g[:, 1] + cy, g[:, 2] + cz
                idx = (((x & M) ^ ((y * P1) & M) ^ ((z * P2) & M)) % lv.size) if lv.hashed else ((x + y * lv.res + z * lv.res * lv.res) % lv.size)
                gt.index_add_(0, idx + lv.offset, (wk.float().double()[:, None] * d.float().double()).float().double())  # fp32 product w * dy, as every kernel forms it
        return gt.view(-1)
    g64 = truth64()
    def acc(gx):
        e = (gx.double() - g64).abs()
        nz = g64 != 0
        rel = (e[nz] / g64[nz].abs())
        return f"max {float(e.max() / g64.abs().max()):.1e} L2 {float(e.norm() / g64.norm()):.1e} rel-median {float(rel.median()):.1e} rel-p99 {float(rel.quantile(0.99)) if rel.numel() < 16e6 else float(rel[:16000000].quantile(0.99)):.1e}"
    g1, gu1 = hashgrid_backward(spec, u, table, dy, None, True, 1, "owner")
    g2, gu2 = hashgrid_backward(spec, u, table, dy, None, True, 1, "atomic")
    err = float((g1 - g2).abs().max() / g2.abs().max()); erru = float((gu1 - gu2).abs().max() / gu2.abs().max())
    # small gradients next to large ones: relative error of the entries below 1e-4 of the maximum
    small = (g2.abs() < 1e-4 * g2.abs().max()) & (g2 != 0)
    rel_small = float(((g1 - g2).abs()[small] / g2.abs()[small]).median()) if bool(small.any()) else 0.0
    # uniform points (nothing merges: the record stream and the owner pass carry everything)
    uu = torch.rand(N, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    t_uni = timeit(lambda: hashgrid_backward(spec, uu, table, dy, gt, True, 1, "owner"), 5)
    print(f"uniform points: backward {t_uni:.3f} ms", end="  ")
    print(f"error against fp64: owner [{acc(g1)}]  atomic fp32 [{acc(g2)}]")
    print(f"aggregate {t_agg:.3f} ms (no input grad {t_agg_noin:.3f})  owner {t_own:.3f}  fwd {t_fwd:.3f}  | owner vs atomic: table {err:.1e} (median rel of small entries {rel_small:.1e}), grad_u {erru:.1e}")
