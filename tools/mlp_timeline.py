"""Who waits for whom in a wave pair of the wave-specialised MLP backward, without a profiler: builds csrc/mlp.hip with
-DNESVOR_MLP_TIMELINE=1 (lane 0 of every wave of the first 64 workgroups records s_memtime - 100 MHz - around its group loop, and
the ticks spent inside pair_sync() and behind its prefetch wait), runs the density-network and sigma-network shapes at N = 2^20
and prints, per role, the loop time and the share spent waiting for the partner / for HBM.

    python tools/mlp_timeline.py [extra -D flags]           # on a gfx950 box
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 1 or sys.argv[1] != "run":
    out = "/tmp/nesvor_mtl"; os.makedirs(out, exist_ok=True)
    libdir = os.path.join(ROOT, "nesvor_amd", "lib")
    others = [os.path.join(libdir, f) for f in os.listdir(libdir) if f.endswith(".o") and f != "mlp.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=off", "-w",
                           "-DNESVOR_MLP_TIMELINE=1", *sys.argv[1:], "-I", os.path.join(ROOT, "include"), "-c",
                           os.path.join(ROOT, "nesvor_amd", "csrc", "mlp.hip"), "-o", f"{out}/mlp.o"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", f"{out}/mlp.o", *others, "-o", f"{out}/libmtl.so"])
    subprocess.check_call([sys.executable, __file__, "run"], env={**os.environ, "NESVOR_HIP_LIB": f"{out}/libmtl.so"})
else:
    sys.path.insert(0, ROOT)
    import ctypes
    import numpy as np
    import torch
    from nesvor_amd import _lib
    from nesvor_amd.mlp import fused_mlp
    from nesvor_amd.models import build_network
    dev = torch.device("cuda:0")
    N = 1 << 20
    lib = _lib.load()
    fn = lib.nesvor_debug_mlp_timeline; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p]
    for name, k_a, k_b, out_dim in (("density 32 -> 64 -> 64 -> 16", 0, 32, 16), ("sigma 16 | 15 -> 64 -> 64 -> 1", 16, 15, 1)):
        torch.manual_seed(0)
        net = build_network(n_input_dims=k_a + k_b, n_output_dims=out_dim, activation="ReLU", output_activation="None", n_neurons=64,
                            n_hidden_layers=2, dtype=torch.float32).to(dev)
        xb = torch.randn(k_b + 1, N, device=dev, requires_grad=True)
        xa = torch.randn(N // 256, k_a, device=dev, requires_grad=True) if k_a else None
        w = torch.randn(out_dim, N, device=dev)
        for _ in range(3):
            y = fused_mlp(net, xa, xb, 1 if k_a else 0, k_b, 256)
            loss = (y * w).sum()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(True), torch.cuda.Event(True)
            s.record(); loss.backward(); e.record(); torch.cuda.synchronize()
        buf = np.zeros((64, 8, 4), dtype=np.uint64)
        assert fn(buf.ctypes.data) == 0
        t = buf.astype(np.int64)
        loop = (t[:, :, 1] - t[:, :, 0]) / 100.0
        sync, wait = t[:, :, 2] / 100.0, t[:, :, 3] / 100.0
        print(f"\n== {name}, N = 2^20: autograd backward {s.elapsed_time(e) * 1e3:.1f} us (incl. the sums); workgroups 0..63, us at 100 MHz")
        for role, nm in ((0, "chain waves (dX)"), (1, "dW waves")):
            sl = slice(4 * role, 4 * role + 4)
            print(f"   {nm:18s}: loop {loop[:, sl].mean():7.1f} us (min {loop[:, sl].min():.1f}, max {loop[:, sl].max():.1f}) | in pair_sync {sync[:, sl].mean():6.1f} us "
                  f"({100 * sync[:, sl].sum() / loop[:, sl].sum():.0f} %) | prefetch wait {wait[:, sl].mean():6.1f} us ({100 * wait[:, sl].sum() / loop[:, sl].sum():.0f} %)")
        first = t[:, :, 0].min(); last = t[:, :, 1].max()
        print(f"   first loop start -> last loop end over these workgroups: {(last - first) / 100.0:.1f} us")
