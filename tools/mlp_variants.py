"""A/B builds of csrc/mlp.hip on the GPU box: every variant = extra -D flags; for each one the forward (with and without
saving the hidden activations) and the fused backward of the two networks of the bench config are timed at N = 2^20 in
the default (split-bf16) evaluation mode.

    python tools/mlp_variants.py base: hidden_in_cache:-DNESVOR_MLP_ABLATE=1 all_in_cache:-DNESVOR_MLP_ABLATE=3
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "run":
    variants = [a.split(":", 1) for a in sys.argv[1:]]
    src = os.path.join(ROOT, "nesvor_amd", "csrc")
    out = "/tmp/nesvor_mlp_variants"
    os.makedirs(out, exist_ok=True)
    libdir = os.path.join(ROOT, "nesvor_amd", "lib")
    others = [os.path.join(libdir, f) for f in os.listdir(libdir) if f.endswith(".o") and f != "mlp.o"]
    procs = []
    for name, flags in variants:
        source = os.path.join(src, "mlp.hip")
        flist = [f for f in flags.split(",") if f]
        for f in list(flist):
            if f.startswith("src="):  # another version of the source file (e.g. an older kernel written out of the history)
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
        print(f"{name:16s} {flags:30s} {r.stdout.strip()} {r.stderr.strip()[-300:] if r.returncode else ''}", flush=True)
else:
    sys.path.insert(0, ROOT)
    import torch
    from nesvor_amd import mlp
    from nesvor_amd.models import build_network
    dev = torch.device("cuda:0")
    N, S = 1 << 20, 256
    torch.manual_seed(0)
    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n
    msg = []
    for name, k_a, k_b, rows, row0, out in (("density", 0, 32, 32, 0, 16), ("sigma", 16, 15, 16, 1, 1)):
        net = build_network(n_input_dims=k_a + k_b, n_output_dims=out, activation="ReLU", output_activation="None", n_neurons=64, n_hidden_layers=2, dtype=torch.float32).to(dev)
        L = mlp.linear_layers(net)
        W, Bs = [l.weight.detach() for l in L], [l.bias.detach() for l in L]
        xa = torch.randn(N // S, k_a, device=dev) if k_a else None
        xb = torch.randn(rows, N, device=dev)
        dy = torch.randn(out, N, device=dev)
        y, saved = mlp.forward_raw(W, Bs, xa, xb, row0, k_b, S, True)
        dxb = torch.empty(k_b, N, device=dev)
        tf = timeit(lambda: mlp.forward_raw(W, Bs, xa, xb, row0, k_b, S, True))
        tf0 = timeit(lambda: mlp.forward_raw(W, Bs, xa, xb, row0, k_b, S, False))
        tb = timeit(lambda: mlp.backward_raw(W, Bs, xa, xb, dy, saved, row0, k_b, S, dxb, k_a > 0))
        msg.append(f"{name}: fwd {tf:.3f} (no save {tf0:.3f}) bwd {tb:.3f} ms")
    print(" | ".join(msg))
