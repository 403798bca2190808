"""Forward / backward of the two MLP shapes in the three operand modes (0 fp32 MFMA, 1 bf16, 2 split) at N = 2^20:
launch times and the error of the output against an fp64 evaluation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for name, k_a, k_b, rows, row0, out in (("density", 0, 32, 32, 0, 16), ("sigma", 16, 15, 16, 1, 1)):
    net = build_network(n_input_dims=k_a + k_b, n_output_dims=out, activation="ReLU", output_activation="None", n_neurons=64, n_hidden_layers=2, dtype=torch.float32).to(dev)
    L = mlp.linear_layers(net)
    W, Bs = [l.weight.detach() for l in L], [l.bias.detach() for l in L]
    xa = torch.randn(N // S, k_a, device=dev) if k_a else None
    xb = torch.randn(rows, N, device=dev)
    dy = torch.randn(out, N, device=dev)
    # fp64 reference on a slice
    M = 1 << 14
    x = xb[row0:row0 + k_b, :M].t().double()
    if k_a: x = torch.cat([xa.double().repeat_interleave(S, 0)[:M], x], 1)
    h = x
    for i, (w, b) in enumerate(zip(W, Bs)):
        h = h @ w.double().t() + b.double()
        if i < len(W) - 1: h = h.relu()
    ref = h.t()
    dxb = torch.empty(k_b, N, device=dev)
    for mode in (0, 1, 2):
        y, saved = mlp.forward_raw(W, Bs, xa, xb, row0, k_b, S, True, mode)
        err = float((y[:, :M].double() - ref).abs().max() / ref.abs().max())
        tf = timeit(lambda: mlp.forward_raw(W, Bs, xa, xb, row0, k_b, S, True, mode))
        tf0 = timeit(lambda: mlp.forward_raw(W, Bs, xa, xb, row0, k_b, S, False, mode))
        tb = timeit(lambda: mlp.backward_raw(W, Bs, xa, xb, dy, saved, row0, k_b, S, dxb, k_a > 0, mode))
        print(f"{name} mode {mode}: fwd {tf:.3f} ms (no save {tf0:.3f})  bwd {tb:.3f} ms   max err vs fp64 / max|y| = {err:.2e}", flush=True)
