"""Forward / fused-backward launch times of the two MLP shapes of the bench config at N = 2^20."""
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
    y, saved = mlp.forward_raw(W, Bs, xa, xb, row0, k_b, S, True)
    dxb = torch.empty(k_b, N, device=dev)
    tf = timeit(lambda: mlp.forward_raw(W, Bs, xa, xb, row0, k_b, S, True))
    tf0 = timeit(lambda: mlp.forward_raw(W, Bs, xa, xb, row0, k_b, S, False))
    print(f"{name}: fwd without saving activations {tf0:.3f} ms")
    tb = timeit(lambda: mlp.backward_raw(W, Bs, xa, xb, dy, saved, row0, k_b, S, dxb, k_a > 0))
    tfb = timeit(lambda: mlp.forward_raw(W, Bs, xa, xb, row0, k_b, S, True, bf16=True))
    yb, savedb = mlp.forward_raw(W, Bs, xa, xb, row0, k_b, S, True, bf16=True)
    tbb = timeit(lambda: mlp.backward_raw(W, Bs, xa, xb, dy, savedb, row0, k_b, S, dxb, k_a > 0, bf16=True))
    print(f"{name}: bf16-operand mode fwd {tfb:.3f} ms  bwd {tbb:.3f} ms")
    fl = 2 * N * (64 * (k_a + k_b) + 64 * 64 + 64 * out)
    print(f"{name}: fwd {tf:.3f} ms ({fl/tf/1e9:.1f} TF)  bwd {tb:.3f} ms ({2*fl/tb/1e9:.1f} TF)", flush=True)
