"""Fixed (size-independent) cost of the MLP launches: time at N = 2^13 .. 2^20 points."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nesvor_amd import mlp
from nesvor_amd.models import build_network
dev = torch.device("cuda:0")
S = 256
torch.manual_seed(0)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
net = build_network(n_input_dims=32, n_output_dims=16, activation="ReLU", output_activation="None", n_neurons=64, n_hidden_layers=2, dtype=torch.float32).to(dev)
L = mlp.linear_layers(net)
W, Bs = [l.weight.detach() for l in L], [l.bias.detach() for l in L]
for lg in (13, 16, 18, 19, 20):
    N = 1 << lg
    xb = torch.randn(32, N, device=dev); dy = torch.randn(16, N, device=dev)
    y, saved = mlp.forward_raw(W, Bs, None, xb, 0, 32, S, True)
    dxb = torch.empty(32, N, device=dev)
    tf = timeit(lambda: mlp.forward_raw(W, Bs, None, xb, 0, 32, S, True))
    tb = timeit(lambda: mlp.backward_raw(W, Bs, None, xb, dy, saved, 0, 32, S, dxb, False))
    print(f"N=2^{lg}: fwd {tf*1e3:.1f} us  bwd {tb*1e3:.1f} us", flush=True)
