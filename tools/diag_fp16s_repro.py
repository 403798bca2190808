"""Diagnostic: run-to-run reproducibility of mode 4 (forward + backward) on the shapes of the test."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nesvor_amd import mlp
from nesvor_amd.models import build_network
dev = torch.device("cuda:0")
N, S = 8192, 256
for (k_a, k_b, b_row0, rows, out_dim, depth) in ((0, 16, 0, 16, 16, 2), (0, 16, 0, 16, 16, 1), (0, 32, 0, 32, 16, 2), (16, 15, 1, 16, 1, 2)):
    torch.manual_seed(3 + depth)
    net = build_network(n_input_dims=k_a + k_b, n_output_dims=out_dim, activation="ReLU", output_activation="None", n_neurons=64, n_hidden_layers=depth, dtype=torch.float32).to(dev)
    L = mlp.linear_layers(net)
    W, Bs = [l.weight.detach() for l in L], [l.bias.detach() for l in L]
    xa = torch.randn(N // S, k_a, device=dev) if k_a else None
    xb = torch.randn(rows, N, device=dev)
    dy = torch.randn(out_dim, N, device=dev)
    ref = None
    for mode in (mlp.FP16S, mlp.SPLIT):
        bad = [0, 0, 0, 0]
        for rep in range(int(os.environ.get("REPS", "20"))):
            y, saved = mlp.forward_raw(W, Bs, xa, xb, b_row0, k_b, S, True, bf16=mode)
            dxb = torch.empty(k_b, N, device=dev)
            dxa, partial = mlp.backward_raw(W, Bs, xa, xb, dy, saved, b_row0, k_b, S, dxb, xa is not None, bf16=mode)
            cur = (y.clone(), saved[0].view(torch.int32).clone(), dxb.clone(), partial.clone())
            if rep == 0: ref = cur
            else:
                for i in range(4):
                    if not torch.equal(ref[i].view(torch.int32), cur[i].view(torch.int32)): bad[i] += 1
        print(f"k_a {k_a} k_b {k_b} out {out_dim} depth {depth} mode {mode}: runs (of 19) that differ from the first in y / bits / dxb / partial: {bad}")
print("---- which parameter blocks differ (k_b 16, depth 2, mode 4)")
k_a, k_b, b_row0, rows, out_dim, depth = 0, 16, 0, 16, 16, 2
torch.manual_seed(3 + depth)
net = build_network(n_input_dims=k_a + k_b, n_output_dims=out_dim, activation="ReLU", output_activation="None", n_neurons=64, n_hidden_layers=depth, dtype=torch.float32).to(dev)
L = mlp.linear_layers(net)
W, Bs = [l.weight.detach() for l in L], [l.bias.detach() for l in L]
xb = torch.randn(rows, N, device=dev); dy = torch.randn(out_dim, N, device=dev)
ref = None
for rep in range(40):
    y, saved = mlp.forward_raw(W, Bs, None, xb, b_row0, k_b, S, True, bf16=mlp.FP16S)
    dxb = torch.empty(k_b, N, device=dev)
    _, partial = mlp.backward_raw(W, Bs, None, xb, dy, saved, b_row0, k_b, S, dxb, False, bf16=mlp.FP16S)
    if ref is None: ref = partial.clone(); continue
    diff = (partial != ref)
    if diff.any():
        rows_bad = diff.any(1).nonzero().flatten().tolist()
        off = 0; blocks = []
        for l, (w, b) in enumerate(zip(W, Bs)):
            if diff[:, off:off + w.numel()].any():
                d2 = diff[:, off:off + w.numel()].any(0).view_as(w)
                blocks.append(f"W{l} rows {d2.any(1).nonzero().flatten().tolist()[:8]} cols {d2.any(0).nonzero().flatten().tolist()[:8]} maxrel {float(((partial-ref)[:, off:off+w.numel()]).abs().max() / ref[:, off:off+w.numel()].abs().max()):.3g}")
            off += w.numel()
            if diff[:, off:off + b.numel()].any(): blocks.append(f"b{l}")
            off += b.numel()
        print(f"rep {rep}: workgroups {rows_bad[:10]} ({len(rows_bad)}): {blocks}")
