"""Diagnostic: what the unstable W_out partial of mode 4 (KB1 = 1, two hidden layers) looks like against candidates."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nesvor_amd import mlp
from nesvor_amd.models import build_network
dev = torch.device("cuda:0")
N, S = 8192, 256
k_a, k_b, b_row0, rows, out_dim, depth = 0, 16, 0, 16, 16, 2
torch.manual_seed(3 + depth)
net = build_network(n_input_dims=k_a + k_b, n_output_dims=out_dim, activation="ReLU", output_activation="None", n_neurons=64, n_hidden_layers=depth, dtype=torch.float32).to(dev)
L = mlp.linear_layers(net)
W, Bs = [l.weight.detach() for l in L], [l.bias.detach() for l in L]
xb = torch.randn(rows, N, device=dev); dy = torch.randn(out_dim, N, device=dev)
X = xb.t().double()
h0 = torch.relu(X @ W[0].t().double() + Bs[0].double()); h1 = torch.relu(h0 @ W[1].t().double() + Bs[1].double())
off2 = W[0].numel() + 64 + W[1].numel() + 64
for rep in range(12):
    y, saved = mlp.forward_raw(W, Bs, None, xb, b_row0, k_b, S, True, bf16=mlp.FP16S)
    dxb = torch.empty(k_b, N, device=dev)
    _, partial = mlp.backward_raw(W, Bs, None, xb, dy, saved, b_row0, k_b, S, dxb, False, bf16=mlp.FP16S)
    p2 = partial[:, off2:off2 + W[2].numel()].view(-1, 16, 64).double()   # per workgroup
    bad = []
    for wg in range(128):
        sl = slice(wg * 64, wg * 64 + 64)  # the workgroup's 4 groups = 64 samples
        ref1 = dy[:, sl].double() @ h1[sl]
        e1 = float((p2[wg] - ref1).abs().max() / ref1.abs().max())
        if e1 > 2e-2:
            ref0 = dy[:, sl].double() @ h0[sl]
            cands = {"dY x h0": float((p2[wg] - ref0).abs().max() / ref0.abs().max())}
            for miss in range(4):  # one of the four groups missing / with h0 instead of h1
                g = slice(wg * 64 + 16 * miss, wg * 64 + 16 * miss + 16)
                r = ref1 - dy[:, g].double() @ h1[g]
                cands[f"group {miss} missing"] = float((p2[wg] - r).abs().max() / ref1.abs().max())
                r2 = r + dy[:, g].double() @ h0[g]
                cands[f"group {miss} with h0"] = float((p2[wg] - r2).abs().max() / ref1.abs().max())
                r3 = r + 2 * (dy[:, g].double() @ h1[g])
                cands[f"group {miss} twice"] = float((p2[wg] - r3).abs().max() / ref1.abs().max())
            best = min(cands, key=cands.get)
            bad.append((wg, round(e1, 3), best, round(cands[best], 4)))
    print(f"rep {rep}: {bad}")
