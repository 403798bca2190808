"""Diagnostic: which samples of the fused MLP (density shape, N = 2^20) differ from an fp64 evaluation, per evaluation mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nesvor_amd import mlp
from nesvor_amd.models import build_network
dev = torch.device("cuda:0")
N, S, k_a, k_b, rows, out_dim, b_row0 = 1 << 20, 256, 0, 32, 32, 16, 0
torch.manual_seed(1)
net = build_network(n_input_dims=k_a + k_b, n_output_dims=out_dim, activation="ReLU", output_activation="None", n_neurons=64, n_hidden_layers=2, dtype=torch.float32).to(dev)
L = mlp.linear_layers(net)
W, Bs = [l.weight.detach() for l in L], [l.bias.detach() for l in L]
xb = torch.randn(rows, N, device=dev); dy = torch.randn(out_dim, N, device=dev)
X = xb.t().double(); Wd, Bd = [w.double() for w in W], [b.double() for b in Bs]
p1 = X @ Wd[0].t() + Bd[0]; h1 = p1.relu(); p2 = h1 @ Wd[1].t() + Bd[1]; h2 = p2.relu()
y_ref = (h2 @ Wd[2].t() + Bd[2]).t(); G = dy.t().double()
d2 = (G @ Wd[2]) * (p2 > 0); d1 = (d2 @ Wd[1]) * (p1 > 0); dxb_ref = (d1 @ Wd[0]).t()
for rep in range(3):
    for mode in (mlp.MFMA_FP32, mlp.SPLIT):
        y, saved = mlp.forward_raw(W, Bs, None, xb, b_row0, k_b, S, True, mode)
        dxb = torch.empty(k_b, N, device=dev)
        _, partial = mlp.backward_raw(W, Bs, None, xb, dy, saved, b_row0, k_b, S, dxb, False, mode)
        torch.cuda.synchronize()
        bad = ((dxb.double() - dxb_ref).abs().amax(0) > 1e-5 * dxb_ref.abs().max())
        ybad = ((y.double() - y_ref).abs().amax(0) > 1e-5 * y_ref.abs().max())
        idx = bad.nonzero().flatten().tolist()
        groups = sorted(set(i // 16 for i in idx))
        print(f"rep {rep} mode {mode}: bad dx samples {len(idx)}, bad y samples {int(ybad.sum())}, groups with bad dx {len(groups)}: {groups[:40]}", flush=True)
        big = [g for g in groups if sum(1 for i in idx if i // 16 == g) >= 8]
        print("   groups with >= 8 bad samples:", big[:40], " (group % 1024:", [g % 1024 for g in big[:40]], ")")
