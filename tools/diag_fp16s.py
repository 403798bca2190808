"""Diagnostic: mode 4 (scaled fp16, leading term) against mode 2 (split) of the fused MLP, piece by piece."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nesvor_amd import mlp
from nesvor_amd.models import build_network
dev = torch.device("cuda:0")
torch.manual_seed(0)
N, S = 8192, 256
for (k_a, k_b, b_row0, rows, out_dim, depth) in ((0, 32, 0, 32, 16, 2), (0, 32, 0, 32, 16, 1), (16, 15, 1, 16, 1, 2)):
    net = build_network(n_input_dims=k_a + k_b, n_output_dims=out_dim, activation="ReLU", output_activation="None", n_neurons=64, n_hidden_layers=depth, dtype=torch.float32).to(dev)
    L = mlp.linear_layers(net)
    W, Bs = [l.weight.detach() for l in L], [l.bias.detach() for l in L]
    xa = torch.randn(N // S, k_a, device=dev) if k_a else None
    xb = torch.randn(rows, N, device=dev)
    dy = torch.randn(out_dim, N, device=dev)
    out = {}
    for mode in (mlp.SPLIT, mlp.FP16S):
        y, saved = mlp.forward_raw(W, Bs, xa, xb, b_row0, k_b, S, True, bf16=mode)
        dxb = torch.empty(k_b, N, device=dev)
        dxa, partial = mlp.backward_raw(W, Bs, xa, xb, dy, saved, b_row0, k_b, S, dxb, xa is not None, bf16=mode)
        out[mode] = (y, dxb, partial.sum(0), saved[0].view(torch.int32))
    a, b = out[mlp.SPLIT], out[mlp.FP16S]
    print(f"== k_a {k_a} k_b {k_b} out {out_dim} depth {depth}")
    print("   y   rel", float((a[0] - b[0]).abs().max() / a[0].abs().max()), " bits equal", float((a[3] == b[3]).float().mean()))
    e = (a[1] - b[1]).abs()
    print("   dxb rel", float(e.max() / a[1].abs().max()), " per row max", [round(float(v), 4) for v in (e.amax(1) / a[1].abs().max())[:8]],
          " per 16-sample position", [round(float(v), 4) for v in (e.view(k_b, -1, 16).amax((0, 1)) / a[1].abs().max())])
    off = 0
    for l, (w, bb) in enumerate(zip(W, Bs)):
        gw_a, gw_b = a[2][off:off + w.numel()], b[2][off:off + w.numel()]; off += w.numel()
        gb_a, gb_b = a[2][off:off + bb.numel()], b[2][off:off + bb.numel()]; off += bb.numel()
        print(f"   layer {l}: dW rel {float((gw_a - gw_b).abs().max() / gw_a.abs().max()):.4g}  db rel {float((gb_a - gb_b).abs().max() / gb_a.abs().max()):.4g}")
