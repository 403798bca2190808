"""Per-parameter-block error of the fused MLP backward against fp64 (split mode vs fp32 MFMA): python tools/diag_mlp_split.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nesvor_amd import mlp
from nesvor_amd.models import build_network

dev = torch.device("cuda:0")
for (k_a, k_b, b_row0, rows, out_dim, nh) in [(0, 32, 0, 32, 16, 2), (16, 15, 1, 16, 1, 2), (0, 32, 0, 32, 16, 1), (16, 8, 0, 32, 1, 2)]:
    torch.manual_seed(1)
    N, S = 8192, 256
    net = build_network(n_input_dims=k_a + k_b, n_output_dims=out_dim, activation="ReLU", output_activation="None",
                        n_neurons=64, n_hidden_layers=nh, dtype=torch.float32).to(dev)
    L = mlp.linear_layers(net)
    W, Bs = [l.weight.detach() for l in L], [l.bias.detach() for l in L]
    xa = torch.randn(N // S, k_a, device=dev) if k_a else None
    xb = torch.randn(rows, N, device=dev) * 0.37
    dy = torch.randn(out_dim, N, device=dev) * 1e-3
    X = xb[b_row0 : b_row0 + k_b].t().double()
    if xa is not None:
        X = torch.cat([xa.double().repeat_interleave(S, 0), X], 1)
    Wd, Bd = [w.double() for w in W], [b.double() for b in Bs]
    acts, pre = [X], []
    for i in range(nh):
        p = acts[-1] @ Wd[i].t() + Bd[i]
        pre.append(p); acts.append(p.relu())
    y_ref = (acts[-1] @ Wd[nh].t() + Bd[nh]).t()
    G = dy.t().double()
    grads = []
    d = G
    for i in range(nh, -1, -1):
        grads.append((f"W{i}", (d.t() @ acts[i]).reshape(-1)))
        grads.append((f"b{i}", d.sum(0)))
        if i > 0:
            d = (d @ Wd[i]) * (pre[i - 1] > 0)
    dx_ref = (d @ Wd[0])[:, k_a:].t()
    order = []
    for i in range(nh + 1):
        order += [g for g in grads if g[0] in (f"W{i}", f"b{i}")]
    print(f"--- k_a={k_a} k_b={k_b} out={out_dim} hidden layers={nh}")
    for mode in (mlp.MFMA_FP32, mlp.SPLIT):
        y, saved = mlp.forward_raw(W, Bs, xa, xb, b_row0, k_b, S, True, mode)
        dxb = torch.empty(k_b, N, device=dev)
        dxa, partial = mlp.backward_raw(W, Bs, xa, xb, dy, saved, b_row0, k_b, S, dxb, xa is not None, mode)
        flat = partial.double().sum(0)
        rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
        msg = [f"y {rel(y, y_ref):.2e}", f"dx {rel(dxb, dx_ref):.2e}"]
        off = 0
        for name, ref in order:
            got = flat[off : off + ref.numel()]
            off += ref.numel()
            msg.append(f"{name} {rel(got, ref):.2e}")
        print("split " if mode == mlp.SPLIT else "fp32  ", "  ".join(msg))
