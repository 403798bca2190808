"""A few forward+backward passes of the fused MLP (density-net shape) at N=2^20 for rocprofv3 --pmc."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nesvor_amd.mlp import fused_mlp
from nesvor_amd.models import build_network
dev = torch.device("cuda:0")
N = 1 << 20
torch.manual_seed(0)
net = build_network(n_input_dims=32, n_output_dims=16, activation="ReLU", output_activation="None", n_neurons=64, n_hidden_layers=2, dtype=torch.float32).to(dev)
xb = torch.randn(32, N, device=dev, requires_grad=True)
w = torch.randn(16, N, device=dev)
for _ in range(3):
    y = fused_mlp(net, None, xb, 0, 32, 256)
    (y * w).sum().backward()
torch.cuda.synchronize()
