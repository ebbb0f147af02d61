"""Fused feature-network kernels (csrc/field_mlp.hip) against the torch path at the bench's size: 2 x 1.2 M points, 32 -> 64 -> 5."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreammat_amd import hipops
from dreammat_amd.geometry import VanillaMLP

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2400000
torch.manual_seed(0)
mlp = VanillaMLP(32, 5, {"n_neurons": 64, "n_hidden_layers": 1}).to(dev)
x = torch.randn(32, M, device=dev)
dy = torch.randn(M, 5, device=dev)


def run(mode):
    hipops.FIELD_MLP_FUSED = mode == "fused"
    xt = x.t().requires_grad_()                                          # feature-major [M, 32] view
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    y = mlp(xt)
    ev[1].record()
    y.backward(dy)
    ev[2].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])


for mode in ("torch", "fused", "torch", "fused", "fused"):
    f, b = run(mode)
    print(json.dumps({"mode": mode, "M": M, "fwd_ms": f, "bwd_ms": b}), flush=True)
