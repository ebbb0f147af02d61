#!/usr/bin/env python3
"""hash-grid backward of the field at the bench's point count (dm_hashgrid_bwd_binned through hipops), per pass-2 split count.
    PYTHONPATH=. python tools/hashgrid_bwd_time.py"""
import time

import torch

from dreammat_amd import hipops

dev = "cuda"
torch.manual_seed(0)
M = 1_200_000
spec = hipops.GridSpec()
table = (torch.rand(spec.n_params, device=dev) * 2e-4 - 1e-4).requires_grad_()
# surface-like points: a displaced sphere shell
d = torch.nn.functional.normalize(torch.randn(M, 3, device=dev), dim=-1)
x = (d * (0.8 + 0.02 * torch.randn(M, 1, device=dev))).clamp(-0.99, 0.99)
g = torch.randn(M, 2 * spec.n_levels, device=dev) * 1e-3


def run():
    table.grad = None
    enc = hipops.hashgrid_encode(x, table, spec, 1.0)
    enc.backward(g)


for _ in range(3):
    run()
torch.cuda.synchronize()
hipops.enable_kernel_timing(True)
for _ in range(10):
    run()
torch.cuda.synchronize()
for k, v in hipops.kernel_times().items():
    if k.startswith("hashgrid"):
        print(k, f"{v['avg_ms'] * 1e3:8.1f} us x {v['launches']}")
ref = table.grad.clone()
print("grad checksum", float(ref.double().abs().sum()))
