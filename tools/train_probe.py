"""Timing of the trainable-layer kernels of the ControlNet training loop (f-4) at the SD-2.1 shapes of a 512^2 batch:
3x3 convolutions with trainable weights on the MFMA route (forward, data gradient, dm_conv3x3_wgrad_nhwc_bf16) against the im2col +
hipBLASLt lowering under torch autograd.  usage: python tools/train_probe.py [iters] -> one JSON line per shape."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreammat_amd import hipops  # noqa: E402
from dreammat_amd.sd import layers  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")


def timed(fn):
    ts = []
    for it in range(iters + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if it >= 2:
            ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


for B, H, Cin, Cout, s in ((4, 64, 320, 320, 1), (4, 64, 320, 320, 2), (4, 32, 640, 640, 1), (4, 16, 1280, 1280, 1), (4, 8, 1280, 1280, 1)):
    conv = layers.Conv2d(Cin, Cout, 3, stride=s, padding=1).to(dev).bfloat16()
    x = torch.randn(B, Cin, H, H, device=dev).bfloat16().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).requires_grad_(True)
    Ho = (H - 1) // s + 1
    g = torch.randn(B, Cout, Ho, Ho, device=dev).bfloat16().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    xn, gn = x.detach().permute(0, 2, 3, 1).contiguous(), g.permute(0, 2, 3, 1).contiguous()

    def step(c=conv):
        c.weight.grad = c.bias.grad = x.grad = None
        c(x).backward(g)

    rec = {"op": "conv3x3_train", "B": B, "HW": H, "Cin": Cin, "Cout": Cout, "stride": s}
    rec["mfma_fwd_bwd_ms"] = round(timed(step), 4)
    rec["wgrad_ms"] = round(timed(lambda: hipops.conv3x3_wgrad(xn, gn, s)), 4)
    rec["wgrad_tflops"] = round(18.0 * B * Ho * Ho * Cin * Cout / rec["wgrad_ms"] * 1e-9, 1)
    layers.CONV_BACKEND = "gemm"
    try:
        rec["im2col_fwd_bwd_ms"] = round(timed(step), 4)
    finally:
        layers.CONV_BACKEND = "mfma"
    print(json.dumps(rec), flush=True)
