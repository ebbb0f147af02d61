#!/usr/bin/env python3
"""the few-channel stem convolutions of the step at the bench shapes (dm_conv3x3_small_*): us per launch.  PYTHONPATH=. python tools/small_conv_time.py"""
import time

import torch

from dreammat_amd import hipops

dev = "cuda"
for dt in (torch.bfloat16, torch.float16):
    for (B, H, Cin, Cout, stride, act) in [(8, 512, 128, 4, 1, 0), (8, 512, 4, 128, 1, 0), (8, 512, 22, 16, 1, 1), (8, 512, 16, 16, 1, 1),
                                           (8, 512, 16, 32, 2, 1), (8, 256, 32, 32, 1, 1), (8, 256, 32, 96, 2, 1), (24, 64, 4, 320, 1, 0)]:
        x = torch.randn(B, H, H, Cin, device=dev).to(dt)
        w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.05).to(dt)
        f = lambda: hipops.conv3x3_small_nhwc(x, w, None, stride, (1, 1), act)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(20):
            f()
        torch.cuda.synchronize()
        print(dt, f"{B}x{H}^2 {Cin}->{Cout} s{stride}: {(time.time() - t) / 20 * 1e6:7.1f} us")
