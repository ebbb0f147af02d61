#!/usr/bin/env python3
"""The products of the VAE mid-block attention's GEMM form (hipops._WideHeadAttention) one by one: dm_gemm_*_batched on its two
shapes under each forced tile variant (DREAMMAT_GEMM_TILE), the softmax / transpose passes, and the whole forward + backward.
    PYTHONPATH=. python tools/wide_attn_time.py [G]"""
import os
import sys
import time

import torch

from dreammat_amd import hipops

dev = "cuda"
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S, D = 4096, 512


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e6


for dt in (torch.float16, torch.bfloat16):
    q = torch.randn(G, S, D, device=dev).to(dt)
    k = torch.randn(G, S, D, device=dev).to(dt)
    s = torch.empty(G, S, S, device=dev, dtype=dt)
    st = torch.empty(G, S, S, device=dev, dtype=dt)
    kt = torch.empty(G, D, S, device=dev, dtype=dt)
    o = torch.empty(G, S, D, device=dev, dtype=dt)
    for tile in ("", "128", "256", "512"):
        if tile:
            os.environ["DREAMMAT_GEMM_TILE"] = tile
        else:
            os.environ.pop("DREAMMAT_GEMM_TILE", None)
        a = timed(lambda: hipops.gemm_batched(q, k, s))
        b = timed(lambda: hipops.gemm_batched(s, kt, o))
        fl = 2.0 * G * S * S * D
        print(f"{dt} G={G} tile={tile or 'auto':>4}: s=qk^T {a:7.1f} us {fl / a / 1e6:6.0f} TF/s | o=p v {b:7.1f} us {fl / b / 1e6:6.0f} TF/s")
    os.environ.pop("DREAMMAT_GEMM_TILE", None)
    print(f"{dt} softmax {timed(lambda: hipops._softmax_rows_(s, 0.05)):7.1f} us | softmax bwd {timed(lambda: hipops._softmax_rows_bwd_(s, st, 0.05)):7.1f} us | "
          f"transpose [S,S] {timed(lambda: hipops.transpose_rows(s, out=st)):7.1f} us | transpose [S,D] {timed(lambda: hipops.transpose_rows(k, out=kt)):7.1f} us")
    B = 8
    qq, kk, vv = (torch.randn(B, S, D, device=dev).to(dt).requires_grad_() for _ in range(3))
    g = torch.randn(B, S, D, device=dev).to(dt)

    def fb():
        o_ = hipops.wide_head_attention(qq, kk, vv, 0.05)
        o_.backward(g)
    print(f"{dt} forward + backward, 8 images: {timed(fb, 5):8.1f} us")
