"""Timing of the differentiated attention (dm_attention_fwd_lse_bf16 / dm_attention_bwd_bf16) on the training shapes.
usage: python tools/attn_bwd_probe.py [iters]   -> one JSON line per shape (forward / backward ms, TF/s; backward FLOP =
2.5 x forward's 4 B S Skv C plus the recomputation = 14 B S Skv C executed, 10 B S Skv C algorithmic)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreammat_amd import hipops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
for B, h, Sq, Skv, D in ((4, 5, 4096, 4096, 64), (4, 10, 1024, 1024, 64), (4, 5, 4096, 77, 64), (4, 20, 256, 256, 64)):
    C = h * D
    q, k, v = (torch.randn(B, S, C, device=dev).bfloat16().requires_grad_(True) for S in (Sq, Skv, Skv))
    do = torch.randn(B, Sq, C, device=dev).bfloat16()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf, tb = [], []
    for it in range(iters + 2):
        ev[0].record()
        out = hipops.attention_train(q, k, v, h)
        ev[1].record()
        out.backward(do)
        ev[2].record()
        torch.cuda.synchronize()
        if it >= 2:
            tf.append(ev[0].elapsed_time(ev[1])); tb.append(ev[1].elapsed_time(ev[2]))
        q.grad = k.grad = v.grad = None
    f, b = sorted(tf)[len(tf) // 2], sorted(tb)[len(tb) // 2]
    # the generic forward kernel alone, V^T operand (plain 16-byte LDS reads) against untransposed V (transposing LDS reads)
    qd, kd, vd = q.detach(), k.detach(), v.detach()
    vt = torch.zeros(B, C, (Skv + 7) // 8 * 8, device=dev, dtype=torch.bfloat16)
    vt[:, :, :Skv] = vd.transpose(1, 2)
    ab = {}
    hipops.attention_select("staged")
    for name, fn in (("staged_vt_ms", lambda: hipops.attention(qd, kd, vt, h)),
                     ("staged_natural_v_ms", lambda: hipops.attention_fwd_lse(qd, kd, vd, h, D ** -0.5))):
        ts = []
        for it in range(iters + 2):
            ev[0].record(); fn(); ev[1].record(); torch.cuda.synchronize()
            if it >= 2:
                ts.append(ev[0].elapsed_time(ev[1]))
        ab[name] = round(sorted(ts)[len(ts) // 2], 4)
    hipops.attention_select(None)
    flop = 4.0 * B * Sq * Skv * C
    print(json.dumps({"op": "attention_train", "B": B, "heads": h, "Sq": Sq, "Skv": Skv, "D": D, "fwd_ms": round(f, 4),
                      "bwd_ms": round(b, 4), "fwd_tflops": round(flop / f * 1e-9, 1),
                      "bwd_tflops_algorithmic": round(2.5 * flop / b * 1e-9, 1),
                      "bwd_tflops_executed": round(3.5 * flop / b * 1e-9, 1), **ab}))
