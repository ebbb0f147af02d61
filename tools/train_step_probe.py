"""One ControlNet training step (row f-4: controlnet_train/diffusers_train_controlnet.py:858-915) at the real size -- SD-2.1-base
shaped UNet + 22-channel ControlNet copy, random weights, bf16, batch B at 512^2 (64^2 latents) -- timed with the trainable-layer
kernels on (MFMA attention forward + backward, trainable 3x3 convolutions with the weight-gradient kernel, GroupNorm affine
gradients) and off (torch autograd over im2col + hipBLASLt, matmul-softmax, ATen GroupNorm).
usage: python tools/train_step_probe.py [B] [iters]  -> JSON lines"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreammat_amd import controlnet_train as ct, hipops  # noqa: E402
from dreammat_amd.sd import ARCHS, AutoencoderKLEncoder, UNet2DConditionModel, layers  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
arch = sys.argv[3] if len(sys.argv) > 3 else "sd21-base"
dev = torch.device("cuda:0")
a = ARCHS[arch]
torch.manual_seed(0)
t0 = time.time()
with torch.device(dev):
    unet, vae = UNet2DConditionModel(a).bfloat16(), AutoencoderKLEncoder(a).bfloat16()
cn = ct.init_controlnet(unet)
tr = ct.ControlNetTrainer(vae, unet, controlnet=cn.to(dev).bfloat16(), lr=1e-5)
print(json.dumps({"build_s": round(time.time() - t0, 1), "controlnet_params_M": round(sum(p.numel() for p in cn.parameters()) / 1e6, 1)}),
      flush=True)
g = torch.Generator(device="cpu").manual_seed(1)
R = 512 if arch != "tiny" else 64
img = (torch.rand(B, 3, R, R, generator=g) * 2 - 1).to(dev).bfloat16()
cond = torch.rand(B, 22, R, R, generator=g).to(dev)
text = torch.randn(B, 77, a.cross_dim, generator=g).to(dev)


def run(flag):
    layers.TRAIN_KERNELS = flag
    ts = []
    for it in range(iters + 1):
        torch.cuda.synchronize()
        t = time.time()
        loss = tr.step(img, cond, text)
        torch.cuda.synchronize()
        if it:
            ts.append(time.time() - t)
    ms = sorted(ts)[len(ts) // 2] * 1e3
    return {"train_kernels": flag, "B": B, "arch": arch, "ms_per_step": round(ms, 1), "images_per_s": round(B / ms * 1e3, 2),
            "loss": round(float(loss), 4), "peak_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}


for flag in (True, False):
    torch.cuda.reset_peak_memory_stats()
    print(json.dumps(run(flag)), flush=True)

# where the hand-written kernels' share of the step goes (HIP events around every launch of one extra step; the rest of the
# step = hipBLASLt Linear layers under autograd, ATen elementwise, AdamW over 364 M parameters, gradient clipping)
layers.TRAIN_KERNELS = True
hipops.enable_kernel_timing(True)
tr.step(img, cond, text)
torch.cuda.synchronize()
groups = {}
for key, v in hipops.kernel_times().items():
    name = key.split("[")[0]
    g_ = groups.setdefault(name, [0, 0.0])
    g_[0] += v["launches"]; g_[1] += v["launches"] * v["avg_ms"]
hipops.enable_kernel_timing(False)
print(json.dumps({"timed_step_kernels_ms": {k: [n, round(ms, 2)] for k, (n, ms) in sorted(groups.items(), key=lambda kv: -kv[1][1])},
                  "sum_ms": round(sum(ms for _, ms in groups.values()), 1)}), flush=True)
