"""Where does the 16-bit noise prediction leave the fp32 one?  Full-size SD-2.1 UNet + ControlNet (seeded random weights), one branch
item: every ResnetBlock2D / Transformer2DModel / Attention / FeedForward / sampler output of the fp32 GPU run is kept, the bf16 and
the f16 runs (hand-written kernels) are compared with it module by module in execution order: relative L2 error of the ACCUMULATED
activation at that point.  A module type at which the f16 error jumps while bf16's does not (or both jump by the same absolute
amount) is an error source that does not scale with the mantissa -- round 5 measured f16 only 2x closer than bf16 at full size
(8x on the tiny nets).   python tools/f16_error_profile.py  ->  gpurun_out/f16_error_profile.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreammat_amd.sd import ARCHS, ControlNetModel, UNet2DConditionModel, layers   # noqa: E402

dev = torch.device("cuda:0")
a = ARCHS["sd21-base"]
torch.manual_seed(0)
with torch.device(dev):
    unet = UNet2DConditionModel(a).eval()
    cn = ControlNetModel.from_unet(unet).eval()
for conv in list(cn.controlnet_down_blocks) + [cn.controlnet_mid_block, cn.controlnet_cond_embedding.conv_out]:
    torch.nn.init.normal_(conv.weight, std=0.02)
for p in list(unet.parameters()) + list(cn.parameters()):
    p.requires_grad_(False)
g = torch.Generator().manual_seed(1)
x = torch.randn(1, 4, 64, 64, generator=g).to(dev); t = torch.tensor([437], device=dev)
ctx = torch.randn(1, 77, a.cross_dim, generator=g).to(dev); cond = torch.rand(1, 22, 512, 512, generator=g).to(dev)

KINDS = (layers.ResnetBlock2D, layers.Transformer2DModel, layers.Attention, layers.FeedForward, layers.Downsample2D, layers.Upsample2D)
names = {m: ("unet." if net is unet else "cn.") + n for net in (unet, cn) for n, m in net.named_modules() if isinstance(m, KINDS)}
trace = []


def hook(m, inp, out):
    trace.append((names[m], out.detach().float().cpu()))


hs = [m.register_forward_hook(hook) for m in names]


def run(dt):
    trace.clear()
    c = lambda v: v.to(dt)
    with torch.no_grad():
        d, m = cn(c(x), t, c(ctx), c(cond), 1.0)
        y = unet(c(x), t, c(ctx), d, m)
    torch.cuda.synchronize()
    return list(trace), y.float().cpu()


master = [{k: v.clone() for k, v in net.state_dict().items()} for net in (unet, cn)]     # fp32 weights: every run casts from THESE
ref, yref = run(torch.float32)
out = {"modules": [n for n, _ in ref], "rel_l2": {}, "final_rel_max": {}}
for dt, nm in ((torch.bfloat16, "bf16"), (torch.float16, "f16")):
    unet.to(dt); cn.to(dt)
    tr, y = run(dt)
    assert [n for n, _ in tr] == out["modules"]
    out["rel_l2"][nm] = [float((v - r).norm() / r.norm().clamp_min(1e-30)) for (_, v), (_, r) in zip(tr, ref)]
    out["final_rel_max"][nm] = float((y - yref).abs().max() / yref.abs().max())
    unet.float(); cn.float()
    unet.load_state_dict(master[0]); cn.load_state_dict(master[1])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "f16_error_profile.json"), "w"))
print("final", out["final_rel_max"])
prev = (0.0, 0.0)
for n, eb, eh in zip(out["modules"], out["rel_l2"]["bf16"], out["rel_l2"]["f16"]):
    flag = "  <-- f16 jump" if eh > 2.5 * max(prev[1], 1e-5) and eh > 1e-3 else ""
    print(f"{n:62s} bf16 {eb:9.2e}  f16 {eh:9.2e}  ratio {eb / max(eh, 1e-30):6.1f}{flag}")
    prev = (eb, eh)
