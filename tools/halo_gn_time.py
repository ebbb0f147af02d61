"""Halo-patch convolution with / without the folded GroupNorm against the per-tap kernel and the apply pass it replaces (HIP events,
back-to-back launches): python tools/halo_gn_time.py"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreammat_amd import hipops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def timed(fn, n=12):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, H, C, Co) in ((8, 512, 128, 128), (8, 256, 256, 256), (8, 128, 512, 512), (8, 64, 512, 512), (24, 16, 1280, 1280)):
    dt = torch.float16
    x = torch.randn(B, H, H, C, device=dev).to(dt); gm = torch.ones(C, device=dev).to(dt); bt = torch.zeros(C, device=dev).to(dt)
    w = (torch.randn(Co, 9 * C, device=dev) * 0.03).to(dt); b = torch.zeros(Co, device=dev).to(dt)
    t_fold = timed(lambda: hipops.gn_conv3x3_nhwc(x, gm, bt, 1e-5, 1, w, w, b))
    t_stats = timed(lambda: hipops._gn_stats(x, gm, bt, 1e-5))
    t_gn = timed(lambda: hipops.groupnorm_nhwc(x, gm, bt, 1e-5, 1))
    t_conv = timed(lambda: hipops.conv3x3_nhwc(x, w, b))
    os.environ["DREAMMAT_CONV_HALO"] = "0"
    t_tap = timed(lambda: hipops.conv3x3_nhwc(x, w, b))
    del os.environ["DREAMMAT_CONV_HALO"]
    fl = 2.0 * B * H * H * Co * 9 * C
    print(json.dumps({"shape": [B, H, C, Co], "fold_stats_plus_conv_us": round(t_fold, 1), "stats_us": round(t_stats, 1), "gn_conv_alone_us": round(t_fold - t_stats, 1),
                      "groupnorm_2launch_us": round(t_gn, 1), "halo_conv_us": round(t_conv, 1), "per_tap_conv_us": round(t_tap, 1),
                      "halo_TFs": round(fl / t_conv / 1e6, 0), "per_tap_TFs": round(fl / t_tap / 1e6, 0), "gn_conv_TFs": round(fl / (t_fold - t_stats) / 1e6, 0)}), flush=True)
