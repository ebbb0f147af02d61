import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dreammat_amd.sd import layers
dev = torch.device("cuda:0")
for (B, C, H, W) in [(8, 128, 512, 512), (8, 256, 256, 256), (8, 512, 128, 128)]:
    ds = layers.Downsample2D(C, asymmetric_pad=True).to(dev, torch.bfloat16)
    for p in ds.parameters(): p.requires_grad_(False)
    x = torch.randn(B, C, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_()
    g = torch.randn(B, C, H // 2, W // 2, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    for mode in ("subpixel", "zeroins", "subpixel", "zeroins"):
        os.environ["DREAMMAT_S2_DGRAD"] = mode
        ts = []
        for it in range(6):
            y = ds(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); y.backward(g); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3); x.grad = None
        print(f"{C}@{H}: {mode}: bwd {sorted(ts)[2]:.0f} us")
