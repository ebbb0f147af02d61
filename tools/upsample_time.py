"""the UNet's three Upsample2D layers: DREAMMAT_UPSAMPLE=subpixel (2 x 2 form) vs =materialize (nearest-2x tensor + 3x3 conv)."""
import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dreammat_amd.sd import layers
dev = torch.device("cuda:0")
for (B, C, h) in [(24, 1280, 8), (24, 1280, 16), (24, 640, 32)]:
    up = layers.Upsample2D(C).to(dev, torch.bfloat16).requires_grad_(False)
    x = torch.randn(B, C, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    for mode in ("subpixel", "materialize", "subpixel", "materialize"):
        os.environ["DREAMMAT_UPSAMPLE"] = mode
        with torch.no_grad():
            for _ in range(3): y = up(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): y = up(x)
            e1.record(); torch.cuda.synchronize()
        print(f"{C}@{h}->{2*h}: {mode}: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us")
