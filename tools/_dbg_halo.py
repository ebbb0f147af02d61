import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreammat_amd import hipops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def run(B, H, W, Cin, Cout, dt, extras, force="1"):
    x = torch.randn(B, H, W, Cin, device=dev).to(dt)
    w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.05).to(dt)
    b = torch.randn(Cout, device=dev).to(dt) if extras else None
    rb = torch.randn(B, Cout, device=dev).to(dt) if extras else None
    res = torch.randn(B, H, W, Cout, device=dev).to(dt) if extras else None
    os.environ["DREAMMAT_CONV_HALO"] = "0"
    y0 = hipops.conv3x3_nhwc(x, w, b, 1, (1, 1), None, rb, res)
    os.environ["DREAMMAT_CONV_HALO"] = force
    y1 = hipops.conv3x3_nhwc(x, w, b, 1, (1, 1), None, rb, res)
    torch.cuda.synchronize()
    d = (y1.float() - y0.float()).abs()
    ref = y0.float().abs().max().item()
    nbad = int((d > 0.02 * ref).sum())
    print(f"B{B} {H}x{W} {Cin}->{Cout} {dt} extras={extras} force={force}: max diff {d.max().item():.4g} (ref max {ref:.3g}) equal={torch.equal(y0, y1)} bad={nbad}", flush=True)
    if nbad:
        idx = (d > 0.02 * ref).nonzero()
        print("  first bad", idx[:5].tolist(), "last", idx[-3:].tolist())
for dt in (torch.bfloat16, torch.float16):
    run(8, 512, 512, 128, 128, dt, False)       # 384 x 128 patches (tile 640)
    run(2, 200, 72, 128, 128, dt, True)         # ragged bands / columns
    run(8, 256, 256, 256, 256, dt, True)        # 256 x 256 patches (tile 512)
    run(8, 128, 128, 512, 512, dt, False)
    run(24, 32, 32, 640, 1280, dt, True)
    run(3, 40, 24, 256, 512, dt, True)
    for force in ("24", "16"):
        run(2, 200, 72, 128, 128, dt, True, force)      # ragged last band (200 = 8 x 24 + 8 = 12 x 16 + 8) and columns (72 = 4.5 x 16)
        run(3, 40, 24, 256, 512, dt, True, force)       # Cout over several / ragged channel tiles
        run(5, 8, 8, 64, 320, dt, True, force)          # images smaller than a patch, 320 = 2.5 x 128
        run(1, 17, 33, 192, 64, dt, False, force)       # odd sizes, Cout below a tile
