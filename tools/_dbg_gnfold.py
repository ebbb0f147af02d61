import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreammat_amd import hipops
from dreammat_amd.sd import layers
dev = torch.device("cuda:0")
torch.manual_seed(0)

def direct(B, H, W, Cin, Cout, dt, act, extras, force):
    os.environ["DREAMMAT_CONV_HALO"] = force
    x = (torch.randn(B, H, W, Cin, device=dev) * 2 + 0.5).to(dt)
    gm, bt = (torch.rand(Cin, device=dev) + 0.5).to(dt), torch.randn(Cin, device=dev).to(dt)
    w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.05).to(dt)
    b = torch.randn(Cout, device=dev).to(dt) if extras else None
    rb = torch.randn(B, Cout, device=dev).to(dt) if extras else None
    res = torch.randn(B, H, W, Cout, device=dev).to(dt) if extras else None
    ok = hipops.gn_conv3x3_ok(x, gm, Cout)
    y_ref = hipops.conv3x3_nhwc(hipops.groupnorm_nhwc(x, gm, bt, 1e-5, act), w, b, 1, (1, 1), None, rb, res)
    if not ok:
        print(f"B{B} {H}x{W} {Cin}->{Cout} force={force}: not served"); return
    y = hipops.gn_conv3x3_nhwc(x, gm, bt, 1e-5, act, w, w, b, rb, res)
    torch.cuda.synchronize()
    d = (y.float() - y_ref.float()).abs(); ref = y_ref.float().abs().max().item()
    print(f"B{B} {H}x{W} {Cin}->{Cout} {dt} act={act} extras={extras} force={force}: max diff {d.max().item():.4g} (ref {ref:.3g}) "
          f"bad={int((d > 0.02 * ref).sum())} equal={torch.equal(y, y_ref)}", flush=True)

for dt in (torch.bfloat16, torch.float16):
    direct(8, 512, 512, 128, 128, dt, 1, False, "1")
    direct(8, 256, 256, 256, 256, dt, 1, True, "1")
    direct(2, 200, 72, 128, 128, dt, 1, True, "24")
    direct(3, 40, 24, 256, 512, dt, 0, True, "16")
    direct(5, 8, 8, 64, 320, dt, 1, True, "24")
    direct(1, 17, 33, 192, 64, dt, 1, False, "16")
os.environ["DREAMMAT_CONV_HALO"] = "1"

# ResnetBlock2D: inference and autograd, fold on / off
for (cin, cout, temb_ch, B, HW, grad) in ((128, 128, 0, 8, 256, True), (128, 256, 0, 8, 256, True), (640, 640, 1280, 24, 32, False), (256, 512, 0, 4, 128, True)):
    torch.manual_seed(1)
    blk = layers.ResnetBlock2D(cin, cout, temb_ch, eps=1e-6).to(dev, torch.float16).eval()
    for p in blk.parameters():
        p.requires_grad_(False)
    x0 = torch.randn(B, cin, HW, HW, device=dev).to(torch.float16).contiguous(memory_format=torch.channels_last)
    temb = torch.randn(B, temb_ch, device=dev).to(torch.float16) if temb_ch else None
    outs = {}
    for fold in (False, True):
        hipops.GN_CONV_FOLD = fold
        x = x0.clone().requires_grad_(grad)
        hipops.enable_kernel_timing(True)
        with torch.set_grad_enabled(grad):
            y = blk(x, temb)
            if grad:
                g = torch.randn_like(y)
                torch.manual_seed(5); g = torch.randn(y.shape, device=dev).to(y.dtype)
                y.backward(g)
        torch.cuda.synchronize()
        keys = sorted(hipops.kernel_times())
        hipops.enable_kernel_timing(False)
        outs[fold] = (y.detach().float(), x.grad.float() if grad else None, keys)
    dy = (outs[True][0] - outs[False][0]).abs().max().item() / outs[False][0].abs().max().item()
    dg = ((outs[True][1] - outs[False][1]).abs().max().item() / outs[False][1].abs().max().item()) if grad else None
    print(f"resnet {cin}->{cout} B{B} {HW}^2 grad={grad}: rel dy {dy:.3g} rel dgrad {dg}  fold kernels: {[k for k in outs[True][2] if 'gn' in k or 'groupnorm' in k]}", flush=True)
hipops.GN_CONV_FOLD = True
