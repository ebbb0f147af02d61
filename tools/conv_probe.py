"""Diagnostic: first-call latency of convolution backends on a fresh MI355X box (MIOpen JIT vs GEMM lowering)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreammat_amd.sd import layers
dev = torch.device("cuda:0")
torch.zeros(1, device=dev); torch.cuda.synchronize()
def run(name, fn, n=5):
    t0 = time.time(); fn(); torch.cuda.synchronize(); t1 = time.time() - t0
    t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); t2 = (time.time() - t0) / n
    print(f"{name}: first {t1:.3f}s steady {t2*1e3:.3f}ms", flush=True)
x = torch.randn(24, 320, 64, 64, device=dev, dtype=torch.bfloat16)
conv = layers.Conv2d(320, 320, 3, padding=1).to(dev, torch.bfloat16)
layers.CONV_BACKEND = "gemm"
with torch.no_grad():
    run("gemm conv3x3 320->320 @64^2 b24", lambda: conv(x))
    x2 = torch.randn(24, 1280, 16, 16, device=dev, dtype=torch.bfloat16)
    conv2 = layers.Conv2d(1280, 1280, 3, padding=1).to(dev, torch.bfloat16)
    run("gemm conv3x3 1280->1280 @16^2 b24", lambda: conv2(x2))
    xv = torch.randn(8, 128, 512, 512, device=dev, dtype=torch.bfloat16)
    convv = layers.Conv2d(128, 128, 3, padding=1).to(dev, torch.bfloat16)
    run("gemm conv3x3 128->128 @512^2 b8", lambda: convv(xv), 3)
if "--miopen" in sys.argv:
    layers.CONV_BACKEND = "miopen"
    with torch.no_grad():
        run("miopen nchw conv3x3 320->320 @64^2 b24", lambda: conv(x))
        xc = x.to(memory_format=torch.channels_last); convc = conv.to(memory_format=torch.channels_last)
        run("miopen nhwc conv3x3 320->320 @64^2 b24", lambda: convc(xc))
        run("miopen nchw conv3x3 1280 @16^2", lambda: conv2(x2))
