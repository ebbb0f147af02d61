"""stem-layer timings: the patch (MFMA) form against the direct kernel (DREAMMAT_STEM_KERNEL=direct), one process each."""
import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dreammat_amd import hipops
dev = torch.device("cuda:0")
def t(B, H, W, Cin, Cout, stride=1, act=1):
    x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
    w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.1).bfloat16()
    b = torch.randn(Cout, device=dev).bfloat16()
    try:
        for _ in range(3): y = hipops.conv3x3_small_nhwc(x, w, b, stride, (1, 1), act)
    except Exception as e:
        print(f"{B}x{H}x{W} {Cin}->{Cout} s{stride}: {type(e).__name__}"); return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y = hipops.conv3x3_small_nhwc(x, w, b, stride, (1, 1), act)
    e1.record(); torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2), b.float(), stride=stride, padding=1)
    if act: ref = torch.nn.functional.silu(ref)
    err = (y.float().permute(0, 3, 1, 2) - ref).abs().max().item() / ref.abs().max().item()
    print(f"{os.environ.get('DREAMMAT_STEM_KERNEL','patch')} {B}x{H}x{W} {Cin}->{Cout} s{stride}: {e0.elapsed_time(e1)/20*1e3:.1f} us  rel err {err:.1e}")
t(8, 512, 512, 4, 128, 1, 0); t(24, 64, 64, 4, 320, 1, 0); t(8, 512, 512, 22, 16); t(8, 512, 512, 16, 16); t(8, 512, 512, 16, 32, 2); t(8, 256, 256, 32, 32); t(8, 256, 256, 32, 96, 2)
t(8, 128, 128, 96, 96) if False else None
t(8, 512, 512, 128, 4, 1, 0); t(2, 100, 70, 22, 16); t(1, 33, 47, 16, 32, 2)
