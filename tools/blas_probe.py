"""Does hipBLASLt / rocBLAS survive the V^T projection shapes of the 1024^2 configurations?  W[C,C] @ X[B,S,C]^T -> [B,C,S] (bf16).
Each shape runs in its own process (a GPU fault kills the context).  usage: blas_probe.py            -> table
                                                                              blas_probe.py B S C      -> one shape"""
import subprocess
import sys

if len(sys.argv) == 4:
    import torch
    B, S, C = (int(v) for v in sys.argv[1:])
    w = torch.randn(C, C, device="cuda").bfloat16()
    x = torch.randn(B, S, C, device="cuda").bfloat16()
    y = torch.matmul(w, x.transpose(1, 2))
    torch.cuda.synchronize()
    print("ok", float((y[0].float() - w.float() @ x[0].float().t()).abs().max()))
    sys.exit(0)
for shape in [(24, 16384, 320), (48, 16384, 320), (24, 4096, 640), (48, 4096, 640), (8, 4096, 640), (24, 2048, 640), (24, 4096, 512),
              (24, 4096, 1280), (24, 1024, 1280), (48, 1024, 1280), (24, 256, 1280), (48, 256, 1280), (3, 4096, 640)]:
    r = subprocess.run([sys.executable, __file__] + [str(v) for v in shape], capture_output=True, text=True)
    out = [ln for ln in r.stdout.splitlines() if ln.startswith("ok")]
    print(shape, out[0] if out else "FAILED rc=%d" % r.returncode, flush=True)
