"""Launches the roofline-target kernels a few times each (shade fwd/bwd at the benchmark's covered-pixel
count, attention S=4096 d=64 B=24, one UNet-sized conv) so that `rocprofv3 --pmc ...` can attribute HBM
traffic per launch.  Usage (one counter group per pass, see MI355X_MICROARCH.md):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python tools/pmc_probe.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -- python tools/pmc_probe.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreammat_amd import _lib, envlight as penv, hipops, mesh as pmesh
from tests import util

dev = torch.device("cuda:0")
B, H, W = 8, 512, 512
m = pmesh.displaced_sphere(160, 160)
batch = util.make_views(B, H, W, seed=0)
v = m.v_pos.to(dev); tri = m.t_pos_idx.to(dev).int().contiguous(); vn = m.v_nrm.to(dev)
pos = hipops.vertex_transform(v, batch["mvp_mtx"].to(dev))
rast = hipops.RasterContext(dev).rasterize(pos, tri, H, W)
gb = hipops.gbuffer_compact(rast, tri, v, vn, batch["rays_d"].to(dev), torch.rand(B, H, W, device=dev), torch.randn(B, H, W, device=dev), 0.05)
N = gb.n
lat = [util.synthetic_latlong(i, 256, 512) for i in range(5)]
atlas = penv.EnvAtlas(lat, scale=2.0, min_res=16, max_res=128, fg_lut=penv.approx_fg_lut(), device=dev)
mat = _lib.MatCfgStruct(0.0, 0.9, 0.1, 0.95)
feat = torch.randn(5, N, device=dev).t().requires_grad_()
env_of_view = torch.randint(0, 5, (B,), dtype=torch.int32, device=dev)
print("covered pixels", N, flush=True)
for _ in range(5):
    out = hipops.shade(feat, gb.nrm.t(), gb.view.t(), gb.pix_idx, gb.n_dev, env_of_view, atlas, mat, H * W, False)
    out[0].backward(torch.ones_like(out[0]))
q = torch.randn(24, 4096, 320, device=dev, dtype=torch.bfloat16); k = torch.randn_like(q)
vt = torch.randn(24, 320, 4096, device=dev, dtype=torch.bfloat16)
for _ in range(5):
    hipops.attention(q, k, vt, 5)
x = torch.randn(24, 64, 64, 320, device=dev, dtype=torch.bfloat16)
w = torch.randn(320, 9 * 320, device=dev, dtype=torch.bfloat16) * 0.02
for _ in range(5):
    hipops.conv3x3_nhwc(x, w, None)
torch.cuda.synchronize()
