"""Diagnostic for the shade kernel: true kernel durations (run under `rocprofv3 --kernel-trace`) for
(A) the benchmark setting (5 envs, 128^2 cube), (B) a single env for all views, (C) a 32^2 cube atlas,
(D) views sorted by env, (E) the benchmark setting with the opt-in fp16 atlas (6 instead of 12 cube-map gathers per
pixel).  Each variant = 6 consecutive k_shade_fwd dispatches; HIP-event averages are printed as well."""
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
fg = penv.approx_fg_lut()
atlas128 = penv.EnvAtlas(lat, scale=2.0, min_res=16, max_res=128, fg_lut=fg, device=dev)
atlas32 = penv.EnvAtlas(lat, scale=2.0, min_res=8, max_res=32, fg_lut=fg, device=dev)
mat = _lib.MatCfgStruct(0.0, 0.9, 0.1, 0.95)
feat = torch.randn(5, N, device=dev).t()
def run(atlas, env, tag=""):
    hipops.shade(feat, gb.nrm.t(), gb.view.t(), gb.pix_idx, gb.n_dev, env, atlas, mat, H * W, False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        hipops.shade(feat, gb.nrm.t(), gb.view.t(), gb.pix_idx, gb.n_dev, env, atlas, mat, H * W, False)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 6 * 1e3
    print(f"{tag}: {us:.1f} us / launch, {56.0 * N / us / 1e6:.2f} TB/s algorithmic ({56.0 * N / us / 1e6 / 8 * 100:.0f} % of 8 TB/s)")
e_rand = torch.tensor([3, 0, 4, 1, 2, 0, 3, 1], dtype=torch.int32, device=dev)
run(atlas128, e_rand, "A fp32 atlas, bench setting")
run(atlas128, torch.zeros(8, dtype=torch.int32, device=dev), "B one env")
run(atlas32, e_rand, "C 32^2 cube")
run(atlas128, torch.tensor([0, 0, 1, 1, 2, 3, 3, 4], dtype=torch.int32, device=dev), "D views sorted by env")
atlas128h = penv.EnvAtlas(lat, scale=2.0, min_res=16, max_res=128, fg_lut=fg, device=dev, texel="fp16")
run(atlas128h, e_rand, "E fp16 atlas, bench setting")
print("N", N)
