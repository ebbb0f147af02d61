"""Per-level timing of the hash-grid backward (atomics) on the MI355X -- diagnostic."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreammat_amd import _lib, hipops, mesh as pmesh
from tests import util
from tools.kernel_bench import timeit

dev = torch.device("cuda:0")
B, H, W = 8, 512, 512
m = pmesh.displaced_sphere(160, 160)
batch = util.make_views(B, H, W, seed=0)
v = m.v_pos.to(dev); tri = m.t_pos_idx.to(dev).int().contiguous(); vn = m.v_nrm.to(dev)
pos = hipops.vertex_transform(v, batch["mvp_mtx"].to(dev))
rast = hipops.RasterContext(dev).rasterize(pos, tri, H, W)
gb = hipops.gbuffer_compact(rast, tri, v, vn, batch["rays_d"].to(dev), torch.rand(B, H, W, device=dev), torch.randn(B, H, W, device=dev), 0.05)
pts2 = torch.cat([gb.pos, gb.pos_jitter], dim=1).t()
M = pts2.shape[0]
spec = hipops.GridSpec()
dt = torch.zeros(spec.n_params, device=dev)
L = _lib.lib()
out = []
for l in range(16):
    dy = torch.randn(2, M, device=dev)
    sc = (ctypes.c_float * 1)(spec.c_scale[l]); rs = (ctypes.c_uint32 * 1)(spec.c_res[l])
    sz = (ctypes.c_uint32 * 1)(spec.c_size[l]); of = (ctypes.c_uint32 * 1)(spec.c_offset[l])
    def f():
        _lib.check(L.dm_hashgrid_bwd(pts2.data_ptr(), pts2.stride(0), pts2.stride(1), None, M, dy.data_ptr(), 1, M, 1, sc, rs, sz, of, 1.0, dt.data_ptr(), hipops._stream()))
    t = timeit(f, 5, 2)
    r = {"level": l, "res": int(spec.c_res[l]), "size": int(spec.c_size[l]), "ms": t * 1e3, "Gatomics_per_s": M * 16 / t / 1e9}
    out.append(r); print(json.dumps(r), flush=True)
json.dump(out, open("gpurun_out/hashgrid_probe.json", "w"), indent=1)
