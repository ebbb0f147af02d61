"""First timing of the Monte-Carlo shading kernels (row f-1): N surface points of the bench mesh, reference sample
counts (200 cosine + 128 GGX directions per point), forward and backward."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreammat_amd import _lib, hipops, mesh as pmesh

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = pmesh.displaced_sphere(160, 160)
bvh = hipops.MeshBvh(m.v_pos, m.t_pos_idx, dev)
mat = _lib.MatCfgStruct(0.0, 0.9, 0.01, 0.9)
tv = m.v_pos.float()[m.t_pos_idx.long()]
fn = torch.nn.functional.normalize(torch.cross(tv[:, 1] - tv[:, 0], tv[:, 2] - tv[:, 0], dim=-1), dim=-1)
pick = torch.randint(0, tv.shape[0], (N,))
n = fn[pick].to(dev)
p = (tv.mean(1)[pick].to(dev) + 1e-4 * n).contiguous()
v = torch.nn.functional.normalize(n + 0.6 * torch.randn(N, 3, device=dev), dim=-1)
f = torch.randn(N, 5, device=dev, requires_grad=True)
pix = (torch.arange(N, device=dev, dtype=torch.int32) % 8) * (512 * 512)
env = torch.tensor([0, 1, 2, 3, 4, 0, 1, 2], dtype=torch.int32, device=dev)
nd = torch.full((1,), N, dtype=torch.int32, device=dev)
rd, rs = torch.rand(N, device=dev), torch.rand(N, device=dev)
lights = [torch.rand(512, 1024, 3) for _ in range(5)]
results = []
# (kernel, tracer, BVH width / grid resolution): the round-2 best (wave x 4-wide tree) next to the occupancy grid
cases = [("wave", "bvh", "4")] + [("wave", "grid", r) for r in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "64", "96"])]
if "--all" in sys.argv:
    cases = [("serial", "bvh", "2"), ("serial", "bvh", "4"), ("wave", "bvh", "2")] + cases
ref_color = None
for kernel, tracer, width in cases:
    if True:
        os.environ["DREAMMAT_BVH"] = width if tracer == "bvh" else "4"
        os.environ["DREAMMAT_MC_KERNEL"] = kernel
        os.environ["DREAMMAT_MC_TRACER"] = tracer
        if tracer == "grid":
            bvh = hipops.MeshBvh(m.v_pos, m.t_pos_idx, dev, grid_res=int(width))
        scene = hipops.McScene(bvh, lights, 200, 128, "schlick")

        def run():
            return hipops.mc_shade(f, p, n, v, pix, nd, env, scene, mat, 512 * 512, rd, rs, False)[0]
        c = run(); torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(); c = run(); e[1].record(); c.sum().backward(); e[2].record(); torch.cuda.synchronize()
        tf, tb = e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])
        if ref_color is None:
            ref_color = c.detach().clone()
        n_diff = int(((c - ref_color).abs().max(-1).values > 1e-6).sum())
        res = {"kernel": kernel, "tracer": tracer, "bvh_width_or_grid_res": int(width), "max_abs_diff_vs_first": float((c - ref_color).abs().max()), "pixels_that_differ": n_diff, "N": N, "samples": 328, "tris": int(m.t_pos_idx.shape[0]),
               "fwd_ms": tf, "bwd_ms": tb, "fwd_Grays_per_s": N * 328 / tf / 1e6, "color_mean": float(c.mean())}
        if hasattr(_lib.lib(), "dm_mc_debug_stats") and "--stats" in sys.argv:
            import ctypes
            st = (ctypes.c_ulonglong * 8)()
            _lib.lib().dm_mc_debug_stats(st)            # clear what the warm-up and the timed run left
            run(); torch.cuda.synchronize()
            _lib.lib().dm_mc_debug_stats(st)
            q = list(st)
            res["stats"] = {"wave_rays": q[0], "rounds": q[1], "cell_walk_trips": q[2], "cell_walk_lanes": q[3], "pair_batches": q[4], "pairs": q[5]}
        print(json.dumps(res), flush=True)
        results.append(res)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(results, open("gpurun_out/mc_probe.json", "w"), indent=1)
