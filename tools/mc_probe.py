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
for kernel in ("serial", "wave"):
    for width in ("2", "4"):
        os.environ["DREAMMAT_BVH"] = width
        os.environ["DREAMMAT_MC_KERNEL"] = kernel
        scene = hipops.McScene(bvh, lights, 200, 128, "schlick")

        def run():
            return hipops.mc_shade(f, p, n, v, pix, nd, env, scene, mat, 512 * 512, rd, rs, False)[0]
        c = run(); torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(); c = run(); e[1].record(); c.sum().backward(); e[2].record(); torch.cuda.synchronize()
        tf, tb = e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])
        res = {"kernel": kernel, "bvh_width": int(width), "N": N, "samples": 328, "tris": int(m.t_pos_idx.shape[0]),
               "fwd_ms": tf, "bwd_ms": tb, "fwd_Grays_per_s": N * 328 / tf / 1e6, "color_mean": float(c.mean())}
        print(json.dumps(res), flush=True)
        results.append(res)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(results, open("gpurun_out/mc_probe.json", "w"), indent=1)
