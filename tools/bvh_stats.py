"""BVH quality on the CPU (no GPU needed): nodes visited / triangles tested per shading ray of the bench mesh, through the
host emulation of the traversal core.  Rays = the Monte-Carlo shading pattern: from surface points along cosine-weighted
directions around the normal."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreammat_amd import hipops, mesh as pmesh
from tests.hostemu import build as hb

emu = ctypes.CDLL(hb.build())
torch.manual_seed(0)
m = pmesh.displaced_sphere(160, 160)
bvh = hipops.MeshBvh(m.v_pos, m.t_pos_idx)
tv = m.v_pos.float()[m.t_pos_idx.long()]
fn = torch.nn.functional.normalize(torch.cross(tv[:, 1] - tv[:, 0], tv[:, 2] - tv[:, 0], dim=-1), dim=-1)
N = 20000
pick = torch.randint(0, tv.shape[0], (N,))
n = fn[pick]
# the displaced sphere's outward side: flip normals that point inwards
c = tv.mean(1)[pick]
n = torch.where(((n * c).sum(-1, keepdim=True) < 0), -n, n)
d = torch.nn.functional.normalize(n + torch.nn.functional.normalize(torch.randn(N, 3), dim=-1), dim=-1)   # cosine-weighted
o = (c + 1e-4 * n + 1e-5 * d).contiguous()
d = d.contiguous()
nv, tt, nh = ctypes.c_longlong(0), ctypes.c_longlong(0), ctypes.c_longlong(0)
p = lambda t: ctypes.c_void_p(t.data_ptr())
emu.emu_bvh_stats(p(bvh.nodes_host), p(bvh.tris_host), p(o), p(d), ctypes.c_longlong(N), ctypes.c_float(10.0),
                  ctypes.byref(nv), ctypes.byref(tt), ctypes.byref(nh))
res = {"tris": int(tv.shape[0]), "rays": N, "hit_frac": nh.value / N,
       "bvh2": {"nodes": bvh.n_nodes, "node_bytes": 32, "node_fetches_per_ray": nv.value / N, "tris_per_ray": tt.value / N}}
emu.emu_bvh4_stats(p(bvh.nodes4_host), p(bvh.tris_host), p(o), p(d), ctypes.c_longlong(N), ctypes.c_float(10.0),
                   ctypes.byref(nv), ctypes.byref(tt), ctypes.byref(nh))
res["bvh4"] = {"nodes": bvh.n_nodes4, "node_bytes": 128, "node_fetches_per_ray": nv.value / N, "tris_per_ray": tt.value / N,
               "hit_frac": nh.value / N}
print(json.dumps(res))
