"""Occupancy-grid traversal cost on the CPU (no GPU needed), next to tools/bvh_stats.py: cells visited / triangles tested per
shading ray of the bench mesh through the host emulation of csrc/grid_core.h, for several grid resolutions."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreammat_amd import hipops, mesh as pmesh
from tests.hostemu import build as hb

emu = ctypes.CDLL(hb.build())
torch.manual_seed(0)
m = pmesh.displaced_sphere(160, 160)
tv = m.v_pos.float()[m.t_pos_idx.long()]
fn = torch.nn.functional.normalize(torch.cross(tv[:, 1] - tv[:, 0], tv[:, 2] - tv[:, 0], dim=-1), dim=-1)
N = 20000
pick = torch.randint(0, tv.shape[0], (N,))
n = fn[pick]
c = tv.mean(1)[pick]
n = torch.where(((n * c).sum(-1, keepdim=True) < 0), -n, n)
d = torch.nn.functional.normalize(n + torch.nn.functional.normalize(torch.randn(N, 3), dim=-1), dim=-1).contiguous()
o = (c + 1e-4 * n + 1e-5 * d).contiguous()
p = lambda t: ctypes.c_void_p(t.data_ptr())
out = {"tris": int(tv.shape[0]), "rays": N}
for res in [int(a) for a in sys.argv[1:]] or [0, 48, 64, 96]:
    b = hipops.MeshBvh(m.v_pos, m.t_pos_idx, grid_res=res)
    g = b.grid_struct(b.grid_blob_host)
    hit = torch.zeros(N, dtype=torch.uint8)
    st = torch.zeros(N, 3, dtype=torch.int32)
    emu.emu_grid_any_hit(ctypes.byref(g), p(o), p(d), ctypes.c_longlong(N), ctypes.c_float(10.0), p(hit), p(st))
    out[f"res{res}"] = {"dim": [int(x) for x in g.dim], "occupied": int(g.n_occ), "entries": int(g.n_entries),
                        "lds_KB": g.n_words * 4 / 1024, "moves_per_ray": float(st[:, 0].float().mean()),
                        "occupied_cells_per_ray": float(st[:, 1].float().mean()), "tris_per_ray": float(st[:, 2].float().mean()), "moves_max": int(st[:, 0].max()), "hit_frac": float(hit.float().mean())}
print(json.dumps(out, indent=1))
