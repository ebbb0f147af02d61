"""Mesh container + loaders (threestudio/models/mesh.py:12-30 `Mesh`, and the load/normalise part of
threestudio/models/geometry/dreammat_mesh.py:142-227 without trimesh/xatlas).  Also the procedural
meshes of the benchmark configurations (quad, displaced UV-sphere)."""
import math

import numpy as np
import torch


class Mesh:
    def __init__(self, v_pos, t_pos_idx, v_nrm=None, v_tex=None, t_tex_idx=None):
        self.v_pos = v_pos
        self.t_pos_idx = t_pos_idx
        self._v_nrm = v_nrm
        self.v_tex = v_tex
        self.t_tex_idx = t_tex_idx if t_tex_idx is not None else t_pos_idx

    @property
    def v_nrm(self):
        if self._v_nrm is None:
            self._v_nrm = compute_vertex_normals(self.v_pos, self.t_pos_idx)
        return self._v_nrm

    def to(self, device):
        self.v_pos = self.v_pos.to(device)
        self.t_pos_idx = self.t_pos_idx.to(device)
        if self._v_nrm is not None:
            self._v_nrm = self._v_nrm.to(device)
        if self.v_tex is not None:
            self.v_tex = self.v_tex.to(device)
        self.t_tex_idx = self.t_tex_idx.to(device)
        return self


def compute_vertex_normals(v, f):
    """area-weighted vertex normals (threestudio/models/mesh.py:117-160)."""
    v = torch.as_tensor(v, dtype=torch.float32)
    f = torch.as_tensor(f).long()
    v0, v1, v2 = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    fn = torch.cross(v1 - v0, v2 - v0, dim=-1)
    n = torch.zeros_like(v)
    for k in range(3):
        n.index_add_(0, f[:, k], fn)
    n = torch.where((n * n).sum(-1, keepdim=True) > 1e-20, n, torch.tensor([0.0, 0.0, 1.0]))
    return torch.nn.functional.normalize(n, dim=-1)


def load_obj(path):
    """Minimal OBJ reader: v / vn / vt / f (triangles or fans); vertices are split so that every
    face corner has one position+normal+uv (the reference's trimesh loader does the same merge)."""
    vs, vns, vts, corners, faces = [], [], [], {}, []
    with open(path) as fh:
        for line in fh:
            if line.startswith("v "):
                vs.append([float(x) for x in line.split()[1:4]])
            elif line.startswith("vn "):
                vns.append([float(x) for x in line.split()[1:4]])
            elif line.startswith("vt "):
                vts.append([float(x) for x in line.split()[1:3]])
            elif line.startswith("f "):
                idx = []
                for tok in line.split()[1:]:
                    parts = tok.split("/")
                    key = (int(parts[0]) - 1,
                           int(parts[1]) - 1 if len(parts) > 1 and parts[1] else -1,
                           int(parts[2]) - 1 if len(parts) > 2 and parts[2] else -1)
                    if key not in corners:
                        corners[key] = len(corners)
                    idx.append(corners[key])
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    keys = sorted(corners, key=corners.get)
    v = torch.tensor([vs[k[0]] for k in keys], dtype=torch.float32)
    f = torch.tensor(faces, dtype=torch.int32)
    vt = torch.tensor([vts[k[1]] if k[1] >= 0 else [0.0, 0.0] for k in keys], dtype=torch.float32) if vts else None
    vn = None
    if vns and all(k[2] >= 0 for k in keys):
        vn = torch.nn.functional.normalize(torch.tensor([vns[k[2]] for k in keys], dtype=torch.float32), dim=-1)
    return Mesh(v, f, vn, vt)


def normalize_mesh(mesh, scale):
    """dreammat_mesh.py:161-193: subtract the vertex CENTROID (`vertices.mean(0)`, not the bbox centre) and divide by
    the largest absolute coordinate, then multiply by `shape_init_params`.  The reference hands this normalised mesh
    to Blender for the pre-rendered condition maps, so any other convention misaligns `condition_source: prerender`."""
    v = mesh.v_pos
    v = v - v.mean(0, keepdim=True)
    v = v / v.abs().max() * scale
    mesh.v_pos = v.contiguous()
    return mesh


def quad_mesh():
    """BASELINE config 1: 2-triangle quad in the z=0 plane, normals +z."""
    v = torch.tensor([[-.5, -.5, 0], [.5, -.5, 0], [.5, .5, 0], [-.5, .5, 0]], dtype=torch.float32)
    f = torch.tensor([[0, 1, 2], [0, 2, 3]], dtype=torch.int32)
    n = torch.tensor([[0, 0, 1.0]] * 4, dtype=torch.float32)
    return Mesh(v, f, n, v[:, :2] + 0.5)


def displaced_sphere(n_lon=160, n_lat=160, radius=0.8, amp=0.15):
    """displaced UV sphere, 2*n_lon*(n_lat-1) triangles (160x160 -> 50 880),
    r(theta,phi) = radius*(1 + amp*sin(5 theta) sin(7 phi)), analytic normals, +z up."""
    th = torch.linspace(0, math.pi, n_lat + 1)[1:-1]            # interior rings
    ph = torch.arange(n_lon, dtype=torch.float32) * (2 * math.pi / n_lon)
    T, P = torch.meshgrid(th, ph, indexing="ij")

    def surf(t, p):
        r = radius * (1 + amp * torch.sin(5 * t) * torch.sin(7 * p))
        return torch.stack([r * torch.sin(t) * torch.cos(p), r * torch.sin(t) * torch.sin(p), r * torch.cos(t)], -1)

    t = T.clone().requires_grad_(True)
    p = P.clone().requires_grad_(True)
    X = surf(t, p)
    dt = torch.stack([torch.autograd.grad(X[..., k].sum(), t, retain_graph=True)[0] for k in range(3)], -1)
    dp = torch.stack([torch.autograd.grad(X[..., k].sum(), p, retain_graph=True)[0] for k in range(3)], -1)
    nrm = torch.nn.functional.normalize(torch.cross(dt, dp, dim=-1), dim=-1).detach()
    ring = X.detach().reshape(-1, 3)
    north = torch.tensor([[0, 0, radius]], dtype=torch.float32)
    south = torch.tensor([[0, 0, -radius]], dtype=torch.float32)
    v = torch.cat([ring, north, south])
    n = torch.cat([nrm.reshape(-1, 3), torch.tensor([[0, 0, 1.0]]), torch.tensor([[0, 0, -1.0]])])
    uv = torch.cat([torch.stack([P / (2 * math.pi), T / math.pi], -1).reshape(-1, 2),
                    torch.tensor([[0.5, 0.0]]), torch.tensor([[0.5, 1.0]])])
    nr = n_lat - 1
    i_n, i_s = nr * n_lon, nr * n_lon + 1
    faces = []
    j = np.arange(n_lon)
    jn = (j + 1) % n_lon
    faces.append(np.stack([np.full(n_lon, i_n), j, jn], -1))                       # north cap
    for i in range(nr - 1):
        a, b, c, d = i * n_lon + j, i * n_lon + jn, (i + 1) * n_lon + j, (i + 1) * n_lon + jn
        faces.append(np.stack([a, c, b], -1))
        faces.append(np.stack([b, c, d], -1))
    faces.append(np.stack([(nr - 1) * n_lon + j, np.full(n_lon, i_s), (nr - 1) * n_lon + jn], -1))  # south cap
    f = torch.from_numpy(np.concatenate(faces).astype(np.int32))
    return Mesh(v.contiguous(), f.contiguous(), n.contiguous(), uv.contiguous())
