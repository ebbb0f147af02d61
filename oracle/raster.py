"""ctypes front-end of oracle/raster_ref.c (TEST INFRASTRUCTURE).

Restates the nvdiffrast calls behind threestudio/utils/rasterize.py:22-78:
  vertex_transform (rasterize.py:22-28), dr.rasterize (:37), dr.interpolate (:66-68),
  dr.antialias (:56).  See raster_ref.c for the rules; parity with real nvdiffrast is UNPINNED.
"""
import ctypes

import numpy as np
import torch

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_build.build())
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def vertex_transform(verts, mvp):
    """rasterize.py:22-28 -- cat(v,1) @ mvp^T in fp32.  The k-summation order is fixed here to
    ((x*m0 + y*m1) + z*m2) + m3, one rounding per op (the reference's cuBLAS order is unspecified),
    so the HIP kernel can reproduce the clip coordinates bit for bit."""
    v = torch.as_tensor(_np(verts), dtype=torch.float32)
    m = torch.as_tensor(_np(mvp), dtype=torch.float32)
    x, y, z = v[None, :, 0:1], v[None, :, 1:2], v[None, :, 2:3]
    r = m[:, None, :, :]  # [B,1,4,4]
    return ((x * r[..., 0] + y * r[..., 1]) + z * r[..., 2]) + r[..., 3]


def rasterize(pos, tri, H, W):
    pos = _f(_np(pos)); tri = _i(_np(tri))
    B, Nv, _ = pos.shape
    out = np.empty((B, H, W, 4), np.float32)
    rc = lib().dmo_rasterize(_p(pos), B, Nv, _p(tri), tri.shape[0], H, W, _p(out))
    assert rc == 0
    return out


def build_topology(tri):
    tri = _i(_np(tri))
    opp = np.empty_like(tri)
    assert lib().dmo_build_topology(_p(tri), tri.shape[0], _p(opp)) == 0
    return opp


def antialias_plan(pos, tri, opp, rast):
    pos = _f(_np(pos)); tri = _i(_np(tri)); opp = _i(_np(opp)); rast = _f(_np(rast))
    B, H, W, _ = rast.shape
    plan = np.empty((B, H, W, 2), np.float32)
    assert lib().dmo_antialias_plan(_p(pos), B, pos.shape[1], _p(tri), _p(opp), tri.shape[0], _p(rast), H, W,
                                    _p(plan)) == 0
    return plan


def antialias_apply(color, plan):
    color = _f(_np(color)); plan = _f(_np(plan))
    B, H, W, C = color.shape
    out = np.empty_like(color)
    assert lib().dmo_antialias_apply(_p(color), _p(plan), B, H, W, C, _p(out)) == 0
    return out


def antialias_grad(dout, plan):
    dout = _f(_np(dout)); plan = _f(_np(plan))
    B, H, W, C = dout.shape
    g = np.empty_like(dout)
    assert lib().dmo_antialias_grad(_p(dout), _p(plan), B, H, W, C, _p(g)) == 0
    return g


def interpolate(attr, rast, tri):
    attr = _f(_np(attr)); rast = _f(_np(rast)); tri = _i(_np(tri))
    B, H, W, _ = rast.shape
    out = np.empty((B, H, W, attr.shape[1]), np.float32)
    assert lib().dmo_interpolate(_p(attr), attr.shape[0], attr.shape[1], _p(tri), _p(rast), B, H, W, _p(out)) == 0
    return out


class Antialias(torch.autograd.Function):
    """dr.antialias with the colour gradient only (mesh is fixed in DreamMat)."""

    @staticmethod
    def forward(ctx, color, plan):
        ctx.plan = plan
        return torch.from_numpy(antialias_apply(color, plan))

    @staticmethod
    def backward(ctx, g):
        return torch.from_numpy(antialias_grad(g.contiguous(), ctx.plan)), None


def brute_force_cover(pos, tri, H, W):
    """Self-check: float64 point-in-triangle on the UNSNAPPED projection.
    Returns (id_map [B,H,W] int, margin [B,H,W] f64 = min |edge distance| in pixels of the winner),
    so tests can compare with dmo_rasterize away from edges."""
    pos = _np(pos).astype(np.float64); tri = _np(tri)
    B = pos.shape[0]
    ids = np.zeros((B, H, W), np.int64)
    depth = np.full((B, H, W), np.inf)
    margin = np.zeros((B, H, W))
    ys, xs = np.meshgrid(np.arange(H) + 0.5, np.arange(W) + 0.5, indexing="ij")
    for b in range(B):
        sx = (pos[b, :, 0] / pos[b, :, 3] * 0.5 + 0.5) * W
        sy = (pos[b, :, 1] / pos[b, :, 3] * 0.5 + 0.5) * H
        zw = pos[b, :, 2] / pos[b, :, 3]
        iw = 1.0 / pos[b, :, 3]
        for t, (i0, i1, i2) in enumerate(tri):
            x0, y0, x1, y1, x2, y2 = sx[i0], sy[i0], sx[i1], sy[i1], sx[i2], sy[i2]
            area = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0)
            if abs(area) < 1e-12:
                continue
            lo_x, hi_x = int(max(0, np.floor(min(x0, x1, x2)) - 1)), int(min(W, np.ceil(max(x0, x1, x2)) + 1))
            lo_y, hi_y = int(max(0, np.floor(min(y0, y1, y2)) - 1)), int(min(H, np.ceil(max(y0, y1, y2)) + 1))
            if lo_x >= hi_x or lo_y >= hi_y:
                continue
            X = xs[lo_y:hi_y, lo_x:hi_x]; Y = ys[lo_y:hi_y, lo_x:hi_x]
            e0 = ((x2 - x1) * (Y - y1) - (y2 - y1) * (X - x1)) / area
            e1 = ((x0 - x2) * (Y - y2) - (y0 - y2) * (X - x2)) / area
            e2 = 1.0 - e0 - e1
            inside = (e0 > 0) & (e1 > 0) & (e2 > 0)
            if not inside.any():
                continue
            # screen-space affine z/w
            z = e0 * zw[i0] + e1 * zw[i1] + e2 * zw[i2]
            L = [np.hypot(x2 - x1, y2 - y1), np.hypot(x0 - x2, y0 - y2), np.hypot(x1 - x0, y1 - y0)]
            dist = np.minimum(np.minimum(e0 * abs(area) / L[0], e1 * abs(area) / L[1]), e2 * abs(area) / L[2])
            D = depth[b, lo_y:hi_y, lo_x:hi_x]
            upd = inside & (z < D)
            D[upd] = z[upd]
            ids[b, lo_y:hi_y, lo_x:hi_x][upd] = t + 1
            margin[b, lo_y:hi_y, lo_x:hi_x][upd] = dist[upd]
    return ids, margin, depth
