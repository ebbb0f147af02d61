"""Environment lighting + texture lookups (TEST INFRASTRUCTURE; fp32 torch, autograd-friendly).

Restates, from their published behaviour (un-vendored deps => PARITY UNPINNED):
  * ashawkey/envlight `EnvLight(path, scale)` / `__call__(dirs[, roughness])`
    (requirements.txt:24; call sites threestudio/models/materials/dreammat_material.py:383,696-697),
    itself a wrapper over nvdiffrec's light.py + renderutils cubemap kernels:
      latlong -> 6 x R x R cubemap (bilinear, wrap), 2x2 average mip chain down to `min_res`,
      GGX-prefiltered specular per mip (roughness linspace [0.08,0.5], last mip 1.0, cutoff 0.99),
      cosine-convolved diffuse cube from the smallest mip, roughness -> mip piecewise-linear.
  * nvdiffrast `dr.texture` in the three modes the path uses: 2-D bilinear/clamp (FG LUT,
    dreammat_material.py:687-692), cube bilinear, cube trilinear with mip_level_bias.

Cube-seam rule of THIS restatement (nvdiffrast's own wrap tables are not reproduced): a bilinear
tap that falls outside its face is replaced by the nearest texel, in whichever face it lands,
along the direction through that tap's centre on the extended face plane (SEAM_MODE = "nearest",
the rule the product's atlas borders implement).

SEAM_MODE = "edge_wrap" is an ALTERNATE mode that follows nvdiffrast's documented seam filtering
(texture.cu wrapCubeMap, restated from its published behaviour): a tap that leaves the face across
ONE edge reads the adjacent face's edge texel with the SAME index along the shared edge, and a tap
that leaves across a corner (both coordinates out: no such texel exists on a cube) is dropped and
the remaining three weights are renormalised.  It exists to MEASURE how far the product's rule is
from that behaviour (tools/seam_delta.py) -- it is not what the parity tests use.
"""
import math

import numpy as np
import torch

MIN_ROUGHNESS = 0.08
MAX_ROUGHNESS = 0.5


# ---------------------------------------------------------------- Radiance .hdr reader
def load_hdr(path):
    """RGBE (.hdr) -> [H,W,3] float32, like cv2.imread(..., IMREAD_UNCHANGED) + BGR2RGB."""
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    hdr_lines = []
    while True:
        end = data.index(b"\n", pos)
        line = data[pos:end]
        pos = end + 1
        if line == b"":
            break
        hdr_lines.append(line)
    end = data.index(b"\n", pos)
    res = data[pos:end].split()
    pos = end + 1
    assert res[0] == b"-Y" and res[2] == b"+X", res
    H, W = int(res[1]), int(res[3])
    img = np.zeros((H, W, 4), np.uint8)
    buf = np.frombuffer(data, np.uint8)
    for y in range(H):
        if buf[pos] == 2 and buf[pos + 1] == 2 and (int(buf[pos + 2]) << 8 | int(buf[pos + 3])) == W:
            pos += 4
            for c in range(4):
                x = 0
                while x < W:
                    n = int(buf[pos]); pos += 1
                    if n > 128:
                        n -= 128
                        img[y, x:x + n, c] = buf[pos]; pos += 1
                    else:
                        img[y, x:x + n, c] = buf[pos:pos + n]; pos += n
                    x += n
        else:  # flat scanline
            img[y] = buf[pos:pos + 4 * W].reshape(W, 4); pos += 4 * W
    e = img[..., 3].astype(np.int32)
    f = np.where(e > 0, np.ldexp(1.0, e - 136), 0.0).astype(np.float32)
    return (img[..., :3].astype(np.float32) * f[..., None]).astype(np.float32)


# ---------------------------------------------------------------- cube geometry
def cube_to_dir(s, x, y):
    one = torch.ones_like(x)
    if s == 0:
        r = (one, -y, -x)
    elif s == 1:
        r = (-one, -y, x)
    elif s == 2:
        r = (x, one, y)
    elif s == 3:
        r = (x, -one, -y)
    elif s == 4:
        r = (x, -y, one)
    else:
        r = (-x, -y, -one)
    return torch.stack(r, dim=-1)


def cube_dirs_all(face, gx, gy):
    """vectorised cube_to_dir for a tensor of face ids."""
    one = torch.ones_like(gx)
    out = torch.zeros(gx.shape + (3,), dtype=gx.dtype)
    tab = [(one, -gy, -gx), (-one, -gy, gx), (gx, one, gy), (gx, -one, -gy), (gx, -gy, one), (-gx, -gy, -one)]
    for s in range(6):
        m = face == s
        if m.any():
            for k in range(3):
                out[..., k] = torch.where(m, tab[s][k], out[..., k])
    return out


def cube_index(d):
    """nvdiffrast indexCubeMap: direction -> (face, u, v) with u,v in [0,1]."""
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    ax, ay, az = x.abs(), y.abs(), z.abs()
    is_z = az > torch.maximum(ax, ay)
    is_y = (~is_z) & (ay > ax)
    is_x = ~(is_z | is_y)
    c = torch.where(is_z, z, torch.where(is_y, y, x))
    face = torch.where(is_z, 4, torch.where(is_y, 2, 0)) + (c < 0).long()
    m = 0.5 / c.abs()
    # per-face (a, b): u = a*m+.5, v = b*m+.5
    a = torch.where(is_x, torch.where(c > 0, -z, z),
                    torch.where(is_y, x, torch.where(c > 0, x, -x)))
    b = torch.where(is_x, -y, torch.where(is_y, torch.where(c > 0, z, -z), -y))
    u = (a * m + 0.5).clamp(0, 1)
    v = (b * m + 0.5).clamp(0, 1)
    return face, u, v


SEAM_MODE = "nearest"      # "nearest" (the parity contract) | "edge_wrap" (measurement only, see the module docstring)


def _resolve_tap(face, ix, iy, R):
    """Seam rule: out-of-face taps -> nearest texel along the extended-plane direction.
    Returns (face, ix, iy, keep): keep = False only in "edge_wrap" mode for corner taps (dropped, weights renormalised)."""
    oobx = (ix < 0) | (ix >= R)
    ooby = (iy < 0) | (iy >= R)
    oob = oobx | ooby
    keep = torch.ones_like(oob)
    if not oob.any():
        return face, ix, iy, keep
    gx = (2.0 * (ix.float() + 0.5) / R - 1.0)
    gy = (2.0 * (iy.float() + 0.5) / R - 1.0)
    if SEAM_MODE == "edge_wrap":
        # the out coordinate becomes the new major axis (magnitude 1 + 1/R): scaling the in-range coordinate by the same
        # factor keeps its position ALONG the shared edge after the re-projection, i.e. the same texel index there
        s = 1.0 + 1.0 / R
        gx = torch.where(ooby & ~oobx, gx * s, gx)
        gy = torch.where(oobx & ~ooby, gy * s, gy)
        keep = ~(oobx & ooby)
    elif SEAM_MODE != "nearest":
        raise ValueError(SEAM_MODE)
    d = cube_dirs_all(face, gx, gy)
    f2, u2, v2 = cube_index(d)
    ix2 = (u2 * R).floor().long().clamp(0, R - 1)
    iy2 = (v2 * R).floor().long().clamp(0, R - 1)
    return torch.where(oob, f2, face), torch.where(oob, ix2, ix), torch.where(oob, iy2, iy), keep


def cube_bilinear(tex, d):
    """tex [6,R,R,C], d [N,3] -> [N,C]  (dr.texture boundary_mode='cube', filter 'linear')."""
    R = tex.shape[1]
    face, u, v = cube_index(d)
    x = u * R - 0.5
    y = v * R - 0.5
    x0 = x.floor(); y0 = y.floor()
    fx = (x - x0)[..., None]; fy = (y - y0)[..., None]
    ix0 = x0.long(); iy0 = y0.long()
    out, wsum = 0, 0
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            f, ix, iy, keep = _resolve_tap(face, ix0 + dx, iy0 + dy, R)
            w = wx * wy
            if SEAM_MODE != "nearest":
                w = w * keep[..., None].to(w.dtype)
                wsum = wsum + w
            out = out + tex[f, iy, ix] * w
    return out if SEAM_MODE == "nearest" else out / wsum


def texture2d_linear_clamp(tex, uv):
    """tex [H,W,C], uv [N,2] -> [N,C]  (dr.texture filter 'linear', boundary 'clamp')."""
    H, W = tex.shape[0], tex.shape[1]
    x = uv[..., 0] * W - 0.5
    y = uv[..., 1] * H - 0.5
    x0 = x.floor(); y0 = y.floor()
    fx = (x - x0)[..., None]; fy = (y - y0)[..., None]
    ix0 = x0.long(); iy0 = y0.long()
    out = 0
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            out = out + tex[(iy0 + dy).clamp(0, H - 1), (ix0 + dx).clamp(0, W - 1)] * (wx * wy)
    return out


def texture2d_linear_wrap(tex, uv):
    H, W = tex.shape[0], tex.shape[1]
    x = uv[..., 0] * W - 0.5
    y = uv[..., 1] * H - 0.5
    x0 = x.floor(); y0 = y.floor()
    fx = (x - x0)[..., None]; fy = (y - y0)[..., None]
    ix0 = x0.long(); iy0 = y0.long()
    out = 0
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            out = out + tex[(iy0 + dy) % H, (ix0 + dx) % W] * (wx * wy)
    return out


# ---------------------------------------------------------------- nvdiffrec light.py / cubemap.cu
def latlong_to_cubemap(latlong, res):
    cm = torch.zeros(6, res, res, latlong.shape[-1], dtype=torch.float32)
    lin = torch.linspace(-1.0 + 1.0 / res, 1.0 - 1.0 / res, res)
    gy, gx = torch.meshgrid(lin, lin, indexing="ij")
    for s in range(6):
        v = torch.nn.functional.normalize(cube_to_dir(s, gx, gy), dim=-1)
        tu = torch.atan2(v[..., 0:1], -v[..., 2:3]) / (2 * np.pi) + 0.5
        tv = torch.acos(torch.clamp(v[..., 1:2], min=-1, max=1)) / np.pi
        uv = torch.cat((tu, tv), dim=-1).reshape(-1, 2)
        cm[s] = texture2d_linear_wrap(latlong, uv).reshape(res, res, -1)
    return cm


def _texel_dirs(R):
    """unit directions [6,R,R,3] and solid-angle weights [R,R] (cubemap.cu cube_to_dir / pixel_area)."""
    c = 2.0 * ((torch.arange(R, dtype=torch.float32) + 0.5) / R) - 1.0
    gy, gx = torch.meshgrid(c, c, indexing="ij")
    dirs = torch.stack([torch.nn.functional.normalize(cube_to_dir(s, gx, gy), dim=-1) for s in range(6)])
    if R > 1:
        Hh = R // 2
        k = (torch.arange(R) - Hh).abs().float()
        dx = torch.atan((k + 1) / Hh) - torch.atan(k / Hh)
        area = dx[None, :] * dx[:, None]
    else:
        area = torch.ones(1, 1)
    return dirs, area


def cubemap_mip(cm):
    return 0.25 * (cm[:, 0::2, 0::2] + cm[:, 1::2, 0::2] + cm[:, 0::2, 1::2] + cm[:, 1::2, 1::2])


def diffuse_cubemap(cm):
    R = cm.shape[1]
    dirs, area = _texel_dirs(R)
    N = dirs.reshape(-1, 3)
    L = dirs.reshape(-1, 3)
    cos = (N @ L.t()).clamp(0.0, 0.999)
    w = cos * area.expand(6, R, R).reshape(1, -1) / 3.141592
    return (w @ cm.reshape(-1, cm.shape[-1])).reshape(6, R, R, -1)


def _ndf_ggx(alpha_sqr, cos_theta):
    c = cos_theta.clamp(0.0, 1.0)
    d = (c * alpha_sqr - c) * c + 1.0
    return alpha_sqr / (d * d * math.pi)


def ggx_cutoff_cos(roughness, cutoff=0.99, n_samples=1000000):
    """nvdiffrec __ndfBounds: cos(theta) at which the cumulative NDF reaches `cutoff`."""
    cos = np.cos(np.linspace(0, np.pi / 2.0, n_samples))
    a2 = roughness ** 4
    c = np.clip(cos, 0.0, 1.0)
    d = (c * a2 - c) * c + 1.0
    D = np.cumsum(a2 / (d * d * np.pi))
    idx = np.argmax(D >= D[-1] * cutoff)
    return float(cos[idx])


def specular_cubemap(cm, roughness, cutoff=0.99, chunk=2048):
    R = cm.shape[1]
    dirs, area = _texel_dirs(R)
    L = dirs.reshape(-1, 3)
    col = cm.reshape(-1, cm.shape[-1])
    aw = area.expand(6, R, R).reshape(-1)
    cos_cut = ggx_cutoff_cos(roughness, cutoff)
    a2 = (roughness * roughness) ** 2
    out = torch.zeros_like(col)
    for s in range(0, L.shape[0], chunk):
        V = L[s:s + chunk]
        ldv = V @ L.t()
        H = torch.nn.functional.normalize(L[None, :, :] + V[:, None, :], dim=-1)
        vdh = (H * V[:, None, :]).sum(-1).clamp(min=0.0)
        w = ldv.clamp(min=0.0) * _ndf_ggx(a2, vdh) * aw[None, :] / 4.0
        w = torch.where(ldv >= cos_cut, w, torch.zeros_like(w))
        out[s:s + chunk] = (w @ col) / w.sum(-1, keepdim=True)
    return out.reshape(cm.shape)


class EnvLight:
    """envlight.EnvLight(path, scale=...)  (min_res=16, max_res=128 upstream defaults)."""

    def __init__(self, latlong, scale=1.0, min_res=16, max_res=128):
        latlong = torch.as_tensor(latlong, dtype=torch.float32) * scale
        self.base = latlong_to_cubemap(latlong, max_res)
        self.specular = [self.base]
        while self.specular[-1].shape[1] > min_res:
            self.specular.append(cubemap_mip(self.specular[-1]))
        self.diffuse = diffuse_cubemap(self.specular[-1])
        n = len(self.specular)
        for idx in range(n - 1):
            r = (idx / (n - 2)) * (MAX_ROUGHNESS - MIN_ROUGHNESS) + MIN_ROUGHNESS if n > 2 else MIN_ROUGHNESS
            self.specular[idx] = specular_cubemap(self.specular[idx], r)
        self.specular[-1] = specular_cubemap(self.specular[-1], 1.0)

    def get_mip(self, roughness):
        n = len(self.specular)
        return torch.where(
            roughness < MAX_ROUGHNESS,
            (roughness.clamp(MIN_ROUGHNESS, MAX_ROUGHNESS) - MIN_ROUGHNESS) / (MAX_ROUGHNESS - MIN_ROUGHNESS) * (n - 2),
            (roughness.clamp(MAX_ROUGHNESS, 1.0) - MAX_ROUGHNESS) / (1.0 - MAX_ROUGHNESS) + n - 2)

    def __call__(self, l, roughness=None):
        if roughness is None:
            return cube_bilinear(self.diffuse, l)
        n = len(self.specular)
        level = self.get_mip(roughness)[..., 0].clamp(0, n - 1)
        l0 = level.floor().clamp(max=n - 1)
        f = (level - l0)[..., None]
        l0 = l0.long()
        l1 = (l0 + 1).clamp(max=n - 1)
        out0 = torch.zeros(l.shape[0], self.specular[0].shape[-1])
        out1 = torch.zeros_like(out0)
        for k in range(n):
            m0 = l0 == k
            if m0.any():
                out0[m0] = cube_bilinear(self.specular[k], l[m0])
            m1 = l1 == k
            if m1.any():
                out1[m1] = cube_bilinear(self.specular[k], l[m1])
        return out0 * (1 - f) + out1 * f
