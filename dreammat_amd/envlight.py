"""Environment-light atlas for the fused split-sum shade kernel.

Plays the role of `envlight.EnvLight(path, scale)` for all env maps at once
(reference: threestudio/models/materials/dreammat_material.py:376-387 builds 5 EnvLight objects and
indexes them with `self.envlight[env_id]`, :696-697).  The pre-filter runs once at configure time in
PyTorch on whatever device the material lives on (ROCm on the GPU box); the per-pixel lookups are NOT
done here -- they are fused into the HIP shade kernel (csrc/shade.hip), which reads the packed atlas
built by `EnvAtlas.pack()`:

  RGBA fp32 texels; every cube face stored with a 1-texel border, so the 2x2 bilinear footprint of
  any direction is one unconditional 4-tap read.  Border texels hold the nearest texel of whichever
  face the tap's direction lands in (the seam rule documented in DESIGN.md).

Algorithm (published behaviour of ashawkey/envlight -> nvdiffrec light.py / renderutils):
  latlong -> cube (bilinear, wrap) at `max_res`; 2x2-average mips down to `min_res`; specular mip i
  GGX-prefiltered at roughness linspace(0.08, 0.5) (last mip 1.0) over the 0.99-energy cone;
  diffuse = cosine convolution of the smallest mip; roughness -> mip level piecewise linear.
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import _lib

MIN_ROUGHNESS = 0.08
MAX_ROUGHNESS = 0.5


def read_hdr(path):
    """Radiance RGBE reader (the reference goes through cv2.imread).  Returns [H,W,3] float32 RGB."""
    raw = open(path, "rb").read()
    end = raw.index(b"\n\n") + 2
    nl = raw.index(b"\n", end)
    tok = raw[end:nl].split()
    if tok[0] != b"-Y" or tok[2] != b"+X":
        raise ValueError(f"unsupported .hdr orientation {tok}")
    H, W = int(tok[1]), int(tok[3])
    buf = np.frombuffer(raw, np.uint8, offset=nl + 1)
    out = np.empty((H, W, 4), np.uint8)
    p = 0
    for y in range(H):
        if W >= 8 and W < 32768 and buf[p] == 2 and buf[p + 1] == 2 and ((int(buf[p + 2]) << 8) | int(buf[p + 3])) == W:
            p += 4
            for c in range(4):
                x = 0
                row = out[y, :, c]
                while x < W:
                    n = int(buf[p]); p += 1
                    if n > 128:
                        n -= 128
                        row[x:x + n] = buf[p]; p += 1
                    else:
                        row[x:x + n] = buf[p:p + n]; p += n
                    x += n
        else:
            out[y] = buf[p:p + 4 * W].reshape(W, 4); p += 4 * W
    e = out[..., 3].astype(np.int32)
    scale = np.where(e > 0, np.ldexp(np.float32(1.0), e - 136), np.float32(0)).astype(np.float32)
    return out[..., :3].astype(np.float32) * scale[..., None]


# --------------------------------------------------------------------------- cube geometry
def _face_dirs(face, gx, gy):
    one = torch.ones_like(gx)
    table = ((one, -gy, -gx), (-one, -gy, gx), (gx, one, gy), (gx, -one, -gy), (gx, -gy, one), (-gx, -gy, -one))
    return torch.stack(table[face], dim=-1)


def _dir_to_face_uv(d):
    x, y, z = d.unbind(-1)
    ax, ay, az = x.abs(), y.abs(), z.abs()
    zmaj = az > torch.maximum(ax, ay)
    ymaj = (~zmaj) & (ay > ax)
    xmaj = ~(zmaj | ymaj)
    c = torch.where(zmaj, z, torch.where(ymaj, y, x))
    pos = c > 0
    face = torch.where(zmaj, 4, torch.where(ymaj, 2, 0)) + (c < 0).long()
    m = 0.5 / c.abs()
    a = torch.where(xmaj, torch.where(pos, -z, z), torch.where(ymaj, x, torch.where(pos, x, -x)))
    b = torch.where(ymaj, torch.where(pos, z, -z), -y)
    return face, (a * m + 0.5).clamp(0, 1), (b * m + 0.5).clamp(0, 1)


def _bilinear_wrap(img, u, v):
    H, W = img.shape[:2]
    x = u * W - 0.5
    y = v * H - 0.5
    x0 = torch.floor(x); y0 = torch.floor(y)
    fx = (x - x0).unsqueeze(-1); fy = (y - y0).unsqueeze(-1)
    x0 = x0.long(); y0 = y0.long()
    x1 = (x0 + 1) % W; y1 = (y0 + 1) % H
    x0 = x0 % W; y0 = y0 % H
    top = img[y0, x0] * (1 - fx) + img[y0, x1] * fx
    bot = img[y1, x0] * (1 - fx) + img[y1, x1] * fx
    return top * (1 - fy) + bot * fy


def latlong_to_cube(latlong, res):
    dev = latlong.device
    lin = torch.linspace(-1.0 + 1.0 / res, 1.0 - 1.0 / res, res, device=dev)
    gy, gx = torch.meshgrid(lin, lin, indexing="ij")
    faces = []
    for s in range(6):
        v = torch.nn.functional.normalize(_face_dirs(s, gx, gy), dim=-1)
        tu = torch.atan2(v[..., 0], -v[..., 2]) / (2 * math.pi) + 0.5
        tv = torch.acos(v[..., 1].clamp(-1, 1)) / math.pi
        faces.append(_bilinear_wrap(latlong, tu, tv))
    return torch.stack(faces)


def _texel_geometry(res, dev):
    c = 2.0 * ((torch.arange(res, device=dev, dtype=torch.float32) + 0.5) / res) - 1.0
    gy, gx = torch.meshgrid(c, c, indexing="ij")
    dirs = torch.stack([torch.nn.functional.normalize(_face_dirs(s, gx, gy), dim=-1) for s in range(6)])
    if res > 1:
        half = res // 2
        k = (torch.arange(res, device=dev) - half).abs().float()
        dx = torch.atan((k + 1) / half) - torch.atan(k / half)
        area = dx[None, :] * dx[:, None]
    else:
        area = torch.ones(1, 1, device=dev)
    return dirs.reshape(-1, 3), area.expand(6, res, res).reshape(-1)


def _ggx_cone_cos(roughness, cutoff):
    cos = np.cos(np.linspace(0, np.pi / 2.0, 1000000))
    a2 = roughness ** 4
    d = (cos * a2 - cos) * cos + 1.0
    cum = np.cumsum(a2 / (d * d * np.pi))
    return float(cos[np.argmax(cum >= cum[-1] * cutoff)])


def prefilter_specular(cube, roughness, cutoff=0.99, rows_per_chunk=4096):
    res = cube.shape[1]
    L, area = _texel_geometry(res, cube.device)
    rad = cube.reshape(-1, cube.shape[-1])
    cos_cut = _ggx_cone_cos(roughness, cutoff)
    a2 = (roughness * roughness) ** 2
    out = torch.empty_like(rad)
    for s in range(0, L.shape[0], rows_per_chunk):
        V = L[s:s + rows_per_chunk]
        ldv = V @ L.t()
        hv = (1.0 + ldv)                         # |L+V|^2 = 2 + 2 L.V ; V.H = (1 + L.V)/|L+V|
        vdh = (hv / torch.sqrt((2.0 * hv).clamp(min=1e-20))).clamp(min=0.0, max=1.0)
        dd = (vdh * a2 - vdh) * vdh + 1.0
        w = ldv.clamp(min=0.0) * (a2 / (dd * dd * math.pi)) * area[None, :] * 0.25
        w = w * (ldv >= cos_cut)
        out[s:s + rows_per_chunk] = (w @ rad) / w.sum(-1, keepdim=True)
    return out.reshape(cube.shape)


def convolve_diffuse(cube):
    res = cube.shape[1]
    L, area = _texel_geometry(res, cube.device)
    w = (L @ L.t()).clamp(0.0, 0.999) * area[None, :] / 3.141592
    return (w @ cube.reshape(-1, cube.shape[-1])).reshape(cube.shape)


def _pad_faces(cube):
    """[6,R,R,3] -> [6,R+2,R+2,4] with the seam-rule border and alpha = 0."""
    R = cube.shape[1]
    dev = cube.device
    idx = torch.arange(-1, R + 1, device=dev)
    iy, ix = torch.meshgrid(idx, idx, indexing="ij")
    out = torch.zeros(6, R + 2, R + 2, 4, device=dev, dtype=torch.float32)
    gx = 2.0 * (ix.float() + 0.5) / R - 1.0
    gy = 2.0 * (iy.float() + 0.5) / R - 1.0
    inside = (ix >= 0) & (ix < R) & (iy >= 0) & (iy < R)
    for f in range(6):
        d = _face_dirs(f, gx, gy)
        f2, u2, v2 = _dir_to_face_uv(d)
        jx = (u2 * R).floor().long().clamp(0, R - 1)
        jy = (v2 * R).floor().long().clamp(0, R - 1)
        f2 = torch.where(inside, torch.full_like(f2, f), f2)
        jx = torch.where(inside, ix.clamp(0, R - 1), jx)
        jy = torch.where(inside, iy.clamp(0, R - 1), jy)
        out[f, :, :, :3] = cube[f2, jy, jx]
    return out


TEXEL_FORMATS = {"fp32": 0, "fp16": 1, "rgb18e8": 2}


def encode_rgb18e8(rgb):
    """[..., 3] non-negative fp32 -> [..., 2] int32 words of the 8-byte shared-exponent texel the shade kernels decode
    (csrc/shade_core.h rgb18e8_decode): bits R[0,18) G[18,36) B[36,54) E[55,63) -- the exponent sits where an fp32 keeps its own, so the
    kernel turns it into the scale 2^(E-127) with ONE and-mask --, value = mantissa * 2^(E-127); the exponent is chosen so that
    the largest channel uses all 18 bits."""
    rgb = rgb.double().clamp(min=0.0)
    mx = rgb.amax(-1)
    _, ex = torch.frexp(mx)                                  # mx = m * 2^ex, m in [0.5, 1)
    ex = ex.clamp(-100, 127).to(torch.int64)
    man = torch.round(torch.ldexp(rgb, (18 - ex)[..., None].expand_as(rgb).to(torch.int32)))
    over = man.amax(-1) >= 2 ** 18                           # rounding carried into bit 18: one exponent up
    ex = ex + over.to(torch.int64)
    man = torch.round(torch.ldexp(rgb, (18 - ex)[..., None].expand_as(rgb).to(torch.int32))).clamp(0, 2 ** 18 - 1).to(torch.int64)
    e = ex - 18 + 127
    assert int(e.min()) >= 1 and int(e.max()) <= 254
    word = man[..., 0] | (man[..., 1] << 18) | (man[..., 2] << 36) | (e << 55)
    lo = word & 0xFFFFFFFF
    hi = (word >> 32) & 0xFFFFFFFF
    lo = torch.where(lo >= 2 ** 31, lo - 2 ** 32, lo)
    hi = torch.where(hi >= 2 ** 31, hi - 2 ** 32, hi)
    return torch.stack([lo, hi], -1).to(torch.int32).contiguous()


def decode_rgb18e8(words):
    """inverse of encode_rgb18e8 (tests): [..., 2] int32 -> [..., 3] fp32."""
    lo = words[..., 0].to(torch.int64) & 0xFFFFFFFF
    hi = words[..., 1].to(torch.int64) & 0xFFFFFFFF
    word = lo | (hi << 32)
    man = torch.stack([word & 0x3FFFF, (word >> 18) & 0x3FFFF, (word >> 36) & 0x3FFFF], -1).double()
    e = ((word >> 55) & 0xFF).to(torch.int32)
    return torch.ldexp(man, (e - 127)[..., None].expand_as(man)).float()


def fg_pair_table(fg_lut):
    """[L, L, 2] -> [L, L+1, 4]: entry (row, x0+1) = (lut[row, max(x0,0)], lut[row, min(x0+1, L-1)]) for x0 in [-1, L-1], the
    clamped x-pair of one bilinear row as one aligned 16 B record (same fp32 values, half the gathers)."""
    L = fg_lut.shape[0]
    x0 = torch.arange(-1, L, device=fg_lut.device)
    return torch.cat([fg_lut[:, x0.clamp(min=0)], fg_lut[:, (x0 + 1).clamp(max=L - 1)]], -1).contiguous()


class EnvAtlas:
    """All environment maps of a DreamMatMaterial, pre-filtered and packed for the shade kernel."""

    def __init__(self, latlongs, scale=1.0, min_res=16, max_res=128, fg_lut=None, device="cpu", texel=None):
        self.device = torch.device(device)
        # texel storage of the packed cube maps (DREAMMAT_ATLAS overrides): "rgb18e8" (default: 8-byte shared-exponent
        # RGB, 18-bit mantissas -- one 16 B load per bilinear row in the shade kernels, relative texel error <= 2^-18),
        # "fp32" (RGBA fp32, 16 B: the exact reference layout, twice the gathers) or "fp16" (RGBA fp16, 8 B, clamped to
        # +-65504, relative error <= 2^-11)
        self.texel = texel or os.environ.get("DREAMMAT_ATLAS", "rgb18e8")
        if self.texel not in TEXEL_FORMATS:
            raise ValueError(f"atlas texel format {self.texel!r}: expected one of {sorted(TEXEL_FORMATS)}")
        self.n_env = len(latlongs)
        spec_all, diff_all = [], []
        self.mip_res = []
        for img in latlongs:
            img = torch.as_tensor(img, dtype=torch.float32, device=self.device) * scale
            mips = [latlong_to_cube(img, max_res)]
            while mips[-1].shape[1] > min_res:
                c = mips[-1]
                mips.append(0.25 * (c[:, 0::2, 0::2] + c[:, 1::2, 0::2] + c[:, 0::2, 1::2] + c[:, 1::2, 1::2]))
            n = len(mips)
            if n < 2:
                raise ValueError("need at least two mips (max_res >= 2*min_res)")
            diffuse = convolve_diffuse(mips[-1])
            for i in range(n - 1):
                r = (i / (n - 2)) * (MAX_ROUGHNESS - MIN_ROUGHNESS) + MIN_ROUGHNESS if n > 2 else MIN_ROUGHNESS
                mips[i] = prefilter_specular(mips[i], r)
            mips[-1] = prefilter_specular(mips[-1], 1.0)
            self.mip_res = [m.shape[1] for m in mips]
            spec_all.append(mips)
            diff_all.append(diffuse)
        self.specular = spec_all          # unpadded, for inspection / export
        self.diffuse = diff_all
        self.fg_lut = torch.as_tensor(fg_lut, dtype=torch.float32, device=self.device).contiguous()
        self.pack()

    def pack(self):
        offs, off = [], 0
        for r in self.mip_res:
            offs.append(off)
            off += 6 * (r + 2) * (r + 2)
        self.mip_off = offs
        self.spec_env_stride = off
        self.spec_packed = torch.stack([torch.cat([_pad_faces(m).reshape(-1, 4) for m in mips]) for mips in self.specular]).contiguous()
        self.diff_res = self.diffuse[0].shape[1]
        self.diff_packed = torch.stack([_pad_faces(d).reshape(-1, 4) for d in self.diffuse]).contiguous()
        self.diff_env_stride = self.diff_packed.shape[1]
        if self.texel == "fp16":
            self.spec_packed = self.spec_packed.clamp(-65504.0, 65504.0).half().contiguous()
            self.diff_packed = self.diff_packed.clamp(-65504.0, 65504.0).half().contiguous()
        elif self.texel == "rgb18e8":
            self.spec_packed = encode_rgb18e8(self.spec_packed[..., :3])
            self.diff_packed = encode_rgb18e8(self.diff_packed[..., :3])
        self.fg_pairs = fg_pair_table(self.fg_lut)
        s = _lib.EnvAtlasStruct()
        s.spec = self.spec_packed.data_ptr()
        s.diff = self.diff_packed.data_ptr()
        s.fg_lut = self.fg_lut.data_ptr()
        s.spec_env_stride = self.spec_env_stride
        s.diff_env_stride = self.diff_env_stride
        for i in range(8):
            s.mip_off[i] = offs[i] if i < len(offs) else 0
            s.mip_res[i] = self.mip_res[i] if i < len(offs) else 0
        s.n_mips = len(self.mip_res)
        s.diff_res = self.diff_res
        s.lut_res = self.fg_lut.shape[0]
        s.min_rough_mip = MIN_ROUGHNESS
        s.max_rough_mip = MAX_ROUGHNESS
        s.texel_format = TEXEL_FORMATS[self.texel]
        s.fg_pairs = self.fg_pairs.data_ptr()
        self.struct = s

    def to(self, device):
        self.device = torch.device(device)
        self.spec_packed = self.spec_packed.to(device)
        self.diff_packed = self.diff_packed.to(device)
        self.fg_lut = self.fg_lut.to(device)
        self.specular = [[m.to(device) for m in mips] for mips in self.specular]
        self.diffuse = [d.to(device) for d in self.diffuse]
        self.pack()
        return self


def approx_fg_lut(res=256):
    """Stand-in FG LUT (Karis' analytic env-BRDF fit) used ONLY when the reference's
    load/lights/bsdf_256_256.bin is not on disk (tests, synthetic benchmark).  [res,res,2], row =
    roughness, col = n.v."""
    t = (torch.arange(res, dtype=torch.float32) + 0.5) / res
    rough, nov = torch.meshgrid(t, t, indexing="ij")
    r0 = rough * -1.0 + 1.0
    r1 = rough * -0.0275 + 0.0425
    r2 = rough * -0.572 + 1.04
    r3 = rough * 0.022 - 0.04
    a004 = torch.minimum(r0 * r0, torch.exp2(-9.28 * nov)) * r0 + r1
    return torch.stack([-1.04 * a004 + r2, 1.04 * a004 + r3], dim=-1).clamp(0, 1).contiguous()


def load_fg_lut(path="load/lights/bsdf_256_256.bin"):
    """dreammat_material.py:399-404: np.fromfile(...).reshape(1,256,256,2)."""
    if os.path.exists(path):
        return torch.from_numpy(np.fromfile(path, dtype=np.float32).reshape(256, 256, 2).copy())
    return None
