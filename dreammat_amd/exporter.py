"""`mesh-exporter` plugin (threestudio/models/exporters/mesh_exporter.py:17-175, exporters/base.py:10-59): bakes the
fitted material field into texture maps over the mesh's UV atlas and hands an OBJ(+MTL) description to the saver.

Texture baking = the reference's recipe with this repo's kernels: rasterize the UV triangles in clip space
(`uv*2-1`, z=0, w=1) at texture resolution, interpolate the world-space position per texel with the POSITION
triangles, query `geometry.export(points)` -> `material.export(**)` and pad the charts.
Divergences, both because the dependency is absent here: no xatlas, so the mesh must already carry UVs
(every OBJ under load/shapes/objs does; procedural meshes get `v_tex` from their parametrisation); chart padding is
an iterative nearest-texel dilation by `xatlas_pack_options.padding` texels instead of cv2.inpaint(TELEA) over all
holes (texels outside the dilated charts stay 0).
"""
from dataclasses import dataclass, field
from typing import Any, Dict, List

import torch
import torch.nn.functional as F

import dreammat_amd
from . import hipops
from .base import BaseModule


@dataclass
class ExporterOutput:
    save_name: str
    save_type: str
    params: Dict[str, Any]


def dilate_charts(image, hole_mask, n_iter):
    """fill hole texels that touch a filled texel with the mean of their filled 8-neighbours, `n_iter` rings deep."""
    img = image.permute(2, 0, 1)[None]                                   # [1,C,H,W]
    filled = (~hole_mask)[None, None].to(img.dtype)
    img = img * filled
    kernel = torch.ones(1, 1, 3, 3, dtype=img.dtype, device=img.device)
    for _ in range(int(n_iter)):
        cnt = F.conv2d(filled, kernel, padding=1)
        acc = F.conv2d(img, kernel.expand(img.shape[1], 1, 3, 3), padding=1, groups=img.shape[1])
        grow = (filled == 0) & (cnt > 0)
        img = torch.where(grow, acc / cnt.clamp(min=1), img)
        filled = torch.where(grow, torch.ones_like(filled), filled)
    return img[0].permute(1, 2, 0)


def per_triangle_atlas(n_faces, texture_size, padding, device):
    """-> (v_tex [3 n_faces, 2] in [0,1], t_tex_idx [n_faces, 3] int64): face f occupies one half of cell f // 2 of a
    ceil(sqrt(n_faces / 2))^2 grid; the two triangles of a cell are separated by a diagonal gutter."""
    import math
    cells = max(1, math.ceil(math.sqrt((n_faces + 1) // 2)))
    cell = 1.0 / cells
    g = min(0.25 * cell, max(float(padding), 0.5) / float(texture_size))      # gutter in uv units
    f = torch.arange(n_faces, device=device)
    c = f // 2
    ox = (c % cells).float() * cell
    oy = (c // cells).float() * cell
    lo, hi = g, cell - g
    upper = (f % 2 == 1)
    # lower triangle: (lo,lo) (hi-g,lo) (lo,hi-g); upper triangle: (hi,hi) (lo+g,hi) (hi,lo+g)
    ax = torch.where(upper, torch.full_like(ox, hi), torch.full_like(ox, lo))
    ay = ax.clone()
    bx = torch.where(upper, torch.full_like(ox, lo + g), torch.full_like(ox, hi - g))
    by = ay.clone()
    cx = ax.clone()
    cy = torch.where(upper, torch.full_like(ox, lo + g), torch.full_like(ox, hi - g))
    uv = torch.stack([torch.stack([ox + ax, oy + ay], -1), torch.stack([ox + bx, oy + by], -1),
                      torch.stack([ox + cx, oy + cy], -1)], 1).reshape(-1, 2)
    return uv.float(), torch.arange(3 * n_faces, device=device, dtype=torch.int64).reshape(n_faces, 3)


@dreammat_amd.register("mesh-exporter")
class MeshExporter(BaseModule):
    @dataclass
    class Config(BaseModule.Config):
        save_video: bool = False
        fmt: str = "obj-mtl"                      # 'obj-mtl' | 'obj'
        save_name: str = "model"
        save_normal: bool = False
        save_uv: bool = True
        save_texture: bool = True
        texture_size: int = 2048
        texture_format: str = "jpg"
        xatlas_chart_options: dict = field(default_factory=dict)
        xatlas_pack_options: dict = field(default_factory=dict)
        context_type: str = "cuda"                # accepted for YAML compatibility; one HIP rasterizer

    cfg: Config

    def configure(self, geometry, material, background) -> None:
        object.__setattr__(self, "_sub", {"geometry": geometry, "material": material, "background": background})
        self.ctx = None

    @property
    def geometry(self):
        return self._sub["geometry"]

    @property
    def material(self):
        return self._sub["material"]

    def __call__(self) -> List[ExporterOutput]:
        mesh = self.geometry.isosurface()
        if self.cfg.fmt == "obj-mtl":
            return self.export_obj_with_mtl(mesh)
        if self.cfg.fmt == "obj":
            return self.export_obj(mesh)
        raise ValueError(f"Unsupported mesh export format: {self.cfg.fmt}")

    def _base_params(self, mesh, save_mat):
        return {"mesh": mesh, "save_mat": save_mat, "save_normal": self.cfg.save_normal, "save_uv": self.cfg.save_uv,
                "save_vertex_color": False, "map_Kd": None, "map_Ks": None, "map_Bump": None, "map_Pm": None,
                "map_Pr": None, "map_format": self.cfg.texture_format}

    def _require_uv(self, mesh):
        """mesh_exporter.py:53-60 unwraps a mesh without UVs through `mesh.unwrap_uv` (xatlas).  xatlas is not available here;
        the stand-in is the simplest valid atlas: one right triangle per face in a square grid of cells (two triangles per
        cell, `padding` texels of gutter), every face its own chart -- no stretch optimisation, no chart merging, but a
        bijective atlas the baker and any OBJ viewer accept."""
        if mesh.v_tex is None:
            mesh.v_tex, mesh.t_tex_idx = per_triangle_atlas(mesh.t_pos_idx.shape[0], int(self.cfg.texture_size),
                                                            int(self.cfg.xatlas_pack_options.get("padding", 2)),
                                                            mesh.v_pos.device)

    @torch.no_grad()
    def bake_textures(self, mesh):
        """-> (maps dict with 'albedo' [S,S,3], 'metallic' [S,S,1], 'roughness' [S,S,1], hole_mask [S,S])."""
        S = int(self.cfg.texture_size)
        dev = mesh.v_pos.device
        if self.ctx is None:
            self.ctx = hipops.RasterContext(dev)
        uv_clip = mesh.v_tex.float() * 2.0 - 1.0
        uv_clip4 = torch.cat([uv_clip, torch.zeros_like(uv_clip[..., 0:1]), torch.ones_like(uv_clip[..., 0:1])], dim=-1)
        rast = self.ctx.rasterize(uv_clip4[None].contiguous(), mesh.t_tex_idx.to(torch.int32).contiguous(), S, S)
        hole_mask = ~(rast[0, :, :, 3] > 0)
        gb_pos = hipops.interpolate(mesh.v_pos.float().contiguous(), rast, mesh.t_pos_idx.to(torch.int32).contiguous())[0]
        pts = gb_pos.reshape(-1, 3)
        geo_out = self.geometry.export(points=pts)
        mat_out = self.material.export(points=pts, **geo_out)
        pad = int(self.cfg.xatlas_pack_options.get("padding", 2))
        maps = {k: dilate_charts(v.reshape(S, S, -1).float(), hole_mask, pad) for k, v in mat_out.items()}
        return maps, hole_mask

    def export_obj_with_mtl(self, mesh) -> List[ExporterOutput]:
        params = self._base_params(mesh, True)
        if self.cfg.save_uv:
            self._require_uv(mesh)
        if self.cfg.save_texture:
            assert self.cfg.save_uv, "save_uv must be True when save_texture is True"
            maps, _ = self.bake_textures(mesh)
            if "albedo" in maps:
                params["map_Kd"] = maps["albedo"]
            else:
                print("[dreammat_amd] save_texture is True but no albedo texture found, using default white texture")
            params["map_Pm"] = maps.get("metallic")
            params["map_Pr"] = maps.get("roughness")
            params["map_Bump"] = maps.get("bump")
        return [ExporterOutput(save_name=f"{self.cfg.save_name}.obj", save_type="obj", params=params)]

    def export_obj(self, mesh) -> List[ExporterOutput]:
        params = self._base_params(mesh, False)
        if self.cfg.save_uv:
            self._require_uv(mesh)
        if self.cfg.save_texture:
            with torch.no_grad():
                geo_out = self.geometry.export(points=mesh.v_pos.float())
                mat_out = self.material.export(points=mesh.v_pos.float(), **geo_out)
            if "albedo" in mat_out:
                mesh.v_rgb = mat_out["albedo"]
                params["save_vertex_color"] = True
            else:
                print("[dreammat_amd] save_texture is True but no albedo texture found, not saving vertex color")
        return [ExporterOutput(save_name=f"{self.cfg.save_name}.obj", save_type="obj", params=params)]
