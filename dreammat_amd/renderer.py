"""`raytracing-renderer` plugin (threestudio/models/renderers/raytracing_renderer.py:85-222).

forward() keeps the reference signature and its 12-key output dict.  The body is the MI355X
pipeline: vertex transform -> tiled rasterize -> antialias plan -> compaction/G-buffer (+jitter) ->
feature field (hash grid HIP + MLP GEMM) x2 -> fused split-sum shade -> scatter -> antialias.
Per-view semantics of the reference at B=1 are kept for every view of the batch.
"""
from dataclasses import dataclass

import torch

import dreammat_amd
from . import hipops
from .base import BaseModule


@dreammat_amd.register("raytracing-renderer")
class RaytraceRender(BaseModule):
    @dataclass
    class Config(BaseModule.Config):
        radius: float = 1.0
        context_type: str = "gl"      # accepted for YAML compatibility (dreammat.yaml:99); one HIP path

    cfg: Config

    def configure(self, geometry, material, background) -> None:
        # non-registered holder, like renderers/base.py:22-35
        object.__setattr__(self, "_sub", {"geometry": geometry, "material": material, "background": background})
        self.ctx = None
        self.change_type = "gaussian"
        self.change_eps = 0.05
        self._opp = None
        self.debug_outputs = True

    @property
    def geometry(self):
        return self._sub["geometry"]

    @property
    def material(self):
        return self._sub["material"]

    @property
    def background(self):
        return self._sub["background"]

    def _mesh(self):
        m = self.geometry.isosurface()
        if self._opp is None or self._opp.device != m.v_pos.device:
            self._opp = hipops.build_topology(m.t_pos_idx)
            if getattr(self.material.cfg, "use_raytracing", False) and m.v_pos.is_cuda:
                # raytracing_renderer.py:103-104: RayTracer(mesh) handed to the material (here: the BVH itself, the
                # occlusion rays are traced inside the Monte-Carlo shading kernel)
                self.material.set_raytracer(hipops.MeshBvh(m.v_pos, m.t_pos_idx, m.v_pos.device))
        return m

    def forward(self, env_id, rays_o, rays_d, w2c, mvp_mtx, camera_positions=None, light_positions=None,
                height: int = 512, width: int = 512, jitter_u=None, jitter_n=None, jitter_uv=None, **kwargs):
        mesh = self._mesh()
        dev = mesh.v_pos.device
        B, H, W = mvp_mtx.shape[0], height, width
        if self.ctx is None:
            self.ctx = hipops.RasterContext(dev)
        tri = mesh.t_pos_idx
        pos_clip = hipops.vertex_transform(mesh.v_pos, mvp_mtx)
        rast = self.ctx.rasterize(pos_clip, tri, H, W, check_overflow=kwargs.get("check_overflow", False))
        plan = hipops.antialias_plan(pos_clip, tri, self._opp, rast)
        mask = (rast[..., 3:] > 0).float()
        mask_aa = hipops.antialias(mask, plan)
        depth, normal_cn = hipops.control_maps(rast, tri, mesh.v_nrm, w2c)
        gb_normal_aa = hipops.antialias(normal_cn, plan)

        if jitter_u is None:   # raytracing_renderer.py:164,168: U[0,1) angle, N(0, eps) radius, drawn per pixel
            jitter_u = torch.rand(B, H, W, device=dev)
            jitter_n = torch.randn(B, H, W, device=dev)
        gb = hipops.gbuffer_compact(rast, tri, mesh.v_pos, mesh.v_nrm, rays_d, jitter_u, jitter_n, self.change_eps)
        N = gb.n
        env_of_view = torch.as_tensor(env_id, device=dev).to(torch.int32).reshape(-1)
        if env_of_view.numel() == 1 and B > 1:
            env_of_view = env_of_view.expand(B)
        env_of_view = env_of_view.contiguous()

        if getattr(self.geometry.cfg, "n_input_dims", 3) == 2:
            # uv-space field (raytracing_renderer.py:177-181): the field is queried at the pixel's interpolated texture
            # coordinate and at that coordinate + N(0, 0.005) per component (`jitter_uv` ~ N(0,1): injectable draw, [B*H*W,2] per
            # pixel or [N,2] per compacted row)
            if mesh.v_tex is None:
                raise ValueError("n_input_dims=2 needs a mesh with texture coordinates")
            texc = hipops.gather_rows(hipops.interpolate(mesh.v_tex.contiguous(), rast, tri).view(B * H * W, 2), gb.pix_idx, gb.n_dev, N)
            if jitter_uv is None:
                jitter_uv = torch.randn(N, 2, device=dev)
            elif jitter_uv.shape[0] == B * H * W:       # injected per PIXEL: independent of the G-buffer's row order
                jitter_uv = jitter_uv[gb.pix_idx.long()]
            pts2 = torch.cat([texc, texc + 0.005 * jitter_uv[:N]], dim=0).t()      # [2, 2N]
        else:
            # both field queries in ONE launch: rows [0,N) = surface points, [N,2N) = jittered points
            pts2 = torch.cat([gb.pos, gb.pos_jitter], dim=1)        # [3, 2N] SoA
        feats2 = self.geometry(pts2.t(), output_normal=False)["features"]   # [2N, 5]
        feat, feat_j = feats2[:N], feats2[N:]
        shade, mat_reg = self.material(gb.pos.t(), feat, feat_j, gb.view.t(), gb.nrm.t(), env_of_view,
                                       pix_idx=gb.pix_idx, n_dev=gb.n_dev, hw=H * W, want_debug=self.debug_outputs)

        ones3 = torch.ones(B * H * W, 3, device=dev)
        color = hipops.scatter_rows(shade["color"], gb.pix_idx, gb.n_dev, ones3)
        comp_rgb = hipops.antialias(color.view(B, H, W, 3), plan)
        out = {"comp_rgb": comp_rgb, "opacity": mask_aa, "comp_depth": depth, "comp_normal": gb_normal_aa,
               "loss_mat_reg": mat_reg}
        if self.debug_outputs:
            ones1 = torch.ones(B * H * W, 1, device=dev)

            def dense(x, init):
                return hipops.scatter_rows(x.detach(), gb.pix_idx, gb.n_dev, init).view(B, H, W, -1)

            out.update({
                "albedo": dense(shade["albedo"], ones3), "metalness": dense(shade["metalness"], ones1),
                "roughness": dense(shade["roughness"], ones1),
                "specular_light": dense(shade["specular_lights"], ones3),
                "diffuse_light": dense(shade["diffuse_lights"], ones3),
                "specular_color": dense(shade["specular_colors"], ones3),
                "diffuse_color": dense(shade["diffuse_colors"], ones3)})
        out["_internals"] = {"rast": rast, "plan": plan, "pos_clip": pos_clip, "gbuffer": gb, "features": feat,
                             "features_jitter": feat_j, "color_pre_aa": color}
        return out
