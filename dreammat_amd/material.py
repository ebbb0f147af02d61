"""`dreammat-material` plugin (threestudio/models/materials/dreammat_material.py:346-797), split-sum
branch.  forward() keeps the reference signature
    forward(pts, features, features_jitter, viewdirs, normals, env_id) -> (outputs dict, mat_reg)
and runs two HIP kernels: the fused activation + FG-LUT + env-lookup shade kernel and the smoothness
regulariser.  `use_raytracing: true` (the reference default, Monte-Carlo shading with BVH visibility, SURVEY
row f-1) runs the fused Monte-Carlo kernels (csrc/mc_shade.hip) once the renderer has handed over the mesh BVH
through `set_raytracer`; parity-tested against the reference on the CPU (tests/hostemu) and on the GPU
(tests/test_mc_gpu.py) but not yet optimised (0.27 G rays/s), so the bench / north-star path keeps
`use_raytracing: false`.
"""
import os
from dataclasses import dataclass
from typing import Any, Optional

import torch

import dreammat_amd
from . import _lib, hipops
from .base import BaseModule
from .envlight import EnvAtlas, approx_fg_lut, load_fg_lut, read_hdr


@dreammat_amd.register("dreammat-material")
class DreamMatMaterial(BaseModule):
    @dataclass
    class Config(BaseModule.Config):
        material_activation: str = "sigmoid"
        environment_texture: str = "load/lights/mud_road_puresky_1k.hdr"
        environment_scale: float = 1.0
        min_metallic: float = 0.0
        max_metallic: float = 0.9
        min_roughness_squre: float = 0.01
        max_roughness_squre: float = 0.9
        min_roughness: float = 0.1
        max_roughness: float = 0.95
        use_bump: bool = True
        diffuse_sample_num: int = 512
        specular_sample_num: int = 256
        geometry_type: str = "schlick"
        random_azimuth: bool = True
        use_raytracing: bool = True
        # additions (not in the reference): envlight cube resolutions and number of env maps
        env_max_res: int = 128
        env_min_res: int = 16
        n_envs: int = 5
        fg_lut_path: str = "load/lights/bsdf_256_256.bin"

    cfg: Config
    requires_normal = True

    def configure(self, latlongs=None) -> None:
        if self.cfg.material_activation != "sigmoid":
            raise NotImplementedError("the fused shade kernel implements the sigmoid activation of dreammat.yaml:83")
        if latlongs is None:
            latlongs = self._load_envmaps()
        fg = load_fg_lut(self.cfg.fg_lut_path)
        self.real_fg_lut = fg is not None
        if fg is None:
            print(f"[dreammat_amd] {self.cfg.fg_lut_path} not found: using the analytic stand-in FG LUT")
            fg = approx_fg_lut()
        dev = self.device if torch.cuda.is_available() else torch.device("cpu")
        self.atlas = EnvAtlas(latlongs, scale=self.cfg.environment_scale, min_res=self.cfg.env_min_res,
                              max_res=self.cfg.env_max_res, fg_lut=fg, device=dev)
        self.register_buffer("FG_LUT", fg.reshape(1, *fg.shape), persistent=False)
        self.mat = _lib.MatCfgStruct(self.cfg.min_metallic, self.cfg.max_metallic, self.cfg.min_roughness,
                                     self.cfg.max_roughness)
        # Monte-Carlo branch: "roughness" is alpha, already squared (dreammat_material.py:740-743)
        self.mat_mc = _lib.MatCfgStruct(self.cfg.min_metallic, self.cfg.max_metallic, self.cfg.min_roughness_squre,
                                        self.cfg.max_roughness_squre)
        # the reference keeps the raw lat-long images next to the envlight cubes (self.light, :384-386)
        self._latlongs = [torch.as_tensor(x, dtype=torch.float32) for x in latlongs]
        self.mc_scene = None

    def _load_envmaps(self):
        """dreammat_material.py:378-386: <environment_texture>/map{i}/map{i}.hdr for i=1..5; a single
        .hdr file is also accepted (replicated), like the Config default suggests."""
        root = self.cfg.environment_texture
        out = []
        if os.path.isdir(root):
            for i in range(1, self.cfg.n_envs + 1):
                out.append(torch.from_numpy(read_hdr(os.path.join(root, f"map{i}", f"map{i}.hdr"))))
        else:
            img = torch.from_numpy(read_hdr(root))
            out = [img for _ in range(self.cfg.n_envs)]
        return out

    def set_raytracer(self, ray_trace_fun):
        """reference: a callable (o, d) -> (inters, normals, depth, hit_mask) (raytracing_renderer.py:104).  Here the
        occlusion rays are fused into the shading kernel, so what is handed over is the mesh BVH (hipops.MeshBvh)."""
        self.ray_trace_fun = ray_trace_fun
        if isinstance(ray_trace_fun, hipops.MeshBvh):
            self.mc_scene = hipops.McScene(ray_trace_fun, self._latlongs, self.cfg.diffuse_sample_num,
                                           self.cfg.specular_sample_num, self.cfg.geometry_type)

    def forward(self, pts, features, features_jitter, viewdirs, normals, env_id, pix_idx=None, n_dev=None,
                hw=None, want_debug=True, **kwargs):
        N = features.shape[0]
        dev = features.device
        if n_dev is None:
            n_dev = torch.full((1,), N, dtype=torch.int32, device=dev)
        env_id = torch.as_tensor(env_id, device=dev).to(torch.int32).reshape(-1)
        if pix_idx is None:
            # reference call style: one env for all rows (B=1 semantics)
            pix_idx = torch.zeros(N, dtype=torch.int32, device=dev)
            hw = 1 << 30
            env_id = env_id[:1].contiguous()
        if self.cfg.use_raytracing:
            if self.mc_scene is None:
                raise _lib.DmError("use_raytracing=true needs the mesh BVH: call set_raytracer(hipops.MeshBvh(...)) first "
                                   "(RaytraceRender.configure does, raytracing_renderer.py:103-104)")
            rand_d, rand_s = kwargs.get("rand_diffuse"), kwargs.get("rand_specular")
            if self.cfg.random_azimuth and self.training:          # is_train=True in the reference's forward (:744)
                rand_d = torch.rand(N, device=dev) if rand_d is None else rand_d
                rand_s = torch.rand(N, device=dev) if rand_s is None else rand_s
            outs = hipops.mc_shade(features, pts, normals, viewdirs, pix_idx, n_dev, env_id.contiguous(), self.mc_scene,
                                   self.mat_mc, int(hw), rand_d, rand_s, want_debug)
        else:
            outs = hipops.shade(features, normals, viewdirs, pix_idx, n_dev, env_id.contiguous(), self.atlas, self.mat,
                                int(hw), want_debug)
        mat_reg = hipops.material_smoothness(features, features_jitter, n_dev)
        out = {"color": outs[0]}
        if want_debug:
            out.update({"albedo": outs[1], "specular_lights": outs[2], "diffuse_lights": outs[3],
                        "specular_colors": outs[4], "diffuse_colors": outs[5], "metalness": outs[6],
                        "roughness": outs[7]})
        return out, mat_reg

    def export(self, features, **kwargs):
        """dreammat_material.py:765-797 (note: uses the *squared* roughness range there)."""
        material = torch.sigmoid(features)
        albedo = material[..., :3]
        metallic = material[..., 3:4] * (self.cfg.max_metallic - self.cfg.min_metallic) + self.cfg.min_metallic
        roughness = material[..., 4:5] * (self.cfg.max_roughness_squre - self.cfg.min_roughness_squre) \
            + self.cfg.min_roughness_squre
        return {"albedo": albedo, "metallic": metallic, "roughness": torch.sqrt(roughness + 1e-7)}
