"""`random-camera-datamodule` (threestudio/data/uncond.py:30-64 config, :340-821
FixCameraIterableDataset, :825-946 RandomCameraDataset, :949-1003 datamodule).

The fixed-view camera tables (:584-645), `collate` (:723-821) and the eval orbit are restated; the
ControlNet condition maps come from one of
  * `condition_source: prerender` -- the reference's Blender/Cycles PNG tree under `pre_render_dir`
    (depth/%03d.png 16-bit, normal/%03d.png, light/%03d_m{0.0,1.0}r{0.0,0.5,1.0}_env{1..5}.png),
    decoded like uncond.py:532-582;
  * `condition_source: render` -- rendered on the fly by this repo's kernels (dreammat_amd/condition.py: depth,
    Blender-convention view normal, 6 probe-material split-sum light maps); needs `attach_renderer(mesh, atlas)`;
  * `condition_source: synthetic` -- seeded U[0,1] maps (Blender is not available on this box;
    SURVEY 8d cfg3), generated per (view, env) on the fly instead of holding the reference's
    [128,5,H,W,18] float table (12.9 GB at 512^2) in host memory.
Extension: `views_per_rank` views per process per step (the reference's batch_size=1 per DDP rank is
the special case), all drawn from the per-rank generator (launch.py:102: seed + rank).
"""
import math
import os
from dataclasses import dataclass, field
from typing import Any, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

import dreammat_amd
from .camera import get_mvp_matrix, get_projection_matrix, get_ray_directions, get_rays
from .config import parse_structured


@dataclass
class RandomCameraDataModuleConfig:
    height: Any = 64
    width: Any = 64
    batch_size: Any = 1
    resolution_milestones: Tuple = ()
    eval_height: int = 512
    eval_width: int = 512
    eval_batch_size: int = 1
    n_val_views: int = 1
    n_test_views: int = 120
    elevation_range: Tuple = (-10, 90)
    azimuth_range: Tuple = (-180, 180)
    camera_distance_range: Tuple = (1, 1.5)
    fovy_range: Tuple = (40, 70)
    camera_perturb: float = 0.1
    center_perturb: float = 0.2
    up_perturb: float = 0.02
    light_position_perturb: float = 1.0
    light_distance_range: Tuple = (0.8, 1.5)
    eval_elevation_deg: float = 15.0
    eval_camera_distance: float = 1.5
    eval_fovy_deg: float = 70.0
    light_sample_strategy: str = "dreamfusion"
    batch_uniform_azimuth: bool = True
    progressive_until: int = 0
    use_fix_views: bool = False
    blender_generate: bool = False
    fix_view_num: int = 128
    fix_env_num: int = 5
    # additions
    condition_source: str = "synthetic"
    pre_render_dir: Optional[str] = None
    seed: int = 0
    # resident: keep the per-view camera tensors and the condition maps of all fix_view_num x fix_env_num combinations in HBM
    # (fp32; 0.4 GB + 14.8 GB at 512^2, 128 views, 5 envs -- 5 % of an MI355X) and collate by a device gather.  None = on
    # when the dataset lives on a GPU.  The reference decodes its PNG tree once at start-up too (uncond.py:559-582).
    resident: Optional[bool] = None


def _lookat_c2w(camera_positions, center, up):
    lookat = F.normalize(center - camera_positions, dim=-1)
    right = F.normalize(torch.cross(lookat, up, dim=-1), dim=-1)
    up = F.normalize(torch.cross(right, lookat, dim=-1), dim=-1)
    c2w3x4 = torch.cat([torch.stack([right, up, -lookat], dim=-1), camera_positions[:, :, None]], dim=-1)
    c2w = torch.cat([c2w3x4, torch.zeros_like(c2w3x4[:, :1])], dim=1)
    c2w[:, 3, 3] = 1.0
    return c2w


class FixCameraIterableDataset:
    def __init__(self, cfg, rank=0, device="cpu"):
        self.cfg = cfg
        self.device = torch.device(device)
        self.height = cfg.height if isinstance(cfg.height, int) else cfg.height[0]
        self.width = cfg.width if isinstance(cfg.width, int) else cfg.width[0]
        self.batch_size = cfg.batch_size if isinstance(cfg.batch_size, int) else cfg.batch_size[0]
        self.directions_unit_focal = get_ray_directions(self.height, self.width, 1.0)
        n = cfg.fix_view_num
        g = torch.Generator().manual_seed(cfg.seed)        # the view tables are shared by all ranks
        er, ar, dr, fr = cfg.elevation_range, cfg.azimuth_range, cfg.camera_distance_range, cfg.fovy_range
        # set_fix_elevs (:584-606): half uniform in degrees, half uniform on the sphere
        e1 = torch.rand(n // 2, generator=g) * (er[1] - er[0]) + er[0]
        pct = [(er[0] + 90.0) / 180.0, (er[1] + 90.0) / 180.0]
        e2 = torch.asin(2 * (torch.rand(n // 2, generator=g) * (pct[1] - pct[0]) + pct[0]) - 1.0) / math.pi * 180.0
        self.elevation_degs = torch.cat((e1, e2))
        # set_fix_azims (:608-616): stratified
        self.azimuth_degs = (torch.rand(n, generator=g) + torch.arange(n)) / n * (ar[1] - ar[0]) + ar[0]
        self.fix_camera_distances = torch.rand(n, generator=g) * (dr[1] - dr[0]) + dr[0]
        self.camera_perturbs = torch.rand(n, 3, generator=g) * 2 * cfg.camera_perturb - cfg.camera_perturb
        self.center_perturbs = torch.randn(n, 3, generator=g) * cfg.center_perturb
        self.up_perturbs = torch.randn(n, 3, generator=g) * cfg.up_perturb
        self.fovy_degs = torch.rand(n, generator=g) * (fr[1] - fr[0]) + fr[0]
        self.gen = torch.Generator().manual_seed(cfg.seed + 1000003 * (rank + 1))   # per-rank draws
        self._prerender = None
        self._cond_renderer = None
        self.resident = self._resident_default() if cfg.resident is None else bool(cfg.resident)
        self._cam_table = None          # device copies of camera_for(all views), built on first use
        self._cond_table = None         # [n_views * n_envs, H, W, 22] fp32 on the device (synthetic / prerender)
        self._cond_cache = {}           # condition_source=render: maps rendered so far, keyed by (view, env)
        if cfg.condition_source == "prerender":
            self._prerender = _PreRendered(cfg.pre_render_dir, n, cfg.fix_env_num, self.height, self.width)
        elif cfg.condition_source not in ("synthetic", "render"):
            raise ValueError(f"condition_source={cfg.condition_source!r}: expected synthetic | prerender | render")

    def attach_renderer(self, mesh, atlas):
        """condition_source: render -- the mesh and the material's environment atlas the maps are rendered with."""
        from .condition import ConditionMapRenderer
        self._cond_renderer = ConditionMapRenderer(mesh, atlas, self.device)

    def camera_for(self, view_id):
        elevation_deg = self.elevation_degs[view_id]
        azimuth_deg = self.azimuth_degs[view_id]
        elevation, azimuth = elevation_deg * math.pi / 180, azimuth_deg * math.pi / 180
        dist = self.fix_camera_distances[view_id]
        B = view_id.shape[0]
        cam = torch.stack([dist * torch.cos(elevation) * torch.cos(azimuth),
                           dist * torch.cos(elevation) * torch.sin(azimuth),
                           dist * torch.sin(elevation)], dim=-1)
        center = torch.zeros_like(cam) + self.center_perturbs[view_id]
        up = torch.as_tensor([0, 0, 1], dtype=torch.float32)[None, :].repeat(B, 1) + self.up_perturbs[view_id]
        cam = cam + self.camera_perturbs[view_id]
        fovy = self.fovy_degs[view_id] * math.pi / 180
        c2w = _lookat_c2w(cam, center, up)
        focal = 0.5 * self.height / torch.tan(0.5 * fovy)
        d = self.directions_unit_focal[None].repeat(B, 1, 1, 1)
        d[:, :, :, :2] = d[:, :, :, :2] / focal[:, None, None, None]
        rays_o, rays_d = get_rays(d, c2w, keepdim=True)
        proj = get_projection_matrix(fovy, self.width / self.height, 0.1, 1000.0)
        mvp, w2c = get_mvp_matrix(c2w, proj)
        return {"rays_o": rays_o, "rays_d": rays_d, "mvp_mtx": mvp, "camera_positions": cam, "c2w": c2w, "w2c": w2c,
                "light_positions": None, "elevation": elevation_deg, "azimuth": azimuth_deg, "camera_distances": dist,
                "height": self.height, "width": self.width}

    def condition_map(self, view_id, env_id, cam=None):
        if self._prerender is not None:
            return self._prerender.get(view_id, env_id)
        if self.cfg.condition_source == "render":
            if self._cond_renderer is None:
                raise RuntimeError("condition_source=render: call attach_renderer(mesh, material.atlas) first "
                                   "(RandomCameraDataModule.setup does when it was given the mesh and the atlas)")
            cam = cam or self.camera_for(view_id)
            return self._cond_renderer(cam["mvp_mtx"], cam["c2w"], cam["rays_d"], env_id)
        out = []
        for v, e in zip(view_id.tolist(), env_id.tolist()):
            g = torch.Generator(device=self.device).manual_seed(7919 * v + e + 17)
            out.append(torch.rand(self.height, self.width, 22, generator=g, device=self.device))
        return torch.stack(out)

    # ---- resident tables (HBM) --------------------------------------------------------------------------------------
    def resident_table_bytes(self):
        """HBM the resident tables take: the 22-channel fp32 condition maps of every (view, env) pair + the per-view rays."""
        n, ne = self.cfg.fix_view_num, self.cfg.fix_env_num
        cond = n * ne * self.height * self.width * 22 * 4 if self.cfg.condition_source in ("synthetic", "prerender") else 0
        return cond + n * self.height * self.width * 3 * 4

    def _resident_default(self, free_bytes=None):
        """resident = None: on when the dataset lives on a GPU AND the tables fit in a quarter of the HBM that is free now
        (14.8 GB at 512^2 and 59 GB at 1024^2 for 128 views x 5 envs: fine on 288 GB, not on a small or busy device);
        otherwise the per-step host collate + copy."""
        if self.device.type != "cuda":
            return False
        if free_bytes is None:
            free_bytes = torch.cuda.mem_get_info(self.device)[0]
        return self.resident_table_bytes() * 4 <= free_bytes

    def _build_tables(self):
        n, ne = self.cfg.fix_view_num, self.cfg.fix_env_num
        dev = self.device
        cam = self.camera_for(torch.arange(n))                      # the reference's per-view camera law, all views at once
        self._cam_table = {k: cam[k].to(dev) for k in ("rays_d", "mvp_mtx", "camera_positions", "c2w", "w2c", "elevation",
                                                        "azimuth", "camera_distances")}
        if self.cfg.condition_source in ("synthetic", "prerender"):
            tab = torch.empty(n * ne, self.height, self.width, 22, device=dev)
            for v in range(n):
                vid = torch.full((ne,), v, dtype=torch.long)
                eid = torch.arange(ne)
                tab[v * ne:(v + 1) * ne] = self.condition_map(vid, eid).to(dev)
            self._cond_table = tab

    def _collate_resident(self, view_id, env_id):
        if self._cam_table is None:
            self._build_tables()
        dev = self.device
        vi = view_id.to(dev, non_blocking=True)
        t = self._cam_table
        B = view_id.shape[0]
        cam_pos = t["camera_positions"].index_select(0, vi)
        out = {"rays_o": cam_pos[:, None, None, :].expand(B, self.height, self.width, 3),
               "rays_d": t["rays_d"].index_select(0, vi), "mvp_mtx": t["mvp_mtx"].index_select(0, vi),
               "camera_positions": cam_pos, "c2w": t["c2w"].index_select(0, vi), "w2c": t["w2c"].index_select(0, vi),
               "light_positions": None, "elevation": t["elevation"].index_select(0, vi),
               "azimuth": t["azimuth"].index_select(0, vi), "camera_distances": t["camera_distances"].index_select(0, vi),
               "height": self.height, "width": self.width}
        if self._cond_table is not None:
            row = (view_id * self.cfg.fix_env_num + env_id).to(dev, non_blocking=True)
            cond = self._cond_table.index_select(0, row)
        else:                                                       # condition_source = render: deterministic per (view, env)
            keys = list(zip(view_id.tolist(), env_id.tolist()))
            miss = [i for i, k in enumerate(keys) if k not in self._cond_cache]
            if miss:
                sel = torch.tensor(miss)
                cam = {k: out[k][sel.to(dev)] for k in ("mvp_mtx", "c2w", "rays_d")}
                maps = self.condition_map(view_id[sel], env_id[sel], cam)
                for i, m in zip(miss, maps):
                    self._cond_cache[keys[i]] = m
            cond = torch.stack([self._cond_cache[k] for k in keys])
        out["condition_map"] = cond
        return out

    def collate(self, batch=None):
        B = self.batch_size
        view_id = (torch.rand(B, generator=self.gen) * self.cfg.fix_view_num).floor().long()
        env_id = (torch.rand(B, generator=self.gen) * self.cfg.fix_env_num).floor().long()
        if self.resident:
            out = self._collate_resident(view_id, env_id)
        else:
            out = self.camera_for(view_id)
            out["condition_map"] = self.condition_map(view_id, env_id, out)
        out.update({"view_id": view_id, "env_id": env_id})
        return out

    def __iter__(self):
        while True:
            yield self.collate()


class _PreRendered:
    """uncond.py:532-582 decode of the Blender pre-render tree."""

    def __init__(self, root, n_views, n_envs, H, W):
        from PIL import Image
        self.root, self.H, self.W = root, H, W
        self.Image = Image
        self.n_envs = n_envs

    def _rgb(self, path):
        img = self.Image.open(path).convert("RGB").resize((self.W, self.H), self.Image.BILINEAR)
        return torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0)

    def _depth(self, path):
        d = np.asarray(self.Image.open(path), dtype=np.float32) / 1000.0
        d = torch.from_numpy(d)[None, None]
        d = F.interpolate(d, (self.H, self.W), mode="nearest")[0, 0]
        mask = d > 0
        out = torch.zeros_like(d)
        if mask.any():
            inv = 1.0 / (d[mask] + 1e-6)
            out[mask] = 0.7 * (inv - inv.min()) / (inv.max() - inv.min() + 1e-6) + 0.3
        return out[..., None]

    def get(self, view_id, env_id):
        outs = []
        for v, e in zip(view_id.tolist(), env_id.tolist()):
            depth = self._depth(os.path.join(self.root, "depth", f"{v:03d}.png"))
            normal = self._rgb(os.path.join(self.root, "normal", f"{v:03d}.png"))
            lights = [self._rgb(os.path.join(self.root, "light", f"{v:03d}_m{m}r{r}_env{e + 1}.png"))
                      for m in ("0.0", "1.0") for r in ("0.0", "0.5", "1.0")]
            outs.append(torch.cat([depth, normal] + lights, dim=-1))
        return torch.stack(outs)


class RandomCameraDataset:
    """Validation / test orbit (uncond.py:825-946): fixed elevation & distance, azimuth sweep, env 4."""

    def __init__(self, cfg, split):
        self.cfg = cfg
        self.n_views = cfg.n_val_views if split == "val" else cfg.n_test_views
        n = self.n_views
        az = torch.linspace(0, 360.0, n + 1)[:n] if split == "val" else torch.linspace(0, 360.0, n)
        el = torch.full_like(az, cfg.eval_elevation_deg)
        dist = torch.full_like(el, cfg.eval_camera_distance)
        e, a = el * math.pi / 180, az * math.pi / 180
        cam = torch.stack([dist * torch.cos(e) * torch.cos(a), dist * torch.cos(e) * torch.sin(a), dist * torch.sin(e)], -1)
        up = torch.as_tensor([0, 0, 1], dtype=torch.float32)[None, :].repeat(n, 1)
        fovy = torch.full_like(el, cfg.eval_fovy_deg) * math.pi / 180
        c2w = _lookat_c2w(cam, torch.zeros_like(cam), up)
        H, W = cfg.eval_height, cfg.eval_width
        focal = 0.5 * H / torch.tan(0.5 * fovy)
        d = get_ray_directions(H, W, 1.0)[None].repeat(n, 1, 1, 1)
        d[:, :, :, :2] = d[:, :, :, :2] / focal[:, None, None, None]
        self.rays_o, self.rays_d = get_rays(d, c2w, keepdim=True)
        self.mvp, self.w2c = get_mvp_matrix(c2w, get_projection_matrix(fovy, W / H, 0.1, 1000.0))
        self.c2w, self.cam, self.el, self.az, self.dist, self.H, self.W = c2w, cam, el, az, dist, H, W

    def __len__(self):
        return self.n_views

    def __getitem__(self, i):
        s = slice(i, i + 1)
        return {"index": i, "env_id": torch.tensor([4]), "rays_o": self.rays_o[s], "rays_d": self.rays_d[s],
                "mvp_mtx": self.mvp[s], "w2c": self.w2c[s], "c2w": self.c2w[s], "camera_positions": self.cam[s],
                "light_positions": None, "elevation": self.el[s], "azimuth": self.az[s],
                "camera_distances": self.dist[s], "height": self.H, "width": self.W}


@dreammat_amd.register("random-camera-datamodule")
class RandomCameraDataModule:
    def __init__(self, mesh=None, prerender_dir=None, cfg=None, rank=0, device="cpu", atlas=None):
        self.device = device
        self.cfg = parse_structured(RandomCameraDataModuleConfig, cfg)
        if prerender_dir is not None and self.cfg.pre_render_dir is None:
            self.cfg.pre_render_dir = prerender_dir
        self.mesh = mesh
        self.atlas = atlas          # the material's EnvAtlas: needed by condition_source=render only
        self.rank = rank

    def setup(self, stage=None):
        if stage in (None, "fit"):
            self.train_dataset = FixCameraIterableDataset(self.cfg, self.rank, self.device)
            if self.cfg.condition_source == "render" and self.mesh is not None and self.atlas is not None:
                self.train_dataset.attach_renderer(self.mesh, self.atlas)
        if stage in (None, "fit", "validate"):
            self.val_dataset = RandomCameraDataset(self.cfg, "val")
        if stage in (None, "test", "predict"):
            self.test_dataset = RandomCameraDataset(self.cfg, "test")
