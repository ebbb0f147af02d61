"""Camera math of the DreamMat data module (TEST INFRASTRUCTURE; fp32 torch).

Follows threestudio/utils/ops.py:179-216 (get_ray_directions), :219-263 (get_rays),
:266-278 (get_projection_matrix), :281-292 (get_mvp_matrix) and the per-view part of
threestudio/data/uncond.py:723-821 (FixCameraIterableDataset.collate).  PINNED by
tests/golden/camera.npz (generated from the reference's own function bodies).
"""
import math

import torch
import torch.nn.functional as F


def get_ray_directions(H, W, focal, use_pixel_centers=True):
    pc = 0.5 if use_pixel_centers else 0.0
    fx = fy = float(focal)
    cx, cy = W / 2, H / 2
    i, j = torch.meshgrid(torch.arange(W, dtype=torch.float32) + pc,
                          torch.arange(H, dtype=torch.float32) + pc, indexing="xy")
    return torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)


def get_rays(directions, c2w):
    """directions [B,H,W,3], c2w [B,4,4] -> rays_o, rays_d [B,H,W,3] (keepdim=True branch)."""
    rays_d = (directions[:, :, :, None, :] * c2w[:, None, None, :3, :3]).sum(-1)
    rays_o = c2w[:, None, None, :3, 3].expand(rays_d.shape)
    return rays_o, F.normalize(rays_d, dim=-1)


def get_projection_matrix(fovy, aspect_wh, near, far):
    B = fovy.shape[0]
    p = torch.zeros(B, 4, 4, dtype=torch.float32)
    p[:, 0, 0] = 1.0 / (torch.tan(fovy / 2.0) * aspect_wh)
    p[:, 1, 1] = -1.0 / torch.tan(fovy / 2.0)
    p[:, 2, 2] = -(far + near) / (far - near)
    p[:, 2, 3] = -2.0 * far * near / (far - near)
    p[:, 3, 2] = -1.0
    return p


def get_mvp_matrix(c2w, proj):
    w2c = torch.zeros(c2w.shape[0], 4, 4).to(c2w)
    w2c[:, :3, :3] = c2w[:, :3, :3].permute(0, 2, 1)
    w2c[:, :3, 3:] = -c2w[:, :3, :3].permute(0, 2, 1) @ c2w[:, :3, 3:]
    w2c[:, 3, 3] = 1.0
    return proj @ w2c, w2c


def camera_batch(elevation_deg, azimuth_deg, camera_distances, fovy_deg, H, W):
    """uncond.py:726-794 with zero perturbations: spherical -> look-at c2w -> rays, proj, mvp."""
    elevation = elevation_deg * math.pi / 180
    azimuth = azimuth_deg * math.pi / 180
    B = elevation.shape[0]
    cam = torch.stack([camera_distances * torch.cos(elevation) * torch.cos(azimuth),
                       camera_distances * torch.cos(elevation) * torch.sin(azimuth),
                       camera_distances * torch.sin(elevation)], dim=-1)
    center = torch.zeros_like(cam)
    up = torch.as_tensor([0, 0, 1], dtype=torch.float32)[None, :].repeat(B, 1)
    fovy = fovy_deg * math.pi / 180
    lookat = F.normalize(center - cam, dim=-1)
    right = F.normalize(torch.cross(lookat, up, dim=-1), dim=-1)
    up = F.normalize(torch.cross(right, lookat, dim=-1), dim=-1)
    c2w3x4 = torch.cat([torch.stack([right, up, -lookat], dim=-1), cam[:, :, None]], dim=-1)
    c2w = torch.cat([c2w3x4, torch.zeros_like(c2w3x4[:, :1])], dim=1)
    c2w[:, 3, 3] = 1.0
    focal = 0.5 * H / torch.tan(0.5 * fovy)
    d = get_ray_directions(H, W, 1.0)[None].repeat(B, 1, 1, 1)
    d[:, :, :, :2] = d[:, :, :, :2] / focal[:, None, None, None]
    rays_o, rays_d = get_rays(d, c2w)
    proj = get_projection_matrix(fovy, W / H, 0.1, 1000.0)
    mvp, w2c = get_mvp_matrix(c2w, proj)
    return {"rays_o": rays_o, "rays_d": rays_d, "mvp_mtx": mvp, "w2c": w2c, "c2w": c2w,
            "camera_positions": cam, "elevation": elevation_deg, "azimuth": azimuth_deg,
            "camera_distances": camera_distances, "fovy": fovy, "height": H, "width": W}
