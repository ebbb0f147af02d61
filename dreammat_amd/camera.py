"""Camera math of the data module: threestudio/utils/ops.py:179-216 (get_ray_directions), :219-263
(get_rays), :266-278 (get_projection_matrix, y-flipped for the row-0-is-bottom raster convention),
:281-292 (get_mvp_matrix -> (mvp, w2c))."""
import torch
import torch.nn.functional as F


def get_ray_directions(H, W, focal, principal=None, use_pixel_centers=True):
    pixel_center = 0.5 if use_pixel_centers else 0
    if isinstance(focal, (float, int)):
        fx, fy = float(focal), float(focal)
        cx, cy = W / 2, H / 2
    else:
        fx, fy = focal
        cx, cy = principal
    i, j = torch.meshgrid(torch.arange(W, dtype=torch.float32) + pixel_center,
                          torch.arange(H, dtype=torch.float32) + pixel_center, indexing="xy")
    return torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)


def get_rays(directions, c2w, keepdim=False, normalize=True):
    if directions.ndim == 3:
        directions = directions[None].expand(c2w.shape[0], -1, -1, -1)
    rays_d = (directions[:, :, :, None, :] * c2w[:, None, None, :3, :3]).sum(-1)
    rays_o = c2w[:, None, None, :3, 3].expand(rays_d.shape)
    if normalize:
        rays_d = F.normalize(rays_d, dim=-1)
    if not keepdim:
        rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
    return rays_o, rays_d


def get_projection_matrix(fovy, aspect_wh, near, far):
    proj = torch.zeros(fovy.shape[0], 4, 4, dtype=torch.float32)
    proj[:, 0, 0] = 1.0 / (torch.tan(fovy / 2.0) * aspect_wh)
    proj[:, 1, 1] = -1.0 / torch.tan(fovy / 2.0)
    proj[:, 2, 2] = -(far + near) / (far - near)
    proj[:, 2, 3] = -2.0 * far * near / (far - near)
    proj[:, 3, 2] = -1.0
    return proj


def get_mvp_matrix(c2w, proj_mtx):
    w2c = torch.zeros(c2w.shape[0], 4, 4).to(c2w)
    w2c[:, :3, :3] = c2w[:, :3, :3].permute(0, 2, 1)
    w2c[:, :3, 3:] = -c2w[:, :3, :3].permute(0, 2, 1) @ c2w[:, :3, 3:]
    w2c[:, 3, 3] = 1.0
    return proj_mtx @ w2c, w2c
