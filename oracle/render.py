"""RaytraceRender.forward restated (TEST INFRASTRUCTURE; fp32 torch + the C raster oracle).

Follows threestudio/models/renderers/raytracing_renderer.py:109-222 step by step, plus
:306-316 get_orthogonal_directions, :326-331 compute_controlnet_normals, :69-83 xfm_vectors.
Generalisation to B>1 keeps the reference's B=1 semantics PER VIEW: per-view env map,
per-view w2c, per-view depth min/max.  All random draws are inputs: `jitter_u` in [0,1) and
`jitter_n` ~ N(0,1), dense [B,H,W] (the reference draws them per covered pixel, :164,168).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import field, raster, shading


def get_orthogonal_directions(directions):
    x, y, z = torch.split(directions, 1, dim=-1)
    otho0 = torch.cat([y, -x, torch.zeros_like(x)], -1)
    otho1 = torch.cat([-z, torch.zeros_like(x), x], -1)
    mask0 = torch.norm(otho0, dim=-1) > torch.norm(otho1, dim=-1)
    otho = torch.where(mask0[:, None], otho0, otho1)
    return F.normalize(otho, dim=-1)


def controlnet_normals(normals_world, w2c_rows):
    """normals [N,3], w2c_rows [N,4,4] (the view's w2c per pixel) -> encoded [N,3]."""
    nv = torch.einsum("nij,nj->ni", w2c_rows[:, :3, :3], normals_world)
    nv = F.normalize(nv, dim=-1)
    enc = 0.5 * (nv + 1)
    enc = torch.cat([1.0 - enc[:, 0:1], enc[:, 1:]], dim=-1)
    return enc


def normalize_depth(rast_zw, mask):
    """raytracing_renderer.py:129-134, per view."""
    depth = rast_zw.clone()
    B = depth.shape[0]
    for b in range(B):
        m = mask[b]
        if m.any():
            d = 1.0 / (depth[b][m] + 1e-6)
            dmax, dmin = d.max(), d.min()
            depth[b][m] = (1 - 0.3) * (d - dmin) / (dmax - dmin + 1e-6) + 0.3
    return depth


def render(mesh, batch, field_params, envs, fg_lut, jitter_u, jitter_n, change_eps=0.05, mat_cfg=None):
    """mesh: dict(v_pos [Nv,3], v_nrm [Nv,3], t_pos_idx [Nf,3], opp [Nf,3]);
    field_params: dict(table, w1, w2, levels, radius) -- torch tensors (may require grad);
    returns the renderer's 12-key dict (+ internals for tests)."""
    mvp, w2c, rays_d = batch["mvp_mtx"], batch["w2c"], batch["rays_d"]
    env_id = torch.as_tensor(batch["env_id"]).long()
    H, W = batch["height"], batch["width"]
    B = mvp.shape[0]
    v_pos = torch.as_tensor(mesh["v_pos"], dtype=torch.float32)
    tri = np.asarray(mesh["t_pos_idx"], dtype=np.int32)
    pos_clip = raster.vertex_transform(v_pos, mvp)
    rast_np = raster.rasterize(pos_clip, tri, H, W)
    rast = torch.from_numpy(rast_np)
    plan = raster.antialias_plan(pos_clip, tri, mesh["opp"], rast_np)
    mask = rast[..., 3] > 0                     # [B,H,W]
    mask_aa = torch.from_numpy(raster.antialias_apply(mask[..., None].float(), plan))
    depth = normalize_depth(rast[..., 2:3], mask[..., None])

    gb_normal = torch.from_numpy(raster.interpolate(mesh["v_nrm"], rast_np, tri))
    gb_normal = F.normalize(gb_normal, dim=-1)
    gb_pos = torch.from_numpy(raster.interpolate(mesh["v_pos"], rast_np, tri))
    sel = mask.reshape(-1)
    view_of = torch.arange(B)[:, None, None].expand(B, H, W).reshape(-1)[sel]
    n_sel = gb_normal.reshape(-1, 3)[sel]
    nc = controlnet_normals(n_sel, w2c[view_of])
    gb_normal_aa = torch.ones(B * H * W, 3)
    gb_normal_aa[sel] = nc
    gb_normal_aa = gb_normal_aa.reshape(B, H, W, 3)
    bg = torch.tensor([0.5, 0.5, 1.0]).reshape(1, 1, 1, 3).expand(B, H, W, 3)
    gb_normal_aa = torch.lerp(bg, gb_normal_aa, mask[..., None].float())
    gb_normal_aa = torch.from_numpy(raster.antialias_apply(gb_normal_aa, plan))

    viewdirs = -rays_d.reshape(-1, 3)[sel]
    positions = gb_pos.reshape(-1, 3)[sel]
    x = get_orthogonal_directions(n_sel)
    y = torch.cross(n_sel, x, dim=-1)
    ang = (jitter_u.reshape(-1)[sel] * np.pi * 2)[:, None]
    eps = (jitter_n.reshape(-1)[sel] * change_eps)[:, None]
    change = (torch.cos(ang) * x + torch.sin(ang) * y) * eps
    positions_jitter = positions + change
    fp = field_params
    if "v_tex" in mesh and fp.get("n_input_dims", 3) == 2:
        # uv-space field (raytracing_renderer.py:177-181): the interpolated texture coordinate and that + N(0, 0.005);
        # `jitter_uv` [B,H,W,2] ~ N(0,1) is the injected draw (the reference draws per covered pixel)
        texc = torch.from_numpy(raster.interpolate(mesh["v_tex"], rast_np, tri)).reshape(-1, 2)[sel]
        positions_q, positions_jitter_q = texc, texc + 0.005 * batch["jitter_uv"]
    else:
        positions_q, positions_jitter_q = positions, positions_jitter
    feat, hid = field.field_forward(positions_q, fp["table"], fp["w1"], fp["w2"], fp["levels"], fp.get("radius", 1.0), True)
    feat_j, hid_j = field.field_forward(positions_jitter_q, fp["table"], fp["w1"], fp["w2"], fp["levels"], fp.get("radius", 1.0), True)
    shade, mat_reg = shading.material_forward(feat, feat_j, viewdirs, n_sel, envs, env_id[view_of], fg_lut, mat_cfg)

    def scatter(vals, C):
        buf = torch.ones(B * H * W, C)
        buf = buf.masked_scatter(sel[:, None].expand(-1, C), vals)
        return buf.reshape(B, H, W, C)

    color = scatter(shade["color"], 3)
    comp_rgb = raster.Antialias.apply(color, plan)
    out = {
        "comp_rgb": comp_rgb, "opacity": mask_aa, "comp_depth": depth, "comp_normal": gb_normal_aa,
        "albedo": scatter(shade["albedo"].detach(), 3), "metalness": scatter(shade["metalness"].detach(), 1),
        "roughness": scatter(shade["roughness"].detach(), 1),
        "specular_light": scatter(shade["specular_lights"].detach(), 3),
        "diffuse_light": scatter(shade["diffuse_lights"].detach(), 3),
        "specular_color": scatter(shade["specular_colors"], 3),
        "diffuse_color": scatter(shade["diffuse_colors"], 3),
        "loss_mat_reg": mat_reg,
        # internals exposed for parity tests
        "_rast": rast, "_plan": torch.from_numpy(plan), "_pos_clip": pos_clip, "_gb_normal": gb_normal,
        "_gb_pos": gb_pos, "_features": feat, "_features_jitter": feat_j, "_color_pre_aa": color,
        "_positions_jitter": positions_jitter, "_hidden": hid, "_hidden_jitter": hid_j,
    }
    return out
