"""Material + split-sum shading (TEST INFRASTRUCTURE; fp32 torch, autograd-friendly).

Follows threestudio/models/materials/dreammat_material.py:
  :110-123 material_smoothness_grad, :679-711 shade_splitsum, :713-763 forward (use_raytracing=False
  branch :746-762), and threestudio/utils/ops.py:83-88 (lin2srgb).  In-tree arithmetic PINNED by
  tests/golden/shading.npz; the texture/envlight lookups it calls are the restatements of
  oracle/envlight.py (unpinned).
"""
import torch

from . import envlight as E


def lin2srgb(x):
    return torch.where(x > 0.0031308,
                       torch.pow(torch.clamp(x, min=0.0031308), 1.0 / 2.4) * 1.055 - 0.055,
                       12.92 * x).clamp(0.0, 1.0)


def material_smoothness_grad(material, material_jitter):
    lambda_kd, lambda_ks = 0.25, 0.1
    kd_grad = torch.abs(material[..., :3] - material_jitter[..., :3])
    ks_grad = torch.abs(material[..., 3:5] - material_jitter[..., 3:5])
    kd_luma_grad = (kd_grad[..., 0] + kd_grad[..., 1] + kd_grad[..., 2]) / 3
    loss = torch.mean(kd_luma_grad * kd_grad[..., -1]) * lambda_kd
    loss = loss + torch.mean(ks_grad[..., :-1] * ks_grad[..., -1:]) * lambda_ks
    return loss


def material_params(features, min_metallic=0.0, max_metallic=0.9, min_roughness=0.1, max_roughness=0.95):
    m = torch.sigmoid(features)
    albedo = m[..., :3].clamp(0.0, 1.0)
    metallic = m[..., 3:4] * (max_metallic - min_metallic) + min_metallic
    roughness = m[..., 4:5] * (max_roughness - min_roughness) + min_roughness
    return m, albedo, metallic, roughness


def shade_splitsum(normals, viewdirs, env, fg_lut, metallic, roughness, albedo):
    """env: oracle.envlight.EnvLight; fg_lut [256,256,2]."""
    v = viewdirs
    n_dot_v = (normals * v).sum(-1, keepdim=True)
    reflective = n_dot_v * normals * 2 - v
    diffuse_albedo = albedo
    fg_uv = torch.cat([n_dot_v, roughness], -1).clamp(0, 1)
    fg = E.texture2d_linear_clamp(fg_lut, fg_uv)
    F0 = (1 - metallic) * 0.04 + metallic * albedo
    specular_albedo = F0 * fg[:, 0:1] + fg[:, 1:2]
    diffuse_light = env(normals)
    specular_light = env(reflective, roughness)
    color = diffuse_albedo * diffuse_light + specular_albedo * specular_light
    color = color.clamp(0.0, 1.0)
    return {"color": color, "albedo": albedo, "roughness": roughness, "metalness": metallic,
            "specular_lights": lin2srgb(specular_light), "diffuse_lights": lin2srgb(diffuse_light),
            "specular_colors": lin2srgb(specular_albedo), "diffuse_colors": lin2srgb(diffuse_albedo)}


def material_forward(features, features_jitter, viewdirs, normals, envs, env_id, fg_lut, cfg=None):
    """envs: list of EnvLight; env_id [N] long per covered pixel (per-view id broadcast)."""
    cfg = cfg or {}
    material, albedo, metallic, roughness = material_params(features, **cfg)
    material_jitter = torch.sigmoid(features_jitter)
    mat_reg = material_smoothness_grad(material, material_jitter)
    keys = ["color", "albedo", "roughness", "metalness", "specular_lights", "diffuse_lights",
            "specular_colors", "diffuse_colors"]
    parts = {}
    N = features.shape[0]
    out = {}
    for e in torch.unique(env_id).tolist():
        m = env_id == e
        o = shade_splitsum(normals[m], viewdirs[m], envs[e], fg_lut, metallic[m], roughness[m], albedo[m])
        parts[e] = (m, o)
    for k in keys:
        C = next(iter(parts.values()))[1][k].shape[-1] if parts else 3
        buf = torch.zeros(N, C)
        for e, (m, o) in parts.items():
            buf = buf.masked_scatter(m[:, None].expand(-1, C), o[k])
        out[k] = buf
    return out, mat_reg
