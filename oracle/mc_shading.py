"""Monte-Carlo ray-traced shading, the reference's DEFAULT material branch (TEST INFRASTRUCTURE; fp32 torch,
autograd-friendly).  Groundwork for SURVEY row f-1: the product path (dreammat_amd/material.py) still raises for
`use_raytracing=true`; this file is the checker the HIP implementation will be held to.

Follows threestudio/models/materials/dreammat_material.py:
  :84-103  sample_sphere / az_el_to_points (Fibonacci lattice)      :64-65   saturate_dot
  :452-470 get_envirmentlight_blender (nearest texel of the lat-long image, z-up)
  :490-507 get_lights (occluded direction => 0, else the environment texel; inner light disabled)
  :509-517 fresnel_schlick(_directions)   :519-530 geometry_schlick(_ggx)   :532-541 geometry_ggx_smith_correlated
  :543-553 get_orthogonal_directions      :554-573 sample_diffuse_directions  :575-596 sample_specular_directions
  :599-604 distribution_ggx               :615-677 shade_raytracing
  :726-744 forward, use_raytracing branch (roughness range min/max_roughness_squre, "already squared")
and threestudio/models/renderers/raytracing_renderer.py:318-324 (hit <=> traced depth < 10).
In-tree arithmetic PINNED by tests/golden/mc_shading.npz (the reference's method bodies executed with only the ray
tracer stubbed); the ray tracer itself (ashawkey/raytracing, un-vendored CUDA BVH, requirements.txt:25, unpinned)
is restated as a brute-force double-sided Moeller-Trumbore any-hit test: parity UNPINNED against the real package.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .shading import lin2srgb, material_smoothness_grad


def sample_sphere(num_samples, begin_elevation=0):
    """Fibonacci lattice restricted to elevations above `begin_elevation` (dreammat_material.py:84-98)."""
    ratio = (begin_elevation + 90) / 180
    num_points = int(num_samples // (1 - ratio))
    phi = (np.sqrt(5) - 1.0) / 2.0
    n = np.arange(num_points - num_samples, num_points)
    z = 2.0 * n / num_points - 1.0
    return 2 * np.pi * n * phi % (2 * np.pi), np.arcsin(z)


def direction_samples(num_samples):
    """the `[sn, 2]` (azimuth, elevation) table in [0,1]^2 that configure() builds (:389-398)."""
    az, el = sample_sphere(num_samples, 0)
    az, el = az * 0.5 / np.pi, 1 - 2 * el / np.pi
    return torch.from_numpy(np.stack([az, el], -1).astype(np.float32))


def saturate_dot(v0, v1):
    return torch.clamp(torch.sum(v0 * v1, dim=-1, keepdim=True), min=0.0, max=1.0)


def get_orthogonal_directions(directions):
    x, y, z = torch.split(directions, 1, dim=-1)
    zero = torch.zeros_like(x)
    o0 = torch.cat([y, -x, zero], -1)
    o1 = torch.cat([-z, zero, x], -1)
    m0 = torch.norm(o0, dim=-1, keepdim=True) > torch.norm(o1, dim=-1, keepdim=True)
    return F.normalize(torch.where(m0, o0, o1), dim=-1)


def sample_diffuse_directions(normals, samples, rand_az=None):
    """cosine-weighted hemisphere around `normals` [pn,3]; rand_az [pn] in [0,1) = the per-point azimuth rotation the
    reference draws with torch.rand (is_train and cfg.random_azimuth), None = no rotation."""
    z = normals
    x = get_orthogonal_directions(normals)
    y = torch.cross(z, x, dim=-1)
    az, el = torch.split(samples, 1, dim=1)
    el, az = el.unsqueeze(0), az.unsqueeze(0)
    az = az * torch.pi * 2
    el_sqrt = torch.sqrt(el + 1e-7)
    if rand_az is not None:
        az = (az + rand_az.view(-1, 1, 1) * torch.pi * 2) % (2 * torch.pi)
    coeff_z = torch.sqrt(1 - el + 1e-7)
    coeff_x = el_sqrt * torch.cos(az)
    coeff_y = el_sqrt * torch.sin(az)
    return coeff_x * x.unsqueeze(1) + coeff_y * y.unsqueeze(1) + coeff_z * z.unsqueeze(1)


def sample_specular_directions(reflections, roughness, samples, rand_az=None):
    """GGX importance samples around the mirror direction; `roughness` [pn,1] is alpha ("already squared")."""
    z = reflections
    x = get_orthogonal_directions(reflections)
    y = torch.cross(z, x, dim=-1)
    a = roughness
    az, el = torch.split(samples, 1, dim=1)
    phi = np.pi * 2 * az
    a, el = a.unsqueeze(1), el.unsqueeze(0)
    cos_theta = torch.sqrt((1.0 - el + 1e-6) / (1.0 + (a ** 2 - 1.0) * el + 1e-6) + 1e-6)
    sin_theta = torch.sqrt(1 - cos_theta ** 2 + 1e-6)
    phi = phi.unsqueeze(0)
    if rand_az is not None:
        phi = (phi + rand_az.view(-1, 1, 1) * np.pi * 2) % (2 * np.pi)
    coeff_x = torch.cos(phi) * sin_theta
    coeff_y = torch.sin(phi) * sin_theta
    coeff_z = cos_theta
    return coeff_x * x.unsqueeze(1) + coeff_y * y.unsqueeze(1) + coeff_z * z.unsqueeze(1)


def distribution_ggx(NoH, roughness):
    a2 = roughness ** 2
    denom = NoH ** 2 * (a2 - 1.0) + 1.0
    return a2 / (np.pi * denom ** 2 + 1e-4)


def geometry_schlick_ggx(NoV, roughness):
    k = roughness / 2
    return NoV / (NoV * (1 - k) + k + 1e-5)


def geometry_schlick(NoV, NoL, roughness):
    return geometry_schlick_ggx(NoV, roughness) * geometry_schlick_ggx(NoL, roughness)


def geometry_ggx_smith_correlated(NoV, NoL, roughness):
    def fun(alpha2, cos_theta):
        cos2 = cos_theta ** 2
        return 0.5 * torch.sqrt(1 + alpha2 * (1 - cos2) / (cos2 + 1e-7)) - 0.5
    a2 = roughness ** 2
    return 1.0 / (1.0 + fun(a2, NoV) + fun(a2, NoL))


def fresnel_schlick(F0, HoV):
    return F0 + (1.0 - F0) * torch.clamp(1.0 - HoV, min=0.0, max=1.0) ** 5.0


def fresnel_schlick_directions(F0, view_dirs, directions):
    H = F.normalize(view_dirs + directions, dim=-1)
    HoV = torch.clamp(torch.sum(H * view_dirs, dim=-1, keepdim=True), min=0.0, max=1.0)
    return fresnel_schlick(F0, HoV), H, HoV


def environment_light_latlong(light, directions):
    """nearest texel of the lat-long image `light` [h,w,3] for z-up directions [n,3] (:452-470)."""
    height, width, _ = light.shape
    d = directions / directions.norm(p=2, dim=-1, keepdim=True)
    x, y, z = d.unbind(-1)
    theta = torch.acos(z)
    phi = torch.atan2(y, x) % (2 * np.pi)
    u = -phi / (2 * np.pi) + 0.5
    v = theta / np.pi
    px = (u * width) % width
    py = (v * height) % height
    return light[py.long(), px.long(), :]


def trace_any_hit(v_pos, tri, origins, dirs, t_max=10.0, chunk=4096):
    """hit mask [n] of rays o + t d against the mesh: double-sided Moeller-Trumbore, hit <=> some 0 < t < t_max
    (raytracing_renderer.py:318-324: miss <=> depth >= 10).  Brute force over all triangles, chunked over rays."""
    p0, p1, p2 = (v_pos[tri[:, k].long()].double() for k in range(3))
    e1, e2 = p1 - p0, p2 - p0
    o, d = origins.detach().double(), dirs.detach().double()
    hit = torch.zeros(o.shape[0], dtype=torch.bool)
    for s in range(0, o.shape[0], chunk):
        oc, dc = o[s:s + chunk, None, :], d[s:s + chunk, None, :]
        pv = torch.cross(dc.expand(-1, e2.shape[0], -1), e2[None].expand(dc.shape[0], -1, -1), dim=-1)
        det = (e1[None] * pv).sum(-1)
        ok = det.abs() > 1e-12
        inv = 1.0 / torch.where(ok, det, torch.ones_like(det))
        tv = oc - p0[None]
        u = (tv * pv).sum(-1) * inv
        qv = torch.cross(tv, e1[None].expand_as(tv), dim=-1)
        v = (dc * qv).sum(-1) * inv
        t = (e2[None] * qv).sum(-1) * inv
        h = ok & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > 0) & (t < t_max)
        hit[s:s + chunk] = h.any(dim=1)
    return hit


def get_lights(points, directions, light, trace_fn):
    """[pn,sn,3] radiance arriving along `directions`: 0 where the offset ray hits the mesh, else the env texel."""
    shape = points.shape[:-1]
    eps = 1e-5
    p, d = points.reshape(-1, 3), directions.reshape(-1, 3)
    hit = trace_fn(p + d * eps, d).reshape(*shape)
    lights = torch.zeros((*shape, 3))
    miss = ~hit
    if miss.any():
        lights[miss] = environment_light_latlong(light, directions[miss])
    return lights


def shade_raytracing(pts, normals, view_dirs, light, metallic, roughness, albedo, diffuse_samples, specular_samples,
                     trace_fn, rand_az_diffuse=None, rand_az_specular=None, geometry_type="schlick"):
    """dreammat_material.py:615-677.  light = the env map of these points [h,w,3]; trace_fn(origins, dirs) -> hit [n].
    Returns the reference's 8-key output dict."""
    reflections = torch.sum(view_dirs * normals, -1, keepdim=True) * normals * 2 - view_dirs
    F0 = 0.04 * (1 - metallic) + metallic * albedo
    diffuse_directions = sample_diffuse_directions(normals, diffuse_samples, rand_az_diffuse)
    diffuse_num = diffuse_directions.shape[1]
    specular_directions = sample_specular_directions(reflections, roughness, specular_samples, rand_az_specular)
    specular_num = specular_directions.shape[1]
    sn = diffuse_num + specular_num

    NoL_d = saturate_dot(diffuse_directions, normals.unsqueeze(1))
    diffuse_probability = NoL_d / np.pi * (diffuse_num / sn)
    H_s = F.normalize(view_dirs.unsqueeze(1) + specular_directions, dim=-1)
    NoH_s = saturate_dot(normals.unsqueeze(1), H_s)
    VoH_s = saturate_dot(view_dirs.unsqueeze(1), H_s)
    specular_probability = distribution_ggx(NoH_s, roughness.unsqueeze(1)) * NoH_s / (4 * VoH_s + 1e-5) * (specular_num / sn)

    directions = torch.cat([diffuse_directions, specular_directions], 1)
    probability = torch.cat([diffuse_probability, specular_probability], 1)

    fresnel, H, HoV = fresnel_schlick_directions(F0.unsqueeze(1), view_dirs.unsqueeze(1), directions)
    NoV = saturate_dot(normals, view_dirs).unsqueeze(1)
    NoL = saturate_dot(normals.unsqueeze(1), directions)
    if geometry_type == "schlick":
        geometry = geometry_schlick(NoV, NoL, roughness.unsqueeze(1))
    elif geometry_type == "ggx_smith":
        geometry = geometry_ggx_smith_correlated(NoV, NoL, roughness.unsqueeze(1))
    else:
        raise NotImplementedError(geometry_type)
    NoH = saturate_dot(normals.unsqueeze(1), H)
    distribution = distribution_ggx(NoH, roughness.unsqueeze(1))
    pts_ = pts.unsqueeze(1).repeat(1, sn, 1)
    lights = get_lights(pts_, directions, light, trace_fn)
    specular_weights = distribution * geometry / (4 * NoV * probability + 1e-5)
    specular_lights = lights * specular_weights
    specular_colors = torch.mean(fresnel * specular_lights, 1)

    diffuse_lights = lights[:, :diffuse_num]
    diffuse_colors = torch.mean(albedo.unsqueeze(1) * diffuse_lights, 1)

    colors = lin2srgb(diffuse_colors + specular_colors)
    return {"color": colors, "albedo": lin2srgb(albedo.detach()), "roughness": torch.sqrt(roughness + 1e-7),
            "metalness": metallic,
            "specular_lights": lin2srgb(torch.mean(lights[:, diffuse_num:, :].detach(), dim=1)),
            "diffuse_lights": lin2srgb(torch.mean(lights[:, :diffuse_num, :].detach(), dim=1)),
            "specular_colors": lin2srgb(specular_colors.detach()), "diffuse_colors": lin2srgb(diffuse_colors.detach())}


def material_forward_mc(pts, features, features_jitter, viewdirs, normals, light, diffuse_samples, specular_samples,
                        trace_fn, rand_az_diffuse=None, rand_az_specular=None, min_metallic=0.0, max_metallic=0.9,
                        min_roughness_squre=0.01, max_roughness_squre=0.9, geometry_type="schlick"):
    """DreamMatMaterial.forward, use_raytracing=True branch (:726-744) -> (outputs, mat_reg)."""
    material = torch.sigmoid(features)
    material_jitter = torch.sigmoid(features_jitter)
    mat_reg = material_smoothness_grad(material, material_jitter)
    albedo = material[..., :3].clamp(0.0, 1.0)
    metallic = material[..., 3:4] * (max_metallic - min_metallic) + min_metallic
    roughness = material[..., 4:5] * (max_roughness_squre - min_roughness_squre) + min_roughness_squre
    out = shade_raytracing(pts, normals, viewdirs, light, metallic, roughness, albedo, diffuse_samples, specular_samples,
                           trace_fn, rand_az_diffuse, rand_az_specular, geometry_type)
    return out, mat_reg

