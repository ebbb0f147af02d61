"""ControlNet condition maps rendered by this repo's own kernels: `condition_source: render`.

The reference pre-renders them in ~15 minutes with Blender / Cycles (threestudio/data/uncond.py:457-526 ->
data/blender_script_fixview.py:308-466) and reads the PNG tree back (uncond.py:528-582).  Here the same 22 channels
    [0]      depth   : per-view inverse-depth, min-max normalised to [0.3, 1] over the object, 0 on the background
                       (loaddepth, uncond.py:540-557; Blender's Z pass in scene units, background masked to 0 :376-398)
    [1:4]    normal  : (0.5 n.right + 0.5, -0.5 n.up + 0.5, -0.5 n.back + 0.5) in Blender camera axes, background
                       (0.5, 0.5, 1) (compositor group create_group_view_normal, blender_script_fixview.py:231-299)
    [4:22]   light   : the object with 6 probe materials -- white base colour, metallic {0, 1} x roughness {0, 0.5, 1}
                       (mat_list :361, create_material :215-229) -- under the step's environment, black background
are produced on the fly for the views of a step: rasterize + G-buffer compaction + six calls of the split-sum shade
kernel with the probe material pinned through a collapsed dm_mat_cfg range (min = max).  Approximations, all because the
original is a path tracer that cannot run here: split-sum instead of Cycles (no self-occlusion / inter-reflection), the
sRGB transfer curve instead of Blender's default view transform, and z-depth for Cycles' Z pass.  Parity with real
Blender output is therefore UNPINNED; what is pinned (tests) is the decode-side conventions above.
"""
import torch

from . import _lib, hipops

PROBE_MATERIALS = [(0.0, 0.0), (0.0, 0.5), (0.0, 1.0), (1.0, 0.0), (1.0, 0.5), (1.0, 1.0)]     # (metallic, roughness)


def lin2srgb(x):
    return torch.where(x > 0.0031308, torch.pow(torch.clamp(x, min=0.0031308), 1.0 / 2.4) * 1.055 - 0.055,
                       12.92 * x).clamp(0.0, 1.0)


class ConditionMapRenderer:
    def __init__(self, mesh, atlas, device):
        self.device = torch.device(device)
        self.v_pos = mesh.v_pos.float().to(self.device).contiguous()
        self.v_nrm = mesh.v_nrm.float().to(self.device).contiguous()
        self.tri = mesh.t_pos_idx.to(torch.int32).to(self.device).contiguous()
        self.atlas = atlas
        self.ctx = hipops.RasterContext(self.device)
        self.probes = [_lib.MatCfgStruct(m, m, r, r) for m, r in PROBE_MATERIALS]

    @torch.no_grad()
    def __call__(self, mvp_mtx, c2w, rays_d, env_id):
        """mvp_mtx [B,4,4], c2w [B,4,4], rays_d [B,H,W,3], env_id [B] -> condition_map [B,H,W,22]."""
        dev = self.device
        mvp, c2w, rays_d = mvp_mtx.to(dev).float(), c2w.to(dev).float(), rays_d.to(dev).float()
        B, H, W, _ = rays_d.shape
        P = B * H * W
        pos_clip = hipops.vertex_transform(self.v_pos, mvp)
        rast = self.ctx.rasterize(pos_clip, self.tri, H, W, check_overflow=True)
        gb = hipops.gbuffer_compact(rast, self.tri, self.v_pos, self.v_nrm, rays_d)
        out = torch.zeros(P, 22, device=dev)
        out[:, 1:4] = torch.tensor([0.5, 0.5, 1.0], device=dev)
        if gb.n == 0:
            return out.view(B, H, W, 22)
        pix = gb.pix_idx.long()
        view_of = pix // (H * W)                                         # [N] view index of each covered pixel
        pos, nrm = gb.pos.t(), torch.nn.functional.normalize(gb.nrm.t(), dim=-1)
        right, up, back, cam = c2w[:, :3, 0], c2w[:, :3, 1], c2w[:, :3, 2], c2w[:, :3, 3]
        # depth: distance along the viewing axis, inverse, per-view min-max to [0.3, 1]
        z = ((cam[view_of] - pos) * back[view_of]).sum(-1)                # > 0 in front of the camera
        inv = 1.0 / (z + 1e-6)
        lo = torch.full((B,), float("inf"), device=dev).scatter_reduce(0, view_of, inv, "amin")
        hi = torch.full((B,), float("-inf"), device=dev).scatter_reduce(0, view_of, inv, "amax")
        out[pix, 0] = 0.7 * (inv - lo[view_of]) / (hi[view_of] - lo[view_of] + 1e-6) + 0.3
        # view normal in Blender camera axes (x right, y up, z towards the viewer), y and z channels inverted
        out[pix, 1] = 0.5 * (nrm * right[view_of]).sum(-1) + 0.5
        out[pix, 2] = -0.5 * (nrm * up[view_of]).sum(-1) + 0.5
        out[pix, 3] = -0.5 * (nrm * back[view_of]).sum(-1) + 0.5
        # light maps: white probe materials under the environment of each view
        env_of_view = torch.as_tensor(env_id, device=dev).to(torch.int32).reshape(-1).contiguous()
        feat = torch.zeros(gb.n, 5, device=dev)
        feat[:, :3] = 30.0                                                # sigmoid -> albedo 1 (white base colour)
        for k, probe in enumerate(self.probes):
            color = hipops.shade(feat, gb.nrm.t(), gb.view.t(), gb.pix_idx, gb.n_dev, env_of_view, self.atlas, probe, H * W,
                                 False)[0]
            out[pix, 4 + 3 * k:7 + 3 * k] = lin2srgb(color)
        return out.view(B, H, W, 22)
