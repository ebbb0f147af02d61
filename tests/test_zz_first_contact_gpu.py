"""First GPU contact of code written after round 1 ran out of GPU budget: SURVEY row f-3 (`mesh-exporter`: texture
baking through the real HIP rasterize / interpolate / hash-grid kernels) and the opt-in wave-parallel Monte-Carlo kernel.
Both are CPU-tested (tests/test_hostlogic_cpu.py, tests/test_golden_cpu.py); this file had no GPU time in round 1, hence
xfail(strict=False): XPASS on success, no red mark if the first contact finds something.  Sorts last on purpose."""
import os

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.xfail(strict=False, reason="first GPU contact pending (written after the round-1 GPU budget was spent)")]


def test_exporter_bakes_the_fitted_field_on_the_gpu(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import dreammat_amd
    from dreammat_amd import saving
    dreammat_amd._import_plugins()
    dev = torch.device("cuda:0")
    enc = {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 14, "base_resolution": 16,
           "per_level_scale": 1.447269237440378}
    geo = dreammat_amd.find("dreammat-mesh")({"shape_init": "quad", "shape_init_params": 1.0, "pos_encoding_config": enc}).to(dev)
    with torch.no_grad():
        geo.encoding.encoding.params.uniform_(-1, 1)
    lat = [torch.full((16, 32, 3), 0.25) for _ in range(5)]
    mat = dreammat_amd.find("dreammat-material")({"use_raytracing": False, "env_max_res": 32, "env_min_res": 8}, latlongs=lat).to(dev)
    ex = dreammat_amd.find("mesh-exporter")({"texture_size": 64, "texture_format": "png"}, geometry=geo, material=mat,
                                            background=None)
    mesh = geo.isosurface()
    maps, holes = ex.bake_textures(mesh)
    assert not bool(holes.any())                                             # the quad's UVs cover the whole atlas
    # texel (j, i) <-> uv ((i+.5)/S, (j+.5)/S) <-> quad position (u-.5, v-.5, 0): query the field there directly
    S = 64
    jj, ii = torch.meshgrid(torch.arange(S, device=dev), torch.arange(S, device=dev), indexing="ij")
    pts = torch.stack([(ii + 0.5) / S - 0.5, (jj + 0.5) / S - 0.5, torch.zeros_like(ii, dtype=torch.float32)], -1).reshape(-1, 3)
    with torch.no_grad():
        ref = mat.export(**geo.export(points=pts.float()))
    for k in ("albedo", "metallic", "roughness"):
        assert (maps[k].reshape(ref[k].shape) - ref[k]).abs().max() < 1e-4, k
    paths = saving.save_obj(str(tmp_path / "model.obj"), **ex()[0].params)
    assert sorted(os.path.basename(p) for p in paths) == ["model.mtl", "model.obj", "texture_kd.png", "texture_metallic.png",
                                                         "texture_roughness.png"]


@pytest.mark.parametrize("bvh_width", ["2", "4"])
@pytest.mark.parametrize("variant", ["schlick", "ggx_smith"])
def test_mc_wave_kernel_matches_the_serial_kernel(variant, bvh_width, monkeypatch):
    """the opt-in one-wave-per-pixel Monte-Carlo kernel (DREAMMAT_MC_KERNEL=wave; samples over the 64 lanes, ballot hit
    bits, butterfly reduction) against the validated one-thread-per-pixel kernel; its decomposition is CPU-checked in
    tests/test_golden_cpu.py, the kernel itself has not run on a GPU yet (hence the file-level xfail(strict=False))."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import numpy as np
    from dreammat_amd import _lib, hipops
    dev = torch.device("cuda:0")
    g = {k: torch.from_numpy(v) if v.ndim else v
         for k, v in np.load(os.path.join(os.path.dirname(__file__), "golden", "mc_shading.npz")).items()}
    bvh = hipops.MeshBvh(g["v_pos"], g["tri"], dev)
    monkeypatch.delenv("DREAMMAT_BVH", raising=False)
    scene_ref = hipops.McScene(bvh, [g["light"]], g[f"{variant}_dsamp"].shape[0], g[f"{variant}_ssamp"].shape[0], variant)
    monkeypatch.setenv("DREAMMAT_BVH", bvh_width)          # "4": trace through the 4-wide collapse of the same tree
    scene_new = hipops.McScene(bvh, [g["light"]], g[f"{variant}_dsamp"].shape[0], g[f"{variant}_ssamp"].shape[0], variant)
    assert (scene_new.nodes4 is not None) == (bvh_width == "4")
    mat = _lib.MatCfgStruct(0.0, 0.9, 0.01, 0.9)
    N = g["pts"].shape[0]
    rd, rs = g[f"{variant}_rand_d"].to(dev).contiguous(), g[f"{variant}_rand_s"].to(dev).contiguous()

    def run(scene):
        feats = g[f"{variant}_feats"].to(dev).requires_grad_()
        outs = hipops.mc_shade(feats, g["pts"].to(dev), g["nrm"].to(dev), g["view"].to(dev),
                               torch.zeros(N, dtype=torch.int32, device=dev), torch.full((1,), N, dtype=torch.int32, device=dev),
                               torch.zeros(1, dtype=torch.int32, device=dev), scene, mat, 1 << 30, rd, rs, True)
        (outs[0] * g[f"{variant}_wgt"].to(dev)).sum().backward()
        return [o.detach().cpu() for o in outs], feats.grad.cpu()
    monkeypatch.delenv("DREAMMAT_MC_KERNEL", raising=False)
    ref_out, ref_grad = run(scene_ref)
    monkeypatch.setenv("DREAMMAT_MC_KERNEL", "wave")
    out, grad = run(scene_new)
    for a, b in zip(out, ref_out):
        assert (a - b).abs().max() < 1e-5
    assert (grad - ref_grad).abs().max() <= 1e-5 * max(1.0, ref_grad.abs().max().item())


def test_condition_map_producer_vs_oracle_composition():
    """SURVEY row f-2 (`condition_source: render`): depth / Blender-convention view normal / 6 probe-material light maps
    from the HIP kernels, against the same recipe composed from the oracle's CPU pieces (C rasterizer + interpolate,
    EnvLight split-sum shading with the probe material)."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import numpy as np
    from dreammat_amd import envlight as penv, mesh as pmesh
    from dreammat_amd.condition import PROBE_MATERIALS, ConditionMapRenderer, lin2srgb
    from oracle import envlight as oenv, raster as oraster, shading as oshade
    from tests import util
    dev = torch.device("cuda:0")
    lat = [util.synthetic_latlong(i) * 0.02 for i in range(3)]
    fg = penv.approx_fg_lut()
    atlas = penv.EnvAtlas(lat, scale=2.0, min_res=8, max_res=32, fg_lut=fg, device=dev)
    oenvs = [oenv.EnvLight(l, scale=2.0, min_res=8, max_res=32) for l in lat]
    m = pmesh.displaced_sphere(48, 40)
    B, H, W = 2, 96, 96
    batch = util.make_views(B, H, W, seed=5)
    env_id = torch.tensor([2, 0])
    cond = ConditionMapRenderer(m, atlas, dev)(batch["mvp_mtx"], batch["c2w"], batch["rays_d"], env_id).cpu()
    assert cond.shape == (B, H, W, 22)
    # ---- oracle composition
    tri = m.t_pos_idx.numpy().astype(np.int32)
    pos_clip = oraster.vertex_transform(m.v_pos.numpy(), batch["mvp_mtx"].numpy())
    rast = torch.from_numpy(oraster.rasterize(pos_clip, tri, H, W))
    mask = rast[..., 3] > 0
    gpos = torch.from_numpy(oraster.interpolate(m.v_pos.numpy(), rast.numpy(), tri))
    gnrm = torch.nn.functional.normalize(torch.from_numpy(oraster.interpolate(m.v_nrm.numpy(), rast.numpy(), tri)), dim=-1)
    ref = torch.zeros(B, H, W, 22)
    ref[..., 1:4] = torch.tensor([0.5, 0.5, 1.0])
    c2w = batch["c2w"]
    for b in range(B):
        mk = mask[b]
        right, up, back, cam = c2w[b, :3, 0], c2w[b, :3, 1], c2w[b, :3, 2], c2w[b, :3, 3]
        inv = 1.0 / (((cam - gpos[b][mk]) * back).sum(-1) + 1e-6)
        ref[b][mk, 0] = 0.7 * (inv - inv.min()) / (inv.max() - inv.min() + 1e-6) + 0.3
        n = gnrm[b][mk]
        ref[b][mk, 1] = 0.5 * (n * right).sum(-1) + 0.5
        ref[b][mk, 2] = -0.5 * (n * up).sum(-1) + 0.5
        ref[b][mk, 3] = -0.5 * (n * back).sum(-1) + 0.5
        view = -torch.nn.functional.normalize(batch["rays_d"][b][mk], dim=-1)
        for k, (met, rough) in enumerate(PROBE_MATERIALS):
            one = torch.ones(n.shape[0], 1)
            out = oshade.shade_splitsum(n, view, oenvs[int(env_id[b])], fg, met * one, rough * one, one.expand(-1, 3))
            ref[b][mk, 4 + 3 * k:7 + 3 * k] = lin2srgb(out["color"])
    assert torch.equal(cond[..., 0] > 0, mask)                              # same coverage as the bit-exact rasterizer
    assert (cond[..., :4] - ref[..., :4]).abs().max() < 2e-4
    assert (cond[..., 4:] - ref[..., 4:]).abs().max() < 3e-3
    fgpix = cond[mask]
    assert fgpix[:, 0].min() >= 0.3 - 1e-6 and fgpix[:, 0].max() <= 1 + 1e-6 and (cond[~mask][:, 4:] == 0).all()
    assert (cond[~mask][:, 1:4] - torch.tensor([0.5, 0.5, 1.0])).abs().max() == 0
    assert (fgpix[:, 4:7] - fgpix[:, 10:13]).abs().max() > 0.02             # the probes do look different


def test_shade_kernels_with_the_fp16_atlas():
    """opt-in DREAMMAT_ATLAS=fp16 (RGBA fp16 texels: 6 instead of 12 cube-map gathers per pixel): forward within the 1e-3
    budget of the fp32 oracle, backward within 2e-3; CPU-checked through tests/hostemu, first GPU contact here."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from dreammat_amd import _lib, envlight as penv, hipops
    from oracle import envlight as oenv, shading as oshade
    from tests import util
    dev = torch.device("cuda:0")
    lat = [util.synthetic_latlong(i) * 0.02 for i in range(3)]
    fg = penv.approx_fg_lut()
    oenvs = [oenv.EnvLight(l, scale=2.0, min_res=8, max_res=32) for l in lat]
    atlas = penv.EnvAtlas(lat, scale=2.0, min_res=8, max_res=32, fg_lut=fg, device=dev, texel="fp16")
    assert atlas.spec_packed.dtype == torch.float16
    torch.manual_seed(0)
    N, HW = 30000, 10000
    n = torch.nn.functional.normalize(torch.randn(N, 3), dim=-1)
    v = torch.nn.functional.normalize(n + 0.8 * torch.randn(N, 3), dim=-1)
    feat = (torch.randn(N, 5) * 1.5).requires_grad_()
    pix = torch.randint(0, 3 * HW, (N,), dtype=torch.int32)
    env_of_view = torch.tensor([2, 0, 1], dtype=torch.int32)
    ref, _ = oshade.material_forward(feat, feat.detach() + 0.1, v, n, oenvs, env_of_view[(pix // HW).long()].long(), fg)
    dcol = torch.randn(N, 3)
    (ref["color"] * dcol).sum().backward()
    fg_ = feat.detach().to(dev).requires_grad_()
    mat = _lib.MatCfgStruct(0.0, 0.9, 0.1, 0.95)
    out = hipops.shade(fg_, n.to(dev), v.to(dev), pix.to(dev), torch.full((1,), N, dtype=torch.int32, device=dev),
                       env_of_view.to(dev), atlas, mat, HW, False)
    (out[0] * dcol.to(dev)).sum().backward()
    assert (out[0].detach().cpu() - ref["color"].detach()).abs().max() < 1e-3
    assert (fg_.grad.cpu() - feat.grad).abs().max() < 2e-3 * feat.grad.abs().max()
