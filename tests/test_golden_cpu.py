"""Pins the oracle (and the CPU-capable product code) to outputs of the REFERENCE'S OWN function bodies
(tests/golden/*.npz, produced by tests/golden/make_golden.py from /root/reference)."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import camera as ocam, envlight as oenv, render as orender, sd_nets as osd, shading as oshade

G = os.path.join(os.path.dirname(__file__), "golden")


def L(name):
    return {k: torch.from_numpy(v) if v.ndim else v for k, v in np.load(os.path.join(G, name)).items()}


def test_camera_math_vs_reference():
    g = L("camera.npz")
    H = W = 16
    from dreammat_amd import camera as pcam
    for mod in (ocam, pcam):
        dirs = mod.get_ray_directions(H, W, 1.0)
        assert torch.equal(dirs, g["dirs"])
        focal = 0.5 * H / torch.tan(0.5 * g["fovy"])
        d = dirs[None].repeat(3, 1, 1, 1)
        d[..., :2] = d[..., :2] / focal[:, None, None, None]
        ro, rd = (mod.get_rays(d, g["c2w"]) if mod is ocam else mod.get_rays(d, g["c2w"], keepdim=True))
        assert torch.allclose(rd, g["rays_d"], atol=1e-7) and torch.allclose(ro, g["rays_o"])
        proj = mod.get_projection_matrix(g["fovy"], 1.0, 0.1, 1000.0)
        assert torch.equal(proj, g["proj"])
        mvp, w2c = mod.get_mvp_matrix(g["c2w"], proj)
        assert torch.allclose(mvp, g["mvp"], atol=1e-6) and torch.allclose(w2c, g["w2c"], atol=1e-7)
    assert torch.allclose(oshade.lin2srgb(g["lin2srgb_x"]), g["lin2srgb_y"], atol=1e-7)


def test_schedule_C_vs_reference():
    from dreammat_amd.config import C
    g = np.load(os.path.join(G, "schedule.npz"))
    specs = [[0, -1.0, -0.5, 2000], [0, 0.0, -0.5, 2000], [500, 0.2, 0.02, 501], [500, 0.8, 0.5, 501], 1.05, [0.1, 0.9, 10.0]]
    for i, s in enumerate(specs):
        for j, st in enumerate(g["steps"]):
            assert abs(C(s, 3, int(st)) - g["values"][i, j]) < 1e-12


def test_material_and_splitsum_vs_reference():
    g = L("shading.npz")
    assert abs(float(oshade.material_smoothness_grad(g["mat"], g["matj"])) - float(g["reg"])) < 1e-7
    env = oenv.EnvLight(g["latlong"], scale=2.0, min_res=8, max_res=16)
    N = g["feats"].shape[0]
    out, reg = oshade.material_forward(g["feats"], g["featsj"], g["view"], g["nrm"], [env],
                                       torch.zeros(N, dtype=torch.long), g["fg"])
    assert abs(float(reg) - float(g["mat_reg"])) < 1e-7
    for k in ["color", "albedo", "roughness", "metalness", "specular_lights", "diffuse_lights", "specular_colors",
              "diffuse_colors"]:
        assert (out[k] - g["out_" + k]).abs().max() < 1e-6, k


def test_material_and_splitsum_product_core_vs_reference(hostemu):
    """the product's shade kernel core (host build) against the reference's shade_splitsum arithmetic."""
    import ctypes
    from dreammat_amd import envlight as penv
    from tests.util import P
    g = L("shading.npz")
    atlas = penv.EnvAtlas([g["latlong"]], scale=2.0, min_res=8, max_res=16, fg_lut=g["fg"])
    N = g["feats"].shape[0]
    color = np.empty((N, 3), np.float32); dbg = np.empty((N, 17), np.float32)
    mat = np.array([0.0, 0.9, 0.1, 0.95], np.float32)
    a = [np.ascontiguousarray(g[k].numpy()) for k in ("nrm", "view", "feats")]
    env = np.zeros(N, np.int32)
    hostemu.emu_shade(ctypes.byref(atlas.struct), P(mat), P(a[0]), P(a[1]), P(a[2]), P(env), ctypes.c_longlong(N),
                      P(color), P(dbg), None, None)
    assert np.abs(color - g["out_color"].numpy()).max() < 1e-5
    assert np.abs(dbg[:, 0:3] - g["out_albedo"].numpy()).max() < 1e-6
    assert np.abs(dbg[:, 3:6] - g["out_specular_lights"].numpy()).max() < 1e-5
    assert np.abs(dbg[:, 16:17] - g["out_roughness"].numpy()).max() < 1e-6


def test_renderer_helpers_vs_reference():
    g = L("renderer.npz")
    assert torch.allclose(orender.get_orthogonal_directions(g["normals"]), g["ortho"], atol=1e-7)
    w2c_rows = g["w2c"].expand(g["normals"].shape[0], 4, 4)
    assert torch.allclose(orender.controlnet_normals(g["normals"], w2c_rows), g["cn_normals"], atol=1e-6)


def test_sds_composition_vs_reference():
    g = L("sds.npz")
    assert torch.allclose(osd.alphas_cumprod(), g["alphas"])
    w = (1 - g["alphas"][g["t"]]).view(-1, 1, 1, 1)
    grad = w * (1.05 * g["eps_text"] + -0.75 * g["eps_uncond"] + -0.25 * g["eps_null"] + 0.1 * g["noise"])
    assert torch.allclose(grad, g["grad"], atol=1e-6)
    assert int(g["min_step"]) == 200 and int(g["max_step"]) == 800
    # the product's guidance composes the same gradient (nets stubbed by the recorded eps)
    from dreammat_amd.guidance import StableDiffusionLightGuidance
    gd = StableDiffusionLightGuidance.__new__(StableDiffusionLightGuidance)
    from dreammat_amd.sd import DDIMScheduler
    gd.scheduler = DDIMScheduler(); gd.alphas = gd.scheduler.alphas_cumprod
    gd.min_step, gd.max_step = 200, 800
    gd.cond_scale, gd.uncond_scale, gd.null_scale, gd.noise_scale = 1.05, -0.75, -0.25, 0.1
    gd.compute_without_perpneg = lambda *a, **k: (g["eps_text"], g["eps_uncond"], g["eps_null"])
    pg, ev = gd.compute_grad_sds(types.SimpleNamespace(use_perp_neg=False), [1.0], g["lat"], [], None, None, None,
                                 rng={"t": g["t"], "noise": g["noise"]})
    assert torch.allclose(pg, g["grad"], atol=1e-6)
    for k in ("uncond_m_noise_norm", "text_m_null_norm", "noise_norm"):
        assert abs(float(ev[k]) - float(g["ev_" + k])) < 1e-4


def test_view_dependent_prompt_selection_vs_reference():
    from dreammat_amd.prompt import shift_azimuth_deg
    g = L("prompt.npz")
    assert torch.equal(shift_azimuth_deg(g["az"]), g["shifted"])


@pytest.mark.skipif(not os.path.exists("/root/reference"), reason="reference tree only exists in the build container")
def test_reference_assets_loaders():
    """The in-tree data fixtures of the reference load through the product's own readers."""
    import hashlib
    from dreammat_amd import envlight as penv, mesh as pmesh
    root = "/root/reference/threestudio_dreammat/load"
    lut = penv.load_fg_lut(os.path.join(root, "lights/bsdf_256_256.bin"))
    raw = open(os.path.join(root, "lights/bsdf_256_256.bin"), "rb").read()
    assert hashlib.sha256(raw).hexdigest() == "aee514f7c7e561a357e529567222da99e84886c31c46a32fe767a5b066bbe196"
    assert lut.shape == (256, 256, 2) and abs(float(lut[0, 0, 0]) - 0.00973) < 1e-4 and abs(float(lut[255, 0, 0]) - 0.94153) < 1e-4
    hdr = penv.read_hdr(os.path.join(root, "lights/mud_road_puresky_1k.hdr"))
    assert hdr.shape == (512, 1024, 3) and np.isfinite(hdr).all()
    assert np.array_equal(hdr, oenv.load_hdr(os.path.join(root, "lights/mud_road_puresky_1k.hdr")))
    m = pmesh.load_obj(os.path.join(root, "shapes/objs/apple.obj"))
    assert m.t_pos_idx.shape[0] == 4164


@pytest.mark.parametrize("variant", ["schlick", "ggx_smith"])
def test_mc_raytraced_shading_oracle_vs_reference(variant):
    """SURVEY row f-1 groundwork: the oracle's restatement of the reference's default (Monte-Carlo, ray-traced)
    material branch against the reference's own shade_raytracing / forward bodies -- outputs AND the gradients wrt
    the material features (autograd through sample directions, pdfs, BRDF terms)."""
    from oracle import mc_shading as omc
    g = L("mc_shading.npz")
    nd, nsp = g[f"{variant}_dsamp"].shape[0], g[f"{variant}_ssamp"].shape[0]
    assert torch.equal(omc.direction_samples(nd), g[f"{variant}_dsamp"])            # Fibonacci tables of configure()
    assert torch.equal(omc.direction_samples(nsp), g[f"{variant}_ssamp"])
    random_az = bool(g[f"{variant}_random"])
    feats = g[f"{variant}_feats"].clone().requires_grad_()
    featsj = g[f"{variant}_featsj"].clone().requires_grad_()
    trace = lambda o, d: omc.trace_any_hit(g["v_pos"], g["tri"], o, d)
    out, reg = omc.material_forward_mc(g["pts"], feats, featsj, g["view"], g["nrm"], g["light"], g[f"{variant}_dsamp"],
                                       g[f"{variant}_ssamp"], trace,
                                       g[f"{variant}_rand_d"] if random_az else None,
                                       g[f"{variant}_rand_s"] if random_az else None, geometry_type=variant)
    assert abs(float(reg) - float(g[f"{variant}_mat_reg"])) < 1e-7
    for k in ["color", "albedo", "roughness", "metalness", "specular_lights", "diffuse_lights", "specular_colors",
              "diffuse_colors"]:
        ref = g[f"{variant}_out_{k}"]
        assert (out[k] - ref).abs().max() <= 1e-5 * max(1.0, ref.abs().max().item()), k
    ((out["color"] * g[f"{variant}_wgt"]).sum() + 3.0 * reg).backward()
    for got, ref in ((feats.grad, g[f"{variant}_dfeats"]), (featsj.grad, g[f"{variant}_dfeatsj"])):
        assert (got - ref).abs().max() <= 1e-4 * max(1e-3, ref.abs().max().item())
    # sanity of the stubbed scene: some directions are occluded by the mesh, most are not
    dirs = omc.sample_diffuse_directions(g["nrm"], g[f"{variant}_dsamp"], None)
    hit = trace((g["pts"][:, None] + 1e-5 * dirs).reshape(-1, 3), dirs.reshape(-1, 3))
    assert 0 < int(hit.sum()) < hit.numel() // 2


def _emu_mc_shade(hostemu, bvh, light, dsamp, ssamp, pts, nrm, view, feats, rand_d, rand_s, ggx_smith, dcolor=None, hit_bits=None,
                  lanes=1, wide=False):
    import ctypes
    N = pts.shape[0]
    cp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    f = lambda t: t.detach().float().contiguous() if t is not None else None
    cfg4 = torch.tensor([0.0, 0.9, 0.01, 0.9])
    light, dsamp, ssamp, pts, nrm, view, feats = map(f, (light, dsamp, ssamp, pts, nrm, view, feats))
    rand_d, rand_s, dcolor = f(rand_d), f(rand_s), f(dcolor)
    if hit_bits is None:
        hit_bits = torch.zeros(N, 32, dtype=torch.int32)
    out = torch.zeros(N, 25)
    dfeat = torch.zeros(N, 5)
    rc = hostemu.emu_mc_shade(cp(cfg4), dsamp.shape[0], ssamp.shape[0], int(ggx_smith), cp(bvh.nodes_host),
                              cp(bvh.nodes4_host) if wide else None, cp(bvh.tris_host),
                              cp(light), light.shape[0], light.shape[1], cp(dsamp), cp(ssamp), ctypes.c_longlong(N), cp(pts),
                              cp(nrm), cp(view), cp(feats), cp(rand_d), cp(rand_s), cp(hit_bits), cp(out), cp(dcolor), cp(dfeat), int(lanes))
    assert rc == 0
    return out, dfeat, hit_bits


@pytest.mark.parametrize("variant", ["schlick", "ggx_smith"])
def test_mc_shading_product_core_vs_reference(hostemu, variant):
    """the product's Monte-Carlo shading core (csrc/mc_shade_core.h: BVH any-hit + sampling + BRDF + forward-mode
    d/d(alpha)), run on the CPU through tests/hostemu, against the REFERENCE's outputs and autograd gradients."""
    from dreammat_amd import hipops
    g = L("mc_shading.npz")
    bvh = hipops.MeshBvh(g["v_pos"], g["tri"])
    random_az = bool(g[f"{variant}_random"])
    rd = g[f"{variant}_rand_d"] if random_az else None
    rs = g[f"{variant}_rand_s"] if random_az else None
    args = (hostemu, bvh, g["light"], g[f"{variant}_dsamp"], g[f"{variant}_ssamp"], g["pts"], g["nrm"], g["view"],
            g[f"{variant}_feats"], rd, rs, variant == "ggx_smith")
    out, _, bits = _emu_mc_shade(*args)
    cols = {"color": (0, 3), "albedo": (3, 6), "roughness": (6, 7), "metalness": (7, 8), "specular_lights": (8, 11),
            "diffuse_lights": (11, 14), "specular_colors": (14, 17), "diffuse_colors": (17, 20)}
    for k, (a, b) in cols.items():
        ref = g[f"{variant}_out_{k}"]
        err = (out[:, a:b] - ref).abs().max().item()
        assert err <= 2e-4 * max(1.0, ref.abs().max().item()), (k, err)
    # backward on the recorded hit bits: d(sum(color * wgt)) / d features  (mat_reg's share is subtracted: it has its
    # own kernel, k_matreg)
    from oracle import shading as oshade
    feats = g[f"{variant}_feats"].clone().requires_grad_()
    featsj = g[f"{variant}_featsj"].clone()
    (3.0 * oshade.material_smoothness_grad(torch.sigmoid(feats), torch.sigmoid(featsj))).backward()
    ref_grad = g[f"{variant}_dfeats"] - feats.grad
    _, dfeat, _ = _emu_mc_shade(*args, dcolor=g[f"{variant}_wgt"], hit_bits=bits)
    scale = ref_grad.abs().max().item()
    assert (dfeat - ref_grad).abs().max().item() <= 2e-3 * scale, ((dfeat - ref_grad).abs().max().item(), scale)
    assert (dfeat[:, 4].abs() > 0).any() and (dfeat[:, 3].abs() > 0).any()           # roughness / metallic do get gradient
    # the wave kernel's decomposition (samples strided over 64 lanes, ballot-packed hit bits, sums combined before the
    # finish) gives the same pixel: identical hit bits, values equal up to fp32 summation order
    out64, _, bits64 = _emu_mc_shade(*args, lanes=64)
    assert torch.equal(bits64, bits) and (out64 - out).abs().max() < 1e-5
    _, dfeat64, _ = _emu_mc_shade(*args, dcolor=g[f"{variant}_wgt"], hit_bits=bits64, lanes=64)
    assert (dfeat64 - dfeat).abs().max() <= 1e-5 * max(1.0, scale)
    # tracing through the 4-wide collapse of the BVH finds the same occlusions
    out4, _, bits4 = _emu_mc_shade(*args, wide=True)
    assert torch.equal(bits4, bits) and torch.equal(out4, out)


def test_perp_neg_prompting_vs_reference():
    """Perp-Neg (non-default `use_perp_neg`): the product's get_text_embeddings_perp_neg / perpendicular_component against the
    reference's own bodies executed on seeded inputs (tests/golden/make_perpneg.py): front-side and side-back interpolation,
    the overhead case, azimuths outside (-180, 180]."""
    from dreammat_amd import prompt as P
    g = L("perpneg.npz")
    shift = P.shift_azimuth_deg
    dirs = [P.DirectionConfig("side", None, None, lambda e, a, d: torch.ones_like(e, dtype=torch.bool)),
            P.DirectionConfig("front", None, None, lambda e, a, d: (shift(a) > -45) & (shift(a) < 45)),
            P.DirectionConfig("back", None, None, lambda e, a, d: (shift(a) > 135) | (shift(a) < -135)),
            P.DirectionConfig("overhead", None, None, lambda e, a, d: e > 60)]
    out = P.PromptProcessorOutput(g["vd"][:1], g["uvd"][:1], g["null"], g["vd"], g["uvd"], dirs,
                                  {"side": 0, "front": 1, "back": 2, "overhead": 3}, True)
    emb, w = out.get_text_embeddings_perp_neg(g["ele"], g["azi"], g["dis"], True, True)
    assert emb.shape == g["emb"].shape and torch.allclose(emb, g["emb"], atol=1e-6)
    assert torch.allclose(w, g["w"], atol=1e-6)
    assert torch.allclose(P.perpendicular_component(g["x"], g["y"]), g["perp"], atol=1e-6)
