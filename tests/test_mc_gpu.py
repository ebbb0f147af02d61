"""GPU checks of SURVEY row f-1 (Monte-Carlo ray-traced shading, the reference's default material branch): the BVH
any-hit kernel against the oracle's brute force, the fused shading kernels (forward + backward) against the
REFERENCE's own outputs and autograd gradients (tests/golden/mc_shading.npz), and the plugin path.  The same
arithmetic is also checked without a GPU through tests/hostemu (tests/test_golden_cpu.py, tests/test_core_cpu.py).
First run on an MI355X: all four passed (round 1, last seconds of the GPU budget)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")


def L(name):
    return {k: torch.from_numpy(v) if v.ndim else v for k, v in np.load(os.path.join(G, name)).items()}


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    return torch.device("cuda:0")


def test_bvh_any_hit_kernel_vs_brute_force(dev):
    from dreammat_amd import hipops, mesh as pmesh
    from oracle import mc_shading as omc
    torch.manual_seed(0)
    m = pmesh.displaced_sphere(48, 40)
    bvh = hipops.MeshBvh(m.v_pos, m.t_pos_idx, dev)
    tv = m.v_pos.float()[m.t_pos_idx.long()]
    fn = torch.nn.functional.normalize(torch.cross(tv[:, 1] - tv[:, 0], tv[:, 2] - tv[:, 0], dim=-1), dim=-1)
    pick = torch.randint(0, tv.shape[0], (4000,))
    d = torch.nn.functional.normalize(torch.randn(4000, 3), dim=-1)
    o = tv.mean(1)[pick] + 1e-4 * fn[pick] + 1e-5 * d
    hit = bvh.any_hit(o.to(dev), d.to(dev)).cpu()
    ref = omc.trace_any_hit(m.v_pos.float(), m.t_pos_idx, o, d)
    assert int((hit != ref).sum()) <= 4
    # the occupancy grid (dm_grid_build + dm_grid_any_hit_rays): the same boolean over the same triangle test => identical
    for res in (0, 9, 40):
        b2 = hipops.MeshBvh(m.v_pos, m.t_pos_idx, dev, grid_res=res)
        assert torch.equal(b2.any_hit_grid(o.to(dev), d.to(dev)).cpu(), hit), res
    # rays that start outside the box, miss it, or run along an axis
    o2 = torch.tensor([[3.0, 0.0, 0.0], [3.0, 3.0, 3.0], [0.0, 0.0, -4.0], [0.0, 5.0, 0.0], [0.0, 0.0, 0.0]])
    d2 = torch.tensor([[-1.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
    assert torch.equal(bvh.any_hit_grid(o2.to(dev), d2.to(dev)).cpu(), bvh.any_hit(o2.to(dev), d2.to(dev)).cpu())
    assert bvh.any_hit(o2.to(dev), d2.to(dev)).cpu().tolist() == [True, False, True, False, True]


@pytest.mark.parametrize("variant", ["schlick", "ggx_smith"])
def test_mc_shade_kernels_vs_reference(dev, variant):
    from dreammat_amd import _lib, hipops
    from oracle import shading as oshade
    g = L("mc_shading.npz")
    bvh = hipops.MeshBvh(g["v_pos"], g["tri"], dev)
    nd, nsp = g[f"{variant}_dsamp"].shape[0], g[f"{variant}_ssamp"].shape[0]
    scene = hipops.McScene(bvh, [g["light"]], nd, nsp, variant)
    assert torch.equal(scene.samples_d.cpu(), g[f"{variant}_dsamp"]) and torch.equal(scene.samples_s.cpu(), g[f"{variant}_ssamp"])
    mat = _lib.MatCfgStruct(0.0, 0.9, 0.01, 0.9)
    N = g["pts"].shape[0]
    random_az = bool(g[f"{variant}_random"])
    rd = g[f"{variant}_rand_d"].to(dev).contiguous() if random_az else None
    rs = g[f"{variant}_rand_s"].to(dev).contiguous() if random_az else None
    feats = g[f"{variant}_feats"].to(dev).requires_grad_()
    outs = hipops.mc_shade(feats, g["pts"].to(dev), g["nrm"].to(dev), g["view"].to(dev),
                           torch.zeros(N, dtype=torch.int32, device=dev), torch.full((1,), N, dtype=torch.int32, device=dev),
                           torch.zeros(1, dtype=torch.int32, device=dev), scene, mat, 1 << 30, rd, rs, True)
    names = ["color", "albedo", "specular_lights", "diffuse_lights", "specular_colors", "diffuse_colors", "metalness", "roughness"]
    for k, o in zip(names, outs):
        ref = g[f"{variant}_out_{k}"]
        assert (o.detach().cpu() - ref).abs().max() <= 2e-4 * max(1.0, ref.abs().max().item()), k
    (outs[0] * g[f"{variant}_wgt"].to(dev)).sum().backward()
    fr = g[f"{variant}_feats"].clone().requires_grad_()
    (3.0 * oshade.material_smoothness_grad(torch.sigmoid(fr), torch.sigmoid(g[f"{variant}_featsj"]))).backward()
    ref_grad = g[f"{variant}_dfeats"] - fr.grad
    assert (feats.grad.cpu() - ref_grad).abs().max() <= 2e-3 * ref_grad.abs().max()


def test_material_plugin_raytracing_branch_runs(dev):
    import dreammat_amd
    from dreammat_amd import hipops, mesh as pmesh
    dreammat_amd._import_plugins()
    lat = [torch.rand(16, 32, 3) for _ in range(5)]
    mat = dreammat_amd.find("dreammat-material")({"use_raytracing": True, "diffuse_sample_num": 32, "specular_sample_num": 16,
                                                  "env_max_res": 32, "env_min_res": 8}, latlongs=lat).to(dev)
    m = pmesh.displaced_sphere(24, 16)
    mat.set_raytracer(hipops.MeshBvh(m.v_pos, m.t_pos_idx, dev))
    tv = m.v_pos.float()[m.t_pos_idx.long()]
    n = torch.nn.functional.normalize(torch.cross(tv[:, 1] - tv[:, 0], tv[:, 2] - tv[:, 0], dim=-1), dim=-1)[:200].to(dev)
    p = (tv.mean(1)[:200].to(dev) + 1e-4 * n).contiguous()
    v = torch.nn.functional.normalize(n + 0.5 * torch.randn(200, 3, device=dev), dim=-1)
    f = torch.randn(200, 5, device=dev, requires_grad=True)
    out, reg = mat(p, f, f.detach() + 0.1, v, n, torch.zeros(1, dtype=torch.int32, device=dev))
    assert out["color"].shape == (200, 3) and torch.isfinite(out["color"]).all() and 0 <= float(out["color"].min())
    (out["color"].sum() + reg).backward()
    assert torch.isfinite(f.grad).all() and f.grad.abs().sum() > 0


@pytest.mark.parametrize("tracer", ["bvh2", "bvh4", "grid", "grid96"])
@pytest.mark.parametrize("variant", ["schlick", "ggx_smith"])
def test_mc_wave_kernel_matches_the_serial_kernel(variant, tracer, monkeypatch):
    """the one-wave-per-pixel Monte-Carlo kernel (the default; samples over the 64 lanes, ballot hit
    bits, butterfly reduction) against the validated one-thread-per-pixel kernel on the binary tree, with each way of
    answering the occlusion queries: binary / 4-wide BVH walked per lane, and (the default) the occupancy grid walked by the
    whole wave together (grid_trace_wave).  The decomposition is CPU-checked in tests/test_golden_cpu.py."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import numpy as np
    from dreammat_amd import _lib, hipops
    dev = torch.device("cuda:0")
    g = {k: torch.from_numpy(v) if v.ndim else v
         for k, v in np.load(os.path.join(os.path.dirname(__file__), "golden", "mc_shading.npz")).items()}
    # "grid96": a 96-cell grid -- its tables and the per-wave scratch do not fit LDS together, the kernel walks the grid per lane
    # with the tables in global memory
    bvh = hipops.MeshBvh(g["v_pos"], g["tri"], dev, grid_res=96 if tracer == "grid96" else 0)
    monkeypatch.setenv("DREAMMAT_BVH", "2")                # reference: the validated serial kernel on the binary tree
    monkeypatch.setenv("DREAMMAT_MC_TRACER", "bvh")
    scene_ref = hipops.McScene(bvh, [g["light"]], g[f"{variant}_dsamp"].shape[0], g[f"{variant}_ssamp"].shape[0], variant)
    assert scene_ref.grid is None and scene_ref.nodes4 is None
    monkeypatch.setenv("DREAMMAT_BVH", "2" if tracer == "bvh2" else "4")
    monkeypatch.setenv("DREAMMAT_MC_TRACER", "grid" if tracer.startswith("grid") else "bvh")
    scene_new = hipops.McScene(bvh, [g["light"]], g[f"{variant}_dsamp"].shape[0], g[f"{variant}_ssamp"].shape[0], variant)
    assert (scene_new.grid is not None) == tracer.startswith("grid") and (scene_new.nodes4 is not None) == (tracer != "bvh2")
    if tracer == "grid96":
        assert max(scene_new.grid.dim) > 88
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
    monkeypatch.setenv("DREAMMAT_MC_KERNEL", "serial")
    ref_out, ref_grad = run(scene_ref)
    monkeypatch.setenv("DREAMMAT_MC_KERNEL", "wave")
    out, grad = run(scene_new)
    for a, b in zip(out, ref_out):
        assert (a - b).abs().max() < 1e-5
    assert (grad - ref_grad).abs().max() <= 1e-5 * max(1.0, ref_grad.abs().max().item())
