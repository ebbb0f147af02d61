"""GPU parity tests (-m gpu): every HIP kernel, called through the C ABI, against the CPU oracle on the
same seeded inputs.  Integer/index work must be bit-exact; floating point within the tolerance written
in each test (north_star: 1e-3 relative, fp32)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dreammat_amd import _lib, envlight as penv, hipops, mesh as pmesh
from oracle import camera, envlight as oenv, field as ofield, raster as oraster, render as orender, shading as oshade
from tests import util

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")



def _rm(gb):
    """Permutation that lists the compacted G-buffer rows in the reference's row-major `x[selector]` order (the product
    compacts in tile order since round 4; every consumer goes through pix_idx)."""
    return torch.argsort(gb.pix_idx.long()).cpu()


class _RowView:
    """rows of a [N, C] tensor (and of its .grad) re-ordered by a permutation, for the per-pixel diagnostics"""

    def __init__(self, t, perm):
        self.t, self.perm = t, perm

    @property
    def grad(self):
        return self.t.grad[self.perm.to(self.t.grad.device)]

    def __getitem__(self, i):
        return self.t[self.perm.to(self.t.device)][i]


def _tile_order(cov):
    """pixel indices of the covered pixels of cov[B,H,W] in the product's documented tile order"""
    B, H, W = cov.shape
    out = []
    lx = np.array([(l & 1) | ((l >> 1) & 2) | ((l >> 2) & 4) for l in range(64)])
    ly = np.array([((l >> 1) & 1) | ((l >> 2) & 2) | ((l >> 3) & 4) for l in range(64)])
    for b in range(B):
        for ty in range((H + 15) // 16):
            for tx in range((W + 15) // 16):
                for w in range(4):
                    x = tx * 16 + (w & 1) * 8 + lx
                    y = ty * 16 + (w >> 1) * 8 + ly
                    ok = (x < W) & (y < H)
                    x, y = x[ok], y[ok]
                    c = cov[b, y, x]
                    out.append(((b * H + y) * W + x)[c])
    return np.concatenate(out)


def _dump(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **{k: np.asarray(v) for k, v in arrs.items()})


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _scene(kind, B, H, W, seed=1):
    if kind == "quad":
        m = pmesh.quad_mesh()
        batch = camera.camera_batch(torch.tensor([80.0]), torch.tensor([10.0]), torch.tensor([3.5]),
                                    torch.tensor([35.0]), H, W)
    else:
        m = pmesh.displaced_sphere(*kind)
        batch = util.make_views(B, H, W, seed)
    return m, util.mesh_dict(m), batch


RASTER_CASES = [("quad", 1, 256, 256), ((48, 40), 3, 128, 128), ((24, 16), 2, 67, 45), ((160, 160), 4, 512, 512)]


@pytest.mark.parametrize("kind,B,H,W", RASTER_CASES)
def test_rasterize_interpolate_antialias_bit_exact(dev, kind, B, H, W):
    m, md, batch = _scene(kind, B, H, W)
    tri_np = md["t_pos_idx"]
    v = m.v_pos.to(dev); tri = m.t_pos_idx.to(dev).int().contiguous()
    pos = hipops.vertex_transform(v, batch["mvp_mtx"].to(dev))
    pos_o = oraster.vertex_transform(md["v_pos"], batch["mvp_mtx"]).numpy()
    assert np.array_equal(pos.cpu().numpy().view(np.uint32), pos_o.view(np.uint32)), "clip positions not bit-exact"
    ctx = hipops.RasterContext(dev)
    rast = ctx.rasterize(pos, tri, H, W, check_overflow=True)
    ro = oraster.rasterize(pos_o, tri_np, H, W)
    rg = rast.cpu().numpy()
    if not np.array_equal(rg[..., 3], ro[..., 3]):
        _dump(f"raster_fail_{B}_{H}", gpu=rg, ref=ro)
    assert np.array_equal(rg[..., 3], ro[..., 3]), f"coverage ids differ at {(rg[..., 3] != ro[..., 3]).sum()} pixels"
    assert np.array_equal(rg.view(np.uint32), ro.view(np.uint32)), "u/v/zw not bit-identical"
    assert (ro[..., 3] > 0).mean() > 0.05
    # interpolate
    it = hipops.interpolate(m.v_nrm.to(dev), rast, tri).cpu().numpy()
    io = oraster.interpolate(md["v_nrm"], ro, tri_np)
    assert np.abs(it - io).max() < 1e-6
    # antialias plan: bit-exact; apply / grad: fp32 sums of <= 5 terms
    opp = hipops.build_topology(tri)
    plan = hipops.antialias_plan(pos, tri, opp, rast)
    po = oraster.antialias_plan(pos_o, tri_np, md["opp"], ro)
    assert np.array_equal(plan.cpu().numpy().view(np.uint32), po.view(np.uint32))
    g = torch.Generator().manual_seed(3)
    for C in (1, 3):
        col = torch.rand(B, H, W, C, generator=g)
        colg = col.to(dev).requires_grad_()
        out = hipops.antialias(colg, plan)
        ref = oraster.antialias_apply(col.numpy(), po)
        assert np.abs(out.detach().cpu().numpy() - ref).max() < 1e-5
        dy = torch.rand(B, H, W, C, generator=g)
        out.backward(dy.to(dev))
        assert np.abs(colg.grad.cpu().numpy() - oraster.antialias_grad(dy.numpy(), po)).max() < 1e-5


def test_gbuffer_and_control_maps(dev):
    B, H, W = 3, 128, 128
    m, md, batch = _scene((48, 40), B, H, W)
    tri = m.t_pos_idx.to(dev).int().contiguous()
    pos = hipops.vertex_transform(m.v_pos.to(dev), batch["mvp_mtx"].to(dev))
    rast = hipops.RasterContext(dev).rasterize(pos, tri, H, W)
    g = torch.Generator().manual_seed(5)
    ju, jn = torch.rand(B, H, W, generator=g), torch.randn(B, H, W, generator=g)
    gb = hipops.gbuffer_compact(rast, tri, m.v_pos.to(dev), m.v_nrm.to(dev), batch["rays_d"].to(dev), ju.to(dev),
                                jn.to(dev), 0.05)
    ro = rast.cpu().numpy()
    sel = torch.from_numpy(ro[..., 3] > 0).reshape(-1)
    assert gb.n == int(sel.sum())
    # documented row order (csrc/raster.hip, include/dreammat_hip.h): 16 x 16 macro tiles row-major inside a view, 8 x 8
    # sub-tiles row-major inside a macro tile, Morton order inside a sub-tile; pix_idx keeps the row-major pixel index
    assert np.array_equal(gb.pix_idx.cpu().numpy(), _tile_order(sel.numpy().reshape(B, H, W)))
    gb_row = hipops.gbuffer_compact(rast, tri, m.v_pos.to(dev), m.v_nrm.to(dev), batch["rays_d"].to(dev), ju.to(dev),
                                    jn.to(dev), 0.05, order="row")
    assert np.array_equal(gb_row.pix_idx.cpu().numpy(), np.nonzero(sel.numpy())[0])     # the reference's x[selector] order
    rm = _rm(gb)
    for nm_t in ("pos", "pos_jitter", "nrm", "view"):                                   # same rows, other order: bit-equal
        assert torch.equal(getattr(gb, nm_t)[:, rm.to(dev)], getattr(gb_row, nm_t)), nm_t
    gb = gb_row
    gn = torch.nn.functional.normalize(torch.from_numpy(oraster.interpolate(md["v_nrm"], ro, md["t_pos_idx"])), dim=-1)
    gp = torch.from_numpy(oraster.interpolate(md["v_pos"], ro, md["t_pos_idx"]))
    n_sel, p_sel = gn.reshape(-1, 3)[sel], gp.reshape(-1, 3)[sel]
    assert (gb.nrm.t().cpu() - n_sel).abs().max() < 1e-5
    assert (gb.pos.t().cpu() - p_sel).abs().max() < 1e-5
    assert (gb.view.t().cpu() + batch["rays_d"].reshape(-1, 3)[sel]).abs().max() == 0
    x = orender.get_orthogonal_directions(n_sel)
    y = torch.cross(n_sel, x, dim=-1)
    ang = (ju.reshape(-1)[sel] * np.pi * 2)[:, None]
    pj = p_sel + (torch.cos(ang) * x + torch.sin(ang) * y) * (jn.reshape(-1)[sel] * 0.05)[:, None]
    assert (gb.pos_jitter.t().cpu() - pj).abs().max() < 2e-5
    depth, normal = hipops.control_maps(rast, tri, m.v_nrm.to(dev), batch["w2c"].to(dev))
    mask = torch.from_numpy(ro[..., 3:] > 0)
    d_ref = orender.normalize_depth(torch.from_numpy(ro[..., 2:3].copy()), mask)
    assert (depth.cpu() - d_ref).abs().max() < 1e-4
    view_of = torch.arange(B)[:, None, None].expand(B, H, W).reshape(-1)[sel]
    nc = orender.controlnet_normals(n_sel, batch["w2c"][view_of])
    ref = torch.tensor([0.5, 0.5, 1.0]).expand(B * H * W, 3).clone()
    ref[sel] = nc
    assert (normal.reshape(-1, 3).cpu() - ref).abs().max() < 1e-5


def test_hashgrid_forward_backward(dev):
    torch.manual_seed(0)
    spec = hipops.GridSpec(n_levels=8, log2_hashmap_size=12)
    lv, tot = ofield.grid_levels(n_levels=8, log2_hashmap_size=12)
    assert tot == spec.total_entries
    table = (torch.rand(tot * 2) * 2 - 1)
    x = torch.rand(5000, 3) * 1.9 - 0.95
    tg = table.to(dev).requires_grad_()
    for layout in ("aos", "soa"):
        xg = x.to(dev) if layout == "aos" else x.t().contiguous().to(dev).t()
        enc = hipops.hashgrid_encode(xg, tg, spec, 1.0)
        to = table.reshape(-1, 2).clone().requires_grad_()
        ref = ofield.hash_encode(ofield.contract_to_unisphere(x), to, lv)
        assert (enc.detach().cpu() - ref.detach()).abs().max() < 1e-4, layout   # |table| <= 1, fp32 pos at scale 4095
        dy = torch.randn(5000, 16)
        tg.grad = None
        enc.backward(dy.to(dev))
        ref.backward(dy)
        err = (tg.grad.cpu().reshape(-1, 2) - to.grad).abs().max()
        assert err < 1e-3 * to.grad.abs().max(), (layout, float(err))


def test_hashgrid_2d_uv_space_field(dev):
    """the uv-space field's grid (n_input_dims = 2, dm_hashgrid2d_fwd / _bwd): tcnn's rules over 2-D points, dense and hashed
    levels (log2_hashmap_size 10 hashes from level 2 on; dreammat.yaml's 16 x 2^19 pyramid is dense up to res 724, level 10)."""
    for n_levels, log2 in ((8, 10), (16, 19)):
        torch.manual_seed(4)
        spec = hipops.GridSpec(n_levels=n_levels, log2_hashmap_size=log2, n_dims=2)
        lv, tot = ofield.grid_levels(n_levels=n_levels, log2_hashmap_size=log2, n_dims=2)
        assert tot == spec.total_entries and [l["size"] for l in lv] == [l["size"] for l in spec.levels]
        first_hashed = next(i for i, l in enumerate(lv) if l["size"] < l["res"] ** 2)
        assert lv[0]["size"] == 16 * 16 and first_hashed == (2 if log2 == 10 else 11)
        table = torch.rand(tot * 2) * 2 - 1
        x = torch.rand(4000, 2) * 1.9 - 0.95
        tg = table.to(dev).requires_grad_()
        for layout in ("aos", "soa"):
            xg = x.to(dev) if layout == "aos" else x.t().contiguous().to(dev).t()
            enc = hipops.hashgrid_encode(xg, tg, spec, 1.0)
            to = table.reshape(-1, 2).clone().requires_grad_()
            ref = ofield.hash_encode(ofield.contract_to_unisphere(x), to, lv)
            # (fp32 positions at scale 4095 carry 2.4e-4 of a cell: the fine levels of the 16-level pyramid see that in the weights)
            assert (enc.detach().cpu() - ref.detach()).abs().max() < (1e-4 if n_levels == 8 else 2e-3), (n_levels, layout)
            dy = torch.randn(4000, 2 * n_levels)
            tg.grad = None
            enc.backward(dy.to(dev))
            ref.backward(dy)
            err = (tg.grad.cpu().reshape(-1, 2) - to.grad).abs().max()
            assert err < 1e-3 * to.grad.abs().max(), (n_levels, layout, float(err))


def test_renderer_uv_space_field_vs_oracle(dev, envs):
    """geometry.n_input_dims = 2 (dreammat_mesh.py:128-135, raytracing_renderer.py:177-181): the field is queried at the
    pixel's interpolated texture coordinate and at that + N(0, 0.005); all outputs and the gradients against oracle/render.py
    with the same injected draws."""
    from dreammat_amd.geometry import DreamMatMesh
    from dreammat_amd.material import DreamMatMaterial
    from dreammat_amd.renderer import RaytraceRender
    from dreammat_amd.background import SolidColorBackground
    lat, fg, oenvs = envs
    torch.manual_seed(0)
    enc_cfg = {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 12,
               "base_resolution": 16, "per_level_scale": 1.447269237440378}
    geom = DreamMatMesh({"shape_init": "sphere:48:40", "shape_init_params": 0.7, "pos_encoding_config": enc_cfg, "n_input_dims": 2}).to(dev)
    with torch.no_grad():
        geom.encoding.encoding.params.copy_((torch.rand_like(geom.encoding.encoding.params) * 2 - 1))
        geom.feature_network.layers[0].weight.copy_(torch.randn(64, 16) * 0.5)
        geom.feature_network.layers[2].weight.copy_(torch.randn(5, 64) * 0.3)
    mat = DreamMatMaterial({"use_raytracing": False, "environment_scale": 2.0, "env_max_res": 32, "env_min_res": 8,
                            "n_envs": 3}, latlongs=lat).to(dev)
    rend = RaytraceRender({}, geometry=geom, material=mat, background=SolidColorBackground({}))
    B, H, W = 2, 96, 96
    batch = util.make_views(B, H, W, seed=5)
    batch["env_id"] = torch.tensor([2, 0])
    g = torch.Generator().manual_seed(11)
    ju, jn = torch.rand(B, H, W, generator=g), torch.randn(B, H, W, generator=g)
    gbatch = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    # the per-covered-pixel uv draws: first a run with zeros to learn the coverage count, then the real one
    out0 = rend(**gbatch, light_positions=None, jitter_u=ju.to(dev), jitter_n=jn.to(dev), jitter_uv=torch.zeros(B * H * W, 2, device=dev))
    N = out0["_internals"]["gbuffer"].n
    juv = torch.randn(N, 2, generator=g)                      # the oracle's draw: one row per covered pixel, row-major
    juv_dense = torch.zeros(B * H * W, 2)                      # the product takes it per PIXEL (independent of the G-buffer's row order)
    juv_dense[torch.sort(out0["_internals"]["gbuffer"].pix_idx.long().cpu()).values] = juv
    out = rend(**gbatch, light_positions=None, jitter_u=ju.to(dev), jitter_n=jn.to(dev), jitter_uv=juv_dense.to(dev), check_overflow=True)
    m = geom.mesh
    md = dict(v_pos=m.v_pos.cpu().numpy(), v_nrm=m.v_nrm.cpu().numpy(), t_pos_idx=m.t_pos_idx.cpu().numpy().astype(np.int32),
              v_tex=m.v_tex.cpu().numpy())
    md["opp"] = oraster.build_topology(md["t_pos_idx"])
    lv, tot = ofield.grid_levels(n_levels=8, log2_hashmap_size=12, n_dims=2)
    table = geom.encoding.encoding.params.detach().cpu().reshape(-1, 2).clone().requires_grad_()
    w1 = geom.feature_network.layers[0].weight.detach().cpu().clone().requires_grad_()
    w2 = geom.feature_network.layers[2].weight.detach().cpu().clone().requires_grad_()
    ob = dict(batch); ob["jitter_uv"] = juv
    ref = orender.render(md, ob, dict(table=table, w1=w1, w2=w2, levels=lv, radius=1.0, n_input_dims=2), oenvs, fg, ju, jn)
    assert np.array_equal(out["_internals"]["rast"].cpu().numpy().view(np.uint32), ref["_rast"].numpy().view(np.uint32))
    rm = _rm(out["_internals"]["gbuffer"])
    assert (out["_internals"]["features"].detach().cpu()[rm] - ref["_features"].detach()).abs().max() < 1e-4
    assert (out["_internals"]["features_jitter"].detach().cpu()[rm] - ref["_features_jitter"].detach()).abs().max() < 1e-4
    assert (out0["_internals"]["features"] - out0["_internals"]["features_jitter"]).abs().max() == 0       # zero draw: same query
    for k in ["comp_rgb", "albedo", "metalness", "roughness", "specular_color", "diffuse_color"]:
        assert (out[k].detach().cpu() - ref[k].detach()).abs().max().item() < 1e-4, k
    assert abs(float(out["loss_mat_reg"]) - float(ref["loss_mat_reg"])) < 1e-5 * max(1, abs(float(ref["loss_mat_reg"])))
    dy = torch.randn(B, H, W, 3, generator=g)
    ((out["comp_rgb"] * dy.to(dev)).sum() + 2.0 * out["loss_mat_reg"]).backward()
    ((ref["comp_rgb"] * dy).sum() + 2.0 * ref["loss_mat_reg"]).backward()
    for a, b, nm in ((geom.encoding.encoding.params.grad.cpu().reshape(-1, 2), table.grad, "table"),
                     (geom.feature_network.layers[0].weight.grad.cpu(), w1.grad, "w1"),
                     (geom.feature_network.layers[2].weight.grad.cpu(), w2.grad, "w2")):
        rel = ((a - b).abs().max() / b.abs().max()).item()
        assert rel < 2e-3, (nm, rel)


@pytest.mark.parametrize("n_in,n_out,M", [(32, 5, 100000), (16, 5, 4097), (32, 8, 63), (16, 3, 1)])
def test_field_mlp_fused_vs_torch(dev, n_in, n_out, M):
    """dm_field_mlp_fwd / _bwd (Linear -> ReLU -> Linear of the feature field in registers, weight gradients on the fp32 matrix
    pipe) against torch fp32 on the CPU: outputs, input gradient, both weight gradients; ragged last batch, a single point."""
    torch.manual_seed(6)
    x = torch.randn(n_in, M)
    w1, w2 = torch.randn(64, n_in) * 0.4, torch.randn(n_out, 64) * 0.3
    dy = torch.randn(M, n_out)
    xr, w1r, w2r = x.clone().requires_grad_(), w1.clone().requires_grad_(), w2.clone().requires_grad_()
    ref = torch.relu(xr.t() @ w1r.t()) @ w2r.t()                        # [M, n_out]
    ref.backward(dy)
    xg, w1g, w2g = x.to(dev).requires_grad_(), w1.to(dev).requires_grad_(), w2.to(dev).requires_grad_()
    assert hipops.field_mlp_ok(xg, w1g, w2g)
    y = hipops.field_mlp(xg, w1g, w2g).t()                              # [M, n_out] view of the feature-major result
    assert (y.detach().cpu() - ref.detach()).abs().max() <= 1e-5 * max(1.0, ref.abs().max().item())
    y.backward(dy.to(dev))
    for got, want, nm in ((xg.grad.cpu(), xr.grad, "dx"), (w1g.grad.cpu(), w1r.grad, "dw1"), (w2g.grad.cpu(), w2r.grad, "dw2")):
        assert (got - want).abs().max() <= 2e-5 * max(1.0, want.abs().max().item()), (nm, float((got - want).abs().max()))


def test_hashgrid_full_size_levels(dev):
    """all 16 levels of dreammat.yaml incl. the hashed ones (uint32 wrap-around index arithmetic)."""
    torch.manual_seed(1)
    spec = hipops.GridSpec()
    lv, tot = ofield.grid_levels()
    table = (torch.rand(tot * 2) * 2 - 1) * 1e-1
    x = torch.rand(3000, 3) * 1.6 - 0.8
    enc = hipops.hashgrid_encode(x.to(dev), table.to(dev), spec, 1.0)
    ref = ofield.hash_encode(ofield.contract_to_unisphere(x), table.reshape(-1, 2), lv)
    assert (enc.cpu() - ref).abs().max() < 1e-4 * 1e-1 * 10


@pytest.mark.parametrize("n_levels,log2,M", [(16, 19, 200000), (12, 15, 70000), (10, 13, 66000)])
def test_hashgrid_backward_binned_route_vs_oracle_and_atomic_route(dev, monkeypatch, n_levels, log2, M):
    """dm_hashgrid_bwd_binned (hashed levels routed through table-region bins, LDS accumulation, one add per entry) against
    the oracle's autograd and against the one-atomic-per-corner route, incl. the full dreammat.yaml grid (11 hashed levels,
    32 bins each), a 2-bin and a 1-bin table, pixel-coherent points, and a bin-overflow run (capacity forced tiny)."""
    torch.manual_seed(3)
    spec = hipops.GridSpec(n_levels=n_levels, log2_hashmap_size=log2)
    lv, tot = ofield.grid_levels(n_levels=n_levels, log2_hashmap_size=log2)
    t = torch.linspace(0, 1, M)[:, None]
    x = torch.cat([-0.7 + 1.4 * t, 0.5 * torch.sin(9 * t), 0.3 * torch.cos(5 * t) + 0.01 * torch.randn(M, 1)], dim=1)
    x[: M // 3] = torch.rand(M // 3, 3) * 1.8 - 0.9
    table = torch.rand(tot * 2) * 2 - 1
    dy = torch.randn(M, 2 * n_levels)
    to = table.reshape(-1, 2).clone().requires_grad_()
    ofield.hash_encode(ofield.contract_to_unisphere(x), to, lv).backward(dy)
    grads = {}
    for route, min_pts in (("binned", 0), ("atomic", 1 << 40)):
        monkeypatch.setattr(hipops, "HASHGRID_BINNED_MIN_POINTS", min_pts)
        tg = table.to(dev).requires_grad_()
        hipops.enable_kernel_timing(True)
        hipops.hashgrid_encode(x.t().contiguous().to(dev).t(), tg, spec, 1.0).backward(dy.to(dev))
        torch.cuda.synchronize()
        assert any(k == f"hashgrid_bwd[{route}]" for k in hipops.kernel_times()), route
        hipops.enable_kernel_timing(False)
        grads[route] = tg.grad.cpu().reshape(-1, 2)
    gmax = to.grad.abs().max()
    for route, g in grads.items():
        assert (g - to.grad).abs().max() < 1e-3 * gmax, route
    assert (grads["binned"] - grads["atomic"]).abs().max() < 1e-4 * gmax


@pytest.fixture(scope="module")
def envs():
    lat = [util.synthetic_latlong(i) * 0.02 for i in range(3)]
    fg = penv.approx_fg_lut()
    oenvs = [oenv.EnvLight(l, scale=2.0, min_res=8, max_res=32) for l in lat]
    return lat, fg, oenvs


def test_shade_forward_backward(dev, envs):
    lat, fg, oenvs = envs
    atlas = penv.EnvAtlas(lat, scale=2.0, min_res=8, max_res=32, fg_lut=fg, device=dev)
    for e in range(3):     # the GPU-side prefilter agrees with the CPU oracle's
        for k in range(3):
            assert (atlas.specular[e][k].cpu() - oenvs[e].specular[k]).abs().max() <= 2e-4 * oenvs[e].specular[k].abs().max()
    torch.manual_seed(0)
    N, HW = 30000, 10000
    n = torch.nn.functional.normalize(torch.randn(N, 3), dim=-1)
    v = torch.nn.functional.normalize(n + 0.8 * torch.randn(N, 3), dim=-1)
    feat = (torch.randn(N, 5) * 1.5).requires_grad_()
    featj = (feat.detach() + 0.3 * torch.randn(N, 5)).requires_grad_()
    pix = torch.arange(N, dtype=torch.int32)
    env_of_view = torch.tensor([2, 0, 1], dtype=torch.int32)
    env = env_of_view[(pix // HW).long()].long()
    out, reg = oshade.material_forward(feat, featj, v, n, oenvs, env, fg)
    dcol = torch.randn(N, 3)
    ((out["color"] * dcol).sum() + 3.0 * reg).backward()
    from dreammat_amd._lib import MatCfgStruct
    mat = MatCfgStruct(0.0, 0.9, 0.1, 0.95)
    fgpu = feat.detach().to(dev).requires_grad_()
    fjgpu = featj.detach().to(dev).requires_grad_()
    n_dev = torch.tensor([N], dtype=torch.int32, device=dev)
    outs = hipops.shade(fgpu, n.to(dev), v.to(dev), pix.to(dev), n_dev, env_of_view.to(dev), atlas, mat, HW, True)
    regg = hipops.material_smoothness(fgpu, fjgpu, n_dev)
    oc = out["color"].detach()
    assert (outs[0].detach().cpu() - oc).abs().max() < 1e-5
    for got, key in zip(outs[1:], ["albedo", "specular_lights", "diffuse_lights", "specular_colors", "diffuse_colors",
                                   "metalness", "roughness"]):
        assert (got.cpu() - out[key].detach()).abs().max() < 1e-5, key
    assert abs(float(regg) - float(reg)) < 1e-5 * max(1.0, abs(float(reg)))
    ((outs[0] * dcol.to(dev)).sum() + 3.0 * regg).backward()
    for a, b, nm in ((fgpu.grad, feat.grad, "feat"), (fjgpu.grad, featj.grad, "featj")):
        assert (a.cpu() - b).abs().max() < 1e-3 * b.abs().max(), nm


@pytest.mark.parametrize("N", [1, 2, 3, 257, 258, 259, 515, 770])
def test_shade_small_launch_boundary_workgroup(dev, envs, N):
    """with N < grid capacity and N % 256 in [1, max(n_mips, n_views)) the boundary workgroup's
    threads past the end used to leave before they had filled their entries of the LDS mip / view->environment tables;
    the tail pixels (last view, any mip) then read uninitialised LDS.  Several views, all mips, forward + backward."""
    lat, fg, oenvs = envs
    atlas = penv.EnvAtlas(lat, scale=2.0, min_res=8, max_res=32, fg_lut=fg, device=dev)
    torch.manual_seed(N)
    HW = max(1, (N + 2) // 3)                              # the N pixels span three views; the tail is in the last one
    n = torch.nn.functional.normalize(torch.randn(N, 3), dim=-1)
    v = torch.nn.functional.normalize(n + 0.8 * torch.randn(N, 3), dim=-1)
    feat = (torch.randn(N, 5) * 2.0).requires_grad_()       # wide roughness range: every specular mip is visited
    pix = torch.arange(N, dtype=torch.int32)
    env_of_view = torch.tensor([1, 2, 0], dtype=torch.int32)
    env = env_of_view[(pix // HW).long()].long()
    out, _ = oshade.material_forward(feat, feat.detach(), v, n, oenvs, env, fg)
    dcol = torch.randn(N, 3)
    (out["color"] * dcol).sum().backward()
    from dreammat_amd._lib import MatCfgStruct
    mat = MatCfgStruct(0.0, 0.9, 0.1, 0.95)
    n_dev = torch.tensor([N], dtype=torch.int32, device=dev)
    soa = lambda t: t.detach().t().contiguous().to(dev).t()     # [C,N] storage: the layout of the kernels' fast pixel loop
    for _ in range(3):                                      # LDS garbage differs from launch to launch
        fgpu = soa(feat).requires_grad_()
        col = hipops.shade(fgpu, soa(n), soa(v), pix.to(dev), n_dev, env_of_view.to(dev), atlas, mat, HW, False)[0]
        assert (col.detach().cpu() - out["color"].detach()).abs().max() < 1e-5
        (col * dcol.to(dev)).sum().backward()
        assert (fgpu.grad.cpu() - feat.grad).abs().max() < 1e-3 * max(float(feat.grad.abs().max()), 1e-6)


@pytest.mark.parametrize("B,Cx,Cs,H,W,with_r", [(3, 640, 320, 16, 16, True), (2, 1280, 1280, 8, 8, True), (1, 320, 320, 33, 17, False),
                                                 (2, 8, 24, 5, 7, True)])
def test_cat_add_skip_connection_vs_torch(dev, B, Cx, Cs, H, W, with_r):
    """dm_cat_add_bf16: torch.cat([x, s + r], dim=1) of the UNet up blocks (ControlNet residual folded in) in one pass,
    channels-last; bit-equal to the unfused pair (both round s + r to bf16 once)."""
    torch.manual_seed(0)
    mk = lambda c: torch.randn(B, c, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    x, sk, r = mk(Cx), mk(Cs), (mk(Cs) if with_r else None)
    y = hipops.cat_add_nhwc(x, sk, r)
    ref = torch.cat([x, sk + r if with_r else sk], dim=1)
    assert y.shape == ref.shape and torch.equal(y, ref)
    assert y.permute(0, 2, 3, 1).is_contiguous()


def test_adam_matches_torch(dev):
    torch.manual_seed(0)
    n = 4096 * 3 + 8
    p0 = torch.randn(n)
    p_ref = p0.clone().requires_grad_()
    opt = torch.optim.Adam([p_ref], lr=0.01, betas=(0.9, 0.99), eps=1e-15)
    p = p0.to(dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    for step in range(1, 6):
        g = torch.randn(n) * (0.0 if step == 3 else 1.0)
        p_ref.grad = g.clone()
        opt.step()
        gg = (g * 4.0).to(dev)                      # world-size 4 sum, averaged in the kernel
        hipops.adam_step(p, gg, m, v, step, 0.01, 0.9, 0.99, 1e-15, grad_scale=0.25, zero_grad=True)
        assert float(gg.abs().max()) == 0.0
        assert (p.cpu() - p_ref.detach()).abs().max() < 2e-6, step


ATTN_CASES = [  # (B, heads, Sq, Skv, D)
    (2, 5, 256, 256, 64), (1, 5, 4096, 4096, 64), (3, 10, 1024, 1024, 64), (2, 20, 64, 64, 64),
    (2, 5, 1024, 77, 64), (2, 20, 256, 77, 64), (2, 8, 256, 256, 40), (1, 8, 192, 77, 80), (1, 8, 128, 128, 160),
    (1, 2, 100, 130, 128),
    # the one-wave-per-SIMD kernel (w64): ragged query blocks, fewer than 256 queries, odd / even tile counts, one tile
    (1, 2, 320, 256, 64), (1, 3, 128, 192, 64), (2, 2, 256, 64, 64), (1, 1, 700, 128, 64),
    # whole 512-row workgroups (the 128-rows-per-wave kernel's domain): 1 tile (prologue + last tile only), 3, 9 (one trip of
    # the unrolled ring + remainder), 16 tiles
    (1, 1, 512, 64, 64), (1, 2, 512, 192, 64), (2, 3, 512, 576, 64), (1, 2, 1024, 1024, 64)]


ATTN_VARIANTS = ["auto", "w128", "w64", "v3l", "staged"]


@pytest.fixture
def attn_variant(request):
    hipops.attention_select(request.param)
    yield request.param
    hipops.attention_select(None)


@pytest.mark.parametrize("attn_variant", ATTN_VARIANTS, indirect=True)
@pytest.mark.parametrize("B,h,Sq,Skv,D", ATTN_CASES)
def test_attention_vs_fp32_reference(dev, B, h, Sq, Skv, D, attn_variant):
    torch.manual_seed(0)
    C = h * D
    q = torch.randn(B, Sq, C); k = torch.randn(B, Skv, C); v = torch.randn(B, Skv, C)
    qb, kb, vb = (t.to(dev).bfloat16() for t in (q, k, v))
    pad = (Skv + 7) // 8 * 8
    vt = torch.zeros(B, C, pad, device=dev, dtype=torch.bfloat16)
    vt[:, :, :Skv] = vb.transpose(1, 2)
    out = hipops.attention(qb, kb, vt, h).float().cpu()
    qf, kf, vf = (t.float().cpu().view(B, -1, h, D).transpose(1, 2) for t in (qb, kb, vb))   # bf16-rounded inputs
    s = qf @ kf.transpose(-1, -2) * D ** -0.5
    ref = (torch.softmax(s, dim=-1) @ vf).transpose(1, 2).reshape(B, Sq, C)
    err = (out - ref).abs().max().item()
    if err >= 2e-2:
        _dump(f"attn_fail_{Sq}_{Skv}_{D}", out=out.numpy(), ref=ref.numpy())
    assert err < 2e-2, err                                 # bf16 P and bf16 output rounding
    assert (out - ref).abs().mean().item() < 2e-3


WGRAD_CASES = [  # (B, H, W, Cin, Cout, stride): the ControlNet's trainable 3x3 layers at 512^2 (64^2 latents) and small shapes
    (2, 64, 64, 320, 320, 1), (2, 64, 64, 320, 320, 2), (2, 32, 32, 320, 640, 1), (1, 16, 16, 1280, 1280, 1),
    (3, 8, 8, 128, 64, 1), (4, 16, 16, 64, 128, 2), (1, 16, 4, 64, 64, 1)]


@pytest.mark.parametrize("B,H,W,Cin,Cout,s", WGRAD_CASES)
def test_conv_trainable_weights_all_three_products_vs_fp32(dev, B, H, W, Cin, Cout, s):
    """hipops.conv3x3_train (the trainable convolutions of the ControlNet training loop): forward, data gradient and the
    weight gradient of dm_conv3x3_wgrad_nhwc_bf16 (transposing LDS reads, per-split fp32 partials) against fp32 conv2d
    autograd on the same bf16-rounded tensors; the weight gradient is bit-reproducible."""
    torch.manual_seed(4)
    x = torch.randn(B, H, W, Cin).to(dev).bfloat16().requires_grad_(True)
    w = (torch.randn(Cout, Cin, 3, 3) * (9 * Cin) ** -0.5).to(dev).bfloat16().requires_grad_(True)
    bias = torch.randn(Cout).to(dev).bfloat16().requires_grad_(True)
    assert hipops.conv3x3_train_ok(x, w, (s, s), (1, 1))
    y = hipops.conv3x3_train(x, w, bias, s)
    g = torch.randn_like(y)
    y.backward(g)
    x32, w32, b32 = (t.detach().float().cpu().requires_grad_(True) for t in (x, w, bias))
    ref = torch.nn.functional.conv2d(x32.permute(0, 3, 1, 2), w32, b32, stride=s, padding=1)
    ref.backward(g.float().cpu().permute(0, 3, 1, 2))
    assert (y.detach().float().cpu() - ref.detach().permute(0, 2, 3, 1)).abs().max().item() < 3e-2
    report = {}
    for name, got, want in (("dx", x.grad, x32.grad), ("dw", w.grad, w32.grad), ("db", bias.grad, b32.grad)):
        got = got.float().cpu()
        report[name] = (((got - want).norm() / want.norm()).item(), ((got - want).abs().max() / want.abs().max()).item())
    print("conv_train", (B, H, W, Cin, Cout, s), report)
    assert all(rel < 6e-3 and worst < 2e-2 for rel, worst in report.values()), report     # outputs rounded to bf16 once
    dw1 = hipops.conv3x3_wgrad(x.detach(), g, s)
    assert torch.equal(dw1, hipops.conv3x3_wgrad(x.detach(), g, s))
    assert ((dw1.cpu() - w32.grad).norm() / w32.grad.norm()).item() < 1e-4                # fp32 partials before the rounding


def test_conv2d_module_with_trainable_weights_takes_the_mfma_route(dev):
    """sd/layers.Conv2d with requires_grad weights on the device in bf16: no im2col lowering -- the autograd node is the
    three-kernel function -- and shapes outside the weight-gradient kernel's domain still lower through im2col."""
    from dreammat_amd.sd import layers
    torch.manual_seed(5)
    conv = layers.Conv2d(64, 128, 3, stride=2, padding=1).to(dev).bfloat16()
    x = torch.randn(2, 64, 16, 16, device=dev).bfloat16().requires_grad_(True)
    y = conv(x)
    assert type(y.grad_fn).__name__ == "PermuteBackward0" and type(y.grad_fn.next_functions[0][0]).__name__ == "_Conv3x3TrainBackward"
    y.float().square().mean().backward()
    c32 = torch.nn.Conv2d(64, 128, 3, stride=2, padding=1)
    c32.load_state_dict({k: v.float().cpu() for k, v in conv.state_dict().items()})
    x32 = x.detach().float().cpu().requires_grad_(True)
    c32(x32).square().mean().backward()
    for got, want in ((x.grad, x32.grad), (conv.weight.grad, c32.weight.grad), (conv.bias.grad, c32.bias.grad)):
        assert ((got.float().cpu() - want).norm() / want.norm()).item() < 2e-2
    odd = layers.Conv2d(32, 64, 3, padding=1).to(dev).bfloat16()                         # Cin = 32: im2col lowering
    yo = odd(torch.randn(1, 32, 8, 8, device=dev).bfloat16())
    assert "_Conv3x3Train" not in type(yo.grad_fn).__name__ and yo.grad_fn is not None


@pytest.mark.parametrize("B,C,H,silu", [(2, 320, 32, True), (3, 64, 8, False), (1, 1280, 8, True), (2, 128, 64, True)])
def test_groupnorm_trainable_affine_gradients_vs_fp32(dev, B, C, H, silu):
    """GroupNorm(32) [+ SiLU] with TRAINABLE weight / bias (the ControlNet copy in the training loop): dx from
    dm_groupnorm_nhwc_bwd, dgamma / dbeta from the per-workgroup channel sums of dm_groupnorm_nhwc_bwd_affine, against fp32
    autograd on the bf16-rounded inputs; bit-reproducible."""
    from dreammat_amd.sd import layers
    torch.manual_seed(6)
    norm = torch.nn.GroupNorm(32, C).to(dev).bfloat16()
    with torch.no_grad():
        norm.weight.copy_(1 + 0.3 * torch.randn(C)); norm.bias.copy_(0.2 * torch.randn(C))
    x = (torch.randn(B, C, H, H) * 1.5 + 0.3).to(dev).bfloat16().requires_grad_(True)
    g = torch.randn(B, C, H, H).to(dev).bfloat16()
    y = layers.group_norm_act(norm, x, silu)
    assert "_GroupNormAct" in type(y.grad_fn.next_functions[0][0]).__name__
    y.backward(g)
    n32 = torch.nn.GroupNorm(32, C)
    n32.load_state_dict({k: v.float().cpu() for k, v in norm.state_dict().items()})
    x32 = x.detach().float().cpu().requires_grad_(True)
    r = n32(x32)
    (torch.nn.functional.silu(r) if silu else r).backward(g.float().cpu())
    for name, got, want in (("dx", x.grad, x32.grad), ("dgamma", norm.weight.grad, n32.weight.grad), ("dbeta", norm.bias.grad, n32.bias.grad)):
        rel = ((got.float().cpu() - want).norm() / want.norm()).item()
        assert rel < 6e-3, (name, rel)
    first = (norm.weight.grad.clone(), norm.bias.grad.clone())
    norm.weight.grad = norm.bias.grad = x.grad = None
    layers.group_norm_act(norm, x, silu).backward(g)
    assert torch.equal(first[0], norm.weight.grad) and torch.equal(first[1], norm.bias.grad)


ATTN_BWD_CASES = [  # (B, heads, Sq, Skv, D): self / cross attention of the SD-2.1 and tiny nets, ragged tails, padded head sizes
    (2, 2, 256, 256, 64), (1, 5, 1024, 1024, 64), (2, 5, 1024, 77, 64), (1, 1, 100, 77, 32), (1, 2, 300, 130, 64),
    (2, 2, 64, 64, 32), (1, 2, 200, 136, 40), (1, 1, 130, 200, 80), (1, 2, 129, 65, 128)]


def _attention_grads_fp32(qb, kb, vb, dob, h):
    """out, dq, dk, dv of softmax(QK^T/sqrt(D))V in fp32 on the bf16-rounded inputs."""
    B, Sq, C = qb.shape
    D = C // h
    q, k, v = (t.float().cpu().requires_grad_(True) for t in (qb, kb, vb))
    qh, kh, vh = (t.view(B, -1, h, D).transpose(1, 2) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) * D ** -0.5
    out = (torch.softmax(s, dim=-1) @ vh).transpose(1, 2).reshape(B, Sq, C)
    out.backward(dob.float().cpu())
    lse2 = torch.logsumexp(s.detach(), dim=-1) * 1.4426950408889634        # [B, h, Sq]
    return out.detach(), q.grad, k.grad, v.grad, lse2


@pytest.mark.parametrize("B,h,Sq,Skv,D", ATTN_BWD_CASES)
def test_attention_backward_vs_fp32_autograd(dev, B, h, Sq, Skv, D):
    """dm_attention_fwd_lse_bf16 + dm_attention_bwd_bf16 (the differentiated attention of the ControlNet training loop)
    against fp32 autograd on the same bf16-rounded inputs: bf16 tolerance (P and dS are rounded to bf16 for the second
    products, like the forward's P), and bit-reproducible (no atomics)."""
    torch.manual_seed(1)
    C = h * D
    qb, kb, vb = (torch.randn(B, S, C).to(dev).bfloat16().requires_grad_(True) for S in (Sq, Skv, Skv))
    dob = torch.randn(B, Sq, C).to(dev).bfloat16()
    assert hipops.attention_train_ok(qb, kb, vb, h)
    out = hipops.attention_train(qb, kb, vb, h)
    out.backward(dob)
    ref_out, rq, rk, rv, lse2 = _attention_grads_fp32(qb.detach(), kb.detach(), vb.detach(), dob, h)
    o2, lse = hipops.attention_fwd_lse(qb.detach(), kb.detach(), vb.detach(), h, D ** -0.5)
    assert torch.equal(o2, out.detach())
    assert (lse.cpu() - lse2).abs().max().item() < 2e-3
    assert (out.detach().float().cpu() - ref_out).abs().max().item() < 2e-2
    report = {}
    for name, got, ref in (("dq", qb.grad, rq), ("dk", kb.grad, rk), ("dv", vb.grad, rv)):
        got = got.float().cpu()
        report[name] = (bool(torch.isfinite(got).all()), ((got - ref).norm() / ref.norm()).item(),
                        ((got - ref).abs().max() / ref.abs().max()).item())
    print("attn_bwd", (B, h, Sq, Skv, D), report)
    assert all(fin and rel < 1e-2 and worst < 3e-2 for fin, rel, worst in report.values()), report
    g1 = [t.grad.clone() for t in (qb, kb, vb)]
    for t in (qb, kb, vb):
        t.grad = None
    hipops.attention_train(qb, kb, vb, h).backward(dob)
    assert all(torch.equal(a, t.grad) for a, t in zip(g1, (qb, kb, vb)))


def test_attention_core_differentiated_path_uses_the_mfma_backward(dev):
    """sd/layers.attention_core under autograd on the device in bf16 (the trainable ControlNet copy): forward + backward
    run dm_attention_fwd_lse_bf16 / dm_attention_bwd_bf16 -- no S x S softmax in the autograd graph -- and the gradients
    of the inputs and of the V projection agree with the composed fp32 product."""
    from dreammat_amd.sd import layers
    torch.manual_seed(2)
    B, S, h, D = 2, 192, 2, 64
    C = h * D
    x = (torch.randn(B, S, C) * 0.5).to(dev).bfloat16()
    q, k = (torch.randn(B, S, C).to(dev).bfloat16().requires_grad_(True) for _ in range(2))
    wv = (torch.randn(C, C) * C ** -0.5).to(dev).bfloat16().requires_grad_(True)
    dout = torch.randn(B, S, C).to(dev).bfloat16()
    out = layers.attention_core(q, k, wv, None, x, h, S)
    assert type(out.grad_fn).__name__ == "_AttentionTrainBackward"
    out.backward(dout)
    q32, k32, w32 = (t.detach().float().cpu().requires_grad_(True) for t in (q, k, wv))
    ref = layers.attention_core(q32, k32, w32, None, x.float().cpu(), h, S)
    ref.backward(dout.float().cpu())
    for name, got, want in (("dq", q.grad, q32.grad), ("dk", k.grad, k32.grad), ("dWv", wv.grad, w32.grad)):
        rel = ((got.float().cpu() - want).norm() / want.norm()).item()
        assert rel < 2e-2, (name, rel)


@pytest.mark.parametrize("attn_variant", ATTN_VARIANTS, indirect=True)
def test_attention_online_softmax_rescale_branch(dev, attn_variant):
    """spiked keys late in the sequence force large running-max jumps (guide rule 26); a row whose first tile is dominated
    by a very negative score exercises the forced first-tile re-base of the v3 kernels."""
    torch.manual_seed(1)
    B, h, S, D = 1, 2, 512, 64
    q = torch.randn(B, S, h * D); k = torch.randn(B, S, h * D); v = torch.randn(B, S, h * D)
    k[:, 300] = q[:, 17] * 4.0
    k[:, 450] = q[:, 99] * 8.0
    k[:, :64] = -q[:, 5:6] * 3.0
    qb, kb, vb = (t.to(dev).bfloat16() for t in (q, k, v))
    out = hipops.attention(qb, kb, vb.transpose(1, 2).contiguous(), h).float().cpu()
    qf, kf, vf = (t.float().cpu().view(B, S, h, D).transpose(1, 2) for t in (qb, kb, vb))
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) * D ** -0.5, -1) @ vf).transpose(1, 2).reshape(B, S, h * D)
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max() < 3e-2


@pytest.mark.parametrize("attn_variant", ["w128", "w64", "v3l"], indirect=True)
def test_attention_lazy_shift_overflow_takes_the_exact_path(dev, attn_variant):
    """The w64 kernel keeps the row shift of the FIRST kv tile and re-bases lazily; a score that outgrows it by more than
    2^100 inside one tile cannot be represented and must send the workgroup through its exact (textbook online softmax) path.
    Rows next to it (same wave, other waves of the workgroup, other workgroups) must be unaffected."""
    torch.manual_seed(2)
    B, h, S, D = 1, 2, 512, 64
    q = torch.randn(B, S, h * D); k = torch.randn(B, S, h * D); v = torch.randn(B, S, h * D)
    q[:, 37, :D] = 6.0                       # head 0, row 37: logit 64*36/8 = 288 nats = 415 in the log2 domain against key 200
    k[:, 200, :D] = 6.0
    q[:, 300, D:] = -5.0                     # head 1, row 300 (second workgroup): 200 nats against key 511 (the last tile)
    k[:, 511, D:] = -5.0
    qb, kb, vb = (t.to(dev).bfloat16() for t in (q, k, v))
    out = hipops.attention(qb, kb, vb.transpose(1, 2).contiguous(), h).float().cpu()
    qf, kf, vf = (t.float().cpu().view(B, S, h, D).transpose(1, 2) for t in (qb, kb, vb))
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) * D ** -0.5, -1) @ vf).transpose(1, 2).reshape(B, S, h * D)
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max() < 3e-2
    assert (out[0, 37, :D] - vf[0, 0, 200]).abs().max() < 2e-2        # that row is its spiked key's value


@pytest.mark.parametrize("attn_variant", ["w128", "w64", "v3l"], indirect=True)
def test_attention_cfg5_sequence_16384_vs_chunked_fp32(dev, attn_variant):
    """BASELINE configs[4] (1024^2 renders -> 128^2 latents): S = 16384, 5 heads of 64 -- 256 kv tiles per row, the longest
    accumulation the kernels see.  Reference: fp32 softmax(QK^T)V on the bf16-rounded inputs, query chunks of 2048
    (1 GB of scores per chunk), plus a float64 CPU spot check of a few rows."""
    torch.manual_seed(4)
    B, h, S, D = 1, 5, 16384, 64
    q = torch.randn(B, S, h * D, device=dev).bfloat16()
    k = torch.randn(B, S, h * D, device=dev).bfloat16()
    v = torch.randn(B, S, h * D, device=dev).bfloat16()
    out = hipops.attention(q, k, v.transpose(1, 2).contiguous(), h).float()
    qf, kf, vf = (t.float().view(B, S, h, D).transpose(1, 2) for t in (q, k, v))
    ref = torch.empty(B, h, S, D, device=dev)
    for c0 in range(0, S, 2048):
        sc = qf[:, :, c0:c0 + 2048] @ kf.transpose(-1, -2) * D ** -0.5
        ref[:, :, c0:c0 + 2048] = torch.softmax(sc, -1) @ vf
    ref = ref.transpose(1, 2).reshape(B, S, h * D)
    err = (out - ref).abs()
    assert err.max().item() < 1e-2 and err.mean().item() < 1e-3, (err.max().item(), err.mean().item())
    rows = [0, 4095, 9001, 16383]
    q64, k64, v64 = qf.double().cpu(), kf.double().cpu(), vf.double().cpu()
    r64 = (torch.softmax(q64[:, :, rows] @ k64.transpose(-1, -2) * D ** -0.5, -1) @ v64).transpose(1, 2).reshape(B, len(rows), h * D)
    assert (out[:, rows].double().cpu() - r64).abs().max().item() < 1e-2
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, f"attn_s16384_{attn_variant}.json"), "w") as f:
        json.dump({"variant": attn_variant, "B": B, "heads": h, "S": S, "D": D, "max_abs_err_vs_fp32": err.max().item(),
                   "mean_abs_err_vs_fp32": err.mean().item()}, f)


def test_renderer_end_to_end_vs_oracle(dev, envs):
    """RaytraceRender.forward (plugin API) against oracle/render.py: all 12 outputs + the gradients of
    a random loss wrt hash table and MLP weights."""
    import dreammat_amd
    from dreammat_amd.geometry import DreamMatMesh
    from dreammat_amd.material import DreamMatMaterial
    from dreammat_amd.renderer import RaytraceRender
    from dreammat_amd.background import SolidColorBackground
    lat, fg, oenvs = envs
    torch.manual_seed(0)
    enc_cfg = {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 14,
               "base_resolution": 16, "per_level_scale": 1.447269237440378}
    geom = DreamMatMesh({"shape_init": "sphere:48:40", "shape_init_params": 0.7, "pos_encoding_config": enc_cfg}).to(dev)
    with torch.no_grad():
        geom.encoding.encoding.params.copy_((torch.rand_like(geom.encoding.encoding.params) * 2 - 1))
        geom.feature_network.layers[0].weight.copy_(torch.randn(64, 16) * 0.5)
        geom.feature_network.layers[2].weight.copy_(torch.randn(5, 64) * 0.3)
    mat = DreamMatMaterial({"use_raytracing": False, "environment_scale": 2.0, "env_max_res": 32, "env_min_res": 8,
                            "n_envs": 3}, latlongs=lat).to(dev)
    rend = RaytraceRender({}, geometry=geom, material=mat, background=SolidColorBackground({}))
    B, H, W = 3, 128, 128
    batch = util.make_views(B, H, W, seed=2)
    batch["env_id"] = torch.tensor([1, 2, 0])
    g = torch.Generator().manual_seed(9)
    ju, jn = torch.rand(B, H, W, generator=g), torch.randn(B, H, W, generator=g)
    gbatch = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    out = rend(**gbatch, light_positions=None, jitter_u=ju.to(dev), jitter_n=jn.to(dev), check_overflow=True)
    # oracle
    m = geom.mesh
    md = dict(v_pos=m.v_pos.cpu().numpy(), v_nrm=m.v_nrm.cpu().numpy(), t_pos_idx=m.t_pos_idx.cpu().numpy().astype(np.int32))
    md["opp"] = oraster.build_topology(md["t_pos_idx"])
    lv, tot = ofield.grid_levels(n_levels=8, log2_hashmap_size=14)
    table = geom.encoding.encoding.params.detach().cpu().reshape(-1, 2).clone().requires_grad_()
    w1 = geom.feature_network.layers[0].weight.detach().cpu().clone().requires_grad_()
    w2 = geom.feature_network.layers[2].weight.detach().cpu().clone().requires_grad_()
    ref = orender.render(md, batch, dict(table=table, w1=w1, w2=w2, levels=lv, radius=1.0), oenvs, fg, ju, jn)
    assert np.array_equal(out["_internals"]["rast"].cpu().numpy().view(np.uint32), ref["_rast"].numpy().view(np.uint32))
    tol = {"comp_rgb": 1e-4, "opacity": 1e-5, "comp_depth": 1e-4, "comp_normal": 1e-5}
    for k in ["comp_rgb", "opacity", "comp_depth", "comp_normal", "albedo", "metalness", "roughness",
              "specular_light", "diffuse_light", "specular_color", "diffuse_color"]:
        err = (out[k].detach().cpu() - ref[k].detach()).abs().max().item()
        assert err < tol.get(k, 1e-4), (k, err)
    assert abs(float(out["loss_mat_reg"]) - float(ref["loss_mat_reg"])) < 1e-5 * max(1, abs(float(ref["loss_mat_reg"])))
    mse = ((out["comp_rgb"].detach().cpu() - ref["comp_rgb"].detach()) ** 2).mean().item()
    psnr = 10 * np.log10(1.0 / max(mse, 1e-20))
    assert psnr > 80, psnr
    dy = torch.randn(B, H, W, 3, generator=g)
    ((out["comp_rgb"] * dy.to(dev)).sum() + 2.0 * out["loss_mat_reg"]).backward()
    ((ref["comp_rgb"] * dy).sum() + 2.0 * ref["loss_mat_reg"]).backward()
    for a, b, nm in ((geom.encoding.encoding.params.grad.cpu().reshape(-1, 2), table.grad, "table"),
                     (geom.feature_network.layers[0].weight.grad.cpu(), w1.grad, "w1"),
                     (geom.feature_network.layers[2].weight.grad.cpu(), w2.grad, "w2")):
        rel = ((a - b).abs().max() / b.abs().max()).item()
        assert rel < 1e-3, (nm, rel)
    with open(os.path.join(OUT, "render_parity.json"), "w") as fh:
        json.dump({"psnr_db": psnr, "coverage_ids_equal": True}, fh)


def test_net_prologue_batched_projections_and_bank_gathers(dev, monkeypatch):
    """layers.NetPrologue: the time-embedding projections of all ResnetBlock2D (one batched product per width) and this step's rows
    of the cross-attention K / V^T banks (one gather per width) against the per-block path, same weights, same bank context:
    the gathered rows are bit-equal, the projections differ by GEMM summation order only; and far fewer launches."""
    from dreammat_amd.sd import ARCHS, ControlNetModel, UNet2DConditionModel, layers
    a = ARCHS["tiny"]
    torch.manual_seed(0)
    unet = UNet2DConditionModel(a).eval()
    cn = ControlNetModel.from_unet(unet).eval()
    for conv in list(cn.controlnet_down_blocks) + [cn.controlnet_mid_block, cn.controlnet_cond_embedding.conv_out]:
        torch.nn.init.normal_(conv.weight, std=0.05)
    unet.to(dev).bfloat16().requires_grad_(False); cn.to(dev).bfloat16().requires_grad_(False)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(6, 4, 32, 32, generator=g).to(dev).bfloat16(); t = torch.tensor([37, 801, 500, 37, 801, 500]).to(dev)
    bank = torch.randn(5, 77, a.cross_dim, generator=g).to(dev).bfloat16()
    ids = torch.tensor([4, 0, 2, 2, 1, 0]).to(dev)
    cond = torch.rand(6, 22, 256, 256, generator=g).to(dev).bfloat16()

    def run():
        ctx = layers.PaddedContext(bank[ids], bank, ids)
        d, m = cn(x, t, ctx, cond, 1.0)
        return unet(x, t, ctx, d, m).float().cpu()

    with torch.no_grad():
        monkeypatch.setattr(layers.NetPrologue, "usable", staticmethod(lambda t_: False))
        run()                                                   # builds the per-layer bank entries
        ref = run()
        monkeypatch.undo()
        from torch.profiler import ProfilerActivity, profile
        run()                                                   # groups the entries
        with profile(activities=[ProfilerActivity.CPU]) as prof:
            y = run()
        names = [e.key for e in prof.key_averages() for _ in range(e.count)]
    assert sum(n == "aten::index_select" for n in names) <= 12, sum(n == "aten::index_select" for n in names)  # K and V^T x 3 widths x 2 nets (per layer: 46)
    assert sum(n == "aten::silu" for n in names) <= 10                    # one per net + the time-embedding MLPs / stem layers (per block: +34)
    rel = ((y - ref).abs().max() / ref.abs().max()).item()
    assert rel < 1e-2, rel
    assert not any("_tproj" in m.__dict__ for m in list(unet.modules()) + list(cn.modules()))      # every projection was consumed


def test_unet_controlnet_gpu_fp32_and_bf16_hip_attention(dev):
    """fp32 on the GPU vs the CPU functional oracle (north_star: noise-pred within 1e-3 rel), and the
    bf16 + MFMA-attention production path vs that fp32 result (bf16 rounding only)."""
    from dreammat_amd.sd import ARCHS, ControlNetModel, UNet2DConditionModel
    from oracle import sd_nets as osd
    for arch_name in ("tiny", "tiny15"):
        a = ARCHS[arch_name]
        torch.manual_seed(0)
        unet = UNet2DConditionModel(a).eval()
        cn = ControlNetModel.from_unet(unet).eval()
        for conv in list(cn.controlnet_down_blocks) + [cn.controlnet_mid_block, cn.controlnet_cond_embedding.conv_out]:
            torch.nn.init.normal_(conv.weight, std=0.05)
        g = torch.Generator().manual_seed(1)
        x = torch.randn(3, 4, 32, 32, generator=g); t = torch.tensor([37, 801, 500])
        ctx = torch.randn(3, 77, a.cross_dim, generator=g); cond = torch.rand(3, 22, 256, 256, generator=g)
        with torch.no_grad():
            od, om = osd.controlnet_forward(cn.state_dict(), x, t, ctx, cond, 1.0, a.heads, a.use_linear_projection)
            oy = osd.unet_forward(unet.state_dict(), x, t, ctx, a.heads, a.use_linear_projection, od, om)
            unet.to(dev); cn.to(dev)
            d, m = cn(x.to(dev), t.to(dev), ctx.to(dev), cond.to(dev), 1.0)
            y = unet(x.to(dev), t.to(dev), ctx.to(dev), d, m).cpu()
            assert (y - oy).abs().max() <= 1e-3 * oy.abs().max(), arch_name
            unet.bfloat16(); cn.bfloat16()
            hipops.enable_kernel_timing(True)
            d, m = cn(x.to(dev).bfloat16(), t.to(dev), ctx.to(dev).bfloat16(), cond.to(dev).bfloat16(), 1.0)
            yb = unet(x.to(dev).bfloat16(), t.to(dev), ctx.to(dev).bfloat16(), d, m).float().cpu()
            torch.cuda.synchronize()
            kt = hipops.kernel_times()
            n_attn = sum(v["launches"] for k, v in kt.items() if k.startswith("attention"))
            n_ln = sum(v["launches"] for k, v in kt.items() if k.startswith("layernorm"))
            n_gg = sum(v["launches"] for k, v in kt.items() if k.startswith(("geglu", "gemm+geglu")))
            hipops.enable_kernel_timing(False)
        assert n_attn == 46, n_attn                    # 2x(16 UNet + 7 ControlNet) attention launches, all on MFMA
        assert n_ln == 69 and n_gg == 23, (n_ln, n_gg)  # 3 LayerNorms + 1 GEGLU (own kernel or GEMM epilogue) per block
        rel = ((yb - oy).abs().max() / oy.abs().max()).item()
        rel_mean = ((yb - oy).abs().mean() / oy.abs().mean()).item()
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, f"tiny_eps_parity_{arch_name}.json"), "w") as fh:
            json.dump({"arch": arch_name, "bf16_rel_max": rel, "bf16_rel_mean": rel_mean}, fh)
        # bf16 storage of every activation through ~60 layers: mean-rel 1.2e-2 (tiny) / 1.4e-2 (tiny15) whatever kernel computes the
        # stem layers; the MAXIMUM over the 12 k outputs is heavy-tailed and moves with the summation order of a single layer
        # (tiny15: 1.94e-2 with the direct stem kernel, 2.37e-2 with the patch kernel, same mean to 3 digits) -- so the mean carries
        # the tight gate (a wrong scale, a dropped residual, a kernel falling back to garbage moves it), the maximum a loose one
        assert rel < 4e-2 and rel_mean < 2e-2, (arch_name, rel, rel_mean)


def test_noise_prediction_hip_graph_replay_equals_eager(dev, tmp_path, monkeypatch):
    """The ControlNet + UNet noise prediction replayed from one captured hipGraph (guidance `hip_graph`) must give the
    eager result bit for bit, on fresh inputs of later steps too (inputs are copied into the capture's static buffers)."""
    monkeypatch.chdir(tmp_path)
    from dreammat_amd.guidance import StableDiffusionLightGuidance
    from dreammat_amd.prompt import StableDiffusionPromptProcessor
    cfg = {"pretrained_model_name_or_path": "tiny", "use_controlnet": True, "control_types": ["light"],
           "condition_scales": [0.9], "width": 128, "height": 128, "cond_scale": 1.05, "uncond_scale": -0.75,
           "null_scale": -0.25}
    torch.manual_seed(0)
    ge = StableDiffusionLightGuidance(dict(cfg, hip_graph=False))      # both: the same seeded synthetic weights, on the GPU
    gg = StableDiffusionLightGuidance(dict(cfg, hip_graph=True))
    assert next(ge.unet.parameters()).is_cuda and next(ge.unet.parameters()).dtype == torch.float16      # (the default: the reference's half_precision_weights type)
    pp = StableDiffusionPromptProcessor({"prompt": "a wooden chair", "negative_prompt": "ugly",
                                         "pretrained_model_name_or_path": "tiny"})
    B = 2
    elev, azim, dist = torch.tensor([10.0, 70.0], device=dev), torch.tensor([5.0, 170.0], device=dev), torch.tensor([3.5, 3.2], device=dev)
    for step in range(3):                                   # step 0 captures, steps 1-2 replay with new inputs
        g = torch.Generator().manual_seed(10 + step)
        rgb = torch.rand(B, 128, 128, 3, generator=g).to(dev)
        cond = torch.rand(B, 128, 128, 22, generator=g).to(dev)
        rng = {"t": torch.tensor([400 - 100 * step, 77 + step], device=dev), "noise": torch.randn(B, 4, 16, 16, generator=g).to(dev),
               "posterior_noise": torch.randn(B, 4, 16, 16, generator=g).to(dev)}
        oe = ge(rgb, pp(), elev, azim, dist, condition_map=cond, rng=rng)
        og = gg(rgb, pp(), elev, azim, dist, condition_map=cond, rng=rng)
        for k in ("e_text", "e_uncond", "e_null", "grad"):
            assert torch.equal(ge._last[k], gg._last[k]), (step, k, (ge._last[k] - gg._last[k]).abs().max().item())
        assert float(oe["loss_sds"]) == float(og["loss_sds"])
    assert len(gg._graphs) == 1 and not hasattr(ge, "_graphs")


def test_full_sds_step_vs_oracle(dev, tmp_path, monkeypatch):
    """BASELINE config 1 shape (2-triangle quad, 1 view, constant env, fp32 nets): one complete optimizer
    step through the plugin API vs the oracle's step: loss, parameter gradients, Adam-updated parameters."""
    monkeypatch.chdir(tmp_path)
    import dreammat_amd
    from dreammat_amd.data import RandomCameraDataModule
    from dreammat_amd.system import Trainer, to_device
    from oracle import sd_nets as osd
    dreammat_amd._import_plugins()
    torch.manual_seed(0)
    H = W = 256
    lat = [torch.full((16, 32, 3), 0.25) for _ in range(5)]
    enc = {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 14,
           "base_resolution": 16, "per_level_scale": 1.447269237440378}
    cfg = {"geometry": {"shape_init": "quad", "shape_init_params": 1.0, "pos_encoding_config": enc},
           "material": {"use_raytracing": False, "environment_scale": 2.0, "env_max_res": 32, "env_min_res": 8},
           "guidance": {"use_controlnet": True, "control_types": ["light"], "condition_scales": [1.0], "width": W,
                        "height": H, "pretrained_model_name_or_path": "tiny15", "half_precision_weights": False,
                        "cond_scale": 1.05, "uncond_scale": -0.6, "null_scale": -0.3},
           "prompt_processor": {"prompt": "a checkered tile", "negative_prompt": "ugly",
                                "pretrained_model_name_or_path": "tiny15"},
           "loss": {"lambda_sds": 1.0, "lambda_mat_reg": 1.0},
           "optimizer": {"name": "Adam", "args": {"lr": 0.01, "betas": [0.9, 0.99], "eps": 1e-15}}}
    system = dreammat_amd.find("dreammat-system")(cfg, material_kwargs={"latlongs": lat})
    with torch.no_grad():
        system.geometry.encoding.encoding.params.uniform_(-1, 1)
        system.geometry.feature_network.layers[0].weight.normal_(0, 0.5)
        system.geometry.feature_network.layers[2].weight.normal_(0, 0.3)
    dm = RandomCameraDataModule(cfg={"height": H, "width": W, "batch_size": 1, "use_fix_views": True,
                                     "camera_distance_range": [3.0, 3.5], "fovy_range": [30, 40], "camera_perturb": 0.0,
                                     "center_perturb": 0.0, "up_perturb": 0.0, "elevation_range": [50, 80]}, device=dev)
    dm.setup("fit")
    system.on_fit_start()
    system.configure_optimizers()
    trainer = Trainer(system, dm, max_steps=1)
    batch = to_device(dm.train_dataset.collate(), dev)
    g = torch.Generator().manual_seed(1)
    ju, jn = torch.rand(1, H, W, generator=g), torch.randn(1, H, W, generator=g)
    batch["jitter_u"], batch["jitter_n"] = ju.to(dev), jn.to(dev)
    rng = {"t": torch.tensor([400]), "noise": torch.randn(1, 4, H // 8, W // 8, generator=g),
           "posterior_noise": torch.randn(1, 4, H // 8, W // 8, generator=g)}
    rng_d = {k: v.to(dev) for k, v in rng.items()}
    # ---- oracle step
    geo, gd = system.geometry, None
    m = geo.mesh
    md = dict(v_pos=m.v_pos.cpu().numpy(), v_nrm=m.v_nrm.cpu().numpy(), t_pos_idx=m.t_pos_idx.cpu().numpy().astype(np.int32))
    md["opp"] = oraster.build_topology(md["t_pos_idx"])
    lv, _ = ofield.grid_levels(n_levels=8, log2_hashmap_size=14)
    table = geo.encoding.encoding.params.detach().cpu().reshape(-1, 2).clone().requires_grad_()
    w1 = geo.feature_network.layers[0].weight.detach().cpu().clone().requires_grad_()
    w2 = geo.feature_network.layers[2].weight.detach().cpu().clone().requires_grad_()
    opt = torch.optim.Adam([table, w1, w2], lr=0.01, betas=(0.9, 0.99), eps=1e-15)
    cb = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    oenvs = [oenv.EnvLight(l, scale=2.0, min_res=8, max_res=32) for l in lat]
    ref = orender.render(md, cb, dict(table=table, w1=w1, w2=w2, levels=lv, radius=1.0), oenvs,
                         system.material.atlas.fg_lut.cpu(), ju, jn)
    gd = system.guidance
    assert gd.weights_dtype == torch.float32
    nets = {"vae": {k: v.cpu() for k, v in gd.vae.state_dict().items()},
            "unet": {k: v.cpu() for k, v in gd.unet.state_dict().items()},
            "controlnet": {k: v.cpu() for k, v in gd.controlnets[0].state_dict().items()}}
    emb = system.prompt_processor().get_text_embeddings(cb["elevation"].to(dev), cb["azimuth"].to(dev),
                                                        cb["camera_distances"].to(dev), True, True).cpu()
    a = gd.arch
    loss_sds, grad, eps, _ = osd.sds_loss(ref["comp_rgb"], nets, emb, cb["condition_map"], rng["t"], rng["noise"],
                                          rng["posterior_noise"], (1.05, -0.6, -0.3, 0.0), a.heads,
                                          a.use_linear_projection, 1.0)
    loss_ref = loss_sds + ref["loss_mat_reg"]
    loss_ref.backward()
    g_ref = [t.grad.clone() for t in (table, w1, w2)]
    opt.step()
    # ---- product step
    system.do_update()
    loss, logs = system.training_step(batch, rng=rng_d)
    loss.backward()
    geo_p = [geo.encoding.encoding.params, geo.feature_network.layers[0].weight, geo.feature_network.layers[2].weight]
    assert abs(float(loss) - float(loss_ref)) <= 1e-3 * abs(float(loss_ref)), (float(loss), float(loss_ref))
    e3 = torch.cat([gd._last["e_text"], gd._last["e_uncond"], gd._last["e_null"]]).cpu()
    assert (e3 - eps).abs().max() <= 1e-3 * eps.abs().max()
    for p, r, nm in zip(geo_p, g_ref, ("table", "w1", "w2")):
        rel = ((p.grad.cpu().reshape(r.shape) - r).abs().max() / r.abs().max()).item()
        assert rel < 2e-3, (nm, rel)
    system.optimizer.step(1)
    for p, r, nm in zip(geo_p, (table, w1, w2), ("table", "w1", "w2")):
        # Adam with eps=1e-15 turns any non-zero gradient into a +-lr step: compare where |g| is not tiny
        gr = g_ref[("table", "w1", "w2").index(nm)]
        gmask = gr.abs() > 1e-2 * gr.abs().max()          # sign of the gradient is certain there
        diff = (p.detach().cpu().reshape(r.shape) - r.detach()).abs()
        assert diff[gmask].max() < 1e-4, (nm, float(diff[gmask].max()))
    assert float(system.flat.grad.abs().max()) == 0.0       # zeroed by the fused kernel for the next step


CONV_CASES = [  # (B, Cin, Cout, H, W, stride)
    (2, 64, 128, 16, 16, 1), (3, 320, 320, 32, 32, 1), (1, 32, 64, 10, 10, 1), (2, 128, 256, 17, 23, 1),
    (2, 320, 320, 32, 32, 2), (1, 640, 1280, 8, 8, 1), (1, 96, 192, 9, 9, 2)]


@pytest.mark.parametrize("B,Cin,Cout,H,W,stride", CONV_CASES)
def test_conv3x3_mfma_vs_fp32_reference(dev, B, Cin, Cout, H, W, stride):
    from dreammat_amd.sd import layers
    torch.manual_seed(0)
    conv = layers.Conv2d(Cin, Cout, 3, stride=stride, padding=1)
    x = torch.randn(B, Cin, H, W)
    xb = x.bfloat16()
    conv_b = layers.Conv2d(Cin, Cout, 3, stride=stride, padding=1)
    conv_b.load_state_dict(conv.state_dict())
    conv_b.to(dev, torch.bfloat16)
    for p in conv_b.parameters():
        p.requires_grad_(False)
    wref = conv_b.weight.float().cpu(); bref = conv_b.bias.float().cpu()
    with_grad = stride == 1 and Cin % 64 == 0          # the data-gradient kernel needs Cin % 64 == 0
    xg = xb.to(dev).requires_grad_(with_grad)
    assert layers.CONV_BACKEND == "mfma"
    hipops.enable_kernel_timing(True)
    y = conv_b(xg)
    torch.cuda.synchronize()
    assert any(k.startswith("conv3x3") for k in hipops.kernel_times()), "MFMA conv kernel was not used"
    hipops.enable_kernel_timing(False)
    xr = xb.float().requires_grad_(True)
    ref = torch.nn.functional.conv2d(xr, wref, bref, stride=stride, padding=1)
    assert y.shape == ref.shape
    err = (y.float().cpu() - ref).abs().max().item()
    assert err < 2e-2 * ref.abs().max().item() + 1e-2, err          # bf16 output rounding
    if with_grad:
        dy = torch.randn_like(ref)
        y.backward(dy.to(dev).bfloat16())
        ref.backward(dy.bfloat16().float())
        gerr = (xg.grad.float().cpu() - xr.grad).abs().max().item()
        assert gerr < 2e-2 * xr.grad.abs().max().item() + 1e-2, gerr


def test_hashgrid_backward_coherent_points(dev):
    """pixel-coherent sample points (long runs of identical cells per wave) exercise the run-combining
    backward; result must equal the oracle's autograd."""
    torch.manual_seed(2)
    spec = hipops.GridSpec(n_levels=12, log2_hashmap_size=15)
    lv, tot = ofield.grid_levels(n_levels=12, log2_hashmap_size=15)
    t = torch.linspace(0, 1, 6000)[:, None]
    x = torch.cat([-0.6 + 1.2 * t, 0.3 * torch.sin(6 * t), 0.1 + 0.002 * torch.randn(6000, 1)], dim=1)
    table = torch.rand(tot * 2) * 2 - 1
    tg = table.to(dev).requires_grad_()
    enc = hipops.hashgrid_encode(x.t().contiguous().to(dev).t(), tg, spec, 1.0)
    to = table.reshape(-1, 2).clone().requires_grad_()
    ref = ofield.hash_encode(ofield.contract_to_unisphere(x), to, lv)
    dy = torch.randn(6000, 24)
    enc.backward(dy.to(dev))
    ref.backward(dy)
    err = (tg.grad.cpu().reshape(-1, 2) - to.grad).abs().max()
    assert err < 1e-3 * to.grad.abs().max(), float(err)


@pytest.mark.parametrize("B,C,H,W,act", [(2, 64, 8, 8, 1), (3, 320, 16, 16, 1), (2, 128, 33, 17, 0), (1, 2560, 4, 4, 1),
                                         (2, 512, 32, 32, 1)])
def test_groupnorm_silu_nhwc_vs_fp32_reference(dev, B, C, H, W, act):
    torch.manual_seed(0)
    x = (torch.randn(B, C, H, W) * 2 + 0.5).bfloat16()
    gn = torch.nn.GroupNorm(32, C, eps=1e-5)
    with torch.no_grad():
        gn.weight.normal_(1.0, 0.3); gn.bias.normal_(0, 0.3)
    gamma, beta = gn.weight.detach().bfloat16(), gn.bias.detach().bfloat16()
    xr = x.float().requires_grad_()
    ref = torch.nn.functional.group_norm(xr, 32, gamma.float(), beta.float(), 1e-5)
    if act:
        ref = torch.nn.functional.silu(ref)
    xg = x.to(dev).permute(0, 2, 3, 1).contiguous().requires_grad_()
    y = hipops.groupnorm_nhwc(xg, gamma.to(dev), beta.to(dev), 1e-5, act)
    got = y.permute(0, 3, 1, 2).float().cpu()
    assert (got - ref).abs().max() < 3e-2 * max(1.0, ref.abs().max().item())
    with torch.no_grad():       # inference entry (coefficients formed inside the apply kernel): same arithmetic as the 3-kernel path
        yi = hipops.groupnorm_nhwc(xg.detach(), gamma.to(dev), beta.to(dev), 1e-5, act)
    assert (yi.float() - y.detach().float()).abs().max().item() <= 2 ** -6 * max(1.0, y.detach().float().abs().max().item())   # <= 1 bf16 ulp
    dy = torch.randn_like(ref).bfloat16()
    y.backward(dy.to(dev).permute(0, 2, 3, 1).contiguous())
    ref.backward(dy.float())
    gerr = (xg.grad.permute(0, 3, 1, 2).float().cpu() - xr.grad).abs().max().item()
    assert gerr < 3e-2 * max(1.0, xr.grad.abs().max().item()), gerr


@pytest.mark.parametrize("tile", ["128", "256", "512", "320", "640"])
@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 320, 320, 32, 32), (1, 128, 128, 40, 24), (3, 64, 192, 16, 16),
                                            (1, 64, 640, 24, 24)])
def test_conv3x3_dma_tile_variants(dev, monkeypatch, tile, B, Cin, Cout, H, W):
    """every tile shape of the LDS-DMA conv kernel (128x{64,128}, 256x128, 256x256 / 256x320 / 512x128 [2-stage ring]), incl.
    ragged last Cout tiles (320 = 2.5 x 128, 192 = 1.5 x 128, 640 = 2.5 x 256 = 2 x 320) and ragged M."""
    monkeypatch.setenv("DREAMMAT_CONV_TILE", tile)
    torch.manual_seed(1)
    x = torch.randn(B, H, W, Cin).bfloat16()
    w = (torch.randn(Cout, Cin, 3, 3) * 0.05).bfloat16()
    bias = torch.randn(Cout).bfloat16()
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    y = hipops.conv3x3_nhwc(x.to(dev), wt.to(dev), bias.to(dev)).float().cpu()
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias.float(), padding=1).permute(0, 2, 3, 1)
    err = (y - ref).abs().max().item()
    assert err < 2e-2 * ref.abs().max().item() + 1e-2, err


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("halo,B,Cin,Cout,H,W", [
    ("24", 2, 128, 128, 200, 72),       # 384 x 128 patches: ragged last band (200 = 8 x 24 + 8) and columns (72 = 4.5 x 16)
    ("16", 2, 128, 128, 200, 72),       # 256 x 256 patches on the same problem: Cout below a channel tile
    ("24", 3, 256, 512, 40, 24),        # several channel tiles of 128
    ("16", 3, 256, 512, 40, 24),
    ("24", 5, 64, 320, 8, 8),           # images smaller than a patch; 320 = 2.5 x 128 (ragged channel tile)
    ("16", 1, 192, 64, 17, 33),         # odd sizes, three channel blocks
    ("128", 3, 256, 512, 40, 24),       # 256 x 128 patches, three weight stages
    ("128", 5, 64, 320, 8, 8),
    (None, 24, 1280, 1280, 16, 16),     # ... the dispatcher's choice for the 1280-channel layers at 16 x 16 (8 views x 3 branches)
    (None, 2, 128, 128, 256, 512),      # the dispatcher's own choice: 384 x 128 patches (the VAE's 512^2 layers' variant)
    (None, 8, 256, 256, 96, 96),        # ... 256 x 256 patches
])
def test_conv3x3_halo_patch_kernel(dev, monkeypatch, dtype, halo, B, Cin, Cout, H, W):
    """k_conv3x3_halo (round 6): the input patch of a 64-channel block requested once and read at nine shifted offsets -- against
    fp32 conv2d with bias, per-image channel bias and residual, both patch variants forced on ragged problems, and against the
    per-tap kernel (same products, taps summed in another order: agreement to the output's rounding)."""
    if halo:
        monkeypatch.setenv("DREAMMAT_CONV_HALO", halo)
    torch.manual_seed(21)
    x = torch.randn(B, H, W, Cin).to(dtype)
    w = (torch.randn(Cout, Cin, 3, 3) * 0.05).to(dtype)
    bias, rowbias, res = torch.randn(Cout).to(dtype), torch.randn(B, Cout).to(dtype), torch.randn(B, H, W, Cout).to(dtype)
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    hipops.enable_kernel_timing(True)
    yd = hipops.conv3x3_nhwc(x.to(dev), wt.to(dev), bias.to(dev), 1, (1, 1), None, rowbias.to(dev), res.to(dev))
    torch.cuda.synchronize()
    hipops.enable_kernel_timing(False)
    assert int(_lib.lib().dm_conv3x3_gn_ok(B, H, W, Cin, Cout)) == 1          # the patch kernel took it
    if halo is None and H == 16 and Cout == 1280:      # ... but not with 3 images (30 items for 256 CUs, no split-K): the per-tap kernel
        assert int(_lib.lib().dm_conv3x3_gn_ok(3, H, W, Cin, Cout)) == 0
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias.float(), padding=1).permute(0, 2, 3, 1)
    ref = ref + rowbias.float()[:, None, None, :] + res.float()
    tol = (2e-2 if dtype == torch.bfloat16 else 3e-3) * ref.abs().max().item() + (1e-2 if dtype == torch.bfloat16 else 2e-3)
    assert (yd.float().cpu() - ref).abs().max().item() < tol
    monkeypatch.setenv("DREAMMAT_CONV_HALO", "0")
    yt = hipops.conv3x3_nhwc(x.to(dev), wt.to(dev), bias.to(dev), 1, (1, 1), None, rowbias.to(dev), res.to(dev))
    ulp = 2.0 ** (-7 if dtype == torch.bfloat16 else -10)
    assert (yd.float() - yt.float()).abs().max().item() <= 2 * ulp * ref.abs().max().item()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("halo,act,B,Cin,Cout,H,W", [
    ("24", 1, 2, 128, 128, 200, 72), ("16", 0, 3, 256, 512, 40, 24), ("24", 1, 5, 64, 320, 8, 8), ("16", 1, 1, 192, 64, 17, 33),
    ("128", 1, 3, 256, 512, 40, 24), (None, 1, 24, 1280, 1280, 16, 16),
    (None, 1, 2, 128, 128, 256, 512), (None, 1, 8, 256, 256, 96, 96)])
def test_conv3x3_with_groupnorm_apply_folded_in(dev, monkeypatch, dtype, halo, act, B, Cin, Cout, H, W):
    """dm_conv3x3_gn_nhwc_*_fused (ABI v13): conv(act(GroupNorm32(x))) with the apply pass done on the patch in LDS -- equal to the
    apply kernel + the same convolution kernel BIT FOR BIT (the patch holds exactly the 16-bit values the apply pass would have
    stored, out-of-image pixels stay zero), and within rounding of fp32 torch."""
    if halo:
        monkeypatch.setenv("DREAMMAT_CONV_HALO", halo)
    monkeypatch.setattr(hipops, "GN_FOLD_MIN_BYTES", 0)           # (the product folds from 192 MB on: where it pays)
    torch.manual_seed(22)
    x = (torch.randn(B, H, W, Cin) * 2 + 0.5).to(dtype)
    gm, bt = (torch.rand(Cin) + 0.5).to(dtype), torch.randn(Cin).to(dtype)
    w = (torch.randn(Cout, Cin, 3, 3) * 0.05).to(dtype)
    bias, rowbias, res = torch.randn(Cout).to(dtype), torch.randn(B, Cout).to(dtype), torch.randn(B, H, W, Cout).to(dtype)
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().to(dev)
    xd, gd, bd = x.to(dev), gm.to(dev), bt.to(dev)
    assert hipops.gn_conv3x3_ok(xd, gd, Cout)
    y = hipops.gn_conv3x3_nhwc(xd, gd, bd, 1e-5, act, wt, wt, bias.to(dev), rowbias.to(dev), res.to(dev))
    # (the three-launch GroupNorm: its coefficient kernel is the one dm_groupnorm_nhwc_stats runs; the two-launch inference entry adds
    # the partial sums in another order and differs in the last bits of A and S)
    hn = hipops._gn_fwd(xd, gd, bd, 1e-5, act, keep_for_backward=True)[0]
    y2 = hipops.conv3x3_nhwc(hn, wt, bias.to(dev), 1, (1, 1), None, rowbias.to(dev), res.to(dev))
    assert torch.equal(y, y2)
    h = torch.nn.functional.group_norm(x.float().permute(0, 3, 1, 2), 32, gm.float(), bt.float(), 1e-5)
    h = torch.nn.functional.silu(h) if act else h
    ref = torch.nn.functional.conv2d(h, w.float(), bias.float(), padding=1).permute(0, 2, 3, 1) + rowbias.float()[:, None, None, :] + res.float()
    tol = (3e-2 if dtype == torch.bfloat16 else 4e-3) * ref.abs().max().item() + (1e-2 if dtype == torch.bfloat16 else 2e-3)
    assert (y.float().cpu() - ref).abs().max().item() < tol


@pytest.mark.parametrize("cin,cout,temb_ch,B,HW,grad", [(128, 128, 0, 8, 128, True), (128, 256, 0, 4, 128, True), (256, 256, 1280, 8, 96, False)])
def test_resnet_block_with_folded_groupnorm_equals_the_two_call_form(dev, monkeypatch, cin, cout, temb_ch, B, HW, grad):
    """ResnetBlock2D with both GroupNorm apply passes folded into its convolutions (hipops.GN_CONV_FOLD) against the block with
    the apply kernels in front of the same convolution kernels: outputs and input gradients bit-equal (VAE encoder: under autograd,
    x reaching norm1 and the skip through one node; frozen nets: time-embedding row bias and residual in the epilogues), and no
    apply / forward GroupNorm launch left."""
    from dreammat_amd.sd import layers
    monkeypatch.setattr(hipops, "GN_FOLD_MIN_BYTES", 0)
    torch.manual_seed(23)
    blk = layers.ResnetBlock2D(cin, cout, temb_ch, eps=1e-6).to(dev, torch.float16).eval()
    for p in blk.parameters():
        p.requires_grad_(False)
    x0 = torch.randn(B, cin, HW, HW, device=dev).to(torch.float16).contiguous(memory_format=torch.channels_last)
    temb = torch.randn(B, temb_ch, device=dev).to(torch.float16) if temb_ch else None
    g = torch.randn(B, cout, HW, HW, device=dev).to(torch.float16).contiguous(memory_format=torch.channels_last)
    outs = {}
    try:
        for fold in (False, True):
            hipops.GN_CONV_FOLD = fold
            x = x0.clone().requires_grad_(grad)
            hipops.enable_kernel_timing(True)
            with torch.set_grad_enabled(grad):
                y = blk(x, temb)
                if grad:
                    y.backward(g)
            torch.cuda.synchronize()
            keys = {k: v["launches"] for k, v in hipops.kernel_times().items()}
            hipops.enable_kernel_timing(False)
            outs[fold] = (y.detach(), x.grad if grad else None, keys)
    finally:
        hipops.GN_CONV_FOLD = True
    if grad:        # the same coefficient kernel on both sides: bit-equal
        assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    else:           # (the frozen nets' two-launch GroupNorm forms its coefficients in another summation order: last-bit differences)
        ref = outs[False][0].float()
        assert (outs[True][0].float() - ref).abs().max().item() <= 2.0 ** -9 * ref.abs().max().item()
    assert sum(n for k, n in outs[True][2].items() if k.startswith("conv3x3[gn+")) == 2 and not any(k.startswith("groupnorm_fwd") for k in outs[True][2])
    assert any(k.startswith("groupnorm_fwd") for k in outs[False][2])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("R,C,batch", [(64, 64, 1), (192, 512, 3), (4096, 512, 2)])
def test_transpose_kernel_is_exact(dev, dtype, R, C, batch):
    """dm_transpose_bf16 / _f16 (ABI v14): [batch, R, C] -> [batch, C, R], bit for bit"""
    torch.manual_seed(33)
    x = torch.randn(batch, R, C).to(dtype).to(dev)
    y = hipops.transpose_rows(x)
    assert y.shape == (batch, C, R) and torch.equal(y, x.transpose(1, 2).contiguous())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,Sq,Skv,D", [(2, 256, 256, 128), (3, 256, 512, 64), (1, 256, 16384, 64), (8, 4096, 4096, 512)])
def test_wide_head_attention_gemm_form_forward_and_gradients(dev, dtype, B, Sq, Skv, D):
    """hipops.wide_head_attention (the VAE mid-block attention's form: per group of images s = q k^T, row softmax, p v on dm_gemm_*_batched /
    dm_softmax_rows_* / dm_transpose_*, probabilities recomputed in the backward) against fp32 softmax attention: output and the
    three gradients; rectangular scores; 16384 columns (the 1024^2 shape's row length); the bench shape (8 x 4096 x 512), where
    forward + backward must allocate less than ONE [B, S, S] score tensor beyond the reusable per-image workspace."""
    torch.manual_seed(34)
    q, k, v = torch.randn(B, Sq, D).to(dtype), torch.randn(B, Skv, D).to(dtype), torch.randn(B, Skv, D).to(dtype)
    g = torch.randn(B, Sq, D).to(dtype)
    scale = 2.0 * D ** -0.5
    qr, kr, vr = (t.float().requires_grad_() for t in (q, k, v))
    ref = torch.softmax(scale * qr @ kr.transpose(1, 2), -1) @ vr
    ref.backward(g.float())
    qd, kd, vd = (t.to(dev).requires_grad_() for t in (q, k, v))
    assert hipops.wide_head_attention_ok(qd, kd, vd)
    gd = g.to(dev)
    if B * Sq * Skv >= 1 << 27:
        hipops.wide_head_attention(qd.detach().requires_grad_(), kd.detach(), vd.detach(), scale).backward(gd)     # (the workspace, once)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    hipops.enable_kernel_timing(True)
    o = hipops.wide_head_attention(qd, kd, vd, scale)
    o.backward(gd)
    torch.cuda.synchronize()
    if B * Sq * Skv >= 1 << 27:
        assert torch.cuda.max_memory_allocated() - base < B * Sq * Skv * 2, (torch.cuda.max_memory_allocated() - base, B * Sq * Skv * 2)
    keys = hipops.kernel_times()
    hipops.enable_kernel_timing(False)
    groups = B // hipops._wide_attn_group(B, Sq, Skv)
    assert sum(r["launches"] for k_, r in keys.items() if k_.startswith("gemm_batched")) == 7 * groups     # 2 forward + 5 backward products per group of images
    tol = 3e-2 if dtype == torch.bfloat16 else 4e-3       # (scores and probabilities are stored in 16 bits between the kernels)
    for got, want in ((o.detach(), ref.detach()), (qd.grad, qr.grad), (kd.grad, kr.grad), (vd.grad, vr.grad)):
        assert torch.isfinite(got).all()
        assert (got.float().cpu() - want).abs().max().item() < tol * want.abs().max().item() + tol * 1e-1


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_vae_attention_runs_on_hand_written_kernels_without_a_batch_score_tensor(dev, dtype):
    """VaeAttention (8 x 512 channels x 96 x 96: S = 9216 -- at the bench's 64 x 64 one [B, S, S] tensor is the size of eight
    activations, which the module's own projections add up to), frozen weights, differentiated input: no call leaves the
    hand-written kernels (it was torch.matmul on hipBLASLt), forward + backward allocate less than ONE [B, S, S] score tensor
    beyond the reusable per-image workspace (the old path held two and made two more in the backward), and output and input
    gradient match the fp32 ATen evaluation of the same module."""
    from dreammat_amd.sd import layers
    from dreammat_amd.sd.models import VaeAttention
    torch.manual_seed(35)
    B, C, Hh = 8, 512, 96
    att32 = VaeAttention(C).to(dev).eval()
    att = VaeAttention(C).to(dev, dtype).eval()
    att.load_state_dict({k: v.to(dtype) for k, v in att32.state_dict().items()})
    att32.load_state_dict({k: v.float() for k, v in att.state_dict().items()})
    for m in (att, att32):
        for p_ in m.parameters():
            p_.requires_grad_(False)
    x = torch.randn(B, C, Hh, Hh, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    g = torch.randn(B, C, Hh, Hh, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)

    def run(xin):
        y = att(xin)
        y.backward(g)
        return y

    run(x.clone().requires_grad_())                       # (workspace of the shape allocated here, once)
    torch.cuda.synchronize()
    layers.fallbacks(clear=True)
    xd = x.clone().requires_grad_()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    y = run(xd)
    torch.cuda.synchronize()
    delta = torch.cuda.max_memory_allocated() - base
    score_tensor = B * (Hh * Hh) ** 2 * 2
    assert not layers.fallbacks(), layers.fallbacks()
    assert delta < score_tensor, (delta, score_tensor)
    x32 = x.float().requires_grad_()
    old = layers.CONV_BACKEND
    try:
        layers.CONV_BACKEND = "gemm"
        y32 = att32(x32)
        y32.backward(g.float())
    finally:
        layers.CONV_BACKEND = old
    tol = 3e-2 if dtype == torch.bfloat16 else 4e-3
    assert ((y.detach().float() - y32.detach()).abs().max() / y32.abs().max()).item() < tol
    assert ((xd.grad.float() - x32.grad).abs().max() / x32.grad.abs().max()).item() < tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_frozen_linear_under_autograd_runs_on_the_fused_gemm(dev, dtype):
    """layers.linear_fused on differentiated activations through a frozen layer (the VAE encoder's 1 x 1 shortcuts and attention
    projections): forward and data gradient on dm_gemm_*_fused (hipops._LinearFrozen), with bias and residual, against fp32 torch."""
    from dreammat_amd.sd import layers
    torch.manual_seed(31)
    M, K, N = 4096, 128, 256
    x = torch.randn(M, K).to(dtype); w = (torch.randn(N, K) * 0.1).to(dtype); b = torch.randn(N).to(dtype); res = torch.randn(M, N).to(dtype)
    g = torch.randn(M, N).to(dtype)
    xr, rr = x.float().requires_grad_(), res.float().requires_grad_()
    (xr @ w.float().t() + b.float() + rr).backward(g.float())
    xd, rd = x.to(dev).requires_grad_(), res.to(dev).requires_grad_()
    wd, bd = w.to(dev), b.to(dev)
    layers.fallbacks(clear=True)
    hipops.enable_kernel_timing(True)
    y = layers.linear_fused(xd, wd, bd, rd)
    y.backward(g.to(dev))
    torch.cuda.synchronize()
    keys = list(hipops.kernel_times())
    hipops.enable_kernel_timing(False)
    assert not layers.fallbacks() and sum(k.startswith("gemm") for k in keys) == 2, keys
    tol = 2e-2 if dtype == torch.bfloat16 else 3e-3
    ref = (x.float() @ w.float().t() + b.float() + res.float())
    assert (y.detach().float().cpu() - ref).abs().max().item() < tol * ref.abs().max().item()
    assert (xd.grad.float().cpu() - xr.grad).abs().max().item() < tol * xr.grad.abs().max().item()
    assert torch.equal(rd.grad.cpu(), g)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N", [8])
def test_few_channel_linear_kernel_forward_and_data_gradient(dev, dtype, N):
    """dm_linear_small (AutoencoderKL's quant_conv, 8 -> 8, differentiated through: dreammat_guidance.py:284-292) through
    layers.linear_fused: forward and data gradient against fp32 torch, nothing left to ATen."""
    from dreammat_amd.sd import layers
    torch.manual_seed(32)
    M = 5000
    x = torch.randn(M, 8).to(dtype); w = torch.randn(N, 8).to(dtype); b = torch.randn(N).to(dtype); g = torch.randn(M, N).to(dtype)
    xr = x.float().requires_grad_()
    (xr @ w.float().t() + b.float()).backward(g.float())
    xd = x.to(dev).requires_grad_()
    layers.fallbacks(clear=True)
    y = layers.linear_fused(xd, w.to(dev), b.to(dev))
    y.backward(g.to(dev))
    assert not layers.fallbacks()
    tol = 2e-2 if dtype == torch.bfloat16 else 3e-3
    ref = x.float() @ w.float().t() + b.float()
    assert (y.detach().float().cpu() - ref).abs().max().item() < tol * ref.abs().max().item()
    assert (xd.grad.float().cpu() - xr.grad).abs().max().item() < tol * xr.grad.abs().max().item()
    with torch.no_grad():
        assert torch.equal(layers.linear_fused(x.to(dev), w.to(dev), b.to(dev)), y.detach())


def test_raster_edge_cases_bit_exact(dev):
    """object partly off-screen, behind the camera (discarded), degenerate and sub-pixel triangles, and a
    camera that sees nothing (empty coverage) -- ids/barycentrics stay bit-identical to the oracle."""
    m = pmesh.displaced_sphere(32, 24)
    md = util.mesh_dict(m)
    H = W = 96
    # view 0: very close (object larger than the frame, some vertices behind the near plane / camera)
    # view 1: looking away (nothing visible)   view 2: far away (sub-pixel triangles)
    b0 = camera.camera_batch(torch.tensor([10.0, 0.0, 25.0]), torch.tensor([30.0, 0.0, -60.0]),
                             torch.tensor([0.9, 3.0, 40.0]), torch.tensor([60.0, 30.0, 25.0]), H, W)
    mvp = b0["mvp_mtx"].clone()
    flip = torch.eye(4); flip[2, 2] = -1; flip[0, 0] = -1
    mvp[1] = mvp[1] @ torch.diag(torch.tensor([1.0, 1.0, 1.0, 1.0]))
    c2w = b0["c2w"][1].clone(); c2w[:3, 0] *= -1; c2w[:3, 2] *= -1          # turn the camera around
    proj = camera.get_projection_matrix(b0["fovy"][1:2], 1.0, 0.1, 1000.0)
    mvp[1] = camera.get_mvp_matrix(c2w[None], proj)[0][0]
    v = torch.cat([m.v_pos, m.v_pos[:3] * 0 + m.v_pos[:1]])                 # 3 coincident vertices
    tri = torch.cat([m.t_pos_idx, torch.tensor([[v.shape[0] - 3, v.shape[0] - 2, v.shape[0] - 1], [0, 0, 1]], dtype=torch.int32)])
    pos_o = oraster.vertex_transform(v.numpy(), mvp).numpy()
    ro = oraster.rasterize(pos_o, tri.numpy().astype(np.int32), H, W)
    pos = hipops.vertex_transform(v.to(dev), mvp.to(dev))
    rg = hipops.RasterContext(dev).rasterize(pos, tri.to(dev).int().contiguous(), H, W, check_overflow=True).cpu().numpy()
    assert np.array_equal(rg.view(np.uint32), ro.view(np.uint32))
    cov = (ro[..., 3] > 0).reshape(3, -1).mean(1)
    assert cov[1] == 0.0 and cov[0] > 0.3 and 0 < cov[2] < 0.01, cov


def test_raster_full_size_1024_and_bin_overflow_retry(dev):
    """BASELINE config 5 size (1024^2) on the 50k mesh, and the workspace-overflow detection + retry path."""
    m = pmesh.displaced_sphere(160, 160)
    md = util.mesh_dict(m)
    batch = util.make_views(2, 1024, 1024, seed=4)
    tri = m.t_pos_idx.to(dev).int().contiguous()
    pos = hipops.vertex_transform(m.v_pos.to(dev), batch["mvp_mtx"].to(dev))
    ctx = hipops.RasterContext(dev)
    ctx.ws = torch.empty(256 + 3 * 2 * 128 * 128 * 4 + 4096 * 4, dtype=torch.uint8, device=dev)   # far too small
    ctx._workspace = lambda B, Nf, H, W, _o=ctx._workspace: (ctx.ws if ctx.ws is not None else _o(B, Nf, H, W))
    rast = ctx.rasterize(pos, tri, 1024, 1024, check_overflow=True)           # first try overflows, retry succeeds
    assert ctx.ws_mult >= 2
    ro = oraster.rasterize(pos.cpu().numpy(), md["t_pos_idx"], 1024, 1024)
    assert np.array_equal(rast.cpu().numpy().view(np.uint32), ro.view(np.uint32))


def test_renderer_empty_view_and_determinism(dev, envs):
    """a batch containing a view that sees nothing still renders (white background, zero opacity) and the
    forward pass is deterministic (no atomics on the forward path)."""
    from dreammat_amd.geometry import DreamMatMesh
    from dreammat_amd.material import DreamMatMaterial
    from dreammat_amd.renderer import RaytraceRender
    from dreammat_amd.background import SolidColorBackground
    lat, fg, oenvs = envs
    torch.manual_seed(0)
    geom = DreamMatMesh({"shape_init": "sphere:32:24", "shape_init_params": 0.7}).to(dev)
    mat = DreamMatMaterial({"use_raytracing": False, "environment_scale": 2.0, "env_max_res": 32, "env_min_res": 8,
                            "n_envs": 3}, latlongs=lat).to(dev)
    rend = RaytraceRender({}, geometry=geom, material=mat, background=SolidColorBackground({}))
    B, H, W = 2, 64, 64
    batch = util.make_views(B, H, W, seed=2)
    c2w = batch["c2w"][1].clone(); c2w[:3, 0] *= -1; c2w[:3, 2] *= -1
    proj = camera.get_projection_matrix(batch["fovy"][1:2], 1.0, 0.1, 1000.0)
    batch["mvp_mtx"][1] = camera.get_mvp_matrix(c2w[None], proj)[0][0]
    batch["env_id"] = torch.tensor([0, 2])
    gb = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    ju, jn = torch.rand(B, H, W, device=dev), torch.randn(B, H, W, device=dev)
    o1 = rend(**gb, light_positions=None, jitter_u=ju, jitter_n=jn)
    o2 = rend(**gb, light_positions=None, jitter_u=ju, jitter_n=jn)
    assert torch.equal(o1["comp_rgb"], o2["comp_rgb"])
    assert float(o1["opacity"][1].abs().max()) == 0.0 and float((o1["comp_rgb"][1] - 1).abs().max()) == 0.0
    assert float(o1["opacity"][0].max()) == 1.0
    # all-empty batch: N = 0 everywhere
    gb["mvp_mtx"] = gb["mvp_mtx"][1:2].repeat(2, 1, 1)
    o3 = rend(**gb, light_positions=None, jitter_u=ju, jitter_n=jn)
    assert float((o3["comp_rgb"] - 1).abs().max()) == 0.0 and float(o3["loss_mat_reg"]) == 0.0


@pytest.mark.parametrize("dgrad", ["subpixel", "zeroins"])
def test_strided_asym_conv_autograd_vs_reference(dev, monkeypatch, dgrad):
    """AutoencoderKL downsampler: conv3x3 stride 2 over F.pad(x,(0,1,0,1)); forward on the MFMA kernel with an
    implied trailing pad; backward = the sub-pixel form (one 2 x 2 convolution at the gradient's resolution,
    dm_conv2x2_nhwc_bf16) or, forced, the 3x3 kernel on the zero-inserted gradient."""
    from dreammat_amd.sd import layers
    monkeypatch.setenv("DREAMMAT_S2_DGRAD", dgrad)
    torch.manual_seed(0)
    for (B, C, H, W) in [(2, 64, 16, 24), (1, 128, 32, 32), (3, 64, 70, 38)]:
        ds = layers.Downsample2D(C, asymmetric_pad=True).to(dev, torch.bfloat16)
        for p in ds.parameters():
            p.requires_grad_(False)
        x = torch.randn(B, C, H, W).bfloat16()
        xg = x.to(dev).requires_grad_()
        y = ds(xg)
        xr = x.float().requires_grad_()
        ref = torch.nn.functional.conv2d(torch.nn.functional.pad(xr, (0, 1, 0, 1)), ds.conv.weight.float().cpu(),
                                         ds.conv.bias.float().cpu(), stride=2)
        assert y.shape == ref.shape
        assert (y.float().cpu() - ref).abs().max() < 2e-2 * ref.abs().max() + 1e-2
        dy = torch.randn_like(ref).bfloat16()
        hipops.enable_kernel_timing(True)
        y.backward(dy.to(dev))
        torch.cuda.synchronize()
        keys = list(hipops.kernel_times())
        hipops.enable_kernel_timing(False)
        assert any(k.startswith("conv2x2_dgrad")for k in keys) == (dgrad == "subpixel"), keys
        ref.backward(dy.float())
        assert (xg.grad.float().cpu() - xr.grad).abs().max() < 2e-2 * xr.grad.abs().max() + 1e-2
        if dgrad == "subpixel" and B > 1:       # tensors past the kernel's 32-bit offsets run as image chunks (16 views at 1024^2)
            g_whole = xg.grad.clone()
            xg.grad = None
            monkeypatch.setattr(hipops, "CONV_MAX_TENSOR_BYTES", 2 * (H // 2) * (W // 2) * 4 * C + 1)
            ds(xg).backward(dy.to(dev))
            monkeypatch.undo()
            monkeypatch.setenv("DREAMMAT_S2_DGRAD", dgrad)
            assert torch.equal(xg.grad, g_whole)


@pytest.mark.parametrize("B,C,h,w", [(2, 64, 8, 8), (3, 128, 13, 9), (1, 64, 32, 16)])
def test_upsample2d_subpixel_conv_vs_torch(dev, monkeypatch, B, C, h, w):
    """diffusers Upsample2D (nearest 2x + conv3x3) without the upsampled tensor: one 2 x 2 convolution at the source resolution
    with summed taps (dm_conv2x2_nhwc_bf16 + interleave) against torch fp32 and against the materialised path."""
    from dreammat_amd.sd import layers
    torch.manual_seed(0)
    up = layers.Upsample2D(C).to(dev, torch.bfloat16).requires_grad_(False)
    x = torch.randn(B, C, h, w).bfloat16()
    monkeypatch.setenv("DREAMMAT_UPSAMPLE", "subpixel")
    with torch.no_grad():
        hipops.enable_kernel_timing(True)
        y = up(x.to(dev).contiguous(memory_format=torch.channels_last)).float().cpu()
        torch.cuda.synchronize()
        keys = list(hipops.kernel_times())
        hipops.enable_kernel_timing(False)
        monkeypatch.setenv("DREAMMAT_UPSAMPLE", "materialize")
        y_mat = up(x.to(dev).contiguous(memory_format=torch.channels_last)).float().cpu()
    assert any(k.startswith("conv2x2_upsample") for k in keys), keys
    ref = torch.nn.functional.conv2d(torch.nn.functional.interpolate(x.float(), scale_factor=2.0, mode="nearest"),
                                     up.conv.weight.float().cpu(), up.conv.bias.float().cpu(), padding=1)
    assert y.shape == ref.shape
    assert (y - ref).abs().max() <= 2e-2 * ref.abs().max()            # bf16 output + bf16-rounded summed taps
    assert (y - y_mat).abs().max() <= 2e-2 * ref.abs().max()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,Cs,Cin,Ho,Wo", [(2, 64, 64, 8, 12), (1, 128, 128, 35, 19), (3, 320, 64, 16, 16), (2, 512, 64, 5, 7)])
def test_conv2x2_subpixel_store_equals_interleaving_copy(dev, monkeypatch, dtype, B, Cs, Cin, Ho, Wo):
    """dm_conv2x2_subpixel_nhwc (ABI v12): the four Cs-channel blocks of an output pixel stored as the four sub-pixels of the 2x finer
    tensor -- mode 1 (stride-2 data gradient: block (py, px) of (u, v) -> (2u + py, 2v + px)) and mode 2 (upsample + conv on its
    (h + 1) x (w + 1) grid: -> (2u - py, 2v - px) where inside) -- must be BIT-equal to the plain kernel followed by the interleaving
    copy it replaces, ragged tiles, image boundaries and Cs = 320 (blocks that straddle the 256-wide N tiles) included."""
    torch.manual_seed(5)
    x = torch.randn(B, Ho, Wo, Cin).to(dtype).to(dev)
    w4 = (torch.randn(4 * Cs, 4 * Cin) * 0.05).to(dtype).to(dev)
    b4 = torch.randn(4 * Cs).to(dtype).to(dev)
    y = hipops.conv2x2_nhwc(x, w4, b4, (Ho, Wo), (1, 1))
    ref1 = y.view(B, Ho, Wo, 2, 2, Cs).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * Ho, 2 * Wo, Cs)
    assert torch.equal(hipops.conv2x2_nhwc(x, w4, b4, (Ho, Wo), (1, 1), subpixel=1), ref1)
    h, w = Ho - 1, Wo - 1                       # mode 2: the same grid read as the (h + 1) x (w + 1) grid of an h x w source
    ref2 = torch.empty(B, h, 2, w, 2, Cs, device=dev, dtype=dtype)
    for py in range(2):
        for px in range(2):
            blk = (2 * py + px) * Cs
            ref2[:, :, py, :, px] = y[:, py:py + h, px:px + w, blk:blk + Cs]
    out2 = hipops.conv2x2_nhwc(x, w4, b4, (Ho, Wo), (1, 1), subpixel=2)
    assert torch.equal(out2, ref2.view(B, 2 * h, 2 * w, Cs))
    if B > 1:                                   # image chunks (tensors past the 32-bit offsets)
        monkeypatch.setattr(hipops, "CONV_MAX_TENSOR_BYTES", 2 * Ho * Wo * 4 * max(Cs, Cin) + 1)
        assert torch.equal(hipops.conv2x2_nhwc(x, w4, b4, (Ho, Wo), (1, 1), subpixel=1), ref1)
        assert torch.equal(hipops.conv2x2_nhwc(x, w4, b4, (Ho, Wo), (1, 1), subpixel=2), out2)


def test_vae_encoder_bf16_gradient_vs_fp32_oracle(dev):
    """the differentiated bf16 VAE-encoder path (MFMA convs incl. data gradients, fused GroupNorm fwd/bwd,
    strided downsamplers) against the fp32 CPU functional oracle."""
    from dreammat_amd.sd import AutoencoderKLEncoder, SDArch
    from oracle import sd_nets as osd
    torch.manual_seed(0)
    arch = SDArch(name="vae-test", vae_block_out=(64, 128, 128, 128))
    vae = AutoencoderKLEncoder(arch).eval()
    for p in vae.parameters():
        p.requires_grad_(False)
    img = torch.rand(2, 3, 64, 64)
    xr = img.clone().requires_grad_()
    mean, logvar = osd.vae_encode_moments(vae.state_dict(), xr * 2 - 1)
    w = torch.randn_like(mean)
    (mean * w).sum().backward()
    vae.to(dev, torch.bfloat16)
    xg = img.to(dev).requires_grad_()
    hipops.enable_kernel_timing(True)
    mg, _ = vae.encode_moments((xg * 2 - 1).bfloat16())
    (mg.float() * w.to(dev)).sum().backward()
    torch.cuda.synchronize()
    keys = hipops.kernel_times().keys()
    hipops.enable_kernel_timing(False)
    assert any(k.startswith("conv3x3") and ",s2]" in k for k in keys), keys        # strided MFMA path was taken
    rel_f = ((mg.float().cpu() - mean).abs().max() / mean.abs().max()).item()
    rel_g = ((xg.grad.cpu() - xr.grad).abs().max() / xr.grad.abs().max()).item()
    assert rel_f < 5e-2 and rel_g < 8e-2, (rel_f, rel_g)


@pytest.mark.parametrize("rows,C", [(1000, 320), (77, 640), (513, 1280), (5, 2048), (64, 8)])
def test_layernorm_rows_vs_torch(dev, rows, C):
    torch.manual_seed(0)
    x = (torch.randn(rows, C) * 3 + 1.5).bfloat16()
    g = torch.randn(C).bfloat16(); b = torch.randn(C).bfloat16()
    y = hipops.layernorm_rows(x.to(dev), g.to(dev), b.to(dev), 1e-5).float().cpu()
    ref = torch.nn.functional.layer_norm(x.float(), (C,), g.float(), b.float(), 1e-5)
    assert (y - ref).abs().max() <= 2e-2 * ref.abs().max()              # one bf16 rounding of the output
    ref_bf = torch.nn.functional.layer_norm(x.to(dev), (C,), g.to(dev), b.to(dev), 1e-5).float().cpu()
    assert ((y - ref_bf).abs() > 1e-6).float().mean() < 0.15           # ATen's bf16 kernel: same up to 1-ulp ties
    assert (y - ref_bf).abs().max() <= 1.6e-2 * ref.abs().max()


@pytest.mark.parametrize("act", [0, 1])
def test_groupnorm_skip_output_gathers_both_gradients(dev, act):
    """groupnorm_nhwc_skip: x leaves the node a second time; the backward kernel adds the skip branch's gradient in its own pass
    (dm_groupnorm_nhwc_bwd_res) -- against fp32 autograd of act(GN(x)) * a + x * b, and with either branch unused."""
    torch.manual_seed(0)
    B, H, W, C = 2, 12, 10, 64
    x = torch.randn(B, H, W, C).bfloat16()
    gamma, beta = torch.randn(C).bfloat16(), torch.randn(C).bfloat16()
    wa, wb = torch.randn(B, H, W, C).bfloat16(), torch.randn(B, H, W, C).bfloat16()
    for use_y, use_skip in ((True, True), (True, False), (False, True)):
        xg = x.to(dev).requires_grad_(True)
        y, xs = hipops.groupnorm_nhwc_skip(xg, gamma.to(dev), beta.to(dev), 1e-6, act)
        loss = (y.float() * wa.to(dev).float()).sum() * float(use_y) + (xs.float() * wb.to(dev).float()).sum() * float(use_skip)
        if not use_y:
            loss = (xs.float() * wb.to(dev).float()).sum()
        if not use_skip:
            loss = (y.float() * wa.to(dev).float()).sum()
        loss.backward()
        xr = x.float().requires_grad_(True)
        yr = torch.nn.functional.group_norm(xr.permute(0, 3, 1, 2), 32, gamma.float(), beta.float(), 1e-6).permute(0, 2, 3, 1)
        if act:
            yr = torch.nn.functional.silu(yr)
        lr = (yr * wa.float()).sum() * float(use_y) + (xr * wb.float()).sum() * float(use_skip)
        lr.backward()
        assert (y.float().cpu() - yr).abs().max() <= 2e-2 * yr.abs().max()
        err = (xg.grad.float().cpu() - xr.grad).abs().max().item()
        assert err <= 2e-2 * xr.grad.abs().max().item(), (use_y, use_skip, err)


def test_conv_and_gemm_balanced_last_round_equals_single_launch(dev, monkeypatch):
    """Cout = 320 at 24 x 64 x 64 is 384 M tiles of 256 rows = 1.5 rounds of 256 CUs: the rows of the under-filled round go to a second
    launch with 128-row tiles (conv.hip launch_320_balanced).  Same arithmetic per output element as the single launch (forced
    through DREAMMAT_CONV_TILE / DREAMMAT_GEMM_TILE, whose kernels the tile-variant tests pin against torch): bit-equal."""
    torch.manual_seed(0)
    x = torch.randn(24, 64, 64, 320, device=dev).bfloat16()
    w = (torch.randn(320, 9 * 320, device=dev) * 0.02).bfloat16()
    b = torch.randn(320, device=dev).bfloat16()
    res = torch.randn(24, 64, 64, 320, device=dev).bfloat16()
    rb = torch.randn(24, 320, device=dev).bfloat16()
    y_bal = hipops.conv3x3_nhwc(x, w, b, 1, (1, 1), None, rb, res)
    monkeypatch.setenv("DREAMMAT_CONV_TILE", "320")
    y_one = hipops.conv3x3_nhwc(x, w, b, 1, (1, 1), None, rb, res)
    monkeypatch.delenv("DREAMMAT_CONV_TILE")
    assert torch.equal(y_bal, y_one)
    ref = torch.nn.functional.conv2d(x[5:6].float().permute(0, 3, 1, 2).cpu(), w.float().view(320, 3, 3, 320).permute(0, 3, 1, 2).cpu(),
                                     b.float().cpu(), padding=1).permute(0, 2, 3, 1) + rb[5].float().cpu() + res[5:6].float().cpu()
    assert (y_bal[5:6].float().cpu() - ref).abs().max() <= 2e-2 * ref.abs().max()
    # the image that straddles the two launches (rows 1024.. of the tall image = image 16) and the last one
    for img in (15, 16, 23):
        ref = torch.nn.functional.conv2d(x[img:img + 1].float().permute(0, 3, 1, 2).cpu(), w.float().view(320, 3, 3, 320).permute(0, 3, 1, 2).cpu(),
                                         b.float().cpu(), padding=1).permute(0, 2, 3, 1) + rb[img].float().cpu() + res[img:img + 1].float().cpu()
        assert (y_bal[img:img + 1].float().cpu() - ref).abs().max() <= 2e-2 * ref.abs().max(), img
    xm = torch.randn(98304, 320, device=dev).bfloat16()
    wm = (torch.randn(320, 320, device=dev) * 0.05).bfloat16()
    rm = torch.randn(98304, 320, device=dev).bfloat16()
    g_bal = hipops.gemm_fused(xm, wm, b, rm)
    monkeypatch.setenv("DREAMMAT_GEMM_TILE", "320")
    g_one = hipops.gemm_fused(xm, wm, b, rm)
    monkeypatch.delenv("DREAMMAT_GEMM_TILE")
    assert torch.equal(g_bal, g_one)
    refm = xm[-300:].float() @ wm.float().t() + b.float() + rm[-300:].float()
    assert (g_bal[-300:].float() - refm).abs().max() <= 2e-2 * refm.abs().max()


def test_conv_stem_residual_shared_by_branches(dev):
    """dm_conv3x3_small_res_nhwc_bf16: ControlNet's conv_in(sample) + conditioning embedding in one pass, the embedding of
    the B views shared by the 3 guidance branches (image b takes residual b % B)."""
    from dreammat_amd.sd import layers
    torch.manual_seed(0)
    conv = layers.Conv2d(4, 320, 3, padding=1).to(dev).bfloat16().requires_grad_(False)
    x = torch.randn(6, 4, 24, 40).bfloat16().to(dev)
    emb = torch.randn(2, 320, 24, 40).bfloat16().to(dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y = conv.forward_small(x, 0, emb).float().cpu()
        ref = torch.nn.functional.conv2d(x.float().cpu(), conv.weight.float().cpu(), conv.bias.float().cpu(), padding=1) + emb.float().cpu().repeat(3, 1, 1, 1)
    assert (y - ref).abs().max() <= 1e-2 * ref.abs().max()


def test_conv_stem_with_image_gradient_vs_torch(dev):
    """the VAE encoder's conv_in (3 -> 128) when the image needs a gradient: direct stem kernel forward, folded g W backward."""
    from dreammat_amd.sd import layers
    torch.manual_seed(0)
    conv = layers.Conv2d(3, 128, 3, padding=1)
    ref = torch.nn.Conv2d(3, 128, 3, padding=1)
    ref.load_state_dict(conv.state_dict())
    ref.weight.data = ref.weight.data.bfloat16().float(); ref.bias.data = ref.bias.data.bfloat16().float()
    conv = conv.to(dev).bfloat16().requires_grad_(False)
    x = torch.randn(2, 40, 56, 3).bfloat16()
    xg = x.to(dev).permute(0, 3, 1, 2).requires_grad_(True)              # logical NCHW over NHWC memory, like the render
    hipops.enable_kernel_timing(True)
    y = conv(xg)
    torch.cuda.synchronize()
    keys = hipops.kernel_times().keys()
    hipops.enable_kernel_timing(False)
    assert any(k.startswith("conv3x3_small[4->128") for k in keys), keys
    g = torch.randn(2, 128, 40, 56).bfloat16()
    y.backward(g.to(dev))
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    yr = ref(xr)
    yr.backward(g.float())
    assert (y.float().cpu() - yr).abs().max() <= 1e-2 * yr.abs().max()
    assert (xg.grad.float().cpu() - xr.grad).abs().max() <= 2e-2 * xr.grad.abs().max()


@pytest.mark.parametrize("shape", [(2, 512, 4096), (3, 77, 2048), (1, 5, 8), (700, 264), (2, 9, 8192)])
def test_softmax_rows_fwd_bwd_vs_torch(dev, shape):
    """dm_softmax_rows_bf16 / _bwd (the VAE mid-block attention's softmax, differentiated) against fp32 autograd of
    softmax(scale * s); more than one row per workgroup at the larger shapes (the grid is capped at 2048 workgroups)."""
    torch.manual_seed(0)
    scale = 512 ** -0.5
    s = (torch.randn(*shape) * 30).bfloat16()
    dp = torch.randn(*shape).bfloat16()
    sg = s.to(dev).requires_grad_(True)
    p = hipops.softmax_rows(sg, scale)
    p.backward(dp.to(dev))
    sr = s.float().requires_grad_(True)
    pr = torch.softmax(sr * scale, dim=-1)
    pr.backward(dp.float())
    assert (p.float().cpu() - pr).abs().max() <= 2 ** -8 * pr.abs().max() + 1e-6            # one bf16 rounding
    assert abs(p.float().sum(-1).mean().item() - 1.0) < 2e-3
    # the backward works from the ROUNDED probabilities (what the second matrix product saw): compare on those too
    pq = p.detach().float().cpu()
    ds_ref_q = scale * pq * (dp.float() - (pq * dp.float()).sum(-1, keepdim=True))
    ds = sg.grad.float().cpu()
    assert (ds - ds_ref_q).abs().max() <= 2 ** -7 * ds_ref_q.abs().max() + 1e-7
    assert (ds - sr.grad).abs().max() <= 2e-2 * sr.grad.abs().max() + 1e-7


@pytest.mark.parametrize("rows,inner", [(4096, 1280), (77, 2560), (3, 8)])
def test_geglu_rows_vs_torch(dev, rows, inner):
    torch.manual_seed(0)
    h = (torch.randn(rows, 2 * inner) * 2).bfloat16()
    y = hipops.geglu_rows(h.to(dev)).float().cpu()
    xv, gate = h.float().chunk(2, dim=-1)
    ref = xv * torch.nn.functional.gelu(gate)
    assert (y - ref).abs().max() <= 2e-2 * ref.abs().max()
    hb = h.to(dev)
    xb, gb = hb.chunk(2, dim=-1)
    ref_bf = (xb * torch.nn.functional.gelu(gb)).float().cpu()
    assert ((y - ref_bf).abs() > 1e-6).float().mean() < 0.05
    assert (y - ref_bf).abs().max() <= 1.6e-2 * ref.abs().max()


@pytest.mark.parametrize("tile", ["", "256", "512", "320", "640"])
@pytest.mark.parametrize("B,Cin,Cout,H,W", [(12, 64, 128, 8, 8), (3, 128, 320, 20, 12)])
def test_conv3x3_fused_epilogue(dev, monkeypatch, tile, B, Cin, Cout, H, W):
    """ResnetBlock2D's `+ temb[:, :, None, None]` and `+ input_tensor` folded into the conv epilogue; workgroup tiles
    that span several images (8x8 images, 256/512-row tiles) exercise the per-row image lookup."""
    if tile:
        monkeypatch.setenv("DREAMMAT_CONV_TILE", tile)
    torch.manual_seed(2)
    x = torch.randn(B, H, W, Cin).bfloat16()
    w = (torch.randn(Cout, Cin, 3, 3) * 0.05).bfloat16()
    bias = torch.randn(Cout).bfloat16()
    rowbias = torch.randn(B, Cout).bfloat16()
    res = torch.randn(B, H, W, Cout).bfloat16()
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    y = hipops.conv3x3_nhwc(x.to(dev), wt.to(dev), bias.to(dev), 1, (1, 1), None, rowbias.to(dev), res.to(dev)).float().cpu()
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias.float(), padding=1).permute(0, 2, 3, 1)
    ref = ref + rowbias.float()[:, None, None, :] + res.float()
    err = (y - ref).abs().max().item()
    assert err < 2e-2 * ref.abs().max().item() + 1e-2, err
    y1 = hipops.conv3x3_nhwc(x.to(dev), wt.to(dev), bias.to(dev), 1, (1, 1), None, rowbias.to(dev), None).float().cpu()
    assert (y1 - (ref - res.float())).abs().max().item() < 2e-2 * ref.abs().max().item() + 1e-2


@pytest.mark.parametrize("tile,M,K,N,res", [
    (None, 4096, 320, 320, True),            # to_out / proj_out @ 64x64 (ragged N tile at 128/256 wide tiles)
    (None, 1024, 1280, 1280, False),
    ("128", 272, 64, 192, True),             # ragged M tile, 128x64 variant
    ("128", 2048, 128, 256, False),
    ("256", 4096, 640, 640, True),
    ("512", 8192, 320, 512, True),
    ("320", 4096, 320, 640, True),
    ("320", 2064, 128, 320, False),
    (None, 32, 1280, 320, False),            # time_emb_proj-sized
])
def test_gemm_fused_vs_fp32_reference(dev, monkeypatch, tile, M, K, N, res):
    """Linear / 1x1-conv layers on the 1-tap LDS-DMA kernel: y = x w^T + bias (+ residual), every tile variant."""
    if tile:
        monkeypatch.setenv("DREAMMAT_GEMM_TILE", tile)
    torch.manual_seed(5)
    x = torch.randn(M, K).bfloat16()
    w = (torch.randn(N, K) * 0.05).bfloat16()
    b = torch.randn(N).bfloat16()
    r = torch.randn(M, N).bfloat16() if res else None
    y = hipops.gemm_fused(x.to(dev), w.to(dev), b.to(dev), r.to(dev) if res else None).float().cpu()
    ref = x.float() @ w.float().t() + b.float() + (r.float() if res else 0.0)
    err = (y - ref).abs().max().item()
    assert err < 1e-2 * ref.abs().max().item() + 1e-2, err
    y0 = hipops.gemm_fused(x.to(dev), w.to(dev), None, None).float().cpu()
    assert (y0 - x.float() @ w.float().t()).abs().max().item() < 1e-2 * ref.abs().max().item() + 1e-2


@pytest.mark.parametrize("tile,M,K,inner", [(None, 4096, 320, 1280), ("128", 1040, 64, 192), ("256", 2048, 640, 2560),
                                             ("512", 4096, 320, 1280)])
def test_gemm_geglu_epilogue_vs_unfused_pair(dev, monkeypatch, tile, M, K, inner):
    """diffusers GEGLU (`hidden, gate = proj(x).chunk(2); hidden * gelu(gate)`) fused into the projection GEMM: the fp32
    value * gelu(gate) of the fp32 projection, rounded once (the unfused pair rounds the projection, the gate and the product;
    rounds 2-4 reproduced the first of those roundings inside the epilogue)."""
    if tile:
        monkeypatch.setenv("DREAMMAT_GEMM_TILE", tile)
    torch.manual_seed(6)
    x = torch.randn(M, K).bfloat16()
    w = (torch.randn(2 * inner, K) * 0.08).bfloat16()
    b = torch.randn(2 * inner).bfloat16()
    y = hipops.gemm_fused(x.to(dev), hipops.geglu_interleave(w).to(dev), hipops.geglu_interleave(b).to(dev), None,
                          geglu=True).float().cpu()
    h = x.float() @ w.float().t() + b.float()
    ref = h[:, :inner] * torch.nn.functional.gelu(h[:, inner:])
    assert tuple(y.shape) == (M, inner)
    err = (y - ref).abs()
    # one bf16 rounding (half an ulp = 2^-9 relative) + the fp32 summation order of the projection (K <= 640 terms of O(1))
    assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-4).all()), (err - 2.0 ** -8 * ref.abs()).max().item()
    assert err.mean().item() < 1.5e-3 * ref.abs().mean().item() + 1e-5, err.mean().item()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_geglu_gate_function_on_exact_projections(dev, dtype):
    """The GEGLU epilogue's gate function alone (round 5: Abramowitz-Stegun erfc on the packed fp32 pipe in place of erff): one-hot
    rows of x make the projection EXACT (an entry of w), so the output differs from value * gelu(gate) evaluated in float64 by the
    final 16-bit rounding only -- on gates from -12 to 12, the negative tail included (where 1 + erf cancels)."""
    torch.manual_seed(61)
    M, K, inner = 256, 64, 256
    x = torch.zeros(M, K)
    x[torch.arange(M), torch.arange(M) % K] = 1.0
    wv = torch.randn(inner, K).to(dtype)
    wg = (torch.rand(inner, K) * 24.0 - 12.0).to(dtype)
    wg[0] = torch.linspace(-12.0, 12.0, K).to(dtype)
    wg[1, :4] = torch.tensor([0.0, -0.0, 1e-3, -1e-3]).to(dtype)
    w = torch.cat([wv, wg])
    y = hipops.gemm_fused(x.to(dtype).to(dev), hipops.geglu_interleave(w).to(dev), None, None, geglu=True).double().cpu()
    v = wv.double().t()[torch.arange(M) % K]              # [M, inner]: the exact projections
    g = wg.double().t()[torch.arange(M) % K]
    ref = v * 0.5 * g * (1.0 + torch.erf(g / 2.0 ** 0.5))
    # final rounding (half an ulp: 2^-9 | 2^-12 relative) + the approximation (relative error < 2e-3 only where |gelu| < 1e-6,
    # < 1e-4 above 1e-3; absolute < 5e-7 per unit of value) + half's subnormal spacing 2^-24
    rel, floor = (2.0 ** -8, 2e-6) if dtype == torch.bfloat16 else (2.0 ** -10, 2e-7)
    err = (y - ref).abs()
    bound = rel * ref.abs() + floor * v.abs().clamp_min(1.0)
    assert bool((err <= bound).all()), (err - bound).max().item()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,heads,S", [(640, 10, 1024), (1280, 20, 256), (640, 10, 208)])
def test_self_attention_fused_qk_projection_vs_two_gemms(dev, monkeypatch, dtype, C, heads, S):
    """frozen self-attention at C >= 640: q | k from one GEMM over [Wq; Wk], handed to the attention kernels as row-strided views of
    its [B, S, 2C] output -- against the same layer with the two projections run on their own (layers.QK_FUSED = False) and
    against fp32."""
    from dreammat_amd.sd import layers
    torch.manual_seed(21)
    att = layers.Attention(C, heads).to(dev)
    for p_ in att.parameters():
        p_.requires_grad_(False)
    x = torch.randn(3, S, C, device=dev)
    res = torch.randn(3, S, C, device=dev)
    with torch.no_grad():
        ref = att(x, None, res).float()
        a16 = att.to(dtype)
        hipops.enable_kernel_timing(True)
        y = a16(x.to(dtype), None, res.to(dtype))
        torch.cuda.synchronize()
        keys = list(hipops.kernel_times())
        hipops.enable_kernel_timing(False)
        monkeypatch.setattr(layers, "QK_FUSED", False)
        y2 = a16(x.to(dtype), None, res.to(dtype))
    assert any(k.startswith(f"gemm[M={3 * S},K={C},N={2 * C}]") for k in keys), keys        # the fused projection ran
    assert (y.float() - y2.float()).abs().max().item() <= 2.0 ** -7 * y2.float().abs().max().item()     # same sums, another tile order at most
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert (y.float() - ref).abs().max().item() < tol * ref.abs().max().item() + tol


@pytest.mark.parametrize("variant", ["auto", "w128", "w64", "v3l", "staged", "fp8"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_attention_kernels_take_row_strided_q_and_k(dev, variant, dtype):
    """q and k as the two halves of ONE [B, S, 2C] tensor (the fused q | k projection: row stride 2C, k's base 2C bytes into a row)
    give bit for bit the output of contiguous copies, in every kernel family and in the MX-FP8 kernel (its quantisation pre-passes
    read through the same strides)."""
    torch.manual_seed(33)
    B, S, heads, D = 2, 1024, 10, 64
    C = heads * D
    qk = torch.randn(B, S, 2 * C, device=dev).to(dtype)
    v = torch.randn(B, S, C, device=dev).to(dtype)
    vt = v.transpose(1, 2).contiguous()
    q, k = qk[..., :C], qk[..., C:]
    fn = hipops.attention_fp8 if variant == "fp8" else hipops.attention
    if variant not in ("auto", "fp8"):
        hipops.attention_select(variant)
    try:
        y_strided = fn(q, k, vt, heads)
        y_contig = fn(q.contiguous(), k.contiguous(), vt, heads)
    finally:
        hipops.attention_select(None)
    assert torch.equal(y_strided, y_contig)
    ref = torch.nn.functional.scaled_dot_product_attention(
        q.float().view(B, S, heads, D).transpose(1, 2), k.float().view(B, S, heads, D).transpose(1, 2),
        v.float().view(B, S, heads, D).transpose(1, 2)).transpose(1, 2).reshape(B, S, C)
    tol = 0.12 if variant == "fp8" else (2e-2 if dtype == torch.bfloat16 else 4e-3)
    assert (y_strided.float() - ref).abs().max().item() < tol * ref.abs().max().item() + tol * 0.1


def test_transformer_block_fused_gemms_vs_aten(dev):
    """BasicTransformerBlock / Transformer2DModel with the Linear layers, their residual adds and GEGLU on the fused GEMM
    kernel vs the same module evaluated with ATen ops in fp32."""
    from dreammat_amd.sd import layers
    torch.manual_seed(7)
    C, heads, B, H, W = 320, 5, 2, 16, 16
    blk = layers.Transformer2DModel(C, heads, 1024, True).to(dev)
    for p_ in blk.parameters():
        p_.requires_grad_(False)
    x = torch.randn(B, C, H, W)
    ctx = torch.randn(B, 77, 1024)
    ref_mod = blk.float()
    with torch.no_grad():
        monkey = layers.CONV_BACKEND
        layers.CONV_BACKEND = "gemm"                    # ATen path
        ref = ref_mod(x.to(dev), layers.PaddedContext(ctx.to(dev))).float().cpu()
        layers.CONV_BACKEND = monkey
        blk16 = blk.to(torch.bfloat16)
        hipops.enable_kernel_timing(True)
        y = blk16(x.to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last),
                  layers.PaddedContext(ctx.to(dev, torch.bfloat16))).float().cpu()
        torch.cuda.synchronize()
        keys = list(hipops.kernel_times())
        hipops.enable_kernel_timing(False)
    assert sum(k.startswith("gemm") for k in keys) >= 5 and any(k.startswith("gemm+geglu") for k in keys), keys
    err = (y - ref).abs().max().item()
    assert err < 3e-2 * ref.abs().max().item() + 3e-2, err


@pytest.mark.parametrize("split,B,Cin,Cout,H,W", [(None, 3, 1280, 1280, 8, 8), ("7", 3, 640, 1280, 8, 8), ("3", 2, 128, 192, 16, 24),
                                                  ("16", 3, 1280, 1280, 16, 16), ("0", 3, 1280, 1280, 8, 8)])
def test_conv3x3_split_k_small_m(dev, monkeypatch, split, B, Cin, Cout, H, W):
    """Small-M layers (1 view per rank: batch 3 at 8x8 / 16x16) split the K loop over workgroups; fp32 partials + reduce
    kernel must reproduce bias + rowbias + residual exactly like the single-pass epilogue, bit for bit the same launch after launch."""
    if split is not None:
        monkeypatch.setenv("DREAMMAT_CONV_SPLITK", split)
    torch.manual_seed(11)
    x = torch.randn(B, H, W, Cin).bfloat16()
    w = (torch.randn(Cout, Cin, 3, 3) * 0.03).bfloat16()
    bias, rowbias, res = torch.randn(Cout).bfloat16(), torch.randn(B, Cout).bfloat16(), torch.randn(B, H, W, Cout).bfloat16()
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    args = (x.to(dev), wt.to(dev), bias.to(dev), 1, (1, 1), None, rowbias.to(dev), res.to(dev))
    y_dev = hipops.conv3x3_nhwc(*args)
    y = y_dev.float().cpu()
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias.float(), padding=1).permute(0, 2, 3, 1)
    ref = ref + rowbias.float()[:, None, None, :] + res.float()
    err = (y - ref).abs().max().item()
    assert err < 1e-2 * ref.abs().max().item() + 1e-2, err
    for _ in range(25):
        assert torch.equal(hipops.conv3x3_nhwc(*args), y_dev)
    # the linear layers of the same resolution (1-tap instantiation)
    xm = torch.randn(B * H * W // 16 * 16, Cin).bfloat16()
    wm = (torch.randn(Cout, Cin) * 0.05).bfloat16()
    rm = torch.randn(xm.shape[0], Cout).bfloat16()
    margs = (xm.to(dev), wm.to(dev), bias.to(dev), rm.to(dev))
    ym_dev = hipops.gemm_fused(*margs)
    ym = ym_dev.float().cpu()
    refm = xm.float() @ wm.float().t() + bias.float() + rm.float()
    assert (ym - refm).abs().max().item() < 1e-2 * refm.abs().max().item() + 1e-2
    for _ in range(25):
        assert torch.equal(hipops.gemm_fused(*margs), ym_dev)


@pytest.mark.parametrize("B,Cin,Cout,H,W,stride,act", [(2, 22, 16, 40, 24, 1, 1), (1, 16, 16, 33, 17, 1, 1), (2, 16, 32, 32, 32, 2, 1),
                                                         (1, 32, 96, 24, 24, 2, 1), (3, 4, 320, 16, 16, 1, 0), (1, 32, 32, 9, 13, 1, 0),
                                                         (1, 8, 48, 12, 12, 1, 1),
                                                         (2, 128, 4, 21, 37, 1, 0), (1, 128, 4, 64, 48, 1, 0), (1, 128, 8, 19, 16, 1, 0)])
def test_conv3x3_small_channel_direct_kernel(dev, B, Cin, Cout, H, W, stride, act):
    """the few-channel stem convs (ControlNet conditioning embedding, conv_in) on the direct kernel, with the fused SiLU."""
    torch.manual_seed(4)
    x = torch.randn(B, H, W, Cin).bfloat16()
    w = (torch.randn(Cout, Cin, 3, 3) * 0.2).bfloat16()
    bias = torch.randn(Cout).bfloat16()
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    y = hipops.conv3x3_small_nhwc(x.to(dev), wt.to(dev), bias.to(dev), stride, (1, 1), act).float().cpu()
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias.float(), stride=stride, padding=1)
    if act:
        ref = torch.nn.functional.silu(ref)
    ref = ref.permute(0, 2, 3, 1)
    assert y.shape == ref.shape
    assert (y - ref).abs().max().item() < 1e-2 * ref.abs().max().item() + 1e-2


def test_conv3x3_small_input_beyond_4gb_goes_in_image_groups(dev):
    """the image gradient of the VAE's conv_in at BASELINE configs[4]'s shape (16 views at 1024^2: 128 channels in = 4.3 GB, past the
    patch kernel's 32-bit buffer offsets): the launcher walks the batch in groups of images -- same values as image-by-image calls."""
    torch.manual_seed(36)
    B, H, Cin, Cout = 17, 1024, 128, 4
    x = torch.randn(B, H, H, Cin, device=dev, dtype=torch.float16)
    assert x.numel() * 2 > 0xffffff00
    w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.05).to(torch.float16)
    y = hipops.conv3x3_small_nhwc(x, w, None, 1, (1, 1), 0)
    for b in (0, 7, 15, 16):
        assert torch.equal(y[b:b + 1], hipops.conv3x3_small_nhwc(x[b:b + 1].contiguous(), w, None, 1, (1, 1), 0))
    assert torch.isfinite(y).all()


def test_controlnet_cond_embedding_stem_kernels_vs_aten(dev):
    """ControlNetConditioningEmbedding (22 -> 16 -> 32 -> 96 -> 256 -> 320, SiLU between) on the direct stem kernel + the
    Cout-padded MFMA kernel vs the ATen fp32 evaluation; no im2col launch may remain."""
    from dreammat_amd.sd import models
    torch.manual_seed(8)
    emb = models.ControlNetConditioningEmbedding(320, 22, (16, 32, 96, 256)).to(dev)
    for p_ in emb.parameters():
        p_.requires_grad_(False)
    c = torch.rand(2, 22, 64, 64)
    with torch.no_grad():
        ref = emb.float()(c.to(dev)).float().cpu()
        e16 = emb.to(torch.bfloat16)
        hipops.enable_kernel_timing(True)
        y = e16(c.to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)).float().cpu()
        torch.cuda.synchronize()
        keys = list(hipops.kernel_times())
        hipops.enable_kernel_timing(False)
    assert sum(k.startswith("conv3x3_small") for k in keys) == 5 and sum(k.startswith("conv3x3[") for k in keys) == 3, keys
    assert (y - ref).abs().max().item() < 3e-2 * ref.abs().max().item() + 3e-2


@pytest.mark.parametrize("sharded", [False, True])
def test_rccl_world1_step_through_bench(dev, sharded):
    """RCCL in the GPU test tier: bench.py with DREAMMAT_FORCE_DIST=1 initialises the `nccl` (= RCCL) process group at world
    size 1 and runs real optimisation steps whose flat-gradient all-reduce / barriers go through it (tiny nets, 64^2); with
    `--sharded-adam` the exchange is RCCL's reduce-scatter + all-gather around the sharded fused Adam (optimizer.sharded)."""
    import json
    import subprocess
    import sys
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:                       # a free port per run: the two parametrisations follow each other closely
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, DREAMMAT_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--views", "2", "--res", "64",
                        "--sd", "tiny", "--mesh", "sphere:24:24", "--env-res", "32", "--no-cpu-baseline"]
                       + (["--sharded-adam"] if sharded else []),
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["value"] > 0
    # the line's second leg: the same step with the nets cast in place to the other 16-bit type (round 6: the main line runs IEEE half, the
    # reference's type, the leg bf16; here under the captured hipGraph path), and the box calibration beside them
    assert res["dtype"] == "f16"
    assert res.get("bf16_leg", {}).get("value") and res["bf16_leg"]["final_loss"] == res["bf16_leg"]["final_loss"], res.get("bf16_leg")
    assert res["box_calibration"]["gemm_bf16_8192_tflops"] > 100 and res["box_calibration"]["copy_1gb_tbps"] > 1, res["box_calibration"]
    assert ("reduce-scatter" if sharded else "all-reduce") in res["config"]["parallelism"]
    assert "ProcessGroupNCCL" in r.stderr or "NCCL" in r.stderr.upper() or res["config"].get("process_group") == "nccl"


_GLOO_CUDA_PROBE = """
import sys, torch, torch.distributed as dist
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d", rank=0, world_size=1)
t = torch.ones(8, device="cuda:0")
dist.all_reduce(t); dist.broadcast(t, src=0)
o = torch.empty(8, device="cuda:0"); dist.all_gather_into_tensor(o, t)
dist.barrier(); torch.cuda.synchronize(); dist.destroy_process_group()
print("gloo-cuda-ok")
"""


def test_bench_world2_control_flow_on_one_gpu(dev):
    """bench.py exactly as the driver launches it at N = 2 (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus
    2`), with both ranks on this box's one GPU and gloo carrying the collectives (DREAMMAT_BENCH_SHARE_GPU / _BACKEND: RCCL
    refuses two ranks on a device).  What it pins is the file's CONTROL FLOW at world > 1: every training step and barrier is
    taken by both ranks (round 5's shade replay once took a step on rank 0 alone -- the all-reduce of that step never returns),
    rank 0 alone prints the line, the line says 2 GPUs and one view per rank."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def free_port():
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            return sk.getsockname()[1]

    probe = subprocess.run([sys.executable, "-c", _GLOO_CUDA_PROBE % free_port()], capture_output=True, text=True, timeout=300)
    if "gloo-cuda-ok" not in probe.stdout:
        pytest.skip("this torch build's gloo does not carry device tensors: " + probe.stderr[-300:])
    env = dict(os.environ, DREAMMAT_BENCH_SHARE_GPU="1", DREAMMAT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "DREAMMAT_FORCE_DIST"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--views", "2", "--res", "64", "--sd", "tiny", "--mesh", "sphere:24:24", "--env-res", "32", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["value"] > 0 and res["scaling"] in ("weak", "strong")
    assert res["config"]["process_group"] == "gloo" and "all-reduce" in res["config"]["parallelism"]
    assert res["roofline_shade_fwd"]["replay_step_features"]["launches_timed"] > 0       # the replay ran (rank 0), after a step both ranks took


def test_conv_and_gemm_chunk_batches_past_the_32bit_addressing_limit(dev, monkeypatch):
    """activations past 4 GB per tensor (cfg5: 16 views at 1024^2 through the VAE) run as image / row chunks of the same
    buffers; exercised here by lowering the limit instead of allocating 4 GB."""
    torch.manual_seed(12)
    B, H, W, Cin, Cout = 5, 16, 16, 64, 128
    x = torch.randn(B, H, W, Cin).bfloat16().to(dev)
    w = (torch.randn(Cout, 9 * Cin) * 0.05).bfloat16().to(dev)
    b = torch.randn(Cout).bfloat16().to(dev)
    rb, res = torch.randn(B, Cout).bfloat16().to(dev), torch.randn(B, H, W, Cout).bfloat16().to(dev)
    ref = hipops.conv3x3_nhwc(x, w, b, 1, (1, 1), None, rb, res)
    xm, wm = torch.randn(1024, 128).bfloat16().to(dev), (torch.randn(256, 128) * 0.05).bfloat16().to(dev)
    rm = torch.randn(1024, 256).bfloat16().to(dev)
    refm = hipops.gemm_fused(xm, wm, None, rm)
    monkeypatch.setattr(hipops, "CONV_MAX_TENSOR_BYTES", 2 * 2 * H * W * Cout + 100)       # two images per launch
    assert torch.equal(hipops.conv3x3_nhwc(x, w, b, 1, (1, 1), None, rb, res), ref)
    monkeypatch.setattr(hipops, "CONV_MAX_TENSOR_BYTES", 2 * 300 * 256)                    # 288-row chunks
    assert torch.equal(hipops.gemm_fused(xm, wm, None, rm), refm)


def test_narrow_head_conv_zero_padded_to_mfma_tile(dev):
    """UNet conv_out (320->4) / VAE conv_out (512->8): Cout zero-padded to 64 for the MFMA kernel, forward + dgrad."""
    from dreammat_amd.sd import layers
    torch.manual_seed(3)
    for (Cin, Cout) in [(128, 4), (64, 8)]:
        conv = layers.Conv2d(Cin, Cout, 3, padding=1).to(dev, torch.bfloat16)
        for p in conv.parameters():
            p.requires_grad_(False)
        x = torch.randn(2, Cin, 24, 16).bfloat16()
        hipops.enable_kernel_timing(True)
        xg = x.to(dev).requires_grad_()
        y = conv(xg)
        with torch.no_grad():
            y2 = conv(x.to(dev))
        torch.cuda.synchronize()
        n_conv = sum(v["launches"] for k, v in hipops.kernel_times().items() if k.startswith("conv3x3"))
        hipops.enable_kernel_timing(False)
        assert n_conv == 2 and tuple(y.shape) == (2, Cout, 24, 16)
        xr = x.float().requires_grad_()
        ref = torch.nn.functional.conv2d(xr, conv.weight.float().cpu(), conv.bias.float().cpu(), padding=1)
        assert (y.float().cpu() - ref).abs().max() < 2e-2 * ref.abs().max() + 1e-2
        assert torch.equal(y2, y.detach())
        dy = torch.randn_like(ref).bfloat16()
        y.backward(dy.to(dev))
        ref.backward(dy.float())
        assert (xg.grad.float().cpu() - xr.grad).abs().max() < 2e-2 * xr.grad.abs().max() + 1e-2


# ---- rows f-3 / f-2 and the fp16 atlas (first ran as XPASS in round 1; the exporter needed `vtex_buffer` on the device)
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


def test_exporter_bake_on_apple_obj_vs_oracle(dev):
    """f-3 (mesh_exporter.py:53-137) on the reference's own apple.obj (4 164 triangles, its own vt atlas) against the ORACLE:
    the UV-space coverage ids / barycentrics of the bake bit-equal to oracle/raster_ref.c, and every covered texel's albedo /
    metallic / roughness equal to oracle/field.py's hash grid + MLP evaluated at the oracle-interpolated position, through
    the reference's export activation (dreammat_material.py:765-797: squared-roughness range, sqrt).  Texels next to a chart
    border get the dilation's mean of covered neighbours (ours; the reference inpaints with cv2): checked for being a convex
    combination of covered texels, not against a reference value."""
    import dreammat_amd
    dreammat_amd._import_plugins()
    torch.manual_seed(3)
    geo = dreammat_amd.find("dreammat-mesh")({"shape_init": "mesh:" + os.path.join(ASSETS, "apple.obj"), "shape_init_params": 0.7,
                                              "shape_init_mesh_up": "+y", "shape_init_mesh_front": "+z"}).to(dev)
    with torch.no_grad():
        geo.encoding.encoding.params.uniform_(-1, 1)
        geo.feature_network.layers[0].weight.copy_(torch.randn(64, 32) * 0.3)
        geo.feature_network.layers[2].weight.copy_(torch.randn(5, 64) * 0.3)
    lat = [torch.full((16, 32, 3), 0.25) for _ in range(5)]
    mat = dreammat_amd.find("dreammat-material")({"use_raytracing": False, "env_max_res": 32, "env_min_res": 8}, latlongs=lat).to(dev)
    S = 256
    ex = dreammat_amd.find("mesh-exporter")({"texture_size": S, "texture_format": "png", "xatlas_pack_options": {"padding": 2}},
                                            geometry=geo, material=mat, background=None)
    mesh = geo.isosurface()
    assert mesh.v_tex is not None and mesh.t_tex_idx.shape[0] == 4164
    maps, holes = ex.bake_textures(mesh)
    # ---- oracle: rasterize the uv triangles, interpolate positions, evaluate the field, activate
    uv = mesh.v_tex.float().cpu() * 2.0 - 1.0
    uv4 = torch.cat([uv, torch.zeros_like(uv[:, :1]), torch.ones_like(uv[:, :1])], -1)[None].numpy()
    t_tex = mesh.t_tex_idx.cpu().numpy().astype(np.int32)
    t_pos = mesh.t_pos_idx.cpu().numpy().astype(np.int32)
    ro = oraster.rasterize(uv4, t_tex, S, S)
    covered = torch.from_numpy(ro[0, :, :, 3] > 0)
    assert torch.equal(~holes.cpu(), covered)
    assert 0.2 < float(covered.float().mean()) < 0.95
    pos = torch.from_numpy(oraster.interpolate(mesh.v_pos.float().cpu().numpy(), ro, t_pos))[0].reshape(-1, 3)
    lv, _ = ofield.grid_levels()
    feats = ofield.field_forward(pos, geo.encoding.encoding.params.detach().cpu().reshape(-1, 2),
                                 geo.feature_network.layers[0].weight.detach().cpu(),
                                 geo.feature_network.layers[2].weight.detach().cpu(), lv, radius=1.0)
    m = torch.sigmoid(feats)
    ref = {"albedo": m[:, :3], "metallic": m[:, 3:4] * 0.9, "roughness": torch.sqrt(m[:, 4:5] * (0.9 - 0.01) + 0.01 + 1e-7)}
    sel = covered.reshape(-1)
    for k in ("albedo", "metallic", "roughness"):
        got = maps[k].reshape(S * S, -1).cpu()
        assert (got[sel] - ref[k][sel]).abs().max() < 1e-4, k
        # dilated ring: inside the range of the covered texels (a mean of covered neighbours), further out: zero
        ring = got[~sel]
        assert float(ring.min()) >= -1e-6 and float(ring.max()) <= float(got[sel].max()) + 1e-6
    assert float((maps["albedo"].reshape(S * S, -1).cpu()[~sel].abs().sum(-1) > 0).float().mean()) > 0.01   # the ring exists


def test_controlnet_training_step_on_the_gpu_vs_fp32_oracle(dev):
    """f-4 (controlnet_train/diffusers_train_controlnet.py:858-915) on the GPU at 64^2 with the tiny architecture: the bf16
    production modules (implicit-GEMM conv data gradients, GroupNorm backward, fused GEMMs, the MFMA attention forward +
    backward of csrc/attn_bwd.hip, and -- where a layer has whole 64-channel tiles -- the trainable-conv route with the
    weight-gradient kernel) under autograd against the functional fp32 CPU oracle differentiated by autograd -- loss and
    the gradient of every ControlNet parameter that receives one -- and a few optimiser steps that must lower the loss."""
    from dreammat_amd import controlnet_train as ct
    from dreammat_amd.sd import ARCHS, AutoencoderKLEncoder, UNet2DConditionModel
    from oracle import sd_nets as osd
    a = ARCHS["tiny"]
    torch.manual_seed(0)
    unet, vae = UNet2DConditionModel(a), AutoencoderKLEncoder(a)
    cn = ct.init_controlnet(unet)
    for conv in list(cn.controlnet_down_blocks) + [cn.controlnet_mid_block, cn.controlnet_cond_embedding.conv_out]:
        torch.nn.init.normal_(conv.weight, std=0.05)          # off the zero initialisation: every layer gets a gradient
    g = torch.Generator().manual_seed(1)
    B = 2
    img = torch.rand(B, 3, 64, 64, generator=g) * 2 - 1
    cond = torch.rand(B, 22, 64, 64, generator=g)
    text = torch.randn(B, 77, a.cross_dim, generator=g)
    t = torch.tensor([300, 700])
    noise = torch.randn(B, 4, 8, 8, generator=g)
    pn = torch.randn(B, 4, 8, 8, generator=g)
    # ---- oracle: functional fp32 nets over the state dicts, ControlNet weights as autograd leaves
    sd_u = {k: v.detach().clone() for k, v in unet.state_dict().items()}
    sd_c = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in cn.state_dict().items()}
    sd_v = {k: v.detach().clone() for k, v in vae.state_dict().items()}
    mean, logvar = osd.vae_encode_moments(sd_v, img)
    lat = (mean + torch.exp(0.5 * logvar.clamp(-30, 20)) * pn) * vae.scaling_factor
    ac = osd.alphas_cumprod()[t].view(B, 1, 1, 1)
    noisy = ac.sqrt() * lat + (1 - ac).sqrt() * noise
    od, om = osd.controlnet_forward(sd_c, noisy, t, text, cond, 1.0, a.heads, a.use_linear_projection)
    loss_o = torch.nn.functional.mse_loss(osd.unet_forward(sd_u, noisy, t, text, a.heads, a.use_linear_projection, od, om), noise)
    loss_o.backward()
    # ---- product: bf16 frozen nets on the GPU, fp32 master ControlNet in bf16 compute
    tr = ct.ControlNetTrainer(vae.to(dev).bfloat16(), unet.to(dev).bfloat16(), controlnet=cn.to(dev).bfloat16(), lr=2e-3)
    fixed = dict(timesteps=t.to(dev), noise=noise.to(dev), posterior_noise=pn.to(dev))
    loss_g = ct.controlnet_training_loss(tr.vae, tr.unet, tr.controlnet, tr.scheduler, img.to(dev).bfloat16(), cond.to(dev),
                                         text.to(dev), **fixed)
    loss_g.backward()
    assert abs(float(loss_g) - float(loss_o)) < 3e-2 * abs(float(loss_o)), (float(loss_g), float(loss_o))
    num = den = 0.0
    n_checked = 0
    per_tensor = {}
    gmax = max(float(v.grad.norm()) for v in sd_c.values() if v.grad is not None)
    for k, p in tr.controlnet.named_parameters():
        go = sd_c[k].grad
        # (biases in front of a GroupNorm / the shift-invariant part of a normalised layer have an analytically ZERO gradient:
        # the oracle's is ~1e-9, bf16 rounding noise would be compared against nothing)
        if go is None or float(go.norm()) < 1e-4 * gmax:
            continue
        gg = p.grad.float().cpu()
        num += float(((gg - go) ** 2).sum()); den += float((go ** 2).sum())
        # per tensor: bf16 activations and weights through ~20 layers (worst: the time embedding, whose gradient sums over
        # every residual block: 0.14 measured)
        per_tensor[k] = (float((gg - go).norm() / go.norm()), float(go.norm()))
        n_checked += 1
    worst = sorted(per_tensor.items(), key=lambda kv: -kv[1][0])[:8]
    with open(os.path.join(OUT, "controlnet_train_gpu_per_tensor.json"), "w") as fh:
        json.dump({"global_rel_l2": (num / den) ** 0.5, "worst": worst}, fh)
    # measured 0.073 over all tensors; the worst single tensors (~0.5) are the first half of mid_block.resnets.1, where the tiny
    # architecture at 64^2 normalises over ONE pixel (8^2 latents, three downsamplings): a GroupNorm over a handful of values
    # amplifies bf16 rounding -- an artefact of the test size, recorded in controlnet_train_gpu_per_tensor.json
    assert n_checked > 40 and (num / den) ** 0.5 < 0.10, (n_checked, (num / den) ** 0.5, worst)
    tr.controlnet.zero_grad(set_to_none=True)
    losses = [float(tr.step(img.to(dev).bfloat16(), cond.to(dev), text.to(dev), **fixed)) for _ in range(8)]
    assert losses[-1] < losses[0], losses
    with open(os.path.join(OUT, "controlnet_train_gpu.json"), "w") as fh:
        json.dump({"loss_gpu_bf16": float(loss_g), "loss_oracle_fp32": float(loss_o), "grad_rel_l2": (num / den) ** 0.5,
                   "tensors_checked": n_checked, "losses": losses}, fh)


def test_controlnet_launcher_main_bf16_on_the_gpu(tmp_path):
    """`python -m dreammat_amd.controlnet_train` crashed on its first GPU step (bf16 latents against an fp32
    ControlNet).  The launcher itself, tiny architecture, two steps: bf16 compute on the MFMA training kernels, fp32 master
    weights in the trainer, a checkpoint with fp32 tensors."""
    from dreammat_amd import controlnet_train as ct
    from tests.test_nets_cpu import _tiny_render_tree
    root = tmp_path / "data"
    _tiny_render_tree(root, tmp_path / "prompts.json")
    tr = ct.main(["--synthetic", "--pretrained_model_name_or_path", "tiny", "--train_data_dir", str(root),
                  "--prompt_file", str(tmp_path / "prompts.json"), "--resolution", "64", "--train_batch_size", "2",
                  "--max_train_steps", "2", "--output_dir", str(tmp_path / "out"), "--checkpointing_steps", "2"])
    assert tr.global_step == 2 and tr.master is not None
    assert next(tr.controlnet.parameters()).dtype == torch.bfloat16 and next(tr.controlnet.parameters()).is_cuda
    sd = torch.load(tmp_path / "out" / "controlnet_step2.pt")
    assert all(v.dtype == torch.float32 for k, v in sd.items() if k.endswith("weight"))


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
    budget of the fp32 oracle, backward within 2e-3; CPU-checked through tests/hostemu."""
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


# ---- BASELINE configs[1] (cfg2) on the reference's own in-tree assets: apple.obj, the CC0 HDR probe and
#      the real split-sum FG LUT, copied to tests/golden/assets by tests/golden/make_assets.py
ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "assets")


def _golden_cfg2_env():
    """oracle EnvLight of the real HDR at the reference's settings (scale 2.0, 16..128), prefiltered once on the CPU by
    tests/golden/make_cfg2_env.py (O(res^4): minutes) and stored."""
    gz = np.load(os.path.join(os.path.dirname(ASSETS), "cfg2_env.npz"))
    env = oenv.EnvLight.__new__(oenv.EnvLight)
    env.specular = [torch.from_numpy(gz[f"spec{i}"]) for i in range(4)]
    env.diffuse = torch.from_numpy(gz["diffuse"])
    env.base = torch.from_numpy(gz["base"])
    return env


def _mask_kink_ambiguous(dy, ref, eps=1e-4, eps_relu=3e-4):
    """Zeroes the upstream gradient wherever it could reach a pixel that sits ON A KINK of the reference's own function, where
    two correct fp32 evaluations (different summation orders) land on different sides and disagree by that pixel's WHOLE
    gradient -- a discontinuity, not a rounding error:
      * the clamp(0, 1) of the shaded colour (dreammat_material.py:700; torch.clamp passes the gradient only inside [0, 1]);
        solidly clamped pixels carry no gradient on either side, so masking them loses nothing;
      * a hidden unit of the field MLP within `eps_relu` of its ReLU kink (networks.py:150-187).  The band is as wide as the two
        sides' hidden pre-activations differ: interpolated positions differ by ulps, the finest grid levels scale that by 4096,
        and these tests fill the table with white noise (adjacent entries O(1) apart), so encodings differ by ~1e-4.  tools/grad_budget.py located the single table entry that made up the 9.8e-4 of round 2 this
        way: identical across atlas formats, fast-math on/off and both hash-grid backward routes, and the ORACLE's field
        backward fed with the PRODUCT's per-pixel feature gradients reproduces the oracle's table gradient to 3e-5.
    The antialias blend hands a pixel's gradient to its 4-neighbours' colours, hence the dilation.  (The jittered rows feed only
    the smoothness regulariser, whose per-row gradient is ~1e-8 of the table's.)  Returns (masked dy, fraction of the COVERED
    pixels' channels masked)."""
    B, H, W, _ = dy.shape
    covered = (ref["_rast"][..., 3] > 0)
    col = ref["_color_pre_aa"].detach()
    amb = ((col > 1.0 - eps) | (col < eps)) & covered[..., None]                      # [B,H,W,3]: the clamp acts per channel
    relu = torch.zeros(B * H * W, dtype=torch.bool)
    relu[covered.reshape(-1)] = (ref["_hidden"].abs() < eps_relu).any(-1)
    amb = amb | relu.reshape(B, H, W, 1)
    d = amb.clone()
    d[:, 1:] |= amb[:, :-1]; d[:, :-1] |= amb[:, 1:]; d[:, :, 1:] |= amb[:, :, :-1]; d[:, :, :-1] |= amb[:, :, 1:]
    return dy * (~d).to(dy.dtype), float((d & covered[..., None]).float().sum() / (3 * covered.float().sum()))


def _kink_companion(out, ref, dy_full, dy_masked, hip_table_param, ref_table, dev, others=()):
    """Bounds what happens INSIDE the kink mask: the backward is linear in the upstream gradient, so the unmasked
    table gradient is the masked one (already compared at 1e-3) plus the gradient of the FLAGGED part dy_full - dy_masked,
    obtained here by a second backward on both sides (graphs retained by the caller).  Asserts, with the mask OFF:
      * every table row whose error exceeds 1e-3 of the largest gradient is reachable from a flagged pixel (it receives a
        non-zero contribution from the flagged part on at least one side);
      * such a row's error is no larger than the flagged contributions it receives (a kink moves at most the flagged pixels'
        own contributions from one side to the other) plus the 1e-3 budget;
      * there are no more such rows than flagged pixels x 2 queries x 16 levels x 8 corners.
    Returns the numbers that go into the parity JSON."""
    g_m = ref_table.grad.detach().clone()
    h_m = hip_table_param.grad.detach().cpu().reshape(-1, 2).clone()
    keep = [(p, p.grad.detach().clone()) for p in others]           # the other parameters' masked gradients (restored below)
    dy_flag = dy_full - dy_masked
    flagged_px = int((dy_flag != 0).any(-1).sum())
    ref_table.grad = None
    hip_table_param.grad = None
    (ref["comp_rgb"] * dy_flag).sum().backward()
    (out["comp_rgb"] * dy_flag.to(dev)).sum().backward()
    g_f = ref_table.grad.detach().clone()
    h_f = hip_table_param.grad.detach().cpu().reshape(-1, 2).clone()
    scale = float((g_m + g_f).abs().max())
    err = ((h_m + h_f) - (g_m + g_f)).abs().max(dim=1).values
    bad = err > 1e-3 * scale
    reach = g_f.abs().max(dim=1).values + h_f.abs().max(dim=1).values
    assert bool((reach[bad] > 0).all()), "a table row outside the reach of every kink-flagged pixel disagrees with the mask off"
    assert bool((err[bad] <= 2.0 * reach[bad] + 1e-3 * scale).all()), "disagreement larger than the flagged pixels' own contributions"
    assert int(bad.sum()) <= flagged_px * 2 * 16 * 8, (int(bad.sum()), flagged_px)
    # restore the masked gradients for the caller's own comparison
    ref_table.grad = g_m
    hip_table_param.grad = h_m.reshape(hip_table_param.shape).to(hip_table_param.device)
    for p, g in keep:
        p.grad = g
    return {"unmasked_table_grad_rel_err": float(err.max() / scale), "table_rows_above_1e-3_with_mask_off": int(bad.sum()),
            "flagged_pixels": flagged_px, "all_such_rows_reachable_from_a_flagged_pixel": True,
            "flagged_part_rel_err": float((h_f - g_f).abs().max() / scale)}



def _cfg2_compare(dev, texel=None, oracle_cache=None, strict=True, mask_clamp=True):
    """scene + comparison of test_cfg2_real_assets_render_vs_oracle; `texel` overrides the atlas storage format (the gradient
    error budget study of tools/grad_budget.py runs it with "fp32"), `oracle_cache` (dict) keeps the CPU oracle's result."""
    import hashlib
    import dreammat_amd
    from dreammat_amd.geometry import DreamMatMesh
    from dreammat_amd.material import DreamMatMaterial
    from dreammat_amd.renderer import RaytraceRender
    from dreammat_amd.background import SolidColorBackground
    sums = json.load(open(os.path.join(ASSETS, "SHA256.json")))
    for name, want in sums.items():
        assert hashlib.sha256(open(os.path.join(ASSETS, name), "rb").read()).hexdigest() == want, name
    assert sums["bsdf_256_256.bin"].startswith("aee514f7c7e5")
    torch.manual_seed(0)
    geom = DreamMatMesh({"shape_init": "mesh:" + os.path.join(ASSETS, "apple.obj"), "shape_init_params": 0.7,
                         "shape_init_mesh_up": "+y", "shape_init_mesh_front": "+z"}).to(dev)
    assert geom.t_buffer.shape[0] == 4164 and geom.encoding.spec.n_params == 12599920
    with torch.no_grad():
        geom.encoding.encoding.params.copy_(torch.rand_like(geom.encoding.encoding.params) * 2 - 1)
        geom.feature_network.layers[0].weight.copy_(torch.randn(64, 32) * 0.3)
        geom.feature_network.layers[2].weight.copy_(torch.randn(5, 64) * 0.3)
    if texel is not None:
        os.environ["DREAMMAT_ATLAS"] = texel
    try:
        mat = DreamMatMaterial({"use_raytracing": False, "environment_scale": 2.0, "n_envs": 1,
                                "environment_texture": os.path.join(ASSETS, "mud_road_puresky_1k.hdr"),
                                "fg_lut_path": os.path.join(ASSETS, "bsdf_256_256.bin")}).to(dev)
    finally:
        if texel is not None:
            del os.environ["DREAMMAT_ATLAS"]
    fg = penv.load_fg_lut(os.path.join(ASSETS, "bsdf_256_256.bin"))
    assert torch.equal(mat.FG_LUT[0].cpu(), fg) and abs(float(fg[0, 0, 0]) - 0.00973) < 1e-5
    golden_env = _golden_cfg2_env()
    assert mat.atlas.mip_res == [128, 64, 32, 16] and mat.atlas.texel == (texel or "rgb18e8")
    for k in range(4):
        a, b = mat.atlas.specular[0][k].cpu(), golden_env.specular[k]
        assert (a - b).abs().max() <= 1e-4 * b.abs().max(), k
    assert (mat.atlas.diffuse[0].cpu() - golden_env.diffuse).abs().max() < 1e-5 * max(1.0, float(golden_env.diffuse.max()))
    rend = RaytraceRender({}, geometry=geom, material=mat, background=SolidColorBackground({}))
    B, H, W = 4, 512, 512
    batch = util.make_views(B, H, W, seed=4)
    batch["env_id"] = torch.zeros(B, dtype=torch.long)
    g = torch.Generator().manual_seed(9)
    ju, jn = torch.rand(B, H, W, generator=g), torch.randn(B, H, W, generator=g)
    gbatch = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    out = rend(**gbatch, light_positions=None, jitter_u=ju.to(dev), jitter_n=jn.to(dev), check_overflow=True)
    m = geom.mesh
    md = dict(v_pos=geom.v_buffer.cpu().numpy(), v_nrm=geom.vnrm_buffer.cpu().numpy(),
              t_pos_idx=geom.t_buffer.cpu().numpy().astype(np.int32))
    md["opp"] = oraster.build_topology(md["t_pos_idx"])
    lv, tot = ofield.grid_levels()
    table = geom.encoding.encoding.params.detach().cpu().reshape(-1, 2).clone().requires_grad_()
    w1 = geom.feature_network.layers[0].weight.detach().cpu().clone().requires_grad_()
    w2 = geom.feature_network.layers[2].weight.detach().cpu().clone().requires_grad_()
    dy = torch.randn(B, H, W, 3, generator=g)
    if oracle_cache is not None and "ref" in oracle_cache:
        ref, table, w1, w2, dy, masked = oracle_cache["ref"]
    else:
        ref = orender.render(md, batch, dict(table=table, w1=w1, w2=w2, levels=lv, radius=1.0), [golden_env], fg, ju, jn)
        dy_full = dy
        dy, masked = _mask_kink_ambiguous(dy, ref) if mask_clamp else (dy, 0.0)
        if oracle_cache is not None:
            ref["_features"].retain_grad(); ref["_features_jitter"].retain_grad()
        ((ref["comp_rgb"] * dy).sum() + 2.0 * ref["loss_mat_reg"]).backward(retain_graph=True)
        if oracle_cache is not None:
            oracle_cache["ref"] = (ref, table, w1, w2, dy, masked)
    assert np.array_equal(out["_internals"]["rast"].cpu().numpy().view(np.uint32), ref["_rast"].numpy().view(np.uint32))
    cover = float((ref["opacity"] > 0).float().mean())
    assert 0.1 < cover < 0.6
    errs = {}
    for k in ["comp_rgb", "opacity", "comp_depth", "comp_normal", "albedo", "metalness", "roughness", "specular_light",
              "diffuse_light", "specular_color", "diffuse_color"]:
        errs[k] = (out[k].detach().cpu() - ref[k].detach()).abs().max().item()
        assert errs[k] < 1e-3, (k, errs[k])
    mse = ((out["comp_rgb"].detach().cpu() - ref["comp_rgb"].detach()) ** 2).mean().item()
    psnr = 10 * np.log10(1.0 / max(mse, 1e-20))
    assert psnr > 80, psnr
    if oracle_cache is not None:          # per-pixel feature gradients, for locating a disagreement (tools/grad_budget.py)
        out["_internals"]["features"].retain_grad(); out["_internals"]["features_jitter"].retain_grad()
    ((out["comp_rgb"] * dy.to(dev)).sum() + 2.0 * out["loss_mat_reg"]).backward(retain_graph=True)
    companion = None
    if mask_clamp and oracle_cache is None:
        companion = _kink_companion(out, ref, dy_full, dy, geom.encoding.encoding.params, table, dev,
                                    others=(w1, w2, geom.feature_network.layers[0].weight, geom.feature_network.layers[2].weight))
    if oracle_cache is not None and ref["_features"].grad is not None:
        rm = _rm(out["_internals"]["gbuffer"])
        for nm_f, hf, of in (("features", out["_internals"]["features"], ref["_features"]),
                             ("features_jitter", out["_internals"]["features_jitter"], ref["_features_jitter"])):
            hf = _RowView(hf, rm)                             # the product's rows in the oracle's row-major order
            d = (hf.grad.cpu() - of.grad).abs().max(dim=1).values
            top = torch.topk(d, 3).indices
            sel = (ref["_rast"][..., 3] > 0).reshape(-1).nonzero()[:, 0]
            oracle_cache["pixel_detail_" + nm_f] = [
                {"row": int(i), "pixel": int(sel[i]), "hip_dfeat": hf.grad[i].cpu().tolist(), "oracle_dfeat": of.grad[i].tolist(),
                 "oracle_feat": of[i].tolist(), "hip_feat": hf[i].detach().cpu().tolist(),
                 "oracle_color_pre_aa": ref["_color_pre_aa"].reshape(-1, 3)[sel[i]].tolist(),
                 "roughness": float(ref["roughness"].reshape(-1)[sel[i]]), "dy": dy.reshape(-1, 3)[sel[i]].tolist()} for i in top]
    if oracle_cache is not None and oracle_cache.get("split_field_backward"):
        # the ORACLE's field backward fed with the PRODUCT's per-pixel feature gradients: separates an upstream disagreement
        # (shade / antialias) from one inside the hash-grid + MLP backward
        sel_b = (ref["_rast"][..., 3] > 0).reshape(-1)
        t2 = table.detach().clone().requires_grad_()
        f2 = ofield.field_forward(ref["_gb_pos"].reshape(-1, 3)[sel_b], t2, w1.detach(), w2.detach(), lv, 1.0)
        fj2 = ofield.field_forward(ref["_positions_jitter"].detach(), t2, w1.detach(), w2.detach(), lv, 1.0)
        rm = _rm(out["_internals"]["gbuffer"])
        ((f2 * out["_internals"]["features"].grad.cpu()[rm]).sum() + (fj2 * out["_internals"]["features_jitter"].grad.cpu()[rm]).sum()).backward()
        th = geom.encoding.encoding.params.grad.cpu().reshape(-1, 2)
        oracle_cache["split"] = {"oracle_bwd_of_hip_dfeat_vs_hip_table": float((t2.grad - th).abs().max() / table.grad.abs().max()),
                                 "oracle_bwd_of_hip_dfeat_vs_oracle_table": float((t2.grad - table.grad).abs().max() / table.grad.abs().max())}
    rels = {}
    for a, b, nm in ((geom.encoding.encoding.params.grad.cpu().reshape(-1, 2), table.grad, "table"),
                     (geom.feature_network.layers[0].weight.grad.cpu(), w1.grad, "w1"),
                     (geom.feature_network.layers[2].weight.grad.cpu(), w2.grad, "w2")):
        rels[nm] = ((a - b).abs().max() / b.abs().max()).item()
        assert rels[nm] < 1e-3 or not strict, (nm, rels[nm])
        if nm == "table" and oracle_cache is not None:       # where the largest table-gradient error sits (tools/grad_budget.py)
            i = int((a - b).abs().max(dim=1).values.argmax())
            offs = [l["offset"] for l in lv] if isinstance(lv[0], dict) else None
            oracle_cache["table_detail"] = {"entry": i, "hip": a[i].tolist(), "oracle": b[i].tolist(),
                                            "oracle_abs_max": float(b.abs().max()), "level_offsets": offs,
                                            "rel_err_of_that_entry": float(((a[i] - b[i]).abs().max() / b[i].abs().max().clamp(min=1e-30)))}
    return {"psnr_db": psnr, "coverage_ids_equal": True, "covered_fraction": cover, "max_abs_err": errs,
            "grad_rel_err": rels, "assets": sums, "atlas_texel": mat.atlas.texel,
            "kink_ambiguous_fraction_of_covered_channels_masked_in_dy": masked, "mask_off_companion": companion}




def test_cfg2_real_assets_render_vs_oracle(dev):
    """run_examples.sh:2 (`shape_init=mesh:load/shapes/objs/apple.obj shape_init_params=0.7`) with dreammat.yaml's geometry
    (+y up, +z front, 16-level 2^19 hash grid), environment_scale 2.0 and the real FG LUT (dreammat_material.py:383,399-404):
    4 views @512^2 through the plugin API against the oracle -- coverage bit-equal, shaded pixels within 1e-3, gradients
    within 1e-3; the product's GPU-built environment atlas against the oracle's prefilter of the same HDR."""
    res = _cfg2_compare(dev)
    with open(os.path.join(OUT, "cfg2_render_parity.json"), "w") as fh:
        json.dump(res, fh)


def test_cfg3_bench_scene_render_vs_oracle(dev):
    """The BENCH configuration end to end (BASELINE configs[2]): the 50 880-triangle displaced sphere
    (`sphere:160:160`), 8 views @512^2, the five seeded synthetic probes of bench.py at cube resolutions 16..128, the real FG
    LUT, the full 16-level 2^19 hash grid, bin-overflow check on -- through the plugin API against the oracle: coverage ids /
    (u, v, z/w) bit-equal, all 11 renderer outputs within 1e-3, hash-table / MLP gradients within 1e-3.
    The oracle's O(res^4) prefilter of ALL FIVE probes is stored (tests/golden/cfg3_env{0..4}.npz, make_cfg3_env.py): the
    product's GPU prefilter is checked against it probe by probe, and the oracle renders from its own cubes."""
    import bench
    from dreammat_amd.geometry import DreamMatMesh
    from dreammat_amd.material import DreamMatMaterial
    from dreammat_amd.renderer import RaytraceRender
    from dreammat_amd.background import SolidColorBackground
    torch.manual_seed(0)
    geom = DreamMatMesh({"shape_init": "sphere:160:160", "shape_init_params": 0.8}).to(dev)
    assert geom.t_buffer.shape[0] == 50880 and geom.encoding.spec.n_params == 12599920
    with torch.no_grad():
        geom.encoding.encoding.params.copy_(torch.rand_like(geom.encoding.encoding.params) * 2 - 1)
        geom.feature_network.layers[0].weight.copy_(torch.randn(64, 32) * 0.3)
        w2_init = torch.randn(5, 64) * 0.3
        # dark albedo (features <= 0 after the ReLU layer): under the bench's probes (radiance ~1.6 x scale 2, 100x sun lobes) a
        # mid-grey material saturates 98 % of the covered channels at the clamp(0, 1), where no gradient flows at all
        w2_init[:3] = -w2_init[:3].abs() * 2.0
        geom.feature_network.layers[2].weight.copy_(w2_init)
    lat = [bench.synthetic_latlong(i) for i in range(5)]
    fg_path = os.path.join(ASSETS, "bsdf_256_256.bin")
    mat = DreamMatMaterial({"use_raytracing": False, "environment_scale": 2.0, "env_max_res": 128, "env_min_res": 16,
                            "n_envs": 5, "fg_lut_path": fg_path}, latlongs=lat).to(dev)
    fg = penv.load_fg_lut(fg_path)
    assert mat.atlas.mip_res == [128, 64, 32, 16] and mat.atlas.texel == "rgb18e8" and mat.real_fg_lut
    # every probe's GPU-side prefilter against the ORACLE's O(res^4) CPU prefilter of the same lat-long map
    # (tests/golden/cfg3_env{0..4}.npz, make_cfg3_env.py); the oracle then renders from its OWN cubes for all five
    oenvs, prefilter_err = [], []
    for e in range(5):
        gz = np.load(os.path.join(os.path.dirname(ASSETS), f"cfg3_env{e}.npz"))
        for k in range(4):
            a, b = mat.atlas.specular[e][k].cpu(), torch.from_numpy(gz[f"spec{k}"])
            # mips 1..3 agree to 1e-6.  Mip 0 is the GGX prefilter at roughness 0.08 (alpha^2 = 4e-5) of a probe with a 100x sun lobe:
            # D(h) = a2 / (pi ((n.h)^2 (a2 - 1) + 1)^2) cancels catastrophically in fp32 next to n.h = 1, so two correct fp32
            # evaluations (CPU oracle, GPU product) differ by up to ~1e-3 of a texel on the lobe's flank (measured 8.6e-4 at one
            # texel, mean 5.5e-5); the real HDR of cfg2 (no such lobe) holds 1e-4 of the maximum
            tol = 5e-4 if k == 0 else 1e-5
            assert (a - b).abs().max() <= tol * b.abs().max(), (e, k, float((a - b).abs().max()), float(b.abs().max()))
            assert ((a - b).abs() / b.abs().clamp(min=1e-3)).mean() < 2e-4, (e, k)
            prefilter_err.append(float((a - b).abs().max() / b.abs().max()))
        assert (mat.atlas.diffuse[e].cpu() - torch.from_numpy(gz["diffuse"])).abs().max() < 1e-5 * max(1.0, float(gz["diffuse"].max()))
        env = oenv.EnvLight.__new__(oenv.EnvLight)
        env.specular = [torch.from_numpy(gz[f"spec{k}"]) for k in range(4)]
        env.diffuse = torch.from_numpy(gz["diffuse"])
        env.base = env.specular[0]
        oenvs.append(env)
    rend = RaytraceRender({}, geometry=geom, material=mat, background=SolidColorBackground({}))
    B, H, W = 8, 512, 512
    batch = util.make_views(B, H, W, seed=0)
    batch["env_id"] = torch.tensor([0, 3, 1, 4, 2, 0, 2, 1], dtype=torch.long)
    g = torch.Generator().manual_seed(11)
    ju, jn = torch.rand(B, H, W, generator=g), torch.randn(B, H, W, generator=g)
    gbatch = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    out = rend(**gbatch, light_positions=None, jitter_u=ju.to(dev), jitter_n=jn.to(dev), check_overflow=True)
    md = dict(v_pos=geom.v_buffer.cpu().numpy(), v_nrm=geom.vnrm_buffer.cpu().numpy(),
              t_pos_idx=geom.t_buffer.cpu().numpy().astype(np.int32))
    md["opp"] = oraster.build_topology(md["t_pos_idx"])
    lv, tot = ofield.grid_levels()
    table = geom.encoding.encoding.params.detach().cpu().reshape(-1, 2).clone().requires_grad_()
    w1 = geom.feature_network.layers[0].weight.detach().cpu().clone().requires_grad_()
    w2 = geom.feature_network.layers[2].weight.detach().cpu().clone().requires_grad_()
    ref = orender.render(md, batch, dict(table=table, w1=w1, w2=w2, levels=lv, radius=1.0), oenvs, fg, ju, jn)
    assert np.array_equal(out["_internals"]["rast"].cpu().numpy().view(np.uint32), ref["_rast"].numpy().view(np.uint32))
    cover = float((ref["opacity"] > 0).float().mean())
    assert 0.3 < cover < 0.7, cover
    errs = {}
    for k in ["comp_rgb", "opacity", "comp_depth", "comp_normal", "albedo", "metalness", "roughness", "specular_light",
              "diffuse_light", "specular_color", "diffuse_color"]:
        errs[k] = (out[k].detach().cpu() - ref[k].detach()).abs().max().item()
        assert errs[k] < 1e-3, (k, errs[k])
    mse = ((out["comp_rgb"].detach().cpu() - ref["comp_rgb"].detach()) ** 2).mean().item()
    psnr = 10 * np.log10(1.0 / max(mse, 1e-20))
    assert psnr > 80, psnr
    dy = torch.randn(B, H, W, 3, generator=g)
    dy_full = dy
    dy, masked = _mask_kink_ambiguous(dy, ref)      # (the 100x sun lobes saturate many pixels)
    assert masked < 0.5, masked                     # the gradient comparison must keep a substantial part of the image
    ((out["comp_rgb"] * dy.to(dev)).sum() + 2.0 * out["loss_mat_reg"]).backward(retain_graph=True)
    ((ref["comp_rgb"] * dy).sum() + 2.0 * ref["loss_mat_reg"]).backward(retain_graph=True)
    companion = _kink_companion(out, ref, dy_full, dy, geom.encoding.encoding.params, table, dev,
                                others=(w1, w2, geom.feature_network.layers[0].weight, geom.feature_network.layers[2].weight))
    rels = {}
    for a, b, nm in ((geom.encoding.encoding.params.grad.cpu().reshape(-1, 2), table.grad, "table"),
                     (geom.feature_network.layers[0].weight.grad.cpu(), w1.grad, "w1"),
                     (geom.feature_network.layers[2].weight.grad.cpu(), w2.grad, "w2")):
        rels[nm] = ((a - b).abs().max() / b.abs().max()).item()
        assert rels[nm] < 1e-3, (nm, rels[nm])
    errs["kink_ambiguous_fraction_of_covered_channels_masked_in_dy"] = masked
    with open(os.path.join(OUT, "cfg3_render_parity.json"), "w") as fh:
        json.dump({"config": "BASELINE configs[2]: sphere:160:160, 8 views @512^2, 5 probes @128, 16 x 2^19 grid",
                   "psnr_db": psnr, "coverage_ids_equal": True, "covered_fraction": cover, "max_abs_err": errs,
                   "grad_rel_err": rels, "mask_off_companion": companion,
                   "prefilter_max_rel_err_per_probe_and_mip": prefilter_err}, fh)


def test_cfg5_mesh_1024_render_vs_oracle(dev, envs):
    """BASELINE configs[4]'s render side (cfg5 had no render-parity evidence): the 200 320-triangle displaced sphere
    (`sphere:320:314`) at 1024^2, two views (two of cfg5's sixteen: the oracle needs minutes per view at this size), the full
    16-level 2^19 hash grid, bin-overflow check on -- through the plugin API against the oracle: coverage ids / (u, v, z/w)
    bit-equal, all 11 renderer outputs within 1e-3, hash-table / MLP gradients within 1e-3 (kink mask + mask-off companion)."""
    from dreammat_amd.geometry import DreamMatMesh
    from dreammat_amd.material import DreamMatMaterial
    from dreammat_amd.renderer import RaytraceRender
    from dreammat_amd.background import SolidColorBackground
    lat, fg, oenvs = envs
    torch.manual_seed(0)
    geom = DreamMatMesh({"shape_init": "sphere:320:314", "shape_init_params": 0.8}).to(dev)
    assert geom.t_buffer.shape[0] == 200320 and geom.encoding.spec.n_params == 12599920
    with torch.no_grad():
        geom.encoding.encoding.params.copy_(torch.rand_like(geom.encoding.encoding.params) * 2 - 1)
        geom.feature_network.layers[0].weight.copy_(torch.randn(64, 32) * 0.3)
        geom.feature_network.layers[2].weight.copy_(torch.randn(5, 64) * 0.3)
    mat = DreamMatMaterial({"use_raytracing": False, "environment_scale": 2.0, "env_max_res": 32, "env_min_res": 8,
                            "n_envs": 3}, latlongs=lat).to(dev)
    rend = RaytraceRender({}, geometry=geom, material=mat, background=SolidColorBackground({}))
    B, H, W = 2, 1024, 1024
    batch = util.make_views(B, H, W, seed=3)
    batch["env_id"] = torch.tensor([2, 0], dtype=torch.long)
    g = torch.Generator().manual_seed(12)
    ju, jn = torch.rand(B, H, W, generator=g), torch.randn(B, H, W, generator=g)
    gbatch = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    out = rend(**gbatch, light_positions=None, jitter_u=ju.to(dev), jitter_n=jn.to(dev), check_overflow=True)
    md = dict(v_pos=geom.v_buffer.cpu().numpy(), v_nrm=geom.vnrm_buffer.cpu().numpy(),
              t_pos_idx=geom.t_buffer.cpu().numpy().astype(np.int32))
    md["opp"] = oraster.build_topology(md["t_pos_idx"])
    lv, tot = ofield.grid_levels()
    table = geom.encoding.encoding.params.detach().cpu().reshape(-1, 2).clone().requires_grad_()
    w1 = geom.feature_network.layers[0].weight.detach().cpu().clone().requires_grad_()
    w2 = geom.feature_network.layers[2].weight.detach().cpu().clone().requires_grad_()
    ref = orender.render(md, batch, dict(table=table, w1=w1, w2=w2, levels=lv, radius=1.0), oenvs, fg, ju, jn)
    assert np.array_equal(out["_internals"]["rast"].cpu().numpy().view(np.uint32), ref["_rast"].numpy().view(np.uint32))
    cover = float((ref["opacity"] > 0).float().mean())
    assert 0.3 < cover < 0.7, cover
    errs = {}
    for k in ["comp_rgb", "opacity", "comp_depth", "comp_normal", "albedo", "metalness", "roughness", "specular_light",
              "diffuse_light", "specular_color", "diffuse_color"]:
        errs[k] = (out[k].detach().cpu() - ref[k].detach()).abs().max().item()
        assert errs[k] < 1e-3, (k, errs[k])
    mse = ((out["comp_rgb"].detach().cpu() - ref["comp_rgb"].detach()) ** 2).mean().item()
    psnr = 10 * np.log10(1.0 / max(mse, 1e-20))
    assert psnr > 80, psnr
    dy_full = torch.randn(B, H, W, 3, generator=g)
    dy, masked = _mask_kink_ambiguous(dy_full, ref)
    assert masked < 0.6, masked
    ((out["comp_rgb"] * dy.to(dev)).sum() + 2.0 * out["loss_mat_reg"]).backward(retain_graph=True)
    ((ref["comp_rgb"] * dy).sum() + 2.0 * ref["loss_mat_reg"]).backward(retain_graph=True)
    companion = _kink_companion(out, ref, dy_full, dy, geom.encoding.encoding.params, table, dev,
                                others=(w1, w2, geom.feature_network.layers[0].weight, geom.feature_network.layers[2].weight))
    rels = {}
    for a, b, nm in ((geom.encoding.encoding.params.grad.cpu().reshape(-1, 2), table.grad, "table"),
                     (geom.feature_network.layers[0].weight.grad.cpu(), w1.grad, "w1"),
                     (geom.feature_network.layers[2].weight.grad.cpu(), w2.grad, "w2")):
        rels[nm] = ((a - b).abs().max() / b.abs().max()).item()
        assert rels[nm] < 1e-3, (nm, rels[nm])
    with open(os.path.join(OUT, "cfg5_render_parity.json"), "w") as fh:
        json.dump({"config": "BASELINE configs[4] render side: sphere:320:314 (200 320 triangles), 2 views @1024^2, 16 x 2^19 grid",
                   "psnr_db": psnr, "coverage_ids_equal": True, "covered_fraction": cover, "max_abs_err": errs,
                   "grad_rel_err": rels, "kink_ambiguous_fraction_of_covered_channels_masked_in_dy": masked,
                   "mask_off_companion": companion}, fh)


def test_full_size_sd15_unet_controlnet_eps_vs_oracle(dev):
    """The SD-1.5 shape set north_star names (8 heads of 40 / 80 / 160 / 160, 768-wide context, conv projections; the reference's
    fallback control types name sd15 ControlNets, dreammat_guidance.py:103-106) at full size, one branch item: fp32 on the GPU within
    1e-3 of the CPU oracle, the IEEE-half production path (staged MFMA attention at D = 40 / 80 / 160, conv / GroupNorm kernels)
    within half's rounding class -- VERDICT r5 row N3 / next #10."""
    from dreammat_amd.sd import ARCHS, ControlNetModel, UNet2DConditionModel
    from dreammat_amd.sd import layers
    from oracle import sd_nets as osd
    a = ARCHS["sd15"]
    torch.manual_seed(0)
    with torch.device(dev):
        unet = UNet2DConditionModel(a).eval()
        cn = ControlNetModel.from_unet(unet).eval()
    for conv in list(cn.controlnet_down_blocks) + [cn.controlnet_mid_block, cn.controlnet_cond_embedding.conv_out]:
        torch.nn.init.normal_(conv.weight, std=0.02)
    for p in list(unet.parameters()) + list(cn.parameters()):
        p.requires_grad_(False)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 4, 64, 64, generator=g); t = torch.tensor([437])
    ctx = torch.randn(1, 77, a.cross_dim, generator=g); cond = torch.rand(1, 22, 512, 512, generator=g)
    with torch.no_grad():
        d, m = cn(x.to(dev), t.to(dev), ctx.to(dev), cond.to(dev), 1.0)
        y = unet(x.to(dev), t.to(dev), ctx.to(dev), d, m).cpu()
        sd_u = {k: v.float().cpu() for k, v in unet.state_dict().items()}
        sd_c = {k: v.float().cpu() for k, v in cn.state_dict().items()}
        od, om = osd.controlnet_forward(sd_c, x, t, ctx, cond, 1.0, a.heads, a.use_linear_projection)
        oy = osd.unet_forward(sd_u, x, t, ctx, a.heads, a.use_linear_projection, od, om)
        del sd_u, sd_c
        rel32 = ((y - oy).abs().max() / oy.abs().max()).item()
        assert rel32 <= 1e-3, rel32
        layers.fallbacks(clear=True)
        unet.half(); cn.half()
        hipops.enable_kernel_timing(True, only=("attention",))
        d, m = cn(x.to(dev).half(), t.to(dev), ctx.to(dev).half(), cond.to(dev).half(), 1.0)
        yh = unet(x.to(dev).half(), t.to(dev), ctx.to(dev).half(), d, m).float().cpu()
        torch.cuda.synchronize()
        kt = hipops.kernel_times()
        hipops.enable_kernel_timing(False)
        left = layers.fallbacks()
    assert sum(v["launches"] for k, v in kt.items() if k.startswith("attention")) == 46
    assert not left, left
    relh = ((yh - oy).abs().max() / oy.abs().max()).item()
    relh_mean = ((yh - oy).abs().mean() / oy.abs().mean()).item()
    with open(os.path.join(OUT, "full_size_eps_parity_sd15.json"), "w") as fh:
        json.dump({"arch": "sd15", "fp32_rel_max": rel32, "f16_rel_max": relh, "f16_rel_mean": relh_mean,
                   "eps_abs_max": float(oy.abs().max())}, fh)
    assert torch.isfinite(yh).all() and relh < 4e-3 and relh_mean < 3e-3, (relh, relh_mean)


def test_full_size_sd21_unet_controlnet_eps_vs_oracle(dev):
    """One branch-item of the guidance's frozen stack at the bench's real shapes (SD-2.1-base UNet 865.9 M parameters + the
    22-channel ControlNet, 64^2 latents, S = 4096 self-attention, 512^2 condition maps; seeded random weights -- no
    checkpoints on this box): fp32 on the GPU within 1e-3 of the functional CPU oracle (north_star), and the bf16 production
    path (MFMA attention / conv / GroupNorm kernels) against the same oracle within bf16 rounding."""
    from dreammat_amd.sd import ARCHS, ControlNetModel, UNet2DConditionModel
    from oracle import sd_nets as osd
    a = ARCHS["sd21-base"]
    torch.manual_seed(0)
    with torch.device(dev):
        unet = UNet2DConditionModel(a).eval()
        cn = ControlNetModel.from_unet(unet).eval()
    for conv in list(cn.controlnet_down_blocks) + [cn.controlnet_mid_block, cn.controlnet_cond_embedding.conv_out]:
        torch.nn.init.normal_(conv.weight, std=0.02)
    for p in list(unet.parameters()) + list(cn.parameters()):
        p.requires_grad_(False)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 4, 64, 64, generator=g); t = torch.tensor([437])
    ctx = torch.randn(1, 77, a.cross_dim, generator=g); cond = torch.rand(1, 22, 512, 512, generator=g)
    with torch.no_grad():
        d, m = cn(x.to(dev), t.to(dev), ctx.to(dev), cond.to(dev), 1.0)
        y = unet(x.to(dev), t.to(dev), ctx.to(dev), d, m).cpu()
        sd_u = {k: v.float().cpu() for k, v in unet.state_dict().items()}
        sd_c = {k: v.float().cpu() for k, v in cn.state_dict().items()}
        od, om = osd.controlnet_forward(sd_c, x, t, ctx, cond, 1.0, a.heads, a.use_linear_projection)
        oy = osd.unet_forward(sd_u, x, t, ctx, a.heads, a.use_linear_projection, od, om)
        del sd_u, sd_c
        rel32 = ((y - oy).abs().max() / oy.abs().max()).item()
        assert rel32 <= 1e-3, rel32
        master = [{k: v.clone() for k, v in net.state_dict().items()} for net in (unet, cn)]     # fp32: each 16-bit leg casts from these
        from dreammat_amd.sd import layers
        layers.fallbacks(clear=True)
        unet.bfloat16(); cn.bfloat16()
        hipops.enable_kernel_timing(True)
        d, m = cn(x.to(dev).bfloat16(), t.to(dev), ctx.to(dev).bfloat16(), cond.to(dev).bfloat16(), 1.0)
        yb = unet(x.to(dev).bfloat16(), t.to(dev), ctx.to(dev).bfloat16(), d, m).float().cpu()
        torch.cuda.synchronize()
        kt = hipops.kernel_times()
        hipops.enable_kernel_timing(False)
        # IEEE half (round 5): the reference's own precision class (half_precision_weights, dreammat_guidance.py:56) through the
        # f16 instantiations of the same kernels -- 11 significant bits where bf16 has 8
        left_bf16 = layers.fallbacks(clear=True)
        unet.float(); cn.float()
        unet.load_state_dict(master[0]); cn.load_state_dict(master[1])        # (half of the bf16-ROUNDED weights would carry bf16's error)
        del master
        unet.half(); cn.half()
        d, m = cn(x.to(dev).half(), t.to(dev), ctx.to(dev).half(), cond.to(dev).half(), 1.0)
        yh = unet(x.to(dev).half(), t.to(dev), ctx.to(dev).half(), d, m).float().cpu()
        left = layers.fallbacks()
        # f16 nets + MX-FP8 self-attention (guidance.attention_precision: fp8, BASELINE configs[4]; VERDICT r5 missing #6: what the
        # switch does to the noise prediction at full size, not to one block) -- the S >= 1024 self-attention layers of both nets
        layers.set_attention_precision(unet, "fp8"); layers.set_attention_precision(cn, "fp8")
        hipops.enable_kernel_timing(True, only=("attention",))
        d, m = cn(x.to(dev).half(), t.to(dev), ctx.to(dev).half(), cond.to(dev).half(), 1.0)
        y8 = unet(x.to(dev).half(), t.to(dev), ctx.to(dev).half(), d, m).float().cpu()
        torch.cuda.synchronize()
        kt8 = hipops.kernel_times()
        hipops.enable_kernel_timing(False)
        layers.set_attention_precision(unet, "16bit"); layers.set_attention_precision(cn, "16bit")
    n_fp8 = sum(v["launches"] for k, v in kt8.items() if k.startswith("attention_fwd_fp8"))
    assert n_fp8 == 14, kt8.keys()              # the S = 4096 and S = 1024 self-attention layers: 5 + 5 in the UNet, 2 + 2 in the ControlNet
    rel8 = ((y8 - oy).abs().max() / oy.abs().max()).item()
    rel8_mean = ((y8 - oy).abs().mean() / oy.abs().mean()).item()
    assert sum(v["launches"] for k, v in kt.items() if k.startswith("attention")) == 46
    assert any(k.startswith("conv3x3") for k in kt)
    assert not left and not left_bf16, (left_bf16, left)   # no layer of the frozen nets left the hand-written kernels, in either type
    rel16 = ((yb - oy).abs().max() / oy.abs().max()).item()
    rel16_mean = ((yb - oy).abs().mean() / oy.abs().mean()).item()
    relh = ((yh - oy).abs().max() / oy.abs().max()).item()
    relh_mean = ((yh - oy).abs().mean() / oy.abs().mean()).item()
    with open(os.path.join(OUT, "full_size_eps_parity.json"), "w") as fh:
        json.dump({"fp32_rel_max": rel32, "bf16_rel_max": rel16, "bf16_rel_mean": rel16_mean, "f16_rel_max": relh,
                   "f16_rel_mean": relh_mean, "f16+fp8attn_rel_max": rel8, "f16+fp8attn_rel_mean": rel8_mean,
                   "fp8_attention_layers": n_fp8, "eps_abs_max": float(oy.abs().max())}, fh)
    assert rel16 < 2e-2 and rel16_mean < 2e-2, (rel16, rel16_mean)     # measured 1.0-1.2e-2 / 0.9e-2 (rounds 2, 3)
    # measured (round 5): 1.27e-3 max / 1.15e-3 mean -- an eighth of bf16's error at every module of the stack
    # (tools/f16_error_profile.py); the first measurement read 5.4e-3 because this test had handed the f16 leg bf16-rounded weights
    assert torch.isfinite(yh).all() and relh < 2.5e-3 and relh_mean < 2e-3 and relh_mean < 0.25 * rel16_mean, (relh, relh_mean)
    # fp8 attention (e4m3 operands: 3 mantissa bits on Q, K, P, V of 14 layers): reported, gated loosely -- an experimental precision
    # class, far outside north_star's 1e-3
    assert torch.isfinite(y8).all() and rel8 < 0.2 and rel8_mean < 0.1, (rel8, rel8_mean)
