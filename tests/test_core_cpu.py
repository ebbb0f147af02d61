"""CPU suite (-m "not gpu"): oracle self-checks and parity of the product's per-thread
kernel cores (compiled for the host by tests/hostemu) against the oracle."""
import ctypes

import numpy as np
import pytest
import torch

from dreammat_amd import _lib, envlight as penv, mesh as pmesh
from oracle import camera, envlight as oenv, field as ofield, raster as oraster, shading as oshade
from tests import util
from tests.util import P


def _scene(kind, B, H, W, seed=1):
    if kind == "quad":
        m = pmesh.quad_mesh()
        batch = camera.camera_batch(torch.tensor([80.0]), torch.tensor([10.0]), torch.tensor([3.5]),
                                    torch.tensor([35.0]), H, W)
    else:
        m = pmesh.displaced_sphere(*kind)
        batch = util.make_views(B, H, W, seed)
    md = util.mesh_dict(m)
    pos = oraster.vertex_transform(md["v_pos"], batch["mvp_mtx"]).numpy()
    return md, batch, np.ascontiguousarray(pos, np.float32)


@pytest.mark.parametrize("kind,B,H,W", [("quad", 1, 64, 64), ((48, 40), 3, 128, 128), ((160, 160), 2, 256, 256),
                                        ((24, 16), 2, 67, 45)])
def test_raster_core_bit_exact_vs_oracle(hostemu, kind, B, H, W):
    md, batch, pos = _scene(kind, B, H, W)
    tri = md["t_pos_idx"]
    ro = oraster.rasterize(pos, tri, H, W)
    re = np.empty_like(ro)
    hostemu.emu_rasterize(P(pos), B, pos.shape[1], P(tri), tri.shape[0], H, W, P(re))
    assert (ro[..., 3] > 0).mean() > 0.05
    assert np.array_equal(ro[..., 3], re[..., 3]), "coverage ids differ"
    assert np.array_equal(ro.view(np.uint32), re.view(np.uint32)), "u/v/zw not bit-identical"
    po = oraster.antialias_plan(pos, tri, md["opp"], ro)
    pe = np.empty_like(po)
    hostemu.emu_aa_plan(P(pos), B, pos.shape[1], P(tri), P(md["opp"]), P(ro), H, W, P(pe))
    assert np.array_equal(po.view(np.uint32), pe.view(np.uint32))
    assert (po != 0).sum() > 10


def test_oracle_raster_vs_bruteforce_float64():
    """away from edges the snapped integer rasterizer equals a float64 point-in-triangle."""
    md, batch, pos = _scene((32, 24), 2, 96, 96)
    tri = md["t_pos_idx"]
    ro = oraster.rasterize(pos, tri, 96, 96)
    ids, margin, _ = oraster.brute_force_cover(pos, tri, 96, 96)
    rid = ro[..., 3].astype(np.int64)
    safe = (ids > 0) & (margin > 0.08)
    assert safe.mean() > 0.2
    assert np.array_equal(rid[safe], ids[safe])
    # and the silhouette agrees within a 1-pixel band
    assert abs((rid > 0).mean() - (ids > 0).mean()) < 0.01


def test_oracle_raster_shared_edge_partition():
    """Top-left rule: a full-screen quad covers every pixel exactly once, including the diagonal."""
    pos = np.array([[[-1, -1, 0, 1], [1, -1, 0, 1], [1, 1, 0, 1], [-1, 1, 0, 1]]], np.float32)
    for tri in ([[0, 1, 2], [0, 2, 3]], [[2, 1, 0], [0, 2, 3]]):
        r = oraster.rasterize(pos, np.array(tri, np.int32), 32, 32)
        assert (r[..., 3] > 0).all()
        # each triangle owns a strict half minus/plus the diagonal, never both
        assert set(np.unique(r[..., 3])) == {1.0, 2.0}


def test_interpolate_and_antialias_identities():
    md, batch, pos = _scene((32, 24), 1, 64, 64)
    tri = md["t_pos_idx"]
    ro = oraster.rasterize(pos, tri, 64, 64)
    ones = np.ones((pos.shape[1], 1), np.float32)
    it = oraster.interpolate(ones, ro, tri)
    m = ro[..., 3] > 0
    assert np.allclose(it[m], 1.0, atol=1e-6) and np.all(it[~m] == 0)
    plan = oraster.antialias_plan(pos, tri, md["opp"], ro)
    const = np.full((1, 64, 64, 3), 0.37, np.float32)
    assert np.allclose(oraster.antialias_apply(const, plan), const)      # AA of a constant image is identity
    op = oraster.antialias_apply(m[..., None].astype(np.float32), plan)
    assert op.min() >= -1e-6 and op.max() <= 1 + 1e-6 and ((op > 0.01) & (op < 0.99)).sum() > 20
    # gradient is the exact transpose of the (linear) forward
    x = torch.randn(1, 64, 64, 3, dtype=torch.float32)
    y = torch.randn(1, 64, 64, 3, dtype=torch.float32)
    lhs = (torch.from_numpy(oraster.antialias_apply(x.numpy(), plan)) * y).sum()
    rhs = (x * torch.from_numpy(oraster.antialias_grad(y.numpy(), plan))).sum()
    assert abs(lhs - rhs) < 1e-3 * abs(lhs)


@pytest.fixture(scope="module")
def env_pair():
    lat = [util.synthetic_latlong(i) * 0.02 for i in range(2)]
    fg = penv.approx_fg_lut()
    atlas = penv.EnvAtlas(lat, scale=2.0, min_res=8, max_res=32, fg_lut=fg)
    oenvs = [oenv.EnvLight(l, scale=2.0, min_res=8, max_res=32) for l in lat]
    return lat, fg, atlas, oenvs


def test_env_prefilter_product_vs_oracle(env_pair):
    lat, fg, atlas, oenvs = env_pair
    for e in range(2):
        for k in range(3):
            a, b = atlas.specular[e][k], oenvs[e].specular[k]
            assert (a - b).abs().max() <= 1e-4 * b.abs().max()
        assert (atlas.diffuse[e] - oenvs[e].diffuse).abs().max() < 1e-5
    c = oenv.EnvLight(torch.full((16, 32, 3), 0.5), min_res=8, max_res=16)
    d = torch.nn.functional.normalize(torch.randn(500, 3), dim=-1)
    assert (c(d, torch.rand(500, 1)) - 0.5).abs().max() < 1e-5      # specular prefilter is normalised


def test_shade_core_vs_oracle(hostemu, env_pair):
    lat, fg, atlas, oenvs = env_pair
    torch.manual_seed(0)
    N = 20000
    n = torch.nn.functional.normalize(torch.randn(N, 3), dim=-1)
    v = torch.nn.functional.normalize(n + 0.8 * torch.randn(N, 3), dim=-1)
    feat = (torch.randn(N, 5) * 1.5).requires_grad_()
    env = torch.randint(0, 2, (N,))
    out, _ = oshade.material_forward(feat, feat.detach() + 0.1, v, n, oenvs, env, fg)
    dcol = torch.randn(N, 3)
    (out["color"] * dcol).sum().backward()
    color = np.empty((N, 3), np.float32); dbg = np.empty((N, 17), np.float32); dfeat = np.empty((N, 5), np.float32)
    mat = np.array([0.0, 0.9, 0.1, 0.95], np.float32)
    arrs = [np.ascontiguousarray(t.detach().numpy()) for t in (n, v, feat, dcol)]
    ee = env.numpy().astype(np.int32)
    hostemu.emu_shade(ctypes.byref(atlas.struct), P(mat), P(arrs[0]), P(arrs[1]), P(arrs[2]), P(ee),
                      ctypes.c_longlong(N), P(color), P(dbg), P(arrs[3]), P(dfeat))
    oc = out["color"].detach().numpy()
    assert 0.2 < ((oc > 0) & (oc < 1)).mean()          # not everything is clamped
    assert np.abs(color - oc).max() < 1e-5
    names = [("albedo", 0, 3), ("specular_lights", 3, 6), ("diffuse_lights", 6, 9), ("specular_colors", 9, 12),
             ("diffuse_colors", 12, 15), ("metalness", 15, 16), ("roughness", 16, 17)]
    for k, a, b in names:
        assert np.abs(dbg[:, a:b] - out[k].detach().numpy()).max() < 1e-5, k
    g = feat.grad.numpy()
    assert np.abs(dfeat - g).max() < 1e-4 * np.abs(g).max()


def test_shade_core_fp16_atlas_stays_inside_the_parity_budget(hostemu, env_pair):
    """opt-in atlas format (DREAMMAT_ATLAS=fp16: RGBA fp16 texels, one 16-byte load per bilinear row): the shaded colour
    stays within the north-star's 1e-3 relative budget of the fp32 oracle, and the default atlas is untouched."""
    lat, fg, atlas, oenvs = env_pair
    half = penv.EnvAtlas(lat, scale=2.0, min_res=8, max_res=32, fg_lut=fg, texel="fp16")
    assert half.struct.texel_format == 1 and half.spec_packed.dtype == torch.float16 and half.diff_packed.dtype == torch.float16
    assert half.spec_packed.shape[:2] == atlas.spec_packed.shape[:2]
    torch.manual_seed(1)
    N = 20000
    n = torch.nn.functional.normalize(torch.randn(N, 3), dim=-1)
    v = torch.nn.functional.normalize(n + 0.8 * torch.randn(N, 3), dim=-1)
    feat = (torch.randn(N, 5) * 1.5).requires_grad_()
    env = torch.randint(0, 2, (N,))
    out, _ = oshade.material_forward(feat, feat.detach() + 0.1, v, n, oenvs, env, fg)
    dcol = torch.randn(N, 3)
    (out["color"] * dcol).sum().backward()
    color = np.empty((N, 3), np.float32); dbg = np.empty((N, 17), np.float32); dfeat = np.empty((N, 5), np.float32)
    mat = np.array([0.0, 0.9, 0.1, 0.95], np.float32)
    arrs = [np.ascontiguousarray(t.detach().numpy()) for t in (n, v, feat, dcol)]
    ee = env.numpy().astype(np.int32)
    hostemu.emu_shade(ctypes.byref(half.struct), P(mat), P(arrs[0]), P(arrs[1]), P(arrs[2]), P(ee),
                      ctypes.c_longlong(N), P(color), P(dbg), P(arrs[3]), P(dfeat))
    oc = out["color"].detach().numpy()
    err = np.abs(color - oc).max()
    assert 1e-7 < err < 1e-3 * max(1.0, np.abs(oc).max()), err            # different from fp32, inside the budget
    g = feat.grad.numpy()
    assert np.abs(dfeat - g).max() < 2e-3 * np.abs(g).max()


def test_white_furnace():
    """constant env L, albedo=1, metallic=0 -> color = L_d + (0.04 fg0 + fg1) L_s."""
    env = oenv.EnvLight(torch.full((16, 32, 3), 0.2), min_res=8, max_res=16)
    fg = penv.approx_fg_lut()
    n = torch.nn.functional.normalize(torch.randn(100, 3), dim=-1)
    feat = torch.tensor([[20.0, 20.0, 20.0, -20.0, 0.3]]).expand(100, 5)
    out, _ = oshade.material_forward(feat, feat, n, n, [env], torch.zeros(100, dtype=torch.long), fg)
    _, alb, met, rough = oshade.material_params(feat)
    f = oenv.texture2d_linear_clamp(fg, torch.cat([torch.ones(100, 1), rough], -1))
    expect = env(n) + (0.04 * f[:, :1] + f[:, 1:]) * 0.2
    assert (out["color"] - expect.clamp(0, 1)).abs().max() < 1e-5


def test_lin2srgb_knots():
    x = torch.tensor([0.0, 0.0031308, 1.0, 2.0, -1.0])
    y = oshade.lin2srgb(x)
    assert y[0] == 0 and abs(y[1] - 12.92 * 0.0031308) < 1e-6 and abs(y[2] - 1.0) < 1e-6 and y[3] == 1 and y[4] == 0


def test_hashgrid_layout():
    lv, tot = ofield.grid_levels()
    assert tot * 2 == 12599920                      # tcnn parameter count of dreammat.yaml:43-49
    from dreammat_amd import hipops
    spec = hipops.GridSpec()
    assert spec.n_params == 12599920
    assert [l["res"] for l in lv] == [l["res"] for l in spec.levels]
    assert [l["size"] for l in lv] == [l["size"] for l in spec.levels]
    # trilinear weights sum to one: a constant table encodes to that constant
    lv4, tot4 = ofield.grid_levels(n_levels=4, log2_hashmap_size=10)
    enc = ofield.hash_encode(torch.rand(64, 3), torch.full((tot4, 2), 0.7), lv4)
    assert (enc - 0.7).abs().max() < 1e-6


def test_c_abi_exports_every_declared_symbol():
    """The library loads on a GPU-less box and exports exactly what include/dreammat_hip.h declares."""
    import os, re
    L = _lib.lib()
    assert L.dm_abi_version() == _lib.ABI_VERSION == 14
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "dreammat_hip.h")).read()
    declared = set(re.findall(r"\b(dm_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    for name in declared:
        assert hasattr(L, name)


def test_topology_native_vs_oracle():
    m = pmesh.displaced_sphere(24, 16)
    tri = m.t_pos_idx.numpy().astype(np.int32)
    from dreammat_amd import hipops
    opp = hipops.build_topology(m.t_pos_idx).numpy()
    assert np.array_equal(opp, oraster.build_topology(tri))
    assert (opp >= 0).all()                         # closed manifold


def test_attention_index_math_on_mfma_model():
    from tests.mfma_sim import attention_wave_sim
    rng = np.random.default_rng(0)
    for D, Skv in ((64, 128), (64, 77), (32, 200)):
        Q = rng.standard_normal((32, D)); K = rng.standard_normal((Skv, D)); V = rng.standard_normal((Skv, D))
        S = Q @ K.T * D ** -0.5
        Pm = np.exp(S - S.max(1, keepdims=True)); Pm /= Pm.sum(1, keepdims=True)
        assert np.abs(attention_wave_sim(Q, K, V, D ** -0.5) - Pm @ V).max() < 1e-12


def test_attention_v3_index_math_and_rebase_logic_on_mfma_model():
    """k_attn_fwd_v3: swapped-row K image + natural V^T + running maximum through the MFMA C operand + deferred re-base
    (threshold 8, forced on the first tile), both with and without the prologue Q scaling; spiked keys force the re-base
    branch at chosen tiles (guide rule 26), ragged tails exercise the permuted mask."""
    from tests.mfma_sim import attention_wave_sim_v3
    rng = np.random.default_rng(1)
    for D, Skv, spike in ((64, 256, True), (64, 77, False), (64, 200, True), (128, 192, False), (64, 64, False), (64, 1, False)):
        Q = rng.standard_normal((32, D)); K = rng.standard_normal((Skv, D)); V = rng.standard_normal((Skv, D))
        if spike:
            K[Skv - 30] = Q[5] * 6.0               # late, large: re-base in the last tile
            K[70] = Q[9] * 3.0                     # second tile
            K[3] = -Q[2] * 50.0                    # a row whose FIRST tile is dominated by a very negative score
        S = Q @ K.T * D ** -0.5
        Pm = np.exp(S - S.max(1, keepdims=True)); Pm /= Pm.sum(1, keepdims=True)
        for prescale in (True, False):
            O, nb = attention_wave_sim_v3(Q, K, V, D ** -0.5, prescale=prescale)
            assert np.isfinite(O).all() and np.abs(O - Pm @ V).max() < 1e-10, (D, Skv, prescale)
            assert nb >= (2 if spike and Skv >= 200 else 1)
        O0, _ = attention_wave_sim_v3(Q, K, V, D ** -0.5, thr=0.0)      # THR = 0 and THR = 8 agree to rounding
        assert np.abs(O0 - O).max() < 1e-10


def test_attention_backward_index_math_on_mfma_model():
    """k_attn_bwd_dq / k_attn_bwd_dkv (csrc/attn_bwd.hip): the row-major staging, the plain and the transposing fragment
    reads (ds_read_b64_tr_b16 as measured by tools/tr_probe.cpp, rows in accumulator order), the accumulator -> B-operand
    packing and the row statistics, followed literally on the numpy MFMA model, against the closed-form gradient of
    softmax(QK^T.scale)V; ragged sequences, D < DP."""
    from tests.mfma_sim import attention_bwd_dkv_wave_sim, attention_bwd_dq_wave_sim
    rng = np.random.default_rng(2)
    for D, DP, Sq, Skv in ((64, 64, 100, 77), (32, 32, 64, 130), (40, 64, 70, 64)):
        Q = rng.standard_normal((Sq, D)); K = rng.standard_normal((Skv, D)); V = rng.standard_normal((Skv, D))
        dO = rng.standard_normal((Sq, D))
        sc = D ** -0.5
        S = Q @ K.T * sc
        m = S.max(1, keepdims=True); P = np.exp(S - m); l = P.sum(1, keepdims=True); P /= l
        O = P @ V
        lse2 = (m[:, 0] + np.log(l[:, 0])) * 1.4426950408889634
        delta = (dO * O).sum(1)
        dS = P * (dO @ V.T - delta[:, None])
        dQ, dK, dV = dS @ K * sc, dS.T @ Q * sc, P.T @ dO
        for q0 in range(0, Sq, 32):
            dq, dl = attention_bwd_dq_wave_sim(Q[q0:q0 + 32], K, V, O[q0:q0 + 32], dO[q0:q0 + 32], lse2[q0:q0 + 32], sc, DP)
            assert np.abs(dq - dQ[q0:q0 + 32]).max() < 1e-12 and np.abs(dl - delta[q0:q0 + 32]).max() < 1e-12
        for k0 in range(0, Skv, 32):
            dk, dv = attention_bwd_dkv_wave_sim(Q, K[k0:k0 + 32], V[k0:k0 + 32], dO, lse2, delta, sc, DP)
            assert np.abs(dk - dK[k0:k0 + 32]).max() < 1e-12 and np.abs(dv - dV[k0:k0 + 32]).max() < 1e-12


def test_conv_weight_gradient_index_math_on_mfma_model():
    """k_conv3x3_wgrad (csrc/conv_wgrad.hip): row-major staging, the per-lane addresses of the transposing LDS read
    (ds_read_b64_tr_b16, modelled as measured by tools/tr_probe.cpp), halo / stride / tap arithmetic and the accumulator
    store, followed literally on the numpy model against torch's conv2d weight gradient (stride 1 and 2, several rows per chunk)."""
    import torch
    from tests.mfma_sim import conv_wgrad_workgroup_sim
    rng = np.random.default_rng(3)
    for B, H, W, s in ((1, 8, 8, 1), (2, 16, 16, 2), (1, 16, 4, 1)):
        x = rng.standard_normal((B, H, W, 64))
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        dy = rng.standard_normal((B, Ho, Wo, 64))
        w = torch.zeros(64, 64, 3, 3, dtype=torch.float64, requires_grad=True)
        y = torch.nn.functional.conv2d(torch.tensor(x).permute(0, 3, 1, 2), w, stride=s, padding=1)
        y.backward(torch.tensor(dy).permute(0, 3, 1, 2))
        assert np.abs(conv_wgrad_workgroup_sim(x, dy, s) - w.grad.permute(0, 2, 3, 1).numpy()).max() < 1e-11


def test_bvh_build_and_traversal_core_vs_brute_force(hostemu):
    """SURVEY row f-1 groundwork: the host BVH builder of the product library (dm_bvh_build) and the traversal core
    the HIP kernels share (csrc/bvh_core.h, run here through tests/hostemu) against the oracle's brute-force
    double-sided any-hit test."""
    import ctypes

    from dreammat_amd import hipops, mesh as pmesh
    from oracle import mc_shading as omc
    torch.manual_seed(0)
    for (n_lon, n_lat) in [(12, 8), (48, 40)]:
        m = pmesh.displaced_sphere(n_lon, n_lat)
        bvh = hipops.MeshBvh(m.v_pos, m.t_pos_idx)
        n_tri = m.t_pos_idx.shape[0]
        nodes = bvh.nodes_host
        assert 1 <= bvh.n_nodes <= 2 * n_tri and sorted(bvh.order.tolist()) == list(range(n_tri))
        # structure: every leaf slot used exactly once; children boxes inside the parent box; triangles inside leaves
        fl = nodes.view(torch.float32)
        covered = torch.zeros(n_tri, dtype=torch.int32)
        tv = m.v_pos.float()[m.t_pos_idx.long()[bvh.order.long()]]                    # [n_tri,3,3] in leaf order
        assert torch.allclose(bvh.tris_host[:, 0:3], tv[:, 0]) and torch.allclose(bvh.tris_host[:, 4:7], tv[:, 1] - tv[:, 0])
        for i in range(bvh.n_nodes):
            a, b = int(nodes[i, 3]), int(nodes[i, 7])
            lo, hi = fl[i, 0:3], fl[i, 4:7]
            if b > 0:
                covered[a:a + b] += 1
                pts = tv[a:a + b].reshape(-1, 3)
                assert (pts >= lo - 1e-6).all() and (pts <= hi + 1e-6).all() and b <= 4
            else:
                for c in (a, a + 1):
                    assert (fl[c, 0:3] >= lo - 1e-6).all() and (fl[c, 4:7] <= hi + 1e-6).all()
        assert (covered == 1).all()
        # rays: from surface points along random directions (the shading pattern) + from outside towards the object
        tri_c = tv.mean(1)
        fn = torch.nn.functional.normalize(torch.cross(tv[:, 1] - tv[:, 0], tv[:, 2] - tv[:, 0], dim=-1), dim=-1)
        pick = torch.randint(0, n_tri, (1500,))
        d1 = torch.nn.functional.normalize(torch.randn(1500, 3), dim=-1)
        o1 = tri_c[pick] + 1e-4 * fn[pick] + 1e-5 * d1
        o2 = torch.nn.functional.normalize(torch.randn(500, 3), dim=-1) * 3.0
        d2 = torch.nn.functional.normalize(-o2 + 0.6 * torch.randn(500, 3), dim=-1)
        o, d = torch.cat([o1, o2]).contiguous(), torch.cat([d1, d2]).contiguous()
        hit = torch.zeros(o.shape[0], dtype=torch.uint8)
        hostemu.emu_bvh_any_hit(ctypes.c_void_p(nodes.data_ptr()), ctypes.c_void_p(bvh.tris_host.data_ptr()),
                                ctypes.c_void_p(o.data_ptr()), ctypes.c_void_p(d.data_ptr()), ctypes.c_longlong(o.shape[0]),
                                ctypes.c_float(10.0), ctypes.c_void_p(hit.data_ptr()))
        ref = omc.trace_any_hit(m.v_pos.float(), m.t_pos_idx, o, d)
        mism = int((hit.bool() != ref).sum())
        assert mism <= 2, (mism, o.shape[0])                                          # fp32 vs fp64 on edge-grazing rays
        # the 4-wide collapse (dm_bvh_collapse4 + dm_bvh4_any_hit): same triangles, same tests => identical answers
        hit4 = torch.zeros(o.shape[0], dtype=torch.uint8)
        hostemu.emu_bvh4_any_hit(ctypes.c_void_p(bvh.nodes4_host.data_ptr()), ctypes.c_void_p(bvh.tris_host.data_ptr()),
                                 ctypes.c_void_p(o.data_ptr()), ctypes.c_void_p(d.data_ptr()), ctypes.c_longlong(o.shape[0]),
                                 ctypes.c_float(10.0), ctypes.c_void_p(hit4.data_ptr()))
        assert torch.equal(hit4, hit) and 1 <= bvh.n_nodes4 <= bvh.n_nodes
        assert 0.2 < ref.float().mean() < 0.95
        # the occupancy grid (dm_grid_build + csrc/grid_core.h): a conservative voxelisation walked by a DDA -- the same
        # boolean over the same triangle test, so the answers must be IDENTICAL to the tree's
        for res in (0, 7, 33):
            b2 = hipops.MeshBvh(m.v_pos, m.t_pos_idx, grid_res=res)
            g = b2.grid_struct(b2.grid_blob_host)
            bits, rank, occ, ids, dist = b2.grid_sections()
            dim = [int(x) for x in g.dim]
            n_cells = dim[0] * dim[1] * dim[2]
            assert g.n_words == (n_cells + 31) // 32 and max(dim) <= 96 and g.n_occ == len(occ) - 1 and int(occ[-1]) == g.n_entries == len(ids)
            assert sorted(set(ids.tolist())) == list(range(n_tri))                    # every triangle is listed somewhere
            pop = torch.tensor([bin(int(w)).count("1") for w in bits.tolist()])
            assert int(pop.sum()) == g.n_occ and torch.equal(rank, torch.cumsum(pop, 0) - pop)
            # block distance field: 0 exactly on the blocks that hold an occupied cell, neighbours differ by at most 1
            bd = [(x + 1) // 2 for x in dim]
            cells = torch.zeros(n_cells, dtype=torch.bool)
            for wi, wv in enumerate(bits.tolist()):
                for bb in range(32):
                    if (wv >> bb) & 1:
                        cells[wi * 32 + bb] = True
            cz = torch.zeros(2 * bd[2], 2 * bd[1], 2 * bd[0], dtype=torch.bool)
            cz[:dim[2], :dim[1], :dim[0]] = cells.reshape(dim[2], dim[1], dim[0])
            blk = cz.reshape(bd[2], 2, bd[1], 2, bd[0], 2).any(5).any(3).any(1)
            D = dist.reshape(bd[2], bd[1], bd[0])
            assert torch.equal(D == 0, blk) and int(D.max()) <= 15
            for ax in range(3):
                assert int((D.narrow(ax, 1, bd[2 - ax] - 1) - D.narrow(ax, 0, bd[2 - ax] - 1)).abs().max()) <= 1
            # every vertex lies in a cell that lists its triangle
            cell = ((tv.reshape(-1, 3) - torch.tensor(list(g.gmin))) * g.inv_cell).floor().long()
            assert (cell >= 0).all() and (cell < torch.tensor(dim)).all()
            hitg = torch.zeros(o.shape[0], dtype=torch.uint8)
            stats = torch.zeros(o.shape[0], 3, dtype=torch.int32)
            hostemu.emu_grid_any_hit(ctypes.byref(g), ctypes.c_void_p(o.data_ptr()),
                                     ctypes.c_void_p(d.data_ptr()), ctypes.c_longlong(o.shape[0]), ctypes.c_float(10.0),
                                     ctypes.c_void_p(hitg.data_ptr()), ctypes.c_void_p(stats.data_ptr()))
            assert torch.equal(hitg, hit), (res, int((hitg != hit).sum()))
            assert int(stats[:, 0].max()) <= 3 * 96 and (res != 0 or float(stats[:1500, 2].float().mean()) < 40)


def test_grid_voxelisation_of_large_and_degenerate_triangles(hostemu):
    """dm_grid_build on the shapes a mesh-sized heuristic gets wrong: the 2-triangle quad of BASELINE configs[0] (triangles
    far larger than a cell, flat in one axis) and a sliver; rays through it against the brute-force test."""
    import ctypes

    from dreammat_amd import hipops
    from oracle import mc_shading as omc
    v = torch.tensor([[-1.0, -1.0, 0.0], [1.0, -1.0, 0.0], [1.0, 1.0, 0.0], [-1.0, 1.0, 0.0], [0.0, 0.0, 0.5], [1e-4, 0.0, 0.5], [0.0, 2.0, 0.7]])
    f = torch.tensor([[0, 1, 2], [0, 2, 3], [4, 5, 6]], dtype=torch.int32)
    torch.manual_seed(1)
    o = torch.cat([torch.randn(3000, 3) * 1.5, torch.tensor([[0.3, 0.2, 1.0], [0.0, 0.0, -1.0]])]).contiguous()
    tgt = torch.cat([torch.rand(3000, 2) * 2.6 - 1.3, torch.rand(3000, 1) * 0.6], -1)          # towards the quad and the sliver
    d = torch.cat([torch.nn.functional.normalize(tgt - o[:3000], dim=-1), torch.tensor([[0.0, 0.0, -1.0], [0.0, 0.0, 1.0]])]).contiguous()
    ref = omc.trace_any_hit(v, f, o, d)
    for res in (0, 16, 96):
        b = hipops.MeshBvh(v, f, grid_res=res)
        g = b.grid_struct(b.grid_blob_host)
        hit = torch.zeros(o.shape[0], dtype=torch.uint8)
        hostemu.emu_grid_any_hit(ctypes.byref(g), ctypes.c_void_p(o.data_ptr()),
                                 ctypes.c_void_p(d.data_ptr()), ctypes.c_longlong(o.shape[0]), ctypes.c_float(10.0),
                                 ctypes.c_void_p(hit.data_ptr()), None)
        hb = torch.zeros(o.shape[0], dtype=torch.uint8)
        hostemu.emu_bvh_any_hit(ctypes.c_void_p(b.nodes_host.data_ptr()), ctypes.c_void_p(b.tris_host.data_ptr()),
                                ctypes.c_void_p(o.data_ptr()), ctypes.c_void_p(d.data_ptr()), ctypes.c_longlong(o.shape[0]),
                                ctypes.c_float(10.0), ctypes.c_void_p(hb.data_ptr()))
        assert torch.equal(hit, hb), (res, int((hit != hb).sum()))
        assert int((hit.bool() != ref).sum()) <= 2 and 0.1 < ref.float().mean() < 0.9


def test_rgb18e8_texels_and_fg_pair_table(hostemu, env_pair):
    """The default atlas texel (8-byte shared-exponent RGB, 18-bit mantissas) and the FG x-pair table: packing round trip
    within 2^-18 of the largest channel, the pair table holds exactly the clamped bilinear taps, and the shading core gives
    the same colours with all three texel formats (fp32 exact layout / rgb18e8 / fp16) within their stated budgets."""
    lat, fg, atlas, oenvs = env_pair
    assert atlas.texel == "rgb18e8" and atlas.struct.texel_format == 2 and atlas.spec_packed.dtype == torch.int32
    g = torch.Generator().manual_seed(0)
    rgb = torch.exp(torch.randn(4000, 3, generator=g) * 4.0)               # 7 decades of dynamic range
    rgb[::7, 1] = 0.0
    rgb[5] = 0.0
    rgb[6] = torch.tensor([1.0 - 2 ** -20, 0.25, 0.5])                    # mantissa rounds up into the next exponent
    dec = penv.decode_rgb18e8(penv.encode_rgb18e8(rgb))
    assert ((dec - rgb).abs().amax(-1) <= rgb.amax(-1) * 2.0 ** -18 + 1e-37).all()
    L = fg.shape[0]
    pairs = penv.fg_pair_table(fg)
    assert pairs.shape == (L, L + 1, 4)
    for x0 in (-1, 0, 100, L - 1):
        assert torch.equal(pairs[:, x0 + 1, :2], fg[:, max(x0, 0)]) and torch.equal(pairs[:, x0 + 1, 2:], fg[:, min(x0 + 1, L - 1)])
    torch.manual_seed(3)
    N = 8000
    n = torch.nn.functional.normalize(torch.randn(N, 3), dim=-1)
    v = torch.nn.functional.normalize(n + 0.8 * torch.randn(N, 3), dim=-1)
    feat = torch.randn(N, 5) * 1.5
    env = torch.randint(0, 2, (N,)).numpy().astype(np.int32)
    mat = np.array([0.0, 0.9, 0.1, 0.95], np.float32)
    arrs = [np.ascontiguousarray(t.numpy()) for t in (n, v, feat, torch.randn(N, 3))]
    res = {}
    for texel in ("fp32", "rgb18e8", "fp16"):
        a = penv.EnvAtlas(lat, scale=2.0, min_res=8, max_res=32, fg_lut=fg, texel=texel)
        color = np.empty((N, 3), np.float32); dfeat = np.empty((N, 5), np.float32)
        hostemu.emu_shade(ctypes.byref(a.struct), P(mat), P(arrs[0]), P(arrs[1]), P(arrs[2]), P(env), ctypes.c_longlong(N),
                          P(color), None, P(arrs[3]), P(dfeat))
        res[texel] = (color, dfeat)
    gmax = np.abs(res["fp32"][1]).max()
    assert np.abs(res["rgb18e8"][0] - res["fp32"][0]).max() < 2e-6 and np.abs(res["rgb18e8"][1] - res["fp32"][1]).max() < 1e-5 * gmax
    assert np.abs(res["fp16"][0] - res["fp32"][0]).max() < 1e-3


def test_oracle_hash_grid_over_two_dimensions():
    """oracle/field.py restated for the uv-space field (n_input_dims = 2): level sizes from res^2, a dense level read at its
    lattice points returns the table rows themselves (index = x + y * res), the four bilinear weights are a partition of unity,
    and the encoding is continuous across cell borders."""
    from oracle import field as ofield
    lv, tot = ofield.grid_levels(n_levels=6, log2_hashmap_size=12, n_dims=2)
    assert [l["res"] for l in lv][:3] == [16, 24, 34] and lv[0]["size"] == 256 and lv[1]["size"] == 24 * 24 and lv[2]["size"] == (34 * 34 + 7) // 8 * 8
    assert all(l["size"] <= 4096 for l in lv) and lv[-1]["size"] == 4096 and tot == sum(l["size"] for l in lv)
    torch.manual_seed(0)
    table = torch.randn(tot, 2)
    l0 = lv[0]
    ij = torch.stack(torch.meshgrid(torch.arange(1, 15), torch.arange(1, 15), indexing="ij"), -1).reshape(-1, 2)
    x = (ij.float() - 0.5) / l0["scale"]                               # pos = x * scale + 0.5 = integer lattice point
    enc = ofield.hash_encode(x, table, lv[:1])
    assert torch.allclose(enc, table[l0["offset"] + ij[:, 0] + ij[:, 1] * l0["res"]], atol=1e-5)
    ones = torch.ones(tot, 2)
    u = torch.rand(500, 2)
    assert torch.allclose(ofield.hash_encode(u, ones, lv), torch.ones(500, 2 * len(lv)), atol=1e-5)
    a = torch.tensor([[0.3, 0.41]])
    assert (ofield.hash_encode(a + 1e-6, table, lv) - ofield.hash_encode(a - 1e-6, table, lv)).abs().max() < 1e-3
