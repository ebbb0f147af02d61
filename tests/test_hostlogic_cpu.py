"""Host-side algebra around the conv C ABI, checked on CPU: the weight layouts `Conv2d._prepared()` hands to the
kernel (tap-major forward weights, flipped / channel-swapped data-gradient weights, zero-padded narrow heads) and
the autograd wrappers in `hipops` (fused residual, stride-2 forward with an implied trailing pad, zero-insert data
gradient).  `hipops.conv3x3_nhwc` is replaced by a torch emulation of the `dm_conv3x3_nhwc_bf16_fused` CONTRACT
(include/dreammat_hip.h) -- the kernel itself is covered by tests/test_hip_gpu.py."""
import os

import pytest
import torch
import torch.nn.functional as F

from dreammat_amd import hipops
from dreammat_amd.geometry import VanillaMLP, _wgrad_splitk
from dreammat_amd.sd import layers


def conv3x3_contract(x_nhwc, w_tap_major, bias, stride=1, pad=(1, 1), out_hw=None, rowbias=None, residual=None):
    """y[b,yo,xo,n] = bias[n] + rowbias[b,n] + residual[b,yo,xo,n] + sum_{dy,dx,c} x[b, yo*s-py+dy, xo*s-px+dx, c] *
    w[n, (dy*3+dx)*Cin + c], zero outside the image; the output size is the caller's (trailing pad implied)."""
    B, H, W, Cin = x_nhwc.shape
    Cout = w_tap_major.shape[0]
    if out_hw is None:
        out_hw = ((H + 2 * pad[0] - 3) // stride + 1, (W + 2 * pad[1] - 3) // stride + 1)
    Ho, Wo = out_hw
    need_h, need_w = (Ho - 1) * stride + 3, (Wo - 1) * stride + 3
    x = x_nhwc.permute(0, 3, 1, 2)
    x = F.pad(x, (pad[1], max(0, need_w - W - pad[1]), pad[0], max(0, need_h - H - pad[0])))[:, :, :need_h, :need_w]
    w = w_tap_major.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    y = F.conv2d(x, w, bias, stride=stride).permute(0, 2, 3, 1)
    assert y.shape[1:3] == (Ho, Wo)
    if rowbias is not None:
        y = y + rowbias[:, None, None, :]
    if residual is not None:
        y = y + residual
    return y.contiguous()


@pytest.fixture
def contract(monkeypatch):
    monkeypatch.setattr(hipops, "conv3x3_nhwc", conv3x3_contract)


def test_prepared_weights_forward_and_data_gradient(contract):
    torch.manual_seed(0)
    conv = layers.Conv2d(6, 64, 3, padding=1).double()
    w_fwd, w_dgrad = conv._prepared()
    x = torch.randn(2, 6, 9, 7, dtype=torch.float64, requires_grad=True)
    ref = F.conv2d(x, conv.weight, conv.bias, padding=1)
    y = conv3x3_contract(x.detach().permute(0, 2, 3, 1).contiguous(), w_fwd, conv._bias_p)
    assert torch.allclose(y.permute(0, 3, 1, 2), ref.detach(), atol=1e-12)
    dy = torch.randn_like(ref)
    ref.backward(dy)
    dx = conv3x3_contract(dy.permute(0, 2, 3, 1).contiguous(), w_dgrad, None)       # same kernel, flipped weights
    assert torch.allclose(dx.permute(0, 3, 1, 2), x.grad, atol=1e-12)


def test_narrow_head_is_zero_padded_to_one_mfma_tile(contract):
    """Cout < 64 (UNet conv_out 320->4, VAE conv_out 512->8): 64 output rows, the extra ones exactly zero, and a
    data gradient that ignores whatever arrives in the padded channels."""
    torch.manual_seed(1)
    conv = layers.Conv2d(8, 4, 3, padding=1).double()
    w_fwd, w_dgrad = conv._prepared()
    assert w_fwd.shape == (64, 72) and w_dgrad.shape == (8, 9 * 64) and conv._bias_p.shape == (64,)
    x = torch.randn(1, 8, 5, 6, dtype=torch.float64, requires_grad=True)
    y = conv3x3_contract(x.detach().permute(0, 2, 3, 1).contiguous(), w_fwd, conv._bias_p)
    ref = F.conv2d(x, conv.weight, conv.bias, padding=1)
    assert torch.allclose(y[..., :4].permute(0, 3, 1, 2), ref.detach(), atol=1e-12)
    assert y[..., 4:].abs().max() == 0
    dy = torch.randn_like(ref)
    ref.backward(dy)
    dy64 = torch.randn(1, 5, 6, 64, dtype=torch.float64)                              # garbage in the padded channels
    dy64[..., :4] = dy.permute(0, 2, 3, 1)
    dx = conv3x3_contract(dy64, w_dgrad, None)
    assert torch.allclose(dx.permute(0, 3, 1, 2), x.grad, atol=1e-12)


def test_stride1_autograd_with_fused_residual(contract):
    torch.manual_seed(2)
    conv = layers.Conv2d(5, 64, 3, padding=1).double()
    w_fwd, w_dgrad = conv._prepared()
    x = torch.randn(2, 6, 8, 5, dtype=torch.float64, requires_grad=True)
    r = torch.randn(2, 6, 8, 64, dtype=torch.float64, requires_grad=True)
    y = hipops.conv3x3_s1_autograd(x, w_fwd, w_dgrad, conv._bias_p, r)
    xr, rr = x.detach().clone().requires_grad_(), r.detach().clone().requires_grad_()
    ref = F.conv2d(xr.permute(0, 3, 1, 2), conv.weight, conv.bias, padding=1).permute(0, 2, 3, 1) + rr
    assert torch.allclose(y, ref, atol=1e-12)
    g = torch.randn_like(ref)
    y.backward(g)
    ref.backward(g)
    assert torch.allclose(x.grad, xr.grad, atol=1e-12) and torch.allclose(r.grad, rr.grad, atol=1e-12)


@pytest.mark.parametrize("lead_pad,H,W", [(0, 8, 12), (1, 8, 12), (1, 7, 9)])
def test_stride2_forward_and_zero_insert_data_gradient(contract, lead_pad, H, W):
    """lead_pad 0 = AutoencoderKL's downsampler over F.pad(x, (0,1,0,1)); lead_pad 1 = the UNet's padding=1 one.
    Backward = stride-1 conv of the zero-inserted gradient with leading pad 2 - lead_pad."""
    torch.manual_seed(3)
    conv = layers.Conv2d(4, 64, 3, stride=2, padding=lead_pad).double()
    w_fwd, w_dgrad = conv._prepared()
    x = torch.randn(2, H, W, 4, dtype=torch.float64, requires_grad=True)
    y = hipops.conv3x3_s2_autograd(x, w_fwd, w_dgrad, conv._bias_p, lead_pad)
    xr = x.detach().clone().requires_grad_()
    xin = xr.permute(0, 3, 1, 2)
    xin = F.pad(xin, (0, 1, 0, 1)) if lead_pad == 0 else xin
    ref = F.conv2d(xin, conv.weight, conv.bias, stride=2, padding=lead_pad).permute(0, 2, 3, 1)
    if lead_pad == 1 and (H % 2 or W % 2):
        assert y.shape == ref.shape
    assert torch.allclose(y, ref, atol=1e-12)
    g = torch.randn_like(ref)
    y.backward(g)
    ref.backward(g)
    assert torch.allclose(x.grad, xr.grad, atol=1e-12)


def test_fused_epilogue_contract_matches_resnet_block_math(contract):
    """ResnetBlock2D inference path: conv1(+temb row bias) and conv2(+residual) through the fused entry point equal
    the unfused module arithmetic."""
    torch.manual_seed(4)
    blk = layers.ResnetBlock2D(64, 64, temb_ch=32).double().eval()
    x = torch.randn(2, 64, 6, 6, dtype=torch.float64)
    temb = torch.randn(2, 32, dtype=torch.float64)
    with torch.no_grad():
        ref = blk(x, temb)                                                          # CPU: unfused torch path
        h = layers.group_norm_act(blk.norm1, x, True).permute(0, 2, 3, 1).contiguous()
        tproj = blk.time_emb_proj(F.silu(temb))
        h = conv3x3_contract(h, blk.conv1._prepared()[0], blk.conv1._bias_p, rowbias=tproj)
        h = layers.group_norm_act(blk.norm2, h.permute(0, 3, 1, 2), True).permute(0, 2, 3, 1).contiguous()
        y = conv3x3_contract(h, blk.conv2._prepared()[0], blk.conv2._bias_p, residual=x.permute(0, 2, 3, 1).contiguous())
    assert torch.allclose(y.permute(0, 3, 1, 2), ref, atol=1e-10)


def test_splitk_weight_gradient_of_the_feature_major_mlp():
    torch.manual_seed(5)
    g = torch.randn(5, 70001, dtype=torch.float64)
    h = torch.randn(64, 70001, dtype=torch.float64)
    assert torch.allclose(_wgrad_splitk(g, h), g @ h.t(), atol=1e-9)
    gt = torch.randn(70001, 5, dtype=torch.float64).t()                              # non-contiguous, as autograd delivers it
    assert torch.allclose(_wgrad_splitk(gt, h), gt @ h.t(), atol=1e-9)
    assert torch.allclose(_wgrad_splitk(g[:, :100], h[:, :100]), g[:, :100] @ h[:, :100].t(), atol=1e-12)   # tiny: plain GEMM
    mlp = VanillaMLP(32, 5, {"n_neurons": 64, "n_hidden_layers": 1})
    xf = torch.randn(32, 40000).t()                                                  # feature-major [N, 32] view
    assert xf.stride(0) == 1
    xa = xf.detach().requires_grad_()
    mlp(xa).square().sum().backward()
    ga = [p.grad.clone() for p in mlp.parameters()]
    gxa = xa.grad.clone()
    for p in mlp.parameters():
        p.grad = None
    xb = xf.contiguous().detach().requires_grad_()
    mlp.layers(xb).square().sum().backward()
    for a, p in zip(ga, mlp.parameters()):
        assert (a - p.grad).abs().max() <= 1e-5 * p.grad.abs().max()
    assert (gxa - xb.grad).abs().max() <= 1e-5 * xb.grad.abs().max()


def test_bench_reads_counter_profiles():
    import bench
    tr = bench.pmc_traffic("k_conv3x3_dma conv3x3[128->128@512x512,s1]", 8)
    assert tr and tr["traffic_unit"] == "bytes/launch" and tr["traffic_source"].startswith("profiles/")
    assert 0.9 * tr["algorithmic_bytes"] < tr["traffic"] < 3.0 * tr["algorithmic_bytes"]
    assert bench.pmc_traffic("k_conv3x3_dma conv3x3[128->128@512x512,s1]", 3) is None      # shape not profiled
    assert bench.pmc_traffic("k_unknown foo", 8) is None


def test_bench_reads_shade_counters_of_the_real_gbuffer():
    import bench
    tr = bench.shade_traffic("shade_fwd", "rgb18e8")
    assert tr and tr["traffic_source"].startswith("profiles/") and 30e6 < tr["traffic"] < 400e6      # ~64 MB algorithmic at 1.15 M px
    assert bench.shade_traffic("shade_bwd", "rgb18e8")["traffic"] > tr["traffic"] * 0.8
    assert bench.shade_traffic("no_such_kernel", "rgb18e8") is None


def test_geglu_interleave_layout_is_the_fused_kernels_contract():
    """dm_gemm_bf16_fused(geglu=1) reads weight rows in blocks of 64: 32 value rows of output channels 32t..32t+31, then their
    32 gate rows; a torch emulation of that contract on the interleaved weight must equal the unfused Linear -> GEGLU."""
    from dreammat_amd import hipops
    torch.manual_seed(0)
    inner, K, M = 96, 40, 7
    w, b, x = torch.randn(2 * inner, K), torch.randn(2 * inner), torch.randn(M, K)
    wi, bi = hipops.geglu_interleave(w), hipops.geglu_interleave(b)
    assert wi.shape == w.shape and torch.equal(wi[:32], w[:32]) and torch.equal(wi[32:64], w[inner:inner + 32])
    h = x @ wi.t() + bi                                        # what the GEMM accumulates, columns in interleaved order
    hv = h.view(M, inner // 32, 2, 32)
    fused = (hv[:, :, 0] * torch.nn.functional.gelu(hv[:, :, 1])).reshape(M, inner)
    ref = x @ w.t() + b
    ref = ref[:, :inner] * torch.nn.functional.gelu(ref[:, inner:])
    assert torch.allclose(fused, ref, atol=1e-5)
    assert hipops.gemm_fused_ok(4096, 320, 2560, geglu=True) and not hipops.gemm_fused_ok(4096, 320, 2560 + 64, geglu=True)
    assert not hipops.gemm_fused_ok(4100, 320, 320) and not hipops.gemm_fused_ok(4096, 300, 320)


def test_step_window_tool_cuts_whole_steps_out_of_a_kernel_trace(tmp_path):
    """tools/step_window.py: per-step table between the starts of a once-per-step kernel (k_hg_acc)."""
    import csv
    import subprocess
    import sys
    rows = []
    t = 1000
    for step in range(5):
        for name, dur in (("setup_only" if step == 0 else "k_a", 10000), ("(anonymous namespace)::k_hg_acc(BinArgs)", 50000), ("k_b", 30000), ("k_b", 30000)):
            rows.append({"Kernel_Name": name, "Start_Timestamp": t, "End_Timestamp": t + dur})
            t += dur + 5
    f = tmp_path / "trace.csv"
    with open(f, "w", newline="") as fh:
        wr = csv.DictWriter(fh, fieldnames=list(rows[0]))
        wr.writeheader()
        wr.writerows(rows)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "step_window.py"), str(f), "1", "4"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    tab = {r[0]: r for r in csv.reader(out.stdout.splitlines())}
    assert float(tab["k_b"][1]) == 2.0 and float(tab["k_a"][1]) == 1.0 and "setup_only" not in tab
    assert abs(float(tab["k_b"][2]) - 0.060) < 1e-9                                  # ms per step


def test_fibonacci_direction_tables_match_the_reference():
    """hipops.fibonacci_direction_samples == the (azimuth, elevation) tables DreamMatMaterial.configure builds from the
    reference's sample_sphere (tests/golden/mc_shading.npz stores the reference's own tables)."""
    import os
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "mc_shading.npz"))
    for key in ("schlick_dsamp", "schlick_ssamp", "ggx_smith_dsamp", "ggx_smith_ssamp"):
        ref = torch.from_numpy(g[key])
        assert torch.equal(hipops.fibonacci_direction_samples(ref.shape[0]), ref), key


def test_mesh_exporter_bakes_textures_and_writes_obj_mtl(monkeypatch, tmp_path):
    """`mesh-exporter` (threestudio/models/exporters/mesh_exporter.py).  The GPU rasterize /
    interpolate entry points are replaced by the oracle's CPU restatement of the same contracts, so the test covers
    the baking recipe (UV clip transform, position interpolation with the POSITION triangles, field queries, chart
    padding) and the OBJ / MTL / texture writers."""
    import numpy as np
    from PIL import Image

    import dreammat_amd
    from dreammat_amd import mesh as pmesh, saving
    from dreammat_amd.exporter import MeshExporter, dilate_charts
    from oracle import raster as oraster

    class CpuRaster:
        def __init__(self, device):
            pass

        def rasterize(self, pos, tri, H, W, check_overflow=False):
            return torch.from_numpy(oraster.rasterize(pos, tri, H, W))
    monkeypatch.setattr(hipops, "RasterContext", CpuRaster)
    monkeypatch.setattr(hipops, "interpolate", lambda attr, rast, tri: torch.from_numpy(oraster.interpolate(attr, rast, tri)))

    m = pmesh.quad_mesh()
    m.v_tex = m.v_tex * 0.5 + 0.125                       # chart covers [0.125, 0.625]^2: the rest of the atlas is holes

    class Geo:
        def isosurface(self):
            return m

        def export(self, points, **kw):
            return {"features": torch.cat([points * 4.0, points[:, :2] * 2.0], dim=-1)}     # 5 "features"

    class Mat:
        def export(self, features, **kw):
            s = torch.sigmoid(features)
            return {"albedo": s[..., :3], "metallic": s[..., 3:4], "roughness": s[..., 4:5]}
    S = 32
    ex = MeshExporter({"texture_size": S, "texture_format": "png", "xatlas_pack_options": {"padding": 3}},
                      geometry=Geo(), material=Mat(), background=None)
    maps, holes = ex.bake_textures(m)
    assert maps["albedo"].shape == (S, S, 3) and maps["metallic"].shape == (S, S, 1)
    jj, ii = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    u, v = (ii + 0.5) / S, (jj + 0.5) / S                 # texel centres; image row 0 = v near 0 (clip y = -1)
    inside = (u > 0.125) & (u < 0.625) & (v > 0.125) & (v < 0.625)
    assert torch.equal(~holes, inside)
    pos = torch.stack([(u - 0.125) / 0.5 - 0.5, (v - 0.125) / 0.5 - 0.5, torch.zeros_like(u)], -1)   # quad: x = u' - .5
    ref = torch.sigmoid(pos * 4.0)
    assert (maps["albedo"][inside] - ref[inside]).abs().max() < 1e-5
    ring = dilate_charts(torch.ones(S, S, 1), holes, 3)[..., 0]
    assert int((ring > 0).sum()) == int((((u > 0.125 - 3 / S) & (u < 0.625 + 3 / S) & (v > 0.125 - 3 / S) & (v < 0.625 + 3 / S))).sum())
    far = ~((u > 0.125 - 3.5 / S) & (u < 0.625 + 3.5 / S) & (v > 0.125 - 3.5 / S) & (v < 0.625 + 3.5 / S))
    assert maps["albedo"][far].abs().max() == 0 and maps["albedo"][~holes].min() > 0
    # padded texels continue the chart: values within the range of the chart border
    border = maps["albedo"][(ring > 0) & holes]
    assert border.min() > 0 and border.max() < 1

    outs = ex()
    assert len(outs) == 1 and outs[0].save_name == "model.obj" and outs[0].save_type == "obj"
    paths = saving.save_obj(str(tmp_path / "exp" / outs[0].save_name), **outs[0].params)
    names = sorted(os.path.basename(p) for p in paths)
    assert names == ["model.mtl", "model.obj", "texture_kd.png", "texture_metallic.png", "texture_roughness.png"]
    mtl = open(tmp_path / "exp" / "model.mtl").read()
    assert "newmtl default" in mtl and "map_Kd texture_kd.png" in mtl and "map_Pm texture_metallic.png" in mtl
    obj = open(tmp_path / "exp" / "model.obj").read().splitlines()
    assert obj[0] == "mtllib model.mtl" and obj[2] == "usemtl default"
    vts = [l for l in obj if l.startswith("vt ")]
    assert len(vts) == 4 and vts[0] == f"vt {float(m.v_tex[0, 0])} {1.0 - float(m.v_tex[0, 1])}"
    assert [l for l in obj if l.startswith("f ")][0] == "f 1/1/ 2/2/ 3/3/"
    kd = np.asarray(Image.open(tmp_path / "exp" / "texture_kd.png"))
    assert kd.shape == (S, S, 3) and np.abs(kd.astype(np.float32) / 255 - maps["albedo"].numpy()).max() < 1 / 255 + 1e-6
    re = pmesh.load_obj(str(tmp_path / "exp" / "model.obj"))
    assert torch.allclose(re.v_pos, m.v_pos) and torch.equal(re.t_pos_idx, m.t_pos_idx)
    # vertex-colour variant (a mesh without UVs gets the per-triangle atlas: test_per_triangle_atlas_fallback_and_gif_writer)
    ex2 = MeshExporter({"fmt": "obj", "save_uv": False}, geometry=Geo(), material=Mat(), background=None)
    p2 = saving.save_obj(str(tmp_path / "exp2" / "m"), **ex2()[0].params)
    first_v = [l for l in open(p2[0]).read().splitlines() if l.startswith("v ")][0].split()
    assert len(first_v) == 7                                # x y z r g b
    assert "mesh-exporter" in dreammat_amd.__modules__ or dreammat_amd.find("mesh-exporter") is MeshExporter


def test_normalize_mesh_follows_the_reference_on_an_asymmetric_mesh(tmp_path):
    """dreammat_mesh.py:161-193 subtracts the vertex centroid and divides by the largest absolute
    coordinate (NOT bbox centre / longest extent); restated here in numpy on an L-shaped, off-centre OBJ."""
    import numpy as np
    import dreammat_amd
    from dreammat_amd import mesh as pmesh
    obj = tmp_path / "ell.obj"
    verts = np.array([[0, 0, 0], [4, 0, 0], [4, 1, 0], [1, 1, 0], [1, 3, 0], [0, 3, 0], [0, 0, 2.5], [4, 0, 0.5]], dtype=np.float64) + [10, -2, 1]
    faces = [[1, 2, 3], [1, 3, 4], [1, 4, 5], [1, 5, 6], [1, 2, 8], [1, 8, 7]]
    with open(obj, "w") as fh:
        for v in verts:
            fh.write(f"v {v[0]} {v[1]} {v[2]}\n")
        for k in range(len(verts)):
            fh.write(f"vt {k / 8.0} {1 - k / 8.0}\n")
        for f in faces:
            fh.write("f " + " ".join(f"{i}/{i}" for i in f) + "\n")
    dreammat_amd._import_plugins()
    enc = {"otype": "HashGrid", "n_levels": 2, "n_features_per_level": 2, "log2_hashmap_size": 8, "base_resolution": 4,
           "per_level_scale": 1.5}
    geo = dreammat_amd.find("dreammat-mesh")({"shape_init": f"mesh:{obj}", "shape_init_params": 0.7, "shape_init_mesh_up": "+y",
                                              "shape_init_mesh_front": "+z", "pos_encoding_config": enc})
    # the reference's arithmetic (centroid, abs-max scale, then std2mesh^-1 with z_=up, x_=front)
    v = verts - verts.mean(0)
    v = v / np.abs(v).max() * 0.7
    z_, x_ = np.array([0, 1, 0.0]), np.array([0, 0, 1.0])
    std2mesh = np.stack([x_, np.cross(z_, x_), z_], 0).T
    ref = (np.linalg.inv(std2mesh) @ v.T).T
    got = geo.v_buffer.numpy().astype(np.float64)       # the loader numbers corners by first use in a face: compare as sets
    key = lambda a: a[np.lexsort(np.round(a, 4).T[::-1])]
    assert np.abs(key(got) - key(ref)).max() < 1e-6
    lo, hi = ref.min(0), ref.max(0)
    assert np.abs((lo + hi) / 2).max() > 0.05            # the bbox centre is NOT at the origin: the two conventions differ
    assert geo.vtex_buffer.shape == (8, 2)               # registered like the reference's vtex_buffer (device-resident)
    assert geo.isosurface().v_tex is geo.vtex_buffer


def test_missing_weights_are_an_error_unless_synthetic_is_requested(tmp_path, monkeypatch):
    """pseudo embeddings / random nets must never stand in silently for a real model name, and
    synthetic embeddings must not land in the cache slot real ones are read from."""
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("DREAMMAT_SD_DIR", raising=False)
    from dreammat_amd.prompt import StableDiffusionPromptProcessor
    real = {"prompt": "a chair", "pretrained_model_name_or_path": "stabilityai/stable-diffusion-2-1-base",
            "pretrained_model_cache_dir": str(tmp_path / "nope"), "cache_dir": str(tmp_path / "cache")}
    with pytest.raises(FileNotFoundError):
        StableDiffusionPromptProcessor(dict(real))
    pp = StableDiffusionPromptProcessor(dict(real, synthetic=True))
    files = os.listdir(tmp_path / "cache")
    assert files and all(f.endswith(".synthetic.pt") for f in files)
    assert pp.text_embeddings.shape == (1, 77, 1024)
    from dreammat_amd.sd.models import ARCHS
    import dreammat_amd.guidance as G
    monkeypatch.setattr(G, "arch_for", lambda name: ARCHS["tiny"])       # keep the construction small
    cfg = {"pretrained_model_name_or_path": "stabilityai/stable-diffusion-2-1-base", "use_controlnet": True,
           "control_types": ["light"], "condition_scales": [1.0], "cache_dir": str(tmp_path / "nope")}
    with pytest.raises(FileNotFoundError):
        G.StableDiffusionLightGuidance(dict(cfg))
    gd = G.StableDiffusionLightGuidance(dict(cfg, synthetic=True))
    assert gd.real_weights == {"vae": False, "unet": False, "controlnet": False}


def test_partial_weight_load_skips_reference_only_keys(tmp_path):
    """`weights: path:module` must accept a reference checkpoint that carries the predictor
    heads this repo does not instantiate."""
    import dreammat_amd
    dreammat_amd._import_plugins()
    enc = {"otype": "HashGrid", "n_levels": 2, "n_features_per_level": 2, "log2_hashmap_size": 8, "base_resolution": 4,
           "per_level_scale": 1.5}
    cfg = {"shape_init": "quad", "shape_init_params": 1.0, "pos_encoding_config": enc}
    g0 = dreammat_amd.find("dreammat-mesh")(dict(cfg))
    with torch.no_grad():
        g0.encoding.encoding.params.uniform_(-1, 1)
    sd = {"geometry." + k: v for k, v in g0.state_dict().items()}
    sd["geometry.albedo_predictor.layers.0.weight"] = torch.zeros(4, 4)   # reference-only head
    torch.save({"state_dict": sd, "epoch": 0, "global_step": 7}, tmp_path / "ref.ckpt")
    g1 = dreammat_amd.find("dreammat-mesh")(dict(cfg, weights=f"{tmp_path / 'ref.ckpt'}:geometry"))
    assert torch.equal(g1.encoding.encoding.params, g0.encoding.encoding.params)
    # anything else is an error, as in the reference (utils/base.py:109 is strict) -- a wrong
    # module prefix (nothing matches => every key missing), a truncated checkpoint, or an unknown extra key
    with pytest.raises(RuntimeError, match="missing"):
        dreammat_amd.find("dreammat-mesh")(dict(cfg, weights=f"{tmp_path / 'ref.ckpt'}:geometri"))
    bad = dict(sd)
    bad["geometry.some_new_head.weight"] = torch.zeros(2)
    torch.save({"state_dict": bad}, tmp_path / "bad.ckpt")
    with pytest.raises(RuntimeError, match="unexpected"):
        dreammat_amd.find("dreammat-mesh")(dict(cfg, weights=f"{tmp_path / 'bad.ckpt'}:geometry"))
    trunc = {k: v for k, v in sd.items() if "feature_network" not in k}
    torch.save({"state_dict": trunc}, tmp_path / "trunc.ckpt")
    with pytest.raises(RuntimeError, match="missing"):
        dreammat_amd.find("dreammat-mesh")(dict(cfg, weights=f"{tmp_path / 'trunc.ckpt'}:geometry"))


def test_per_triangle_atlas_fallback_and_gif_writer(monkeypatch, tmp_path):
    """f-3 remainder: a mesh without UVs gets the per-triangle atlas (mesh_exporter.py:53-60 would call xatlas):
    every face its own chart, charts disjoint and inside [0,1]^2, each chart covered by at least one texel at the default
    resolution; and saving.save_gif (saving.py:401-408) assembles `<i>.png` frames into eval.gif."""
    import numpy as np
    from PIL import Image

    from dreammat_amd import mesh as pmesh, saving
    from dreammat_amd.exporter import MeshExporter, per_triangle_atlas
    from oracle import raster as oraster

    nf, S = 37, 256
    uv, tt = per_triangle_atlas(nf, S, 2, torch.device("cpu"))
    assert uv.shape == (3 * nf, 2) and tt.shape == (nf, 3) and float(uv.min()) > 0 and float(uv.max()) < 1
    clip = torch.cat([uv * 2 - 1, torch.zeros(3 * nf, 1), torch.ones(3 * nf, 1)], -1)[None].numpy()
    ro = oraster.rasterize(clip, tt.numpy().astype(np.int32), S, S)
    ids = ro[0, :, :, 3].astype(np.int64)
    assert set(np.unique(ids)) == set(range(nf + 1))             # 0 = gutter; every face owns texels
    # charts are disjoint with a gutter: no two DIFFERENT faces in 4-neighbouring texels
    for a, b in ((ids[:, 1:], ids[:, :-1]), (ids[1:], ids[:-1])):
        touch = (a != b) & (a > 0) & (b > 0)
        assert not touch.any()

    class CpuRaster:
        def __init__(self, device):
            pass

        def rasterize(self, pos, tri, H, W, check_overflow=False):
            return torch.from_numpy(oraster.rasterize(pos, tri, H, W))
    monkeypatch.setattr(hipops, "RasterContext", CpuRaster)
    monkeypatch.setattr(hipops, "interpolate", lambda attr, rast, tri: torch.from_numpy(oraster.interpolate(attr, rast, tri)))
    m = pmesh.quad_mesh()
    m.v_tex, m.t_tex_idx = None, None

    class Geo:
        def isosurface(self):
            return m

        def export(self, points, **kw):
            return {"features": torch.cat([points * 4.0, points[:, :2] * 2.0], dim=-1)}

    class Mat:
        def export(self, features, **kw):
            s = torch.sigmoid(features)
            return {"albedo": s[..., :3], "metallic": s[..., 3:4], "roughness": s[..., 4:5]}
    ex = MeshExporter({"texture_size": 64, "texture_format": "png"}, geometry=Geo(), material=Mat(), background=None)
    outs = ex()                                                  # used to raise NotImplementedError without UVs
    assert m.v_tex is not None and outs[0].params["map_Kd"].shape == (64, 64, 3)
    paths = saving.save_obj(str(tmp_path / "m.obj"), **outs[0].params)
    assert any(p.endswith("texture_kd.png") for p in paths)
    # gif
    d = tmp_path / "view"
    d.mkdir()
    for i in range(5):
        Image.fromarray(np.full((8, 12, 3), 40 * i, np.uint8)).save(d / f"{i}.png")
    g = saving.save_gif(str(d), n_frames=120, fps=30)
    im = Image.open(g)
    assert im.n_frames == 5 and im.size == (12, 8)
    with pytest.raises(FileNotFoundError):
        saving.save_gif(str(tmp_path / "nothing"))


def test_resident_condition_tables_are_gated_on_free_hbm():
    """data.RandomCameraDataModule: `resident: null` keeps the (view, env) condition maps in HBM only when they fit in a
    quarter of the memory that is free; an explicit true / false is obeyed; a CPU dataset is never resident."""
    import types

    from dreammat_amd.data import RandomCameraDataModule
    dm = RandomCameraDataModule(cfg={"height": 512, "width": 512, "batch_size": 1, "use_fix_views": True})
    dm.setup("fit")
    ds = dm.train_dataset
    assert ds.resident is False                                       # lives on the CPU here
    n, ne = ds.cfg.fix_view_num, ds.cfg.fix_env_num
    assert ds.resident_table_bytes() == n * ne * 512 * 512 * 22 * 4 + n * 512 * 512 * 3 * 4
    fake = types.SimpleNamespace(device=types.SimpleNamespace(type="cuda"), resident_table_bytes=ds.resident_table_bytes)
    assert type(ds)._resident_default(fake, free_bytes=280 << 30) is True       # an idle MI355X
    assert type(ds)._resident_default(fake, free_bytes=32 << 30) is False       # 14.9 GB of tables in 32 GB free: host collate


def test_prompt_processor_runs_the_real_clip_path_on_a_locally_built_checkpoint(tmp_path, monkeypatch):
    """stable_diffusion_prompt_processor.py:56-106 (tokenizer + CLIPTextModel through transformers, cached per prompt): no CLIP
    weights exist on this box, so every other test runs on md5-seeded pseudo embeddings -- here a one-layer CLIP text model with
    random weights and a character-level byte-pair vocabulary is written in the `from_pretrained` layout and the processor's
    REAL branch encodes the prompt, the negative prompt, "" and the four view-dependent prompts, caches them as `.pt` (not
    `.synthetic.pt`), and a second processor is served from the cache without touching the encoder."""
    import json

    from transformers import AutoTokenizer, CLIPTextConfig, CLIPTextModel, CLIPTokenizer

    from dreammat_amd.prompt import StableDiffusionPromptProcessor
    root = tmp_path / "SD" / "runwayml" / "stable-diffusion-v1-5"
    (root / "tokenizer").mkdir(parents=True)
    chars = list("abcdefghijklmnopqrstuvwxyz0123456789,.' ")
    vocab = {c: i for i, c in enumerate(chars)}
    vocab.update({c + "</w>": len(chars) + i for i, c in enumerate(chars)})
    vocab["<|startoftext|>"], vocab["<|endoftext|>"] = len(vocab), len(vocab) + 1
    (root / "tokenizer" / "vocab.json").write_text(json.dumps(vocab))
    (root / "tokenizer" / "merges.txt").write_text("#version: 0.2\n")
    CLIPTokenizer(str(root / "tokenizer" / "vocab.json"), str(root / "tokenizer" / "merges.txt"),
                  model_max_length=77).save_pretrained(str(root / "tokenizer"))
    torch.manual_seed(0)
    CLIPTextModel(CLIPTextConfig(vocab_size=len(vocab), hidden_size=768, intermediate_size=64, num_hidden_layers=1,
                                 num_attention_heads=2, max_position_embeddings=77, bos_token_id=vocab["<|startoftext|>"],
                                 eos_token_id=vocab["<|endoftext|>"], pad_token_id=vocab["<|endoftext|>"])
                  ).save_pretrained(str(root / "text_encoder"))
    cfg = {"prompt": "a wooden chair", "negative_prompt": "ugly", "pretrained_model_name_or_path": "runwayml/stable-diffusion-v1-5",
           "pretrained_model_cache_dir": str(tmp_path / "SD"), "cache_dir": str(tmp_path / "cache")}
    pp = StableDiffusionPromptProcessor(dict(cfg))
    files = sorted(os.listdir(tmp_path / "cache"))
    assert len(files) == 7 and all(f.endswith(".pt") and not f.endswith(".synthetic.pt") for f in files)   # prompt, "ugly", "", 4 views
    tok = AutoTokenizer.from_pretrained(str(root / "tokenizer"))
    enc = CLIPTextModel.from_pretrained(str(root / "text_encoder"))
    with torch.no_grad():
        want = enc(tok(["a wooden chair", "a wooden chair, back view"], padding="max_length", max_length=77,
                       return_tensors="pt").input_ids)[0]
    assert pp.text_embeddings.shape == (1, 77, 768) and torch.allclose(pp.text_embeddings[0], want[0], atol=1e-6)
    assert torch.allclose(pp.text_embeddings_vd[pp.direction2idx["back"]], want[1], atol=1e-6)
    assert not torch.allclose(want[0], want[1])                       # the view suffix reaches the encoder
    out = pp()
    emb = out.get_text_embeddings(torch.tensor([10.0]), torch.tensor([170.0]), torch.tensor([3.0]))
    assert torch.allclose(emb[0], want[1], atol=1e-6)                 # azimuth 170 degrees: the "back" prompt
    monkeypatch.setattr(StableDiffusionPromptProcessor, "_encode", lambda self, p: (_ for _ in ()).throw(AssertionError("cache miss")))
    pp2 = StableDiffusionPromptProcessor(dict(cfg))
    assert torch.equal(pp2.text_embeddings, pp.text_embeddings)


def conv3x3_wgrad_contract(x_nhwc, dy_nhwc, stride):
    """dm_conv3x3_wgrad_nhwc_bf16's contract as hipops.conv3x3_wgrad hands it on: dW [Cout, Cin, 3, 3] (fp32 there) of a 3x3 /
    pad 1 convolution from its NHWC input and output gradient."""
    with torch.enable_grad():                       # called from inside an autograd backward
        x = x_nhwc.detach().permute(0, 3, 1, 2)
        w = torch.zeros(dy_nhwc.shape[3], x_nhwc.shape[3], 3, 3, dtype=x.dtype, requires_grad=True)
        F.conv2d(x, w, stride=stride, padding=1).backward(dy_nhwc.detach().permute(0, 3, 1, 2))
    return w.grad


@pytest.mark.parametrize("stride", [1, 2])
def test_trainable_conv_autograd_algebra_around_the_three_kernels(contract, monkeypatch, stride):
    """hipops._Conv3x3Train (the trainable convolutions of the ControlNet training loop): with the forward / data-gradient kernel
    and the weight-gradient kernel replaced by their torch contracts, the host algebra -- tap-major weights, flipped and
    channel-swapped weights for the data gradient, the zero-inserted output gradient at stride 2, dW layout, bias gradient --
    reproduces F.conv2d's autograd for x, weight and bias."""
    monkeypatch.setattr(hipops, "conv3x3_wgrad", conv3x3_wgrad_contract)
    torch.manual_seed(7)
    x = torch.randn(2, 6, 5, 8, dtype=torch.float64, requires_grad=True)          # NHWC
    w = torch.randn(7, x.shape[3], 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.randn(7, dtype=torch.float64, requires_grad=True)
    y = hipops.conv3x3_train(x, w, b, stride)
    g = torch.randn_like(y)
    y.backward(g)
    x2, w2, b2 = (t.detach().clone().requires_grad_(True) for t in (x, w, b))
    ref = F.conv2d(x2.permute(0, 3, 1, 2), w2, b2, stride=stride, padding=1).permute(0, 2, 3, 1)
    assert torch.allclose(y.detach(), ref.detach(), atol=1e-12)
    ref.backward(g)
    for got, want in ((x.grad, x2.grad), (w.grad, w2.grad), (b.grad, b2.grad)):
        assert torch.allclose(got, want, atol=1e-11)


def test_training_route_gates_accept_the_kernel_domains_only():
    """hipops.attention_train_ok / conv3x3_train_ok (pure host logic): what the differentiated MFMA routes serve, and what is
    handed back to torch autograd (im2col, matmul-softmax) -- decided before any kernel is called, never by a failing launch."""
    from types import SimpleNamespace as NS

    def t(shape, dtype=torch.bfloat16, cuda=True):
        return NS(shape=torch.Size(shape), dtype=dtype, is_cuda=cuda)
    q, k = t((2, 256, 320)), t((2, 77, 320))
    assert hipops.attention_train_ok(q, k, t((2, 77, 320)), 5)                       # D = 64
    assert hipops.attention_train_ok(t((2, 64, 1280)), t((2, 64, 1280)), t((2, 64, 1280)), 10)     # D = 128
    assert not hipops.attention_train_ok(q, k, t((2, 77, 320)), 2)                   # D = 160: beyond the backward's head sizes
    assert not hipops.attention_train_ok(q, k, t((2, 77, 320), torch.float32), 5)    # fp32 values
    assert not hipops.attention_train_ok(t((2, 256, 320), cuda=False), k, t((2, 77, 320)), 5)
    assert not hipops.attention_train_ok(q, k, t((2, 80, 320)), 5)                   # k / v of different lengths
    w = lambda co, ci, dtype=torch.bfloat16: NS(shape=torch.Size((co, ci, 3, 3)), dtype=dtype)
    x = lambda b, h, wd, c: NS(shape=torch.Size((b, h, wd, c)), dtype=torch.bfloat16, is_cuda=True)
    ok = hipops.conv3x3_train_ok
    assert ok(x(4, 64, 64, 320), w(320, 320), (1, 1), (1, 1)) and ok(x(4, 64, 64, 320), w(640, 320), (2, 2), (1, 1))
    assert ok(x(2, 8, 8, 1280), w(1280, 1280), (1, 1), (1, 1))                       # 64 pixels per image: one chunk
    assert not ok(x(4, 64, 64, 320), w(320, 320), (1, 1), (0, 0))                    # padding other than 1
    assert not ok(x(4, 64, 64, 96), w(256, 96), (2, 2), (1, 1))                      # Cin not a multiple of 64 (conditioning embedding)
    assert not ok(x(4, 4, 4, 1280), w(1280, 1280), (1, 1), (1, 1))                   # 16 pixels per image: below one chunk
    assert not ok(x(1, 128, 128, 64), w(64, 64), (1, 1), (1, 1))                     # rows wider than one 64-pixel chunk
    assert not ok(x(4, 64, 64, 320), w(320, 320, torch.float32), (1, 1), (1, 1))     # fp32 master weights
    assert not ok(x(1, 64, 48, 64), w(64, 64), (1, 1), (1, 1))                       # width not a power of two


def test_static_isa_guards_on_the_lds_dma_main_loops():
    """No GPU needed: the steady-state blocks of the two LDS-DMA kernels that carry the step must keep their counted waits.
    hipcc inserts `s_waitcnt vmcnt(0)` in front of any LDS read it believes may alias a `buffer_load ... lds` destination, which
    silently drains the DMA ring every tile (round 1: 620 -> 815 TF/s when removed; round 3: the untransposed-V attention
    experiment, profiles/r03_experiments/attn_w64_natural_v.json) -- a source edit that brings it back fails here."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("isa_waits", os.path.join(root, "tools", "isa_waits.py"))
    iw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(iw)

    for defines in ((), ("-DDM_F16",)):         # the bf16 build and the IEEE-half instantiation of the same sources (csrc/dm_elem.h)
        asm = {src: iw.assembly(os.path.join(root, "dreammat_amd", "csrc", src), defines) for src in ("attn_w64.hip", "conv.hip", "attn_w128.hip")}

        def steady_blocks(src, pattern, n_mfma):
            (name, body), = list(iw.kernels(asm[src], pattern))
            return [ins for _, ins in iw.blocks(body) if sum(i.startswith("v_mfma") for i in ins) == n_mfma]
        tiles = steady_blocks("attn_w64.hip", "k_attn_fwd_w64ILi2E", 32)
        tiles = [t for t in tiles if len(t) < 300]                      # the unrolled ring of the main loop (not the exact path)
        assert len(tiles) >= 5
        for t in tiles:
            waits = [i.split(None, 1)[1] for i in t if i.startswith("s_waitcnt")]
            assert not any("vmcnt(0)" in w for w in waits), waits
            assert 12 <= sum(i.startswith("ds_read_b128") for i in t) <= 16 and sum(i.startswith("buffer_load") for i in t) <= 4
        # (round 4: + the 2 x 2 instance of the stride-2 data gradients / upsample convs, the 128-row tile of the balanced Cout = 320 launches)
        for pattern in ("k_conv3x3_dmaILi512ELi128ELi8ELi4ELi2ELi9ELi0", "k_conv3x3_dmaILi256ELi256ELi8ELi2ELi2ELi9ELi0",
                        "k_conv3x3_dmaILi256ELi256ELi8ELi2ELi2ELi4ELi0"):
            chunks = steady_blocks("conv.hip", pattern, 8)
            assert chunks
            for c in chunks:
                assert not any(i.startswith("s_waitcnt") and "vmcnt(0)" in i for i in c)
                assert not any("scratch_" in i for i in c)
        # the 128-row attention kernel: one unrolled block of five tiles, 64 MFMAs each, no scratch, no drained DMA ring
        (name, body), = list(iw.kernels(asm["attn_w128.hip"], "k_attn_fwd_w128ILi2ELi4E"))
        main = max((ins for _, ins in iw.blocks(body)), key=lambda ins: sum(i.startswith("v_mfma") for i in ins))
        assert sum(i.startswith("v_mfma") for i in main) == 320
        assert not any("scratch_" in i for i in body)
        assert not any(i.startswith("s_waitcnt") and "vmcnt(0)" in i for i in main)
        assert sum(i.startswith("v_accvgpr") for i in main) <= 8, "register copies in the main loop: the AGPR / VGPR split of attn_w128 broke"


def test_half_dtype_entry_points_resolve_to_exported_symbols():
    """hipops._sym: the IEEE-half instantiation of a net kernel carries f16 in place of bf16 (or an _f16 suffix where the bf16 name
    has no dtype token); every such name is exported by the library and declared with the bf16 entry point's signature."""
    from dreammat_amd import _lib, hipops
    for name in ("dm_conv3x3_nhwc_bf16_fused", "dm_conv2x2_nhwc_bf16", "dm_conv2x2_subpixel_nhwc_bf16", "dm_conv3x3_small_res_nhwc_bf16", "dm_gemm_bf16_fused",
                 "dm_attention_fwd_bf16", "dm_layernorm_bf16", "dm_geglu_bf16", "dm_cat_add_bf16", "dm_softmax_rows_bf16",
                 "dm_softmax_rows_bwd_bf16", "dm_groupnorm_nhwc_fwd", "dm_groupnorm_nhwc_infer", "dm_groupnorm_nhwc_bwd_res",
                 "dm_gemm_bf16_batched", "dm_transpose_bf16", "dm_linear_small_bf16", "dm_conv3x3_gn_nhwc_bf16_fused", "dm_groupnorm_nhwc_stats"):
        fb, nb = hipops._sym(name, torch.bfloat16)
        fh, nh = hipops._sym(name, torch.float16)
        assert nb == name and nh != name and "bf16" not in nh and "f16" in nh
        assert _lib._SIGS[nh] == _lib._SIGS[name] and fb is not fh
    with pytest.raises(TypeError):
        hipops._sym("dm_layernorm_bf16", torch.float32)
    with pytest.raises(AssertionError):
        hipops._same_half(torch.zeros(1, dtype=torch.bfloat16), torch.zeros(1, dtype=torch.float16))


def test_wide_head_attention_groups_and_domain():
    """hipops._WideHeadAttention's host side: images per launch (a power of two that divides the batch, score bytes per group bounded
    whatever the batch: the workspace never scales with B) and the shapes its kernels take; a CPU tensor is refused loudly."""
    grp = hipops._wide_attn_group
    assert grp(8, 4096, 4096) == 4 and grp(16, 4096, 4096) == 4 and grp(2, 4096, 4096) == 2 and grp(1, 4096, 4096) == 1
    assert grp(16, 16384, 16384) == 1                          # BASELINE configs[4]: one 512 MB score matrix at a time
    assert grp(3, 256, 256) == 1 and grp(6, 256, 256) == 2 and grp(8, 256, 512) == 8
    for B in (1, 2, 8, 16, 24):
        g = grp(B, 4096, 4096)
        assert B % g == 0 and g * 4096 * 4096 * 2 <= hipops.WIDE_ATTN_GROUP_BYTES
    q = torch.zeros(2, 256, 128, dtype=torch.float16)
    assert not hipops.wide_head_attention_ok(q, q, q)          # (CPU tensor)
    with pytest.raises(Exception):
        hipops.wide_head_attention(q, q, q, 1.0)


def test_recorded_exits_from_the_hand_written_kernels(monkeypatch, capsys):
    """layers.note_fallback: every 16-bit CUDA call that leaves the kernels of csrc/ is counted, announced ONCE per (kind, shape),
    and an error under DREAMMAT_STRICT_KERNELS=1 -- there is no silent lowering to ATen."""
    from dreammat_amd.sd import layers
    layers.fallbacks(clear=True)
    layers.note_fallback("conv", "3x3 7->9 s1 p1")
    layers.note_fallback("conv", "3x3 7->9 s1 p1")
    layers.note_fallback("linear", "K=8 N=8 M=16 autograd")
    assert layers.fallbacks() == {("conv", "3x3 7->9 s1 p1"): 2, ("linear", "K=8 N=8 M=16 autograd"): 1}
    err = capsys.readouterr().err
    assert err.count("3x3 7->9") == 1 and "ATen / hipBLASLt" in err
    monkeypatch.setenv("DREAMMAT_STRICT_KERNELS", "1")
    with pytest.raises(RuntimeError):
        layers.note_fallback("groupnorm", "C=48 groups=16")
    assert layers.fallbacks(clear=True) and not layers.fallbacks()
    # CPU / fp32 tensors are not "expected native": the fp32 plumbing tier never reports
    assert not layers._native_expected(torch.zeros(1))


def test_bank_cast_cache_matches_the_bank_by_identity_not_by_address():
    """guidance._bank_cast: an entry keeps its source bank and is matched by identity + version; evicting the pool
    drops everything derived from the casts (graphs, per-layer K / V^T banks, grouped gathers)."""
    from dreammat_amd.guidance import StableDiffusionLightGuidance as G
    from dreammat_amd.sd import layers

    class Stub:
        weights_dtype = torch.float64
        BANK_CAST_POOL = G.BANK_CAST_POOL
        _bank_cast = G._bank_cast
        _drop_bank_projections = G._drop_bank_projections

        def __init__(self):
            self.attn = layers.Attention(8, 1, 8)
            self.attn.__dict__["_kv_banks"] = {"stale": 1}
            self.unet = torch.nn.Sequential(self.attn)
            self.unet.__dict__["_net_prologue"] = layers.NetPrologue(self.unet)
            self.unet.__dict__["_net_prologue"]._kv["stale"] = 1
            self.controlnets = []
            self._graphs = {"g": 1}
    s = Stub()
    a = torch.ones(3, 4)
    ca = s._bank_cast(a)
    assert s._bank_cast(a) is ca and ca.dtype == torch.float64
    b = torch.ones(3, 4)                                   # an equal tensor is NOT the same bank
    assert s._bank_cast(b) is not ca
    a.add_(1)                                              # a changed bank is re-cast
    assert s._bank_cast(a) is not ca and float(s._bank_cast(a)[0, 0]) == 2.0
    assert s._graphs and s.attn.__dict__.get("_kv_banks")
    for _ in range(G.BANK_CAST_POOL):                      # overflow: casts, graphs and every projection derived from them go
        s._bank_cast(torch.zeros(2, 2))
    assert not s._graphs and "_kv_banks" not in s.attn.__dict__ and not s.unet.__dict__["_net_prologue"]._kv


def test_bench_takes_no_collective_step_under_a_rank_test():
    """bench.py at world > 1: a training step (flat all-reduce) or `sync()` (barrier) inside an `if ... rank == 0 ...` body is taken
    by rank 0 alone and hangs the job -- the shade replay's extra step was written that way once."""
    import ast
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tree = ast.parse(open(os.path.join(root, "bench.py")).read())

    def names_rank(test):
        return any(isinstance(n, ast.Name) and n.id == "rank" for n in ast.walk(test))

    bad = []
    for node in ast.walk(tree):
        if isinstance(node, ast.If) and names_rank(node.test):
            for stmt in node.body:
                for c in ast.walk(stmt):
                    if isinstance(c, ast.Call):
                        f = c.func
                        nm = f.attr if isinstance(f, ast.Attribute) else getattr(f, "id", "")
                        if nm in ("train_one_step", "sync", "barrier", "all_reduce"):
                            bad.append((node.lineno, nm))
    assert not bad, bad
