"""GPU parity tests (-m gpu) of BASELINE configs[4]'s precisions: the IEEE-half instantiations of the net kernels (the same sources
as the bf16 ones compiled with -DDM_F16, csrc/dm_elem.h -- what the reference's `half_precision_weights` nets run in,
dreammat_guidance.py:56,92-94) and the MX-FP8 attention (csrc/attn_fp8.hip).  Every kernel through the C ABI against an fp32 torch
reference on the SAME 16-bit-rounded inputs; the tolerance of each comparison is the rounding of its output type and is written
where it is asserted."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from dreammat_amd import hipops

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
H16 = torch.float16


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture
def attn_variant(request):
    hipops.attention_select(request.param)
    yield request.param
    hipops.attention_select(None)


def _attn_ref(qb, kb, vb, h):
    B, Sq, C = qb.shape
    D = C // h
    qf, kf, vf = (t.float().cpu().view(B, -1, h, D).transpose(1, 2) for t in (qb, kb, vb))
    s = qf @ kf.transpose(-1, -2) * D ** -0.5
    return (torch.softmax(s, dim=-1) @ vf).transpose(1, 2).reshape(B, Sq, C)


F16_ATTN_CASES = [(2, 5, 256, 256, 64), (1, 5, 4096, 4096, 64), (3, 10, 1024, 1024, 64), (2, 5, 1024, 77, 64), (2, 8, 256, 256, 40),
                  (1, 8, 128, 128, 160), (1, 2, 100, 130, 128), (1, 2, 320, 256, 64), (1, 1, 700, 128, 64), (2, 3, 512, 576, 64)]


@pytest.mark.parametrize("attn_variant", ["auto", "w128", "w64", "v3l", "staged"], indirect=True)
@pytest.mark.parametrize("B,h,Sq,Skv,D", F16_ATTN_CASES)
def test_f16_attention_vs_fp32_reference(dev, B, h, Sq, Skv, D, attn_variant):
    torch.manual_seed(0)
    C = h * D
    qb, kb, vb = (torch.randn(B, S, C).to(dev).to(H16) for S in (Sq, Skv, Skv))
    pad = (Skv + 7) // 8 * 8
    vt = torch.zeros(B, C, pad, device=dev, dtype=H16)
    vt[:, :, :Skv] = vb.transpose(1, 2)
    out = hipops.attention(qb, kb, vt, h)
    assert out.dtype == H16
    ref = _attn_ref(qb, kb, vb, h)
    err = (out.float().cpu() - ref).abs()
    # half carries 11 significant bits through P and the output (bf16: 8): an eighth of the bf16 gates of test_hip_gpu.py
    assert err.max().item() < 3e-3 and err.mean().item() < 3e-4, (err.max().item(), err.mean().item())


@pytest.mark.parametrize("attn_variant", ["w128", "w64", "v3l"], indirect=True)
def test_f16_attention_numerator_range_takes_the_exact_path(dev, attn_variant):
    """The one-wave kernels keep un-normalised probabilities exp2(s - m_first_tile) in 16 bits.  bf16 holds them up to 2^127; half
    overflows at 65504 = 2^16 and loses precision when a shared shift scales a row below 2^-14: a logit that outgrows the first
    tile's maximum by 30 in the log2 domain (harmless for bf16) must send the workgroup through its exact path, and the rows
    around it must be unaffected."""
    torch.manual_seed(2)
    B, h, S, D = 1, 2, 512, 64
    q = torch.randn(B, S, h * D); k = torch.randn(B, S, h * D); v = torch.randn(B, S, h * D)
    q[:, 37, :D] = 1.5                       # head 0, row 37 against key 200 (4th tile): 64 * 2.25 / 8 = 18 nats = 26 in the log2 domain
    k[:, 200, :D] = 1.5
    q[:, 300, D:] = -2.0                     # head 1, row 300 (second workgroup of the 256-row kernels) against the last key: 32 nats
    k[:, 511, D:] = -2.0
    qb, kb, vb = (t.to(dev).to(H16) for t in (q, k, v))
    out = hipops.attention(qb, kb, vb.transpose(1, 2).contiguous(), h).float().cpu()
    ref = _attn_ref(qb, kb, vb, h)
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max() < 5e-3
    assert (out[0, 300, D:] - vb.float().cpu()[0, 511, D:]).abs().max() < 5e-3        # that row is its spiked key's value


def test_f16_attention_rows_far_below_a_shared_shift(dev):
    """attn_w128 shifts the four rows of a lane by the LARGEST of their first-tile maxima: a row 2^-20 below its neighbours keeps
    only subnormal half numerators -- the row-sum floor (DM_P_SUM_MIN) must catch it."""
    torch.manual_seed(3)
    B, h, S, D = 1, 1, 512, 64
    q = torch.randn(B, S, D) * 0.05; k = torch.randn(B, S, D); v = torch.randn(B, S, D)
    q[:, 32] = k[:, 3] * 0.45                # row 32 shares a lane with rows 0, 64, 96 of its wave: its maximum sits ~28 (log2) above theirs
    qb, kb, vb = (t.to(dev).to(H16) for t in (q, k, v))
    hipops.attention_select("w128")
    try:
        out = hipops.attention(qb, kb, vb.transpose(1, 2).contiguous(), h).float().cpu()
    finally:
        hipops.attention_select(None)
    ref = _attn_ref(qb, kb, vb, h)
    assert (out - ref).abs().max() < 3e-3, (out - ref).abs().max()


F16_CONV_CASES = [(2, 64, 128, 16, 16, 1), (3, 320, 320, 32, 32, 1), (1, 32, 64, 10, 10, 1), (2, 128, 256, 17, 23, 1), (2, 320, 320, 32, 32, 2),
                  (1, 640, 1280, 8, 8, 1), (1, 1280, 1280, 8, 8, 1)]


@pytest.mark.parametrize("B,Cin,Cout,H,W,stride", F16_CONV_CASES)
def test_f16_conv3x3_and_data_gradient_vs_fp32(dev, B, Cin, Cout, H, W, stride):
    from dreammat_amd.sd import layers
    torch.manual_seed(0)
    conv = layers.Conv2d(Cin, Cout, 3, stride=stride, padding=1).to(dev, H16)
    for p in conv.parameters():
        p.requires_grad_(False)
    xb = torch.randn(B, Cin, H, W).to(H16)
    with_grad = stride == 1 and Cin % 64 == 0
    xg = xb.to(dev).requires_grad_(with_grad)
    layers.fallbacks(clear=True)
    hipops.enable_kernel_timing(True)
    y = conv(xg)
    torch.cuda.synchronize()
    assert any(k.startswith("conv3x3") for k in hipops.kernel_times()) and y.dtype == H16 and not layers.fallbacks()
    hipops.enable_kernel_timing(False)
    xr = xb.float().requires_grad_(True)
    ref = torch.nn.functional.conv2d(xr, conv.weight.float().cpu(), conv.bias.float().cpu(), stride=stride, padding=1)
    err = (y.float().cpu() - ref).abs().max().item()
    assert err < 2e-3 * ref.abs().max().item() + 1e-3, err            # one rounding to half (2^-11) of an fp32 accumulation
    if with_grad:
        dy = torch.randn_like(ref).to(H16)
        y.backward(dy.to(dev))
        ref.backward(dy.float())
        gerr = (xg.grad.float().cpu() - xr.grad).abs().max().item()
        assert gerr < 2e-3 * xr.grad.abs().max().item() + 1e-3, gerr


@pytest.mark.parametrize("tile", ["", "256", "512", "320", "640"])
def test_f16_conv_fused_epilogue_and_split_k(dev, monkeypatch, tile):
    if tile:
        monkeypatch.setenv("DREAMMAT_CONV_TILE", tile)
    torch.manual_seed(1)
    for (B, Cin, Cout, H, W) in [(3, 128, 320, 20, 12), (3, 1280, 1280, 8, 8)]:
        x = torch.randn(B, H, W, Cin).to(dev).to(H16)
        w = (torch.randn(Cout, 3, 3, Cin) * 0.05).to(dev).to(H16)
        bias = torch.randn(Cout).to(dev).to(H16); rb = torch.randn(B, Cout).to(dev).to(H16); res = torch.randn(B, H, W, Cout).to(dev).to(H16)
        y = hipops.conv3x3_nhwc(x, w.reshape(Cout, 9 * Cin).contiguous(), bias, 1, (1, 1), None, rb, res).float().cpu()
        ref = torch.nn.functional.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float().cpu().permute(0, 3, 1, 2), bias.float().cpu(), padding=1)
        ref = ref.permute(0, 2, 3, 1) + rb.float().cpu()[:, None, None, :] + res.float().cpu()
        assert (y - ref).abs().max().item() < 2e-3 * ref.abs().max().item() + 1e-3


@pytest.mark.parametrize("tile,M,K,N,res", [(None, 4096, 320, 320, True), (None, 98304, 320, 320, True), ("128", 1040, 64, 192, False),
                                            ("256", 2048, 640, 1280, True), ("512", 2048, 1280, 1280, False), (None, 48, 1280, 320, False)])
def test_f16_gemm_fused_and_geglu_vs_fp32(dev, monkeypatch, tile, M, K, N, res):
    if tile:
        monkeypatch.setenv("DREAMMAT_GEMM_TILE", tile)
    torch.manual_seed(2)
    x = torch.randn(M, K).to(dev).to(H16); w = (torch.randn(N, K) * K ** -0.5).to(dev).to(H16); b = torch.randn(N).to(dev).to(H16)
    r = torch.randn(M, N).to(dev).to(H16) if res else None
    y = hipops.gemm_fused(x, w, b, r).float().cpu()
    ref = x.float().cpu() @ w.float().cpu().t() + b.float().cpu() + (r.float().cpu() if res else 0)
    assert (y - ref).abs().max().item() < 2e-3 * ref.abs().max().item() + 1e-3
    if N % 128 == 0:
        yg = hipops.gemm_fused(x, hipops.geglu_interleave(w), hipops.geglu_interleave(b), None, geglu=True).float().cpu()
        hh = x.float().cpu() @ w.float().cpu().t() + b.float().cpu()       # fp32 projection -> gate -> product, rounded once
        val, gate = hh.chunk(2, dim=-1)
        refg = val * torch.nn.functional.gelu(gate)
        assert (yg - refg).abs().max().item() < 2e-3 * refg.abs().max().item() + 1e-3


@pytest.mark.parametrize("B,C,H,W,act", [(2, 64, 8, 8, 1), (3, 320, 16, 16, 1), (2, 128, 33, 17, 0), (1, 2560, 4, 4, 1), (2, 128, 64, 64, 1)])
def test_f16_groupnorm_forward_backward_and_skip_vs_fp32(dev, B, C, H, W, act):
    torch.manual_seed(3)
    x = (torch.randn(B, H, W, C) * 2 + 0.5).to(H16)
    gamma = (1 + 0.1 * torch.randn(C)).to(H16); beta = (0.1 * torch.randn(C)).to(H16)
    xg = x.to(dev).requires_grad_(True)
    y, xs = hipops.groupnorm_nhwc_skip(xg, gamma.to(dev), beta.to(dev), 1e-5, act)
    yi = hipops.groupnorm_nhwc(x.to(dev), gamma.to(dev), beta.to(dev), 1e-5, act)       # the 2-launch inference entry
    xr = x.float().requires_grad_(True)
    ref = torch.nn.functional.group_norm(xr.permute(0, 3, 1, 2), 32, gamma.float(), beta.float(), 1e-5)
    ref = (torch.nn.functional.silu(ref) if act else ref).permute(0, 2, 3, 1)
    for got in (y, yi):
        assert got.dtype == H16 and (got.float().cpu() - ref).abs().max().item() < 4e-3
    dy = torch.randn_like(ref).to(H16); ds = torch.randn_like(ref).to(H16)
    (y.float() * dy.to(dev).float()).sum().backward(retain_graph=True)
    g1 = xg.grad.clone(); xg.grad = None
    ((y.float() * dy.to(dev).float()).sum() + (xs.float() * ds.to(dev).float()).sum()).backward()
    (ref * dy.float()).sum().backward()
    gref = xr.grad
    tol = 4e-3 * gref.abs().max().item() + 2e-3
    assert (g1.float().cpu() - gref).abs().max().item() < tol
    assert (xg.grad.float().cpu() - (gref + ds.float())).abs().max().item() < tol + 2e-3


def test_f16_row_kernels_vs_fp32(dev):
    torch.manual_seed(4)
    for rows, C in [(1000, 320), (77, 640), (513, 1280), (5, 2048)]:
        x = (torch.randn(rows, C) * 3 + 1).to(H16); g = torch.randn(C).to(H16); b = torch.randn(C).to(H16)
        y = hipops.layernorm_rows(x.to(dev), g.to(dev), b.to(dev), 1e-5).float().cpu()
        ref = torch.nn.functional.layer_norm(x.float(), (C,), g.float(), b.float(), 1e-5)
        assert (y - ref).abs().max().item() < 2e-3 * ref.abs().max().item() + 1e-3
    for shape in [(2, 512, 4096), (3, 77, 2048), (700, 264)]:
        s = (torch.randn(*shape) * 4).to(H16)
        sg = s.to(dev).requires_grad_(True)
        p = hipops.softmax_rows(sg, 0.37)
        sr = s.float().requires_grad_(True)
        pr = torch.softmax(sr * 0.37, -1)
        assert (p.float().cpu() - pr).abs().max().item() < 1e-3
        dp = torch.randn(*shape).to(H16)
        p.backward(dp.to(dev)); pr.backward(dp.float())
        assert (sg.grad.float().cpu() - sr.grad).abs().max().item() < 2e-3 * sr.grad.abs().max().item() + 1e-3
    h = torch.randn(4096, 2560).to(H16)
    y = hipops.geglu_rows(h.to(dev)).float().cpu()
    val, gate = h.float().chunk(2, -1)
    refg = val * torch.nn.functional.gelu(gate)
    assert (y - refg).abs().max().item() < 2e-3 * refg.abs().max().item() + 1e-3        # gelu(gate) and the product are each rounded to half
    x = torch.randn(3, 640, 16, 16).to(dev).to(H16).contiguous(memory_format=torch.channels_last)
    s = torch.randn(3, 320, 16, 16).to(dev).to(H16).contiguous(memory_format=torch.channels_last)
    r = torch.randn(3, 320, 16, 16).to(dev).to(H16).contiguous(memory_format=torch.channels_last)
    y = hipops.cat_add_nhwc(x, s, r)
    ref = torch.cat([x.float(), (s.float() + r.float())], 1)
    assert y.dtype == H16 and (y.float() - ref).abs().max().item() < 4e-3


def test_f16_stem_kernels_vs_fp32(dev):
    """the few-channel stem layers (patch kernel on the matrix pipe + the direct kernel behind it) and the VAE conv_in with its image gradient"""
    from dreammat_amd.sd import layers
    torch.manual_seed(5)
    for (B, Cin, Cout, H, W, stride, act) in [(2, 22, 16, 40, 24, 1, 1), (2, 16, 32, 32, 32, 2, 1), (1, 32, 96, 17, 9, 2, 0), (3, 4, 320, 16, 16, 1, 0)]:
        x = torch.randn(B, H, W, Cin).to(H16); w = (torch.randn(Cout, 3, 3, Cin) * 0.1).to(H16); b = torch.randn(Cout).to(H16)
        y = hipops.conv3x3_small_nhwc(x.to(dev), w.reshape(Cout, 9 * Cin).contiguous().to(dev), b.to(dev), stride, (1, 1), act).float().cpu()
        ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b.float(), stride=stride, padding=1)
        ref = (torch.nn.functional.silu(ref) if act else ref).permute(0, 2, 3, 1)
        assert (y - ref).abs().max().item() < 2e-3 * ref.abs().max().item() + 1e-3, (Cin, Cout)
    conv = layers.Conv2d(3, 128, 3, padding=1).to(dev, H16)
    for p in conv.parameters():
        p.requires_grad_(False)
    img = torch.rand(2, 3, 64, 48).to(H16)
    ig = img.to(dev).requires_grad_(True)
    y = conv(ig)
    ir = img.float().requires_grad_(True)
    ref = torch.nn.functional.conv2d(ir, conv.weight.float().cpu(), conv.bias.float().cpu(), padding=1)
    assert (y.float().cpu() - ref).abs().max().item() < 2e-3 * ref.abs().max().item() + 1e-3
    dy = torch.randn_like(ref).to(H16)
    y.backward(dy.to(dev)); ref.backward(dy.float())
    assert (ig.grad.float().cpu() - ir.grad).abs().max().item() < 2e-3 * ir.grad.abs().max().item() + 2e-3


def test_f16_tiny_nets_and_vae_gradient_vs_fp32_oracle(dev):
    """UNet + ControlNet (tiny, tiny15) and the differentiated VAE encoder in IEEE half on the hand-written kernels against the fp32
    CPU oracle -- the same comparison test_hip_gpu.py makes for bf16, whose mean error is 1.2-1.4e-2: half's three extra mantissa
    bits must show (gate: a quarter of it)."""
    from dreammat_amd.sd import ARCHS, AutoencoderKLEncoder, ControlNetModel, UNet2DConditionModel
    from dreammat_amd.sd import layers
    from oracle import sd_nets as osd
    res = {}
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
            unet.to(dev, H16); cn.to(dev, H16)
            for p in list(unet.parameters()) + list(cn.parameters()):
                p.requires_grad_(False)
            layers.fallbacks(clear=True)
            d, m = cn(x.to(dev, H16), t.to(dev), ctx.to(dev, H16), cond.to(dev, H16), 1.0)
            yh = unet(x.to(dev, H16), t.to(dev), ctx.to(dev, H16), d, m).float().cpu()
        # (the tiny test architectures have 8- / 22-channel layers the kernels' gates exclude: those run on ATen and are recorded;
        # the full-size nets leave nothing behind -- asserted in test_full_size_sd21_unet_controlnet_eps_vs_oracle)
        assert all(k in ("conv", "geglu", "linear", "groupnorm", "layernorm") for k, _ in layers.fallbacks()), layers.fallbacks()
        rel = ((yh - oy).abs().max() / oy.abs().max()).item()
        rel_mean = ((yh - oy).abs().mean() / oy.abs().mean()).item()
        res[arch_name] = {"f16_rel_max": rel, "f16_rel_mean": rel_mean}
        assert rel < 1e-2 and rel_mean < 4e-3, (arch_name, rel, rel_mean)
    # the VAE encoder under autograd (SDS differentiates through it, dreammat_guidance.py:284-292): same comparison as
    # test_vae_encoder_bf16_gradient_vs_fp32_oracle (bf16 gates 5e-2 / 8e-2)
    from dreammat_amd.sd import SDArch
    torch.manual_seed(0)
    vae = AutoencoderKLEncoder(SDArch(name="vae-test", vae_block_out=(64, 128, 128, 128))).eval()
    for p in vae.parameters():
        p.requires_grad_(False)
    img = torch.rand(2, 3, 64, 64)
    xr = img.clone().requires_grad_()
    mean, _ = osd.vae_encode_moments(vae.state_dict(), xr * 2 - 1)
    w = torch.randn_like(mean)
    (mean * w).sum().backward()
    vae.to(dev, H16)
    xg = img.to(dev).requires_grad_()
    mg, _ = vae.encode_moments((xg * 2 - 1).to(H16))
    (mg.float() * w.to(dev)).sum().backward()
    rel_l = ((mg.float().cpu() - mean).abs().max() / mean.abs().max()).item()
    rel_g = ((xg.grad.cpu() - xr.grad).abs().max() / xr.grad.abs().max()).item()
    res["vae_test_arch"] = {"f16_moments_rel_max": rel_l, "f16_image_grad_rel_max": rel_g}
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "f16_tiny_parity.json"), "w") as fh:
        json.dump(res, fh)
    assert rel_l < 1e-2 and rel_g < 2e-2, (rel_l, rel_g)


# ------------------------------------------------------------------------------------------ MX-FP8 attention
def _vt(vb):
    return vb.transpose(1, 2).contiguous()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fp8_attention_operand_layout_is_exact_on_representable_inputs(dev, dtype):
    """Layout proof, independent of quantisation noise: every operand is exactly representable in MX-FP8 (keys in {-1, +1} on the
    first half of the head dimension and {-2, +2} on the second -- two different block scales per row --, queries = 4 x their target
    key, values = small integers times a power of two that changes from one 32-row kv block to the next), and each query's
    target key out-scores all others by > 20 nats, so the output row must BE the target's value row.  A wrong lane / byte / block
    mapping anywhere (K or Q fragments, the v_permlane32_swap hand-over of P, the byte permutation of V^T, a scale byte applied
    to the wrong block) selects a different row or a different power of two."""
    torch.manual_seed(7)
    B, h, Sq, Skv, D = 2, 3, 256, 320, 64
    k = torch.where(torch.rand(B, Skv, h, D) < 0.5, -1.0, 1.0)
    k[..., 32:] *= 2.0
    tgt = torch.randint(0, Skv, (B, Sq, h))
    q = torch.gather(k, 1, tgt[..., None].expand(B, Sq, h, D)) * 4.0
    q[..., 32:] *= 0.25                      # q.k_target = 32 * 4 + 32 * 4 = 256 -> 32 nats; a random key: N(0, 16^2) / 8 = 2 nats sigma
    v = torch.randint(-8, 9, (B, Skv, h, D)).float() * (2.0 ** ((torch.arange(Skv) // 32) % 5 - 2))[None, :, None, None]
    qb, kb, vb = (t.reshape(B, -1, h * D).to(dev).to(dtype) for t in (q, k, v))
    out = hipops.attention_fp8(qb, kb, _vt(vb), h).float().cpu().view(B, Sq, h, D)
    want = torch.gather(v, 1, tgt[..., None].expand(B, Sq, h, D))
    err = (out - want).abs().max().item()
    assert err < 0.05, err                   # (values up to 32; a mis-mapped row or scale is off by whole units)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,h,Sq,Skv", [(1, 5, 4096, 4096), (2, 3, 1024, 1024), (1, 2, 256, 64), (3, 2, 256, 1984)])
def test_fp8_attention_vs_fp32_reference(dev, dtype, B, h, Sq, Skv):
    """randn inputs: the error is the e4m3 rounding (3 mantissa bits, up to 2^-4 relative per element) of Q, K, P and V -- a few per
    cent of the output's rms, which is what an fp8 attention is; the gates are on the relative rms and on the bias."""
    torch.manual_seed(0)
    D, C = 64, h * 64
    qb, kb, vb = (torch.randn(B, S, C).to(dev).to(dtype) for S in (Sq, Skv, Skv))
    out = hipops.attention_fp8(qb, kb, _vt(vb), h).float().cpu()
    ref = _attn_ref(qb, kb, vb, h)
    rel_rms = ((out - ref).norm() / ref.norm()).item()
    bias = ((out - ref).mean().abs() / ref.abs().mean()).item()
    ref16 = hipops.attention(qb, kb, _vt(vb), h).float().cpu()
    rel16 = ((ref16 - ref).norm() / ref.norm()).item()
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, f"fp8_attn_parity_{str(dtype).split('.')[-1]}_{Sq}_{Skv}.json"), "w") as fh:
        json.dump({"B": B, "heads": h, "Sq": Sq, "Skv": Skv, "fp8_rel_rms_vs_fp32": rel_rms, "fp8_bias_over_mean_abs": bias,
                   "same_inputs_16bit_kernel_rel_rms": rel16}, fh)
    assert torch.isfinite(out).all()
    assert rel_rms < 8e-2 and bias < 1e-2, (rel_rms, bias)


def test_fp8_attention_running_maximum_and_wide_magnitudes(dev):
    """spiked keys late in the sequence move the running maximum by hundreds (the O / l rescale branch); queries and values of
    very different magnitude per row exercise the per-block scales (a per-tensor scale would flush the small rows)."""
    torch.manual_seed(1)
    B, h, S, D = 1, 2, 512, 64
    q = torch.randn(B, S, h * D); k = torch.randn(B, S, h * D); v = torch.randn(B, S, h * D)
    k[:, 300] = q[:, 17] * 4.0
    k[:, 450] = q[:, 99] * 8.0
    k[:, :64] = -q[:, 5:6] * 3.0
    v[:, ::2] *= 1e-3                        # every other value row three orders of magnitude smaller
    v[:, 450] = 100.0
    qb, kb, vb = (t.to(dev).bfloat16() for t in (q, k, v))
    out = hipops.attention_fp8(qb, kb, _vt(vb), h).float().cpu()
    ref = _attn_ref(qb, kb, vb, h)
    assert torch.isfinite(out).all()
    assert ((out - ref).norm() / ref.norm()).item() < 8e-2
    assert (out[0, 99] - 100.0).abs().max() < 7.0          # row 99 is its spiked key's value row (e4m3: 100 -> 96 or 104)


def test_f16_vae_attention_large_logits_stay_finite(dev):
    """the VAE mid-block attention (1 head of 512) stored the UNSCALED q.k in 16 bits; with q, k of magnitude ~16 the
    unscaled product of 512 channels passes half's 65504 (inf -> NaN through the softmax) although the scaled logits are modest.
    The scale now enters inside the product: forward and gradient stay finite and match fp32 math on the same half inputs."""
    from dreammat_amd.sd.models import VaeAttention
    torch.manual_seed(3)
    C, Hh = 512, 16
    att = VaeAttention(C).to(dev, H16).eval()
    with torch.no_grad():
        for lin in (att.to_q, att.to_k):          # projections that keep the magnitude of their (normalised) input, times 16
            lin.weight.copy_(torch.eye(C) * 16.0); lin.bias.zero_()
    for p in att.parameters():
        p.requires_grad_(False)
    x = torch.randn(1, C, Hh, Hh, device=dev).to(H16).requires_grad_()
    y = att(x)
    y.float().sum().backward()
    assert torch.isfinite(y).all() and torch.isfinite(x.grad).all()
    # (unscaled, row maxima of q.k sit near 16 * 16 * 512 = 131072 > 65504)
    att32 = VaeAttention(C).to(dev).eval()
    att32.load_state_dict({k: v.float() for k, v in att.state_dict().items()})
    x32 = x.detach().float().requires_grad_()
    from dreammat_amd.sd import layers
    old = layers.CONV_BACKEND
    try:
        layers.CONV_BACKEND = "gemm"
        y32 = att32(x32)
    finally:
        layers.CONV_BACKEND = old
    assert ((y.float() - y32).abs().max() / y32.abs().max()).item() < 2e-2


def test_fp8_attention_through_the_guidance_switch(dev):
    """layers.set_attention_precision(net, "fp8") (what guidance.attention_precision does to ITS nets) routes the S >= 1024
    self-attention of a Transformer block to the fp8 kernel and nothing else (cross-attention, short sequences); a second
    block in the same process keeps its own setting (the switch used to be a module global)."""
    from dreammat_amd.sd import layers
    torch.manual_seed(0)
    blk = layers.BasicTransformerBlock(320, 5, 1024).to(dev, torch.float16).eval()
    for p in blk.parameters():
        p.requires_grad_(False)
    x = torch.randn(2, 1024, 320, device=dev, dtype=torch.float16)
    ctx = layers.PaddedContext(torch.randn(2, 77, 1024, device=dev, dtype=torch.float16))
    with torch.no_grad():
        y16 = blk(x, ctx)
        import copy
        other = copy.deepcopy(blk)                    # a second net of the process: stays on the 16-bit kernels
        layers.set_attention_precision(blk, "fp8")
        try:
            hipops.enable_kernel_timing(True)
            y8 = blk(x, ctx)
            torch.cuda.synchronize()
            kt = hipops.kernel_times()
            hipops.enable_kernel_timing(False)
            assert torch.equal(other(x, ctx), y16)
        finally:
            layers.set_attention_precision(blk, "16bit")
    assert sum(v["launches"] for k, v in kt.items() if k.startswith("attention_fwd_fp8")) == 1
    assert sum(v["launches"] for k, v in kt.items() if k.startswith("attention_fwd_bf16")) == 1       # (the timing key of the 16-bit kernels)
    assert ((y8.float() - y16.float()).norm() / y16.float().norm()).item() < 5e-2
