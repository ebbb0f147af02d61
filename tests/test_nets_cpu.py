"""CPU suite: the product's SD modules (fp32, plain-math attention path) against the oracle's functional
restatement on identical seeded state_dicts; guidance arithmetic against oracle.sd_nets.sds_loss."""
import os

import numpy as np
import pytest
import torch

from dreammat_amd.sd import ARCHS, AutoencoderKLEncoder, ControlNetModel, UNet2DConditionModel
from oracle import sd_nets as osd


def _nets(arch_name, seed=0):
    a = ARCHS[arch_name]
    torch.manual_seed(seed)
    unet = UNet2DConditionModel(a).eval()
    cn = ControlNetModel.from_unet(unet).eval()
    for conv in list(cn.controlnet_down_blocks) + [cn.controlnet_mid_block, cn.controlnet_cond_embedding.conv_out]:
        torch.nn.init.normal_(conv.weight, std=0.05)
    vae = AutoencoderKLEncoder(a).eval()
    return a, unet, cn, vae


@pytest.mark.parametrize("arch_name", ["tiny", "tiny15"])
def test_unet_controlnet_vae_match_functional_oracle(arch_name):
    a, unet, cn, vae = _nets(arch_name)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 4, 16, 16, generator=g)
    t = torch.tensor([37, 801])
    ctx = torch.randn(2, 77, a.cross_dim, generator=g)
    cond = torch.rand(2, 22, 128, 128, generator=g)
    with torch.no_grad():
        d, m = cn(x, t, ctx, cond, 0.8)
        y = unet(x, t, ctx, d, m)
        od, om = osd.controlnet_forward(cn.state_dict(), x, t, ctx, cond, 0.8, a.heads, a.use_linear_projection)
        oy = osd.unet_forward(unet.state_dict(), x, t, ctx, a.heads, a.use_linear_projection, od, om)
        assert len(d) == 12
        for p, q in zip(d + [m], od + [om]):
            assert (p - q).abs().max() <= 1e-4 * max(1.0, q.abs().max())
        assert (y - oy).abs().max() <= 1e-3 * oy.abs().max()          # north_star: noise-pred within 1e-3 rel
        img = torch.rand(2, 3, 128, 128, generator=g) * 2 - 1
        mean, logvar = vae.encode_moments(img)
        omean, ologvar = osd.vae_encode_moments(vae.state_dict(), img)
        assert (mean - omean).abs().max() < 1e-4 and (logvar - ologvar).abs().max() < 1e-4


def test_full_size_parameter_counts_match_diffusers():
    """The only externally known constants: parameter counts of the published checkpoints."""
    with torch.device("meta"):
        assert sum(p.numel() for p in UNet2DConditionModel(ARCHS["sd21-base"]).parameters()) == 865910724
        assert sum(p.numel() for p in UNet2DConditionModel(ARCHS["sd15"]).parameters()) == 859520964
        assert sum(p.numel() for p in AutoencoderKLEncoder(ARCHS["sd21-base"]).parameters()) == 34163664


def test_ddim_alphas_and_sds_identity():
    from dreammat_amd.sd import DDIMScheduler
    s = DDIMScheduler()
    assert torch.allclose(s.alphas_cumprod, osd.alphas_cumprod())
    assert abs(float(s.alphas_cumprod[0]) - 0.99915) < 1e-5 and abs(float(s.alphas_cumprod[-1]) - 0.0047) < 1e-4
    # eps_text = eps_uncond = eps, scales (1,-1,0,0) -> grad = 0
    e = torch.randn(2, 4, 8, 8)
    w = (1 - s.alphas_cumprod[torch.tensor([10, 500])]).view(-1, 1, 1, 1)
    assert (w * (1.0 * e + -1.0 * e + 0 * e + 0 * e)).abs().max() == 0


def test_guidance_call_matches_oracle_sds(tmp_path, monkeypatch):
    """StableDiffusionLightGuidance.__call__ (plugin API, CPU fp32) vs oracle.sd_nets.sds_loss with the
    same weights and injected randomness: loss, SDS grad, eps-pred and d loss / d rgb."""
    monkeypatch.chdir(tmp_path)
    import dreammat_amd
    from dreammat_amd.guidance import StableDiffusionLightGuidance
    from dreammat_amd.prompt import StableDiffusionPromptProcessor
    gd = StableDiffusionLightGuidance({"pretrained_model_name_or_path": "tiny", "use_controlnet": True,
                                       "control_types": ["light"], "condition_scales": [0.9], "width": 128,
                                       "height": 128, "cond_scale": 1.05, "uncond_scale": [0, -1.0, -0.5, 2000],
                                       "null_scale": [0, 0.0, -0.5, 2000], "half_precision_weights": True})
    assert gd.weights_dtype == torch.float32            # no GPU here: fp32 plumbing path
    gd.update_step(0, 1000)
    assert abs(gd.uncond_scale - (-0.75)) < 1e-9 and abs(gd.null_scale - (-0.25)) < 1e-9
    pp = StableDiffusionPromptProcessor({"prompt": "a wooden chair", "negative_prompt": "ugly",
                                         "pretrained_model_name_or_path": "tiny"})
    B = 2
    g = torch.Generator().manual_seed(3)
    rgb = torch.rand(B, 128, 128, 3, generator=g).requires_grad_()
    cond = torch.rand(B, 128, 128, 22, generator=g)
    elev, azim, dist = torch.tensor([10.0, 70.0]), torch.tensor([5.0, 170.0]), torch.tensor([3.5, 3.2])
    rng = {"t": torch.tensor([400, 77]), "noise": torch.randn(B, 4, 16, 16, generator=g),
           "posterior_noise": torch.randn(B, 4, 16, 16, generator=g)}
    out = gd(rgb, pp(), elev, azim, dist, env_id=torch.tensor([0, 1]), condition_map=cond, rng=rng)
    out["loss_sds"].backward()
    emb = pp().get_text_embeddings(elev, azim, dist, True, True)
    # view-dependent selection: elev 70 > 60 -> overhead; az 5 -> front
    assert torch.equal(emb[0], pp.text_embeddings_vd[1]) and torch.equal(emb[1], pp.text_embeddings_vd[3])
    rgb2 = rgb.detach().clone().requires_grad_()
    nets = {"vae": gd.vae.state_dict(), "unet": gd.unet.state_dict(), "controlnet": gd.controlnets[0].state_dict()}
    a = gd.arch
    loss, grad, eps, lat = osd.sds_loss(rgb2, nets, emb, cond, rng["t"], rng["noise"], rng["posterior_noise"],
                                        (1.05, -0.75, -0.25, 0.0), a.heads, a.use_linear_projection, 0.9)
    loss.backward()
    assert abs(float(out["loss_sds"]) - float(loss)) <= 1e-4 * abs(float(loss))
    assert (gd._last["grad"] - grad).abs().max() <= 1e-3 * grad.abs().max()
    e3 = torch.cat([gd._last["e_text"], gd._last["e_uncond"], gd._last["e_null"]])
    assert (e3 - eps).abs().max() <= 1e-3 * eps.abs().max()
    assert (rgb.grad - rgb2.grad).abs().max() <= 1e-3 * rgb2.grad.abs().max()


def test_controlnet_training_step_and_cfg_dropout():
    """the ControlNet training loss / step of controlnet_train/diffusers_train_controlnet.py:858-915 on
    the tiny architecture -- zero-initialised ControlNet == plain UNet prediction, only ControlNet parameters receive
    gradients (frozen UNet / VAE), the loss goes down on a fixed batch, and the dataset's CFG dropout rates."""
    from dreammat_amd import controlnet_train as ct
    from dreammat_amd.sd import ARCHS, AutoencoderKLEncoder, UNet2DConditionModel
    from dreammat_amd.sd.layers import PaddedContext
    torch.manual_seed(0)
    a = ARCHS["tiny"]
    unet, vae = UNet2DConditionModel(a), AutoencoderKLEncoder(a)
    tr = ct.ControlNetTrainer(vae, unet, lr=2e-3)
    assert all(not p.requires_grad for p in list(unet.parameters()) + list(vae.parameters()))
    assert all(p.requires_grad for p in tr.controlnet.parameters())
    g = torch.Generator().manual_seed(1)
    B = 2
    img = torch.rand(B, 3, 64, 64, generator=g) * 2 - 1
    cond = torch.rand(B, 22, 64, 64, generator=g)
    text = torch.randn(B, 77, a.cross_dim, generator=g)
    t = torch.tensor([300, 700])
    noise = torch.randn(B, 4, 8, 8, generator=g)
    pn = torch.randn(B, 4, 8, 8, generator=g)
    fixed = dict(timesteps=t, noise=noise, posterior_noise=pn)
    # zero convs: the ControlNet contributes nothing at initialisation
    loss0 = ct.controlnet_training_loss(vae, unet, tr.controlnet, tr.scheduler, img, cond, text, **fixed)
    with torch.no_grad():
        lat = vae.sample(img, pn) * vae.scaling_factor
        noisy = tr.scheduler.add_noise(lat, noise, t)
        plain = unet(noisy, t, PaddedContext(text))
    assert abs(float(loss0) - float(torch.nn.functional.mse_loss(plain, noise))) < 1e-6
    loss0.backward()
    assert all(p.grad is None for p in unet.parameters())
    got = [n for n, p in tr.controlnet.named_parameters() if p.grad is not None and p.grad.abs().sum() > 0]
    assert any(n.startswith("controlnet_down_blocks") for n in got) and any(n.startswith("controlnet_mid_block") for n in got)
    tr.controlnet.zero_grad(set_to_none=True)
    losses = [float(tr.step(img, cond, text, **fixed)) for _ in range(12)]
    assert losses[-1] < losses[0] and tr.global_step == 12
    assert set(tr.state_dict()) == set(tr.controlnet.state_dict())
    # CFG dropout of the dataset: 5 % all conditions, 5 % depth, 5 % normal, 5 % light maps, 30 % empty prompt
    n = 40000
    te, ne = torch.ones(n, 1, 1), torch.zeros(1, 1, 1)
    cd = torch.ones(n, 22, 1, 1)
    t2, c2 = ct.cfg_dropout(te, ne, cd, torch.Generator().manual_seed(3))
    z = (c2[:, :, 0, 0] == 0)
    all0 = z.all(1)
    depth0 = z[:, 0] & ~z[:, 1] & ~z[:, 4]
    normal0 = ~z[:, 0] & z[:, 1:4].all(1) & ~z[:, 4]
    light0 = ~z[:, 0] & ~z[:, 1] & z[:, 4:].all(1)
    for m in (all0, depth0, normal0, light0):
        assert abs(float(m.float().mean()) - 0.05) < 0.006
    assert int((z.any(1) & ~(all0 | depth0 | normal0 | light0)).sum()) == 0
    dt = (t2.view(-1) == 0)
    assert abs(float(dt.float().mean()) - 0.30) < 0.01 and int((dt & z.any(1)).sum()) == 0   # never both


def _tiny_render_tree(root, prompts_path, S=64, n_views=16):
    """a complete object directory in the reference's render-tree layout (random images)"""
    import json

    import numpy as np
    from PIL import Image
    rs = np.random.RandomState(0)
    for sub in ("color", "depth", "normal", "light"):
        (root / "obj_a" / sub).mkdir(parents=True)
    for view in range(n_views):
        d = np.zeros((S, S), np.uint16); d[8:56, 8:56] = rs.randint(1500, 2500, (48, 48))
        Image.fromarray(d).save(root / "obj_a" / "depth" / f"{view:03d}.png")
        n = rs.randint(0, 255, (S, S, 4)).astype(np.uint8); n[..., 3] = 255
        Image.fromarray(n, "RGBA").save(root / "obj_a" / "normal" / f"{view:03d}.png")
        for env in range(1, 6):
            Image.fromarray(rs.randint(0, 255, (S, S, 3)).astype(np.uint8), "RGB").save(root / "obj_a" / "color" / f"{view:03d}_color_env{env}.png")
            for m in ("0.0", "1.0"):
                for r in ("0.0", "0.5", "1.0"):
                    Image.fromarray(rs.randint(0, 255, (S, S, 3)).astype(np.uint8), "RGB").save(
                        root / "obj_a" / "light" / f"{view:03d}_m{m}r{r}_env{env}.png")
    prompts_path.write_text(json.dumps({"obj_a": "a red teapot"}))


def test_controlnet_launcher_main_runs_two_steps(tmp_path):
    """nothing called controlnet_train.main(); its first step crashed on a GPU (bf16 latents against an
    fp32 ControlNet).  Here on the CPU (fp32 throughout), tests/test_hip_gpu.py runs the same launcher in bf16 on the GPU."""
    from dreammat_amd import controlnet_train as ct
    root = tmp_path / "data"
    _tiny_render_tree(root, tmp_path / "prompts.json")
    tr = ct.main(["--synthetic", "--pretrained_model_name_or_path", "tiny", "--train_data_dir", str(root),
                  "--prompt_file", str(tmp_path / "prompts.json"), "--resolution", "64", "--train_batch_size", "2",
                  "--max_train_steps", "2", "--output_dir", str(tmp_path / "out"), "--checkpointing_steps", "2"])
    assert tr.global_step == 2 and tr.master is None
    assert (tmp_path / "out" / "controlnet_step2.pt").exists()


def test_controlnet_trainer_keeps_fp32_master_weights_under_bf16_compute():
    """bf16 ControlNet parameters + AdamW at lr 1e-5 lose most updates to rounding: the trainer keeps fp32 masters (created
    when gradients are first applied, also when the module was cast AFTER construction), steps them, casts back, and its
    state dict carries the fp32 values."""
    from dreammat_amd import controlnet_train as ct
    from dreammat_amd.sd import ARCHS, AutoencoderKLEncoder, UNet2DConditionModel
    torch.manual_seed(0)
    a = ARCHS["tiny"]
    unet, vae = UNet2DConditionModel(a).bfloat16(), AutoencoderKLEncoder(a).bfloat16()
    tr = ct.ControlNetTrainer(vae, unet, lr=1e-5)
    assert tr.master is None                                    # built in fp32 ...
    tr.controlnet.bfloat16()                                    # ... and cast by hand afterwards, as the round-3 tests did
    g = torch.Generator().manual_seed(1)
    img = (torch.rand(2, 3, 64, 64, generator=g) * 2 - 1).bfloat16()
    cond = torch.rand(2, 22, 64, 64, generator=g)
    text = torch.randn(2, 77, a.cross_dim, generator=g)
    fixed = dict(timesteps=torch.tensor([300, 700]), noise=torch.randn(2, 4, 8, 8, generator=g),
                 posterior_noise=torch.randn(2, 4, 8, 8, generator=g))
    w0 = [p.detach().float().clone() for p in tr.controlnet.parameters()]
    for _ in range(3):
        tr.step(img, cond, text, **fixed)
    assert tr.master is not None and all(m.dtype == torch.float32 for m in tr.master)
    params = [p for p in tr.controlnet.parameters() if p.requires_grad]
    assert all(p.dtype == torch.bfloat16 and torch.equal(p, m.bfloat16()) for p, m in zip(params, tr.master))
    moved = sum(float((m - w).abs().sum()) for m, w in zip(tr.master, w0))
    assert moved > 0                                            # updates of ~1e-5 survive in the masters ...
    sd = tr.state_dict()
    assert all(sd[k].dtype == torch.float32 for k, p in tr.controlnet.named_parameters() if p.requires_grad)


def test_controlnet_render_dataset_reads_the_reference_tree(tmp_path):
    """f-4 remainder: the dataset of controlnet_train/diffusers_dataset.py:84-159 on a synthetic tree in the
    reference's layout -- index -> (object, environment, view) mapping, 22 channels in the reference's order, depth decoding
    (16-bit mm -> inverse min-max to [0.3, 1], background 0), transparent pixels (condition -> 0, target -> white),
    target without alpha whitened by the depth mask, and the pinned CFG dropout."""
    import json
    import random

    import numpy as np
    from PIL import Image

    from dreammat_amd import controlnet_train as ct
    S = 8
    root = tmp_path / "data"
    for sub in ("color", "depth", "normal", "light"):
        (root / "obj_a" / sub).mkdir(parents=True)
    (root / "not_listed").mkdir()
    rs = np.random.RandomState(0)
    depth_mm = np.zeros((S, S), np.uint16)
    depth_mm[2:6, 2:6] = rs.randint(1500, 2500, (4, 4))
    view, env = 3, 2
    Image.fromarray(depth_mm).save(root / "obj_a" / "depth" / f"{view:03d}.png")
    nrm = rs.randint(0, 255, (S, S, 4)).astype(np.uint8); nrm[..., 3] = 255; nrm[0, 0, 3] = 0
    Image.fromarray(nrm, "RGBA").save(root / "obj_a" / "normal" / f"{view:03d}.png")
    col = rs.randint(0, 255, (S, S, 3)).astype(np.uint8)
    Image.fromarray(col, "RGB").save(root / "obj_a" / "color" / f"{view:03d}_color_env{env}.png")
    lights = {}
    for m in ("0.0", "1.0"):
        for r in ("0.0", "0.5", "1.0"):
            a = rs.randint(0, 255, (S, S, 3)).astype(np.uint8)
            lights[(m, r)] = a
            Image.fromarray(a, "RGB").save(root / "obj_a" / "light" / f"{view:03d}_m{m}r{r}_env{env}.png")
    (tmp_path / "prompts.json").write_text(json.dumps({"obj_a": "a red teapot", "missing_dir": "x"}))
    ds = ct.ControlNetRenderDataset(str(root), str(tmp_path / "prompts.json"), size=S)
    assert len(ds) == 5 * 16                                        # one listed object with a directory
    item = ds[(env - 1) * 16 + view]
    assert item["input_ids"] == "a red teapot"
    src, tgt = item["conditioning_pixel_values"].numpy(), item["pixel_values"].numpy()
    assert src.shape == (S, S, 22) and tgt.shape == (S, S, 3)
    mask = depth_mm > 0
    d = depth_mm.astype(np.float64) / 1000
    inv = 1 / (d + 1e-6)
    ref_d = np.where(mask, 0.7 * (inv - inv[mask].min()) / (inv[mask].max() - inv[mask].min() + 1e-6) + 0.3, 0)
    assert np.abs(src[..., 0] - ref_d).max() < 1e-6 and src[..., 0][mask].min() >= 0.3 - 1e-6
    ref_n = nrm[..., :3].astype(np.float32) / 255; ref_n[0, 0] = 0            # transparent -> 0
    assert np.abs(src[..., 1:4] - ref_n).max() < 1e-6
    k = 4
    for m in ("0.0", "1.0"):
        for r in ("0.0", "0.5", "1.0"):
            assert np.abs(src[..., k:k + 3] - lights[(m, r)].astype(np.float32) / 255).max() < 1e-6, (m, r)
            k += 3
    ref_t = col.astype(np.float32) / 127.5 - 1; ref_t[~mask] = 1.0          # no alpha: whitened outside the depth mask
    assert np.abs(tgt - ref_t).max() < 1e-6
    b = ct.collate([item, item])
    assert b["pixel_values"].shape == (2, 3, S, S) and b["conditioning_pixel_values"].shape == (2, 22, S, S)
    # pinned CFG dropout: the same draw as the reference's `random.random()` thresholds
    for seed, check in ((1, None), (5, None), (8, None)):
        r = random.Random(seed).random()
        it = ct.ControlNetRenderDataset(str(root), str(tmp_path / "prompts.json"), S, True, random.Random(seed))[(env - 1) * 16 + view]
        s2 = it["conditioning_pixel_values"].numpy()
        if r < 0.05:
            assert not s2.any()
        elif 0.05 < r < 0.1:
            assert not s2[..., 0].any() and np.array_equal(s2[..., 1:], src[..., 1:])
        elif 0.1 < r < 0.15:
            assert not s2[..., 1:4].any()
        elif 0.15 < r < 0.2:
            assert not s2[..., 4:].any()
        elif 0.2 < r < 0.5:
            assert it["input_ids"] == "" and np.array_equal(s2, src)
        else:
            assert it["input_ids"] == "a red teapot" and np.array_equal(s2, src)


def test_guidance_perp_neg_branch_matches_oracle_composition(tmp_path, monkeypatch):
    """Non-default `use_perp_neg` (dreammat_guidance.py:319-386, 440-486) at the reference's B = 1: five branch items through
    ControlNet + UNet, the two view-prompt predictions enter perpendicular to (text - uncond) with the prompt processor's
    negative weights, scaled by perpneg_scale.  Oracle: the functional nets + the reference's formulas (golden-pinned pieces:
    tests/golden/perpneg.npz)."""
    monkeypatch.chdir(tmp_path)
    from dreammat_amd.guidance import StableDiffusionLightGuidance
    from dreammat_amd.prompt import StableDiffusionPromptProcessor, perpendicular_component
    gd = StableDiffusionLightGuidance({"pretrained_model_name_or_path": "tiny", "use_controlnet": True,
                                       "control_types": ["light"], "condition_scales": [1.0], "width": 128,
                                       "height": 128, "cond_scale": 1.05, "uncond_scale": -0.75, "null_scale": -0.25,
                                       "perpneg_scale": 0.6, "half_precision_weights": True})
    gd.update_step(0, 0)
    assert abs(gd.perpneg_scale - 0.6) < 1e-9
    pp = StableDiffusionPromptProcessor({"prompt": "a wooden chair", "negative_prompt": "ugly", "use_perp_neg": True,
                                         "pretrained_model_name_or_path": "tiny"})
    pu = pp()
    assert pu.use_perp_neg
    g = torch.Generator().manual_seed(5)
    rgb = torch.rand(1, 128, 128, 3, generator=g)
    cond = torch.rand(1, 128, 128, 22, generator=g)
    elev, azim, dist = torch.tensor([10.0]), torch.tensor([60.0]), torch.tensor([3.5])
    rng = {"t": torch.tensor([400]), "noise": torch.randn(1, 4, 16, 16, generator=g),
           "posterior_noise": torch.randn(1, 4, 16, 16, generator=g)}
    out = gd(rgb, pu, elev, azim, dist, env_id=torch.tensor([0]), condition_map=cond, rng=rng)
    emb, w = pu.get_text_embeddings_perp_neg(elev, azim, dist, True, True)          # [5, 77, D], [1, 2]
    assert emb.shape[0] == 5 and float(w.abs().sum()) > 0
    nets = {"vae": gd.vae.state_dict(), "unet": gd.unet.state_dict(), "controlnet": gd.controlnets[0].state_dict()}
    a = gd.arch
    with torch.no_grad():
        mean, logvar = osd.vae_encode_moments(nets["vae"], rgb.permute(0, 3, 1, 2) * 2 - 1)
        lat = (mean + torch.exp(0.5 * logvar) * rng["posterior_noise"]) * 0.18215
        ac = osd.alphas_cumprod()[rng["t"]].view(-1, 1, 1, 1)
        noisy = ac.sqrt() * lat + (1 - ac).sqrt() * rng["noise"]
        lat5, t5 = torch.cat([noisy] * 5), torch.cat([rng["t"]] * 5)
        cond5 = torch.cat([cond.permute(0, 3, 1, 2)] * 5)
        d, m = osd.controlnet_forward(nets["controlnet"], lat5, t5, emb, cond5, 1.0, a.heads, a.use_linear_projection)
        eps = osd.unet_forward(nets["unet"], lat5, t5, emb, a.heads, a.use_linear_projection, d, m)
    e_text, e_unc, e_n0, e_n1, e_null = eps[0:1], eps[1:2], eps[2:3], eps[3:4], eps[4:5]
    e_pos = e_text - e_unc
    perp = w[:, 0].view(-1, 1, 1, 1) * perpendicular_component(e_n0 - e_unc, e_pos) + \
        w[:, 1].view(-1, 1, 1, 1) * perpendicular_component(e_n1 - e_unc, e_pos)
    wt = (1 - osd.alphas_cumprod()[rng["t"]]).view(-1, 1, 1, 1)
    grad = wt * (1.05 * e_text - 0.75 * e_unc - 0.25 * e_null) + wt * 0.6 * perp
    assert (gd._last["grad"] - grad).abs().max() <= 1e-3 * grad.abs().max()
    loss = 0.5 * ((lat - (lat - grad)) ** 2).sum() / 1
    assert abs(float(out["loss_sds"]) - float(loss)) <= 1e-3 * abs(float(loss))


def test_guidance_multi_controlnet_light_depth_normal(tmp_path, monkeypatch):
    """Non-default control_types (dreammat_guidance.py:99-119, 205-241, 518-534): one ControlNet per type -- the 22-channel
    light-geo net plus 3-channel depth / normal nets -- whose residuals are summed with their own conditioning scales before
    the UNet.  Against the functional oracle with the same (random) weights; the renderer's depth map is repeated to three
    channels, an unknown type is an error."""
    monkeypatch.chdir(tmp_path)
    from dreammat_amd.guidance import StableDiffusionLightGuidance
    from dreammat_amd.prompt import StableDiffusionPromptProcessor
    cfg = {"pretrained_model_name_or_path": "tiny", "use_controlnet": True, "control_types": ["light", "depth", "normal"],
           "condition_scales": [1.0, 0.5, 0.7], "width": 128, "height": 128, "cond_scale": 1.05, "uncond_scale": -0.75,
           "null_scale": -0.25, "half_precision_weights": True}
    gd = StableDiffusionLightGuidance(dict(cfg))
    assert [cn.cond_channels for cn in gd.controlnets] == [22, 3, 3]
    with pytest.raises(ValueError):
        StableDiffusionLightGuidance(dict(cfg, control_types=["canny"], condition_scales=[1.0]))
    gd.update_step(0, 0)
    pp = StableDiffusionPromptProcessor({"prompt": "a wooden chair", "negative_prompt": "ugly", "pretrained_model_name_or_path": "tiny"})
    B = 1
    g = torch.Generator().manual_seed(9)
    rgb = torch.rand(B, 128, 128, 3, generator=g)
    light = torch.rand(B, 128, 128, 22, generator=g)
    depth = torch.rand(B, 128, 128, 1, generator=g)
    normal = torch.rand(B, 128, 128, 3, generator=g)
    elev, azim, dist = torch.tensor([10.0]), torch.tensor([5.0]), torch.tensor([3.5])
    rng = {"t": torch.tensor([300]), "noise": torch.randn(B, 4, 16, 16, generator=g),
           "posterior_noise": torch.randn(B, 4, 16, 16, generator=g)}
    gd(rgb, pp(), elev, azim, dist, env_id=torch.tensor([0]), condition_map=light, cond_depth=depth, cond_normal=normal, rng=rng)
    emb = pp().get_text_embeddings(elev, azim, dist, True, True)
    a = gd.arch
    with torch.no_grad():
        mean, logvar = osd.vae_encode_moments(gd.vae.state_dict(), rgb.permute(0, 3, 1, 2) * 2 - 1)
        lat = (mean + torch.exp(0.5 * logvar) * rng["posterior_noise"]) * 0.18215
        ac = osd.alphas_cumprod()[rng["t"]].view(-1, 1, 1, 1)
        noisy = ac.sqrt() * lat + (1 - ac).sqrt() * rng["noise"]
        lat3, t3 = torch.cat([noisy] * 3), torch.cat([rng["t"]] * 3)
        down = mid = None
        conds = [light.permute(0, 3, 1, 2), depth.permute(0, 3, 1, 2).repeat(1, 3, 1, 1), normal.permute(0, 3, 1, 2)]
        for cn, c, sc in zip(gd.controlnets, conds, (1.0, 0.5, 0.7)):
            d, m = osd.controlnet_forward(cn.state_dict(), lat3, t3, emb, torch.cat([c] * 3), sc, a.heads, a.use_linear_projection)
            down, mid = (d, m) if down is None else ([x + y for x, y in zip(down, d)], mid + m)
        eps = osd.unet_forward(gd.unet.state_dict(), lat3, t3, emb, a.heads, a.use_linear_projection, down, mid)
    e3 = torch.cat([gd._last["e_text"], gd._last["e_uncond"], gd._last["e_null"]])
    assert (e3 - eps).abs().max() <= 1e-3 * eps.abs().max()
    # the three nets must actually differ in what they contribute
    assert float((gd.controlnets[1].controlnet_cond_embedding.conv_in.weight).abs().sum()) > 0


def _conv2x2_ref(x_nhwc, w4, bias, out_hw, pad):
    """what dm_conv2x2_nhwc_bf16 computes, in torch: y[b,yo,xo,n] = sum_{dy,dx} x[b, yo - pad + dy, xo - pad + dx, :] . w4[n][2 dy + dx]."""
    B, H, W, Cin = x_nhwc.shape
    Ho, Wo = out_hw
    N = w4.shape[0]
    wv = w4.view(N, 2, 2, Cin)
    xp = torch.nn.functional.pad(x_nhwc, (0, 0, pad, Wo + 1 - W, pad, Ho + 1 - H))
    y = torch.zeros(B, Ho, Wo, N)
    for dy in range(2):
        for dx in range(2):
            y += torch.einsum("bhwc,nc->bhwn", xp[:, dy:dy + Ho, dx:dx + Wo], wv[:, dy, dx])
    return y if bias is None else y + bias


def test_subpixel_dgrad_weights_reproduce_the_strided_convs_data_gradient():
    """hipops.subpixel_dgrad_weights: the data gradient of conv3x3(F.pad(x, (0,1,0,1)), stride 2) as ONE 2 x 2 convolution at the
    gradient's resolution whose 4 Cin output channels are the four parities of dx (host logic of _Conv3x3S2.backward)."""
    from dreammat_amd import hipops
    torch.manual_seed(0)
    B, Cin, Cout, H, W = 2, 6, 10, 8, 12
    x = torch.randn(B, Cin, H, W, requires_grad=True)
    w = torch.randn(Cout, Cin, 3, 3)
    y = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (0, 1, 0, 1)), w, stride=2)
    g = torch.randn_like(y)
    y.backward(g)
    w_fwd = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)                    # tap-major, as Conv2d._prepared builds it
    ws = hipops.subpixel_dgrad_weights(w_fwd, Cin)
    assert tuple(ws.shape) == (4 * Cin, 4 * Cout)
    Ho, Wo = H // 2, W // 2
    yy = _conv2x2_ref(g.permute(0, 2, 3, 1), ws, None, (Ho, Wo), 1)
    dx = yy.view(B, Ho, Wo, 2, 2, Cin).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, Cin).permute(0, 3, 1, 2)
    assert (dx - x.grad).abs().max() < 1e-4


def test_subpixel_upsample_weights_reproduce_upsample_plus_conv():
    """hipops.subpixel_upsample_weights: conv3x3(nearest-2x(x)) as a 2 x 2 convolution at the source resolution on an (h+1) x (w+1)
    grid, parity (py, px) of output (u, v) = channel block 2 py + px at grid (u + py, v + px) (host logic of Upsample2D)."""
    from dreammat_amd import hipops
    torch.manual_seed(1)
    B, Cin, Cout, h, w_ = 2, 5, 7, 6, 9
    x = torch.randn(B, Cin, h, w_)
    w = torch.randn(Cout, Cin, 3, 3)
    b = torch.randn(Cout)
    ref = torch.nn.functional.conv2d(torch.nn.functional.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)
    w4, b4 = hipops.subpixel_upsample_weights(w, b)
    assert tuple(w4.shape) == (4 * Cout, 4 * Cin) and tuple(b4.shape) == (4 * Cout,)
    y = _conv2x2_ref(x.permute(0, 2, 3, 1), w4, b4, (h + 1, w_ + 1), 1)
    out = torch.zeros(B, h, 2, w_, 2, Cout)
    for py in range(2):
        for px in range(2):
            blk = (2 * py + px) * Cout
            out[:, :, py, :, px] = y[:, py:py + h, px:px + w_, blk:blk + Cout]
    out = out.view(B, 2 * h, 2 * w_, Cout).permute(0, 3, 1, 2)
    assert (out - ref).abs().max() < 1e-4


def test_net_prologue_batched_time_embedding_projections_equal_the_per_block_ones(monkeypatch):
    """layers.NetPrologue.project_temb on the CPU (the device check lifted): every ResnetBlock2D consumes the projection the
    prologue left for it, and the net's output equals the per-block path."""
    from dreammat_amd.sd import ARCHS, UNet2DConditionModel, layers
    torch.manual_seed(0)
    a = ARCHS["tiny"]
    unet = UNet2DConditionModel(a).eval().requires_grad_(False)
    x = torch.randn(2, 4, 16, 16); t = torch.tensor([37, 801]); ctx = torch.randn(2, 77, a.cross_dim)
    with torch.no_grad():
        ref = unet(x, t, ctx)
        monkeypatch.setattr(layers.NetPrologue, "usable", staticmethod(lambda t_: True))
        calls = []
        orig = layers.NetPrologue.project_temb
        monkeypatch.setattr(layers.NetPrologue, "project_temb", lambda self, temb: (calls.append(len(self.resnets)), orig(self, temb))[1])
        y = unet(x, t, ctx)
    assert calls and calls[0] == sum(1 for m in unet.modules() if isinstance(m, layers.ResnetBlock2D))
    assert not any("_tproj" in m.__dict__ for m in unet.modules())          # every projection was consumed
    assert (y - ref).abs().max() <= 1e-4 * ref.abs().max() + 1e-6
