"""Stable-Diffusion UNet / ControlNet / VAE-encoder forward passes as pure functions over a diffusers
state_dict (TEST INFRASTRUCTURE; fp32 torch, CPU).

diffusers (requirements.txt:7, unpinned, ~0.16-0.19) is not installable here; the functions restate
the published architecture of UNet2DConditionModel / ControlNetModel / AutoencoderKL that
threestudio/models/guidance/dreammat_guidance.py:110-154, 205-292 runs.  They share NO code with
dreammat_amd/sd (only the checkpoint key names), so agreement between the two is an independent check
of both; PARITY with real diffusers is UNPINNED except for the total parameter counts (865 910 724 /
859 520 964 for the SD-2.1 / SD-1.5 UNets), which the product's modules reproduce exactly.
"""
import math

import torch
import torch.nn.functional as F


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _gn(sd, p, x, eps):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def time_embed(sd, t, ch):
    half = ch // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)     # flip_sin_to_cos=True, shift 0
    return _lin(sd, "time_embedding.linear_2", F.silu(_lin(sd, "time_embedding.linear_1", emb)))


def resnet(sd, p, x, temb, eps=1e-5):
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x, eps)))
    if temb is not None:
        h = h + _lin(sd, p + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h, eps)))
    if (p + ".conv_shortcut.weight") in sd:
        x = _conv(sd, p + ".conv_shortcut", x, padding=0)
    return x + h


def mha(sd, p, x, ctx, heads):
    q = _lin(sd, p + ".to_q", x)
    k = _lin(sd, p + ".to_k", ctx)
    v = _lin(sd, p + ".to_v", ctx)
    B, S, Cc = q.shape
    d = Cc // heads
    split = lambda z: z.reshape(B, -1, heads, d).permute(0, 2, 1, 3)
    w = torch.softmax(split(q) @ split(k).transpose(-1, -2) / math.sqrt(d), dim=-1)
    o = (w @ split(v)).permute(0, 2, 1, 3).reshape(B, S, Cc)
    return _lin(sd, p + ".to_out.0", o)


def transformer(sd, p, x, ctx, heads, linear):
    B, Cc, H, W = x.shape
    h = _gn(sd, p + ".norm", x, 1e-6)
    if linear:
        h = _lin(sd, p + ".proj_in", h.permute(0, 2, 3, 1).reshape(B, H * W, Cc))
    else:
        h = _conv(sd, p + ".proj_in", h, padding=0).permute(0, 2, 3, 1).reshape(B, H * W, Cc)
    b = p + ".transformer_blocks.0"
    h = h + mha(sd, b + ".attn1", _ln(sd, b + ".norm1", h), _ln(sd, b + ".norm1", h), heads)
    h = h + mha(sd, b + ".attn2", _ln(sd, b + ".norm2", h), ctx, heads)
    n3 = _ln(sd, b + ".norm3", h)
    a, g = _lin(sd, b + ".ff.net.0.proj", n3).chunk(2, dim=-1)
    h = h + _lin(sd, b + ".ff.net.2", a * F.gelu(g))
    if linear:
        h = _lin(sd, p + ".proj_out", h).reshape(B, H, W, Cc).permute(0, 3, 1, 2)
    else:
        h = _conv(sd, p + ".proj_out", h.reshape(B, H, W, Cc).permute(0, 3, 1, 2), padding=0)
    return h + x


def _encoder(sd, x, temb, ctx, heads, linear, n_blocks=4):
    skips = [x]
    for i in range(n_blocks):
        for j in range(2):
            x = resnet(sd, f"down_blocks.{i}.resnets.{j}", x, temb)
            if f"down_blocks.{i}.attentions.{j}.norm.weight" in sd:
                x = transformer(sd, f"down_blocks.{i}.attentions.{j}", x, ctx, heads[i], linear)
            skips.append(x)
        if f"down_blocks.{i}.downsamplers.0.conv.weight" in sd:
            x = _conv(sd, f"down_blocks.{i}.downsamplers.0.conv", x, stride=2)
            skips.append(x)
    x = resnet(sd, "mid_block.resnets.0", x, temb)
    x = transformer(sd, "mid_block.attentions.0", x, ctx, heads[-1], linear)
    x = resnet(sd, "mid_block.resnets.1", x, temb)
    return x, skips


def controlnet_forward(sd, sample, t, ctx, cond, scale, heads, linear):
    temb = time_embed(sd, t, sd["conv_in.weight"].shape[0])
    x = _conv(sd, "conv_in", sample)
    c = F.silu(_conv(sd, "controlnet_cond_embedding.conv_in", cond))
    i = 0
    while f"controlnet_cond_embedding.blocks.{i}.weight" in sd:
        c = F.silu(_conv(sd, f"controlnet_cond_embedding.blocks.{i}", c, stride=2 if i % 2 else 1))
        i += 1
    x = x + _conv(sd, "controlnet_cond_embedding.conv_out", c)
    mid, skips = _encoder(sd, x, temb, ctx, heads, linear)
    down = [_conv(sd, f"controlnet_down_blocks.{k}", s, padding=0) * scale for k, s in enumerate(skips)]
    return down, _conv(sd, "controlnet_mid_block", mid, padding=0) * scale


def unet_forward(sd, sample, t, ctx, heads, linear, down_res=None, mid_res=None):
    temb = time_embed(sd, t, sd["conv_in.weight"].shape[0])
    x = _conv(sd, "conv_in", sample)
    x, skips = _encoder(sd, x, temb, ctx, heads, linear)
    if down_res is not None:
        skips = [s + r for s, r in zip(skips, down_res)]
    if mid_res is not None:
        x = x + mid_res
    rheads = list(reversed(heads))
    for i in range(4):
        for j in range(3):
            x = resnet(sd, f"up_blocks.{i}.resnets.{j}", torch.cat([x, skips.pop()], dim=1), temb)
            if f"up_blocks.{i}.attentions.{j}.norm.weight" in sd:
                x = transformer(sd, f"up_blocks.{i}.attentions.{j}", x, ctx, rheads[i], linear)
        if f"up_blocks.{i}.upsamplers.0.conv.weight" in sd:
            x = _conv(sd, f"up_blocks.{i}.upsamplers.0.conv", F.interpolate(x, scale_factor=2.0, mode="nearest"))
    return _conv(sd, "conv_out", F.silu(_gn(sd, "conv_norm_out", x, 1e-5)))


def vae_encode_moments(sd, x):
    h = _conv(sd, "encoder.conv_in", x)
    i = 0
    while f"encoder.down_blocks.{i}.resnets.0.conv1.weight" in sd:
        for j in range(2):
            h = resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, None, eps=1e-6)
        if f"encoder.down_blocks.{i}.downsamplers.0.conv.weight" in sd:
            h = _conv(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", F.pad(h, (0, 1, 0, 1)), stride=2, padding=0)
        i += 1
    h = resnet(sd, "encoder.mid_block.resnets.0", h, None, eps=1e-6)
    p = "encoder.mid_block.attentions.0"
    B, Cc, H, W = h.shape
    n = _gn(sd, p + ".group_norm", h, 1e-6).reshape(B, Cc, H * W).transpose(1, 2)
    q, k, v = _lin(sd, p + ".to_q", n), _lin(sd, p + ".to_k", n), _lin(sd, p + ".to_v", n)
    a = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(Cc), dim=-1) @ v
    h = h + _lin(sd, p + ".to_out.0", a).transpose(1, 2).reshape(B, Cc, H, W)
    h = resnet(sd, "encoder.mid_block.resnets.1", h, None, eps=1e-6)
    h = _conv(sd, "encoder.conv_out", F.silu(_gn(sd, "encoder.conv_norm_out", h, 1e-6)))
    mean, logvar = _conv(sd, "quant_conv", h, padding=0).chunk(2, dim=1)
    return mean, logvar.clamp(-30.0, 20.0)


# --------------------------------------------------------------------------- SDS composition
def alphas_cumprod(n=1000, b0=0.00085, b1=0.012):
    betas = torch.linspace(b0 ** 0.5, b1 ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def sds_loss(rgb, nets, text_emb3, cond_map, t, noise, posterior_noise, scales, heads, linear, cond_scale_cn=1.0,
             vae_scaling=0.18215):
    """dreammat_guidance.py:536-602 with every random draw supplied.
    rgb [B,H,W,3] (requires grad), text_emb3 [3B,77,D] (text, negative, null), cond_map [B,H,W,22],
    scales = (cond, uncond, null, noise).  Returns (loss_sds, grad, eps_pred[3B,...])."""
    B = rgb.shape[0]
    x = rgb.permute(0, 3, 1, 2) * 2.0 - 1.0
    mean, logvar = vae_encode_moments(nets["vae"], x)
    latents = (mean + torch.exp(0.5 * logvar) * posterior_noise) * vae_scaling
    ac = alphas_cumprod()
    a = ac[t].view(-1, 1, 1, 1)
    noisy = a.sqrt() * latents + (1 - a).sqrt() * noise
    with torch.no_grad():
        lat3 = torch.cat([noisy] * 3)
        t3 = torch.cat([t] * 3)
        down = mid = None
        if "controlnet" in nets:
            cond3 = torch.cat([cond_map.permute(0, 3, 1, 2)] * 3)
            down, mid = controlnet_forward(nets["controlnet"], lat3, t3, text_emb3, cond3, cond_scale_cn, heads, linear)
        eps = unet_forward(nets["unet"], lat3, t3, text_emb3, heads, linear, down, mid)
    e_text, e_uncond, e_null = eps.chunk(3)
    w = (1 - ac[t]).view(-1, 1, 1, 1)
    grad = w * (scales[0] * e_text + scales[1] * e_uncond + scales[2] * e_null + scales[3] * noise)
    grad = torch.nan_to_num(grad)
    target = (latents - grad).detach()
    loss = 0.5 * F.mse_loss(latents, target, reduction="sum") / B
    return loss, grad, eps, latents
