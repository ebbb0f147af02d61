"""UNet2DConditionModel, ControlNetModel and the AutoencoderKL encoder in plain torch with
diffusers' parameter names (see layers.py).  Architecture tables:
  sd21-base: block_out (320,640,1280,1280), heads (5,10,20,20) [head_dim 64], cross_dim 1024, linear proj
  sd15     : block_out (320,640,1280,1280), heads 8 [head_dim 40/80/160/160], cross_dim 768, conv proj
Both are what SURVEY D1 asks for: the reference ships SD-2.1-base (configs/dreammat.yaml:60,76), the
task statement names SD-1.5.
"""
from dataclasses import dataclass, field
from typing import Optional, Tuple

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import (Attention, Conv2d, Downsample2D, group_norm_act, NetPrologue, PaddedContext, ResnetBlock2D, TimestepEmbedding,
                     Transformer2DModel, Upsample2D, attention_core, timestep_embedding)


@dataclass
class SDArch:
    name: str = "sd21-base"
    block_out: Tuple[int, ...] = (320, 640, 1280, 1280)
    heads: Tuple[int, ...] = (5, 10, 20, 20)
    cross_dim: int = 1024
    use_linear_projection: bool = True
    layers_per_block: int = 2
    in_channels: int = 4
    out_channels: int = 4
    cond_channels: int = 22                       # light-geo ControlNet: depth1 + normal3 + light18
    cond_embed_channels: Tuple[int, ...] = (16, 32, 96, 256)
    vae_block_out: Tuple[int, ...] = (128, 256, 512, 512)
    vae_scaling_factor: float = 0.18215
    prediction_type: str = "epsilon"


ARCHS = {
    "sd21-base": SDArch(),
    "sd15": SDArch(name="sd15", heads=(8, 8, 8, 8), cross_dim=768, use_linear_projection=False),
    # tiny variant for fast tests (same topology, 32-divisible channels)
    "tiny": SDArch(name="tiny", block_out=(32, 64, 128, 128), heads=(1, 2, 2, 2), cross_dim=64,
                   cond_embed_channels=(8, 8, 16, 32), vae_block_out=(32, 32, 64, 64)),
    "tiny15": SDArch(name="tiny15", block_out=(32, 64, 128, 128), heads=(2, 2, 2, 2), cross_dim=48,
                     use_linear_projection=False, cond_embed_channels=(8, 8, 16, 32), vae_block_out=(32, 32, 64, 64)),
}


def arch_for(name_or_path: str) -> SDArch:
    n = name_or_path.lower()
    if n in ARCHS:
        return ARCHS[n]
    if "stable-diffusion-2" in n or "sd21" in n or "sd2" in n:
        return ARCHS["sd21-base"]
    if "v1-5" in n or "v1-4" in n or "sd15" in n or "stable-diffusion-v1" in n:
        return ARCHS["sd15"]
    raise ValueError(f"unknown Stable-Diffusion architecture for '{name_or_path}'")


class DownBlock(nn.Module):
    def __init__(self, in_ch, out_ch, temb, n_layers, heads, cross_dim, linear, add_down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_ch if i == 0 else out_ch, out_ch, temb) for i in range(n_layers)])
        if heads:
            self.attentions = nn.ModuleList([Transformer2DModel(out_ch, heads, cross_dim, linear) for _ in range(n_layers)])
        else:
            self.attentions = None
        self.downsamplers = nn.ModuleList([Downsample2D(out_ch)]) if add_down else None

    def forward(self, x, temb, ctx):
        outs = []
        for i, r in enumerate(self.resnets):
            x = r(x, temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, ch, temb, heads, cross_dim, linear):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb), ResnetBlock2D(ch, ch, temb)])
        self.attentions = nn.ModuleList([Transformer2DModel(ch, heads, cross_dim, linear)])

    def forward(self, x, temb, ctx):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, ctx)
        return self.resnets[1](x, temb)


def _cat_skip(x, skip):
    """torch.cat([x, skip], dim=1) with skip = s or the pair (s, ControlNet residual r) -> s + r"""
    s, r = skip if isinstance(skip, tuple) else (skip, None)
    fused = (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and not (torch.is_grad_enabled() and (x.requires_grad or s.requires_grad
             or (r is not None and r.requires_grad))) and x.shape[1] % 8 == 0 and s.shape[1] % 8 == 0
             and x.permute(0, 2, 3, 1).is_contiguous() and s.permute(0, 2, 3, 1).is_contiguous()
             and (r is None or (r.dtype == x.dtype and r.permute(0, 2, 3, 1).is_contiguous())))
    if fused:
        from .. import hipops
        return hipops.cat_add_nhwc(x, s, r)
    return torch.cat([x, s if r is None else s + r], dim=1)


class UpBlock(nn.Module):
    def __init__(self, in_ch, prev_ch, out_ch, temb, n_layers, heads, cross_dim, linear, add_up):
        super().__init__()
        res = []
        for i in range(n_layers):
            skip = in_ch if i == n_layers - 1 else out_ch
            rin = prev_ch if i == 0 else out_ch
            res.append(ResnetBlock2D(rin + skip, out_ch, temb))
        self.resnets = nn.ModuleList(res)
        self.attentions = nn.ModuleList([Transformer2DModel(out_ch, heads, cross_dim, linear) for _ in range(n_layers)]) if heads else None
        self.upsamplers = nn.ModuleList([Upsample2D(out_ch)]) if add_up else None

    def forward(self, x, skips, temb, ctx):
        for i, r in enumerate(self.resnets):
            x = r(_cat_skip(x, skips.pop()), temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class _Encoder(nn.Module):
    """conv_in + time embedding + down blocks + mid block shared by UNet and ControlNet."""

    def _build_encoder(self, a: SDArch):
        bo = a.block_out
        temb = bo[0] * 4
        self.arch = a
        self.conv_in = Conv2d(a.in_channels, bo[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(bo[0], temb)
        downs = []
        ch = bo[0]
        for i, oc in enumerate(bo):
            last = i == len(bo) - 1
            downs.append(DownBlock(ch, oc, temb, a.layers_per_block, 0 if last else a.heads[i], a.cross_dim,
                                   a.use_linear_projection, not last))
            ch = oc
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = MidBlock(bo[-1], temb, a.heads[-1], a.cross_dim, a.use_linear_projection)

    def _temb(self, t, dtype):
        return self.time_embedding(timestep_embedding(t, self.arch.block_out[0]).to(dtype))

    def _prologue(self, temb, ctx):
        """the per-block time-embedding projections and K / V^T bank gathers of this forward, batched (layers.NetPrologue)"""
        pro = self.__dict__.get("_net_prologue")
        if pro is None:
            pro = self.__dict__["_net_prologue"] = NetPrologue(self)
        pro.project_temb(temb)
        pro.gather_kv(ctx)


class UNet2DConditionModel(_Encoder):
    def __init__(self, arch: SDArch):
        super().__init__()
        self._build_encoder(arch)
        bo = arch.block_out
        temb = bo[0] * 4
        rev = list(reversed(bo))
        rheads = list(reversed(arch.heads))
        ups = []
        out_ch = rev[0]
        for i in range(len(rev)):
            prev = out_ch
            out_ch = rev[i]
            in_ch = rev[min(i + 1, len(rev) - 1)]
            ups.append(UpBlock(in_ch, prev, out_ch, temb, arch.layers_per_block + 1, 0 if i == 0 else rheads[i],
                               arch.cross_dim, arch.use_linear_projection, i != len(rev) - 1))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(32, bo[0], eps=1e-5)
        self.conv_out = Conv2d(bo[0], arch.out_channels, 3, padding=1)

    def forward(self, sample, timestep, encoder_hidden_states, down_block_additional_residuals=None,
                mid_block_additional_residual=None):
        ctx = encoder_hidden_states if isinstance(encoder_hidden_states, PaddedContext) else PaddedContext(encoder_hidden_states)
        temb = self._temb(timestep, sample.dtype)
        self._prologue(temb, ctx)
        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, temb, ctx)
            skips += outs
        if down_block_additional_residuals is not None:
            # (skip, ControlNet residual) pairs: the sum is formed where the up block concatenates it (one fused pass on the
            # GPU instead of an add pass here and a cat pass there)
            skips = [(s, r) for s, r in zip(skips, down_block_additional_residuals)]
        x = self.mid_block(x, temb, ctx)
        if mid_block_additional_residual is not None:
            x = x + mid_block_additional_residual
        for blk in self.up_blocks:
            x = blk(x, skips, temb, ctx)
        return self.conv_out(group_norm_act(self.conv_norm_out, x, True))


class ControlNetConditioningEmbedding(nn.Module):
    def __init__(self, out_ch, cond_ch, chans):
        super().__init__()
        self.conv_in = Conv2d(cond_ch, chans[0], 3, padding=1)
        blocks = []
        for i in range(len(chans) - 1):
            blocks.append(Conv2d(chans[i], chans[i], 3, padding=1))
            blocks.append(Conv2d(chans[i], chans[i + 1], 3, padding=1, stride=2))
        self.blocks = nn.ModuleList(blocks)
        self.conv_out = Conv2d(chans[-1], out_ch, 3, padding=1)

    def forward(self, c):
        x = self.conv_in.forward_silu(c)
        for b in self.blocks:
            x = b.forward_silu(x)
        return self.conv_out(x)


class ControlNetModel(_Encoder):
    """ControlNetModel.from_unet(unet, conditioning_channels=22) (controlnet_train/diffusers_train_controlnet.py:638)."""

    def __init__(self, arch: SDArch, cond_channels: Optional[int] = None):
        """cond_channels: 22 (arch default: the light-geo ControlNet) or 3 (the sd15 depth / normal ControlNets the
        reference names for control_types 'depth' / 'normal', dreammat_guidance.py:103-106)"""
        super().__init__()
        self._build_encoder(arch)
        bo = arch.block_out
        self.cond_channels = cond_channels or arch.cond_channels
        self.controlnet_cond_embedding = ControlNetConditioningEmbedding(bo[0], self.cond_channels, arch.cond_embed_channels)
        chans = [bo[0]]
        for i, oc in enumerate(bo):
            chans += [oc] * arch.layers_per_block
            if i != len(bo) - 1:
                chans.append(oc)
        self.controlnet_down_blocks = nn.ModuleList([Conv2d(c, c, 1) for c in chans])
        self.controlnet_mid_block = Conv2d(bo[-1], bo[-1], 1)

    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale=1.0):
        ctx = encoder_hidden_states if isinstance(encoder_hidden_states, PaddedContext) else PaddedContext(encoder_hidden_states)
        temb = self._temb(timestep, sample.dtype)
        self._prologue(temb, ctx)
        # the conditioning stem depends on the image only: the B views are embedded once, the (text / negative / null) branches
        # share the result instead of embedding the same image three times
        emb = self.controlnet_cond_embedding(controlnet_cond)
        if self.conv_in.small_ok(sample) and emb.dtype == sample.dtype and sample.shape[0] % emb.shape[0] == 0:
            x = self.conv_in.forward_small(sample, 0, emb)          # conv_in + bias + emb[b % B] in one pass, one rounding
        else:
            x = self.conv_in(sample)
            if emb.shape[0] != x.shape[0]:
                emb = emb.repeat(x.shape[0] // emb.shape[0], 1, 1, 1)
            x = x + emb
        outs = [x]
        for blk in self.down_blocks:
            x, o = blk(x, temb, ctx)
            outs += o
        x = self.mid_block(x, temb, ctx)
        unit = isinstance(conditioning_scale, (int, float)) and float(conditioning_scale) == 1.0    # (x * 1.0 still launches a kernel: 14 per forward)
        down = [conv(o) if unit else conv(o) * conditioning_scale for conv, o in zip(self.controlnet_down_blocks, outs)]
        mid = self.controlnet_mid_block(x) if unit else self.controlnet_mid_block(x) * conditioning_scale
        return down, mid

    @classmethod
    def from_unet(cls, unet: UNet2DConditionModel, cond_channels: Optional[int] = None):
        cn = cls(unet.arch, cond_channels)
        sd = {k: v for k, v in unet.state_dict().items()
              if k.startswith(("conv_in.", "time_embedding.", "down_blocks.", "mid_block."))}
        cn.load_state_dict(sd, strict=False)
        for conv in list(cn.controlnet_down_blocks) + [cn.controlnet_mid_block, cn.controlnet_cond_embedding.conv_out]:
            nn.init.zeros_(conv.weight)
            nn.init.zeros_(conv.bias)
        return cn


# ------------------------------------------------------------------------------------------ VAE encoder
class VaeAttention(nn.Module):
    """mid-block self attention of AutoencoderKL (1 head, d = 512; needs autograd: the SDS gradient
    flows through the VAE encoder, dreammat_guidance.py:284-292)."""

    def __init__(self, ch):
        super().__init__()
        self.group_norm = nn.GroupNorm(32, ch, eps=1e-6)
        self.to_q = nn.Linear(ch, ch)
        self.to_k = nn.Linear(ch, ch)
        self.to_v = nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch), nn.Identity()])
        self._q16 = None

    def _q_sixteenth(self):
        """(to_q.weight / 16, to_q.bias / 16) of the frozen projection, formed once per weight version (persistent tensors: the
        transposed-weight cache of layers.linear_fused is keyed by the tensor)"""
        w, b = self.to_q.weight, self.to_q.bias
        key = (w.data_ptr(), w._version, b._version, w.dtype)
        if self._q16 is None or self._q16[0] != key:
            with torch.no_grad():
                self._q16 = (key, (w.detach() * 0.0625).contiguous(), (b.detach() * 0.0625).contiguous())
        return self._q16[1], self._q16[2]

    def forward(self, x):
        B, C, H, W = x.shape
        from . import layers
        from .. import hipops
        if torch.is_grad_enabled() and x.requires_grad and layers._gn_kernel_ok(self.group_norm, x):
            # x feeds the norm and the residual: both gradients meet in the GroupNorm backward kernel (see ResnetBlock2D)
            h, x = hipops.groupnorm_nhwc_skip(x.permute(0, 2, 3, 1).contiguous(), self.group_norm.weight, self.group_norm.bias,
                                              self.group_norm.eps, 0)
            x = x.permute(0, 3, 1, 2)
            h = h.reshape(B, H * W, C)
        else:
            h = group_norm_act(self.group_norm, x, False).permute(0, 2, 3, 1).reshape(B, H * W, C)
        lin = layers.linear_fused        # (frozen projections under autograd: forward + data gradient on the hand-written GEMM, round 6)
        frozen = not (self.to_q.weight.requires_grad or self.to_q.bias.requires_grad)
        if (layers._native_expected(h) and frozen and h.shape[1] % 256 == 0 and C % 64 == 0 and h.shape[1] <= 16384
                and hipops.gemm_fused_ok(B * H * W, C, C) and os.environ.get("DREAMMAT_VAE_ATTENTION") != "blas"):
            # the one differentiated attention of the path (1 head of 512): a few images at a time on the hand-written GEMM, the
            # row-softmax kernels and dm_transpose over one reused [G, S, S] buffer (hipops._WideHeadAttention; round 6 -- it was
            # torch.matmul on hipBLASLt over a [B, S, S] score tensor).  to_q carries 2^-4 of the scale -- exact, and it keeps the unscaled q.k of 512
            # channels (22.6x the logits) inside IEEE half -- the softmax kernel the other 16 / sqrt(C), in fp32.
            wq, bq = self._q_sixteenth()
            q, k, v = lin(h, wq, bq), lin(h, self.to_k.weight, self.to_k.bias), lin(h, self.to_v.weight, self.to_v.bias)
            o = hipops.wide_head_attention(q, k, v, 16.0 * C ** -0.5)
            o = lin(o, self.to_out[0].weight, self.to_out[0].bias)
            return o.reshape(B, H, W, C).permute(0, 3, 1, 2) + x
        if layers._native_expected(h):
            layers.note_fallback("vae_attention", f"C={C} S={H * W} autograd: QK^T + PV (and their four backward products)")
        q, k, v = lin(h, self.to_q.weight, self.to_q.bias), lin(h, self.to_k.weight, self.to_k.bias), lin(h, self.to_v.weight, self.to_v.bias)
        if q.dtype == torch.float16:
            # IEEE half: the UNSCALED q.k of 512 channels (22.6x the scaled logits) can pass 65504 -> inf -> NaN in the stored
            # score matrix; the scale is applied inside the product, in fp32, before the rounding (what diffusers'
            # attention does), and the row kernel gets scale 1
            s = torch.baddbmm(q.new_empty(B, H * W, H * W), q, k.transpose(1, 2), beta=0.0, alpha=C ** -0.5)
            s_scale = 1.0
        else:
            s = torch.matmul(q, k.transpose(1, 2))
            s_scale = C ** -0.5
        if hipops.softmax_rows_ok(s) and layers.CONV_BACKEND == "mfma":
            p = hipops.softmax_rows(s, s_scale)         # scale, fp32 softmax and the rounding in one pass (and one pass back)
        else:
            p = torch.softmax((s * s_scale).float(), dim=-1).to(q.dtype)
        o = lin(torch.matmul(p, v), self.to_out[0].weight, self.to_out[0].bias)
        return o.reshape(B, H, W, C).permute(0, 3, 1, 2) + x


class VaeEncoder(nn.Module):
    def __init__(self, block_out):
        super().__init__()
        self.conv_in = Conv2d(3, block_out[0], 3, padding=1)
        downs = []
        ch = block_out[0]
        for i, oc in enumerate(block_out):
            blk = nn.Module()
            blk.resnets = nn.ModuleList([ResnetBlock2D(ch if j == 0 else oc, oc, 0, eps=1e-6) for j in range(2)])
            if i != len(block_out) - 1:
                blk.downsamplers = nn.ModuleList([Downsample2D(oc, asymmetric_pad=True)])
            downs.append(blk)
            ch = oc
        self.down_blocks = nn.ModuleList(downs)
        mid = nn.Module()
        mid.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, 0, eps=1e-6), ResnetBlock2D(ch, ch, 0, eps=1e-6)])
        mid.attentions = nn.ModuleList([VaeAttention(ch)])
        self.mid_block = mid
        self.conv_norm_out = nn.GroupNorm(32, ch, eps=1e-6)
        self.conv_out = Conv2d(ch, 8, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for blk in self.down_blocks:
            for r in blk.resnets:
                x = r(x)
            if hasattr(blk, "downsamplers"):
                x = blk.downsamplers[0](x)
        x = self.mid_block.resnets[0](x)
        x = self.mid_block.attentions[0](x)
        x = self.mid_block.resnets[1](x)
        return self.conv_out(group_norm_act(self.conv_norm_out, x, True))


class AutoencoderKLEncoder(nn.Module):
    """`vae.encode(x).latent_dist` half of AutoencoderKL (the decoder is not on the training path)."""

    def __init__(self, arch: SDArch):
        super().__init__()
        self.encoder = VaeEncoder(arch.vae_block_out)
        self.quant_conv = Conv2d(8, 8, 1)
        self.scaling_factor = arch.vae_scaling_factor

    def encode_moments(self, x):
        m = self.quant_conv(self.encoder(x))
        mean, logvar = m.chunk(2, dim=1)
        return mean, logvar.clamp(-30.0, 20.0)

    def sample(self, x, noise):
        """posterior.sample() with the N(0,1) draw supplied by the caller."""
        mean, logvar = self.encode_moments(x)
        return mean + torch.exp(0.5 * logvar) * noise.to(mean.dtype)


class DDIMScheduler:
    """The only parts of diffusers' DDIMScheduler the path touches (dreammat_guidance.py:148-154,
    191-193, 463): scaled_linear betas 0.00085..0.012 over 1000 steps, alphas_cumprod, add_noise."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
        self.num_train_timesteps = num_train_timesteps
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)

    def add_noise(self, x, noise, t):
        ac = self.alphas_cumprod.to(device=x.device, dtype=x.dtype)
        a = ac[t] ** 0.5
        s = (1 - ac[t]) ** 0.5
        while a.dim() < x.dim():
            a, s = a.unsqueeze(-1), s.unsqueeze(-1)
        return a * x + s * noise
