"""Training of the 22-channel geometry- and light-aware ControlNet: the loss and the optimisation loop of
controlnet_train/diffusers_train_controlnet.py (reference repository root): ControlNet initialised from the UNet
(`ControlNetModel.from_unet(unet, conditioning_channels=22)`, :535-548) with zero output convolutions, everything else
frozen (:638), per step (:858-915)

    latents = vae.encode(img).sample() * scaling_factor;  noise ~ N;  t ~ U{0..999};  x_t = add_noise(latents, noise, t)
    down, mid = controlnet(x_t, t, text, cond);  eps = unet(x_t, t, text, down, mid);  loss = mse(eps, noise)

and the classifier-free-guidance dropout of the dataset (controlnet_train/diffusers_dataset.py:148-159: 5 % each for
dropping the whole condition / the depth / the normal / the light maps, 30 % for an empty prompt).  What is NOT here: the accelerate launcher, the CLIP text
encoder (text embeddings are an input) and the Blender dataset producer.  The frozen UNet / VAE run on the same HIP
kernels as the SDS path wherever those are differentiable (implicit-GEMM conv data gradients, GroupNorm backward); the
attention of the UNet falls back to the matmul-softmax path under autograd (the MFMA attention kernel is forward-only)
and the trainable ControlNet convolutions use the im2col lowering, so this is a correctness-first implementation.
"""
import torch
import torch.nn.functional as F

from .sd import ControlNetModel, DDIMScheduler
from .sd.layers import PaddedContext


def freeze(*modules):
    for m in modules:
        for p in m.parameters():
            p.requires_grad_(False)
        m.eval()


def init_controlnet(unet):
    """ControlNetModel.from_unet: encoder copy of the UNet, zero convs (training script :535-548)."""
    cn = ControlNetModel.from_unet(unet)
    cn.train()
    for p in cn.parameters():
        p.requires_grad_(True)
    return cn


def cfg_dropout(text_emb, null_emb, cond, generator=None):
    """diffusers_dataset.py:148-159 (use_cfg), one uniform draw r per sample:
    r < .05 zero the whole condition | .05-.1 zero depth | .1-.15 zero normal | .15-.2 zero the light maps |
    .2-.5 empty prompt.  text_emb [B,77,D], null_emb [1|B,77,D] (embedding of ""), cond [B,22,H,W]."""
    B = text_emb.shape[0]
    r = torch.rand(B, generator=generator, device="cpu")
    cond = cond.clone()
    k = (r < 0.05).to(cond.device)[:, None, None, None]
    cond = torch.where(k, torch.zeros_like(cond), cond)
    for lo, hi, sl in ((0.05, 0.10, slice(0, 1)), (0.10, 0.15, slice(1, 4)), (0.15, 0.20, slice(4, None))):
        m = ((r > lo) & (r < hi)).to(cond.device)[:, None, None, None]
        cond[:, sl] = torch.where(m, torch.zeros_like(cond[:, sl]), cond[:, sl])
    dt = ((r > 0.2) & (r < 0.5)).to(text_emb.device)[:, None, None]
    return torch.where(dt, null_emb.expand_as(text_emb), text_emb), cond


def controlnet_training_loss(vae, unet, controlnet, scheduler, pixel_values, cond, text_emb, timesteps=None, noise=None,
                             posterior_noise=None, generator=None):
    """pixel_values [B,3,H,W] in [-1,1], cond [B,22,H,W] in [0,1], text_emb [B,77,D] -> scalar MSE (fp32)."""
    dev, dt = pixel_values.device, pixel_values.dtype
    B = pixel_values.shape[0]
    with torch.no_grad():
        lh, lw = pixel_values.shape[2] // 8, pixel_values.shape[3] // 8
        if posterior_noise is None:
            posterior_noise = torch.randn(B, 4, lh, lw, generator=generator, device="cpu").to(dev)
        latents = (vae.sample(pixel_values, posterior_noise.to(dt)) * vae.scaling_factor).to(dt)
        if noise is None:
            noise = torch.randn(latents.shape, generator=generator, device="cpu").to(dev)
        if timesteps is None:
            timesteps = torch.randint(0, scheduler.num_train_timesteps, (B,), generator=generator, device="cpu").to(dev)
        noisy = scheduler.add_noise(latents, noise.to(dt), timesteps.long())
    ctx = PaddedContext(text_emb.to(dt))
    down, mid = controlnet(noisy, timesteps, ctx, cond.to(dt), 1.0)
    pred = unet(noisy, timesteps, ctx, down_block_additional_residuals=down, mid_block_additional_residual=mid)
    return F.mse_loss(pred.float(), noise.float(), reduction="mean")


class ControlNetTrainer:
    """AdamW(lr 1e-5 by default in the reference's train.sh) over the ControlNet parameters only; gradient clipping at
    max_grad_norm (:899-902)."""

    def __init__(self, vae, unet, controlnet=None, lr=1e-5, weight_decay=1e-2, max_grad_norm=1.0):
        freeze(vae, unet)
        self.vae, self.unet = vae, unet
        self.controlnet = controlnet if controlnet is not None else init_controlnet(unet)
        self.scheduler = DDIMScheduler()
        self.opt = torch.optim.AdamW([p for p in self.controlnet.parameters() if p.requires_grad], lr=lr,
                                     weight_decay=weight_decay)
        self.max_grad_norm = max_grad_norm
        self.global_step = 0

    def step(self, pixel_values, cond, text_emb, null_emb=None, generator=None, **fixed):
        if null_emb is not None:
            text_emb, cond = cfg_dropout(text_emb, null_emb, cond, generator)
        loss = controlnet_training_loss(self.vae, self.unet, self.controlnet, self.scheduler, pixel_values, cond, text_emb,
                                        generator=generator, **fixed)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.controlnet.parameters(), self.max_grad_norm)
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        self.global_step += 1
        return loss.detach()

    def state_dict(self):
        """diffusers key layout (loadable by dreammat_amd.sd.loading and by diffusers' ControlNetModel)."""
        return self.controlnet.state_dict()
