"""Training of the 22-channel geometry- and light-aware ControlNet: the loss and the optimisation loop of
controlnet_train/diffusers_train_controlnet.py (reference repository root): ControlNet initialised from the UNet
(`ControlNetModel.from_unet(unet, conditioning_channels=22)`, :535-548) with zero output convolutions, everything else
frozen (:638), per step (:858-915)

    latents = vae.encode(img).sample() * scaling_factor;  noise ~ N;  t ~ U{0..999};  x_t = add_noise(latents, noise, t)
    down, mid = controlnet(x_t, t, text, cond);  eps = unet(x_t, t, text, down, mid);  loss = mse(eps, noise)

and the classifier-free-guidance dropout of the dataset (controlnet_train/diffusers_dataset.py:148-159: 5 % each for
dropping the whole condition / the depth / the normal / the light maps, 30 % for an empty prompt), the dataset itself
(`ControlNetRenderDataset`: the reference's Blender output tree color/ depth/ normal/ light/ per object, read with PIL,
diffusers_dataset.py:10-159) and a one-process-per-GPU launcher (`python -m dreammat_amd.controlnet_train`, the reference
uses HF accelerate, diffusers_train_controlnet.py:858-915: same loop, gradients all-reduced over RCCL).  What is NOT here: the CLIP
text encoder (text embeddings come from prompt.py's encoder or its synthetic stand-in) and Blender (the tree is an input).  The frozen UNet / VAE run on the same HIP
kernels as the SDS path wherever those are differentiable (implicit-GEMM conv data gradients, GroupNorm backward); the
differentiated attention runs the MFMA forward-with-statistics + backward kernels (csrc/attn_bwd.hip), the trainable 3x3
convolutions forward / data gradient / weight gradient on the conv kernels (csrc/conv_wgrad.hip; layers with whole 64-channel
tiles, the others through im2col) and GroupNorm returns its affine gradients; Linear / LayerNorm / GEGLU are torch autograd
(hipBLASLt + ATen).  A full-size step (batch 4, 512^2): tools/train_step_probe.py.
"""
import json
import os

import numpy as np
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
    max_grad_norm (:899-902).

    Mixed precision as the reference's `accelerate --mixed_precision` does it: when the ControlNet computes in bf16 (the MFMA
    training kernels need bf16 weights), the optimizer runs on fp32 MASTER copies of its parameters -- pure bf16 parameters
    under AdamW at lr 1e-5 lose most updates to rounding (a bf16 ulp at |w| = 0.02 is 1.2e-4).  The masters are created the
    first time gradients are applied, from whatever dtype the module has then (so `trainer.controlnet.bfloat16()` after
    construction works); the state dict is the masters' values."""

    def __init__(self, vae, unet, controlnet=None, lr=1e-5, weight_decay=1e-2, max_grad_norm=1.0, compute_dtype=None):
        freeze(vae, unet)
        self.vae, self.unet = vae, unet
        self.controlnet = controlnet if controlnet is not None else init_controlnet(unet)
        if compute_dtype is not None:
            self.controlnet.to(dtype=compute_dtype)
        self.scheduler = DDIMScheduler()
        self.lr, self.weight_decay = lr, weight_decay
        self.opt, self.master = None, None
        self.max_grad_norm = max_grad_norm
        self.global_step = 0
        self._build_optimizer()

    def _params(self):
        return [p for p in self.controlnet.parameters() if p.requires_grad]

    def _build_optimizer(self):
        params = self._params()
        self._opt_dtype = params[0].dtype
        self._opt_device = params[0].device
        if any(p.dtype != torch.float32 for p in params):
            self.master = [p.detach().float().clone().requires_grad_() for p in params]
            target = self.master
        else:
            self.master, target = None, params
        self.opt = torch.optim.AdamW(target, lr=self.lr, weight_decay=self.weight_decay)

    def apply_gradients(self, world=1):
        """all-reduce (ONE flat collective) -> clip -> AdamW on the fp32 masters -> cast back; called after loss.backward()."""
        params = self._params()
        if params[0].dtype != self._opt_dtype or params[0].device != self._opt_device:     # module cast / moved after construction
            assert self.global_step == 0, "the ControlNet changed dtype / device after the first optimizer step"
            self._build_optimizer()
        targets = self.master if self.master is not None else params
        if self.master is not None:
            for m, p in zip(self.master, params):
                m.grad = None if p.grad is None else p.grad.float()
                p.grad = None
        if world > 1:
            import torch.distributed as dist
            live = [t for t in targets if t.grad is not None]
            flat = torch.cat([t.grad.reshape(-1) for t in live])
            dist.all_reduce(flat)
            flat /= world
            o = 0
            for t in live:
                t.grad.copy_(flat[o:o + t.numel()].view_as(t.grad)); o += t.numel()
        torch.nn.utils.clip_grad_norm_(targets, self.max_grad_norm)
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        if self.master is not None:
            with torch.no_grad():
                for m, p in zip(self.master, params):
                    p.copy_(m)
        self.global_step += 1

    def step(self, pixel_values, cond, text_emb, null_emb=None, generator=None, **fixed):
        if null_emb is not None:
            text_emb, cond = cfg_dropout(text_emb, null_emb, cond, generator)
        loss = controlnet_training_loss(self.vae, self.unet, self.controlnet, self.scheduler, pixel_values, cond, text_emb,
                                        generator=generator, **fixed)
        loss.backward()
        self.apply_gradients()
        return loss.detach()

    def state_dict(self):
        """diffusers key layout (loadable by dreammat_amd.sd.loading and by diffusers' ControlNetModel); trainable tensors
        come from the fp32 masters when there are any."""
        sd = self.controlnet.state_dict()
        if self.master is not None:
            by_id = {id(p): m for p, m in zip(self._params(), self.master)}
            for k, p in self.controlnet.named_parameters():
                if id(p) in by_id:
                    sd[k] = by_id[id(p)].detach()
        return sd


# ------------------------------------------------------------------------------------------------ dataset
def _imread(path):
    from PIL import Image
    if not os.path.exists(path):
        return None
    return Image.open(path)


def _resize_area(img, size):
    """cv2.INTER_AREA for a downscale = box filter"""
    from PIL import Image
    return img if img.size == (size, size) else img.resize((size, size), Image.BOX)


def load_rgb(path, size):
    """diffusers_dataset.py:27-42: RGB(A) PNG -> [size,size,3] in [0,1]; transparent pixels -> 0; a missing file -> zeros."""
    img = _imread(path)
    if img is None:
        return np.zeros((size, size, 3), np.float32)
    a = np.asarray(img.convert("RGBA") if img.mode in ("RGBA", "LA", "P") else img.convert("RGB")).copy()
    if a.shape[2] == 4:
        a[a[..., 3] == 0] = 0
        a = a[..., :3]
    from PIL import Image
    a = np.asarray(_resize_area(Image.fromarray(a), size))
    return a.astype(np.float32) / 255.0


def load_target(path, size):
    """:44-60: -> ([size,size,3] in [-1,1], had_alpha); transparent pixels -> white; a missing file -> ones."""
    img = _imread(path)
    if img is None:
        return np.ones((size, size, 3), np.float32), False
    alpha = img.mode in ("RGBA", "LA")
    a = np.asarray(img.convert("RGBA") if alpha else img.convert("RGB")).copy()
    if alpha:
        a[a[..., 3] == 0] = 255
        a = a[..., :3]
    from PIL import Image
    a = np.asarray(_resize_area(Image.fromarray(a), size))
    return a.astype(np.float32) / 127.5 - 1.0, alpha


def load_depth(path, size):
    """:62-82: 16-bit PNG in millimetres -> [size,size,1]: 0 on the background, else the inverse depth min-max normalised to
    [0.3, 1] over the object; also the object mask."""
    from PIL import Image
    img = Image.open(path)
    if img.size != (size, size):
        img = img.resize((size, size), Image.NEAREST)
    depth = np.asarray(img).astype(np.float64) / 1000.0
    mask = depth > 0
    if mask.sum() <= 0:
        return depth[..., None].astype(np.float32), mask
    inv = 1.0 / (depth + 1e-6)
    dmax, dmin = inv[mask].max(), inv[mask].min()
    depth[mask] = (1 - 0.3) * (inv[mask] - dmin) / (dmax - dmin + 1e-6) + 0.3
    return depth[..., None].astype(np.float32), mask


class ControlNetRenderDataset(torch.utils.data.Dataset):
    """diffusers_dataset.py:84-159 (DiffusersDataset): <datadir>/<object>/{color,depth,normal,light}/..., 5 environments x 16
    views per object, prompts from a JSON {object: prompt}.  Item = pixel_values [H,W,3] in [-1,1], the prompt string, and
    conditioning_pixel_values [H,W,22] = depth | normal | light m0r0 m0r.5 m0r1 m1r0 m1r.5 m1r1.  `use_cfg`: the reference's
    condition / prompt dropout, drawn from `rng` (random.Random) so that tests can pin it."""
    env_num, view_num = 5, 16

    def __init__(self, datadir, promptfile, size=512, use_cfg=False, rng=None):
        import random
        self.size, self.use_cfg, self.rng = size, use_cfg, rng or random.Random()
        with open(promptfile) as fh:
            content = json.load(fh)
        self.obj_info = [{"path": os.path.join(datadir, k), "prompt": v} for k, v in content.items()
                         if os.path.isdir(os.path.join(datadir, k))]

    def __len__(self):
        return len(self.obj_info) * self.env_num * self.view_num

    def __getitem__(self, idx):
        per = self.env_num * self.view_num
        info = self.obj_info[idx // per]
        env, view = (idx % per) // self.view_num + 1, (idx % per) % self.view_num
        p, S = info["path"], self.size
        target, alpha = load_target(f"{p}/color/{view:03d}_color_env{env}.png", S)
        depth, mask = load_depth(f"{p}/depth/{view:03d}.png", S)
        normal = load_rgb(f"{p}/normal/{view:03d}.png", S)
        if not alpha:
            target[~mask] = 1.0
        lights = [load_rgb(f"{p}/light/{view:03d}_m{m}r{r}_env{env}.png", S)
                  for m in ("0.0", "1.0") for r in ("0.0", "0.5", "1.0")]
        source = np.concatenate([depth, normal] + lights, axis=-1)
        prompt = info["prompt"]
        if self.use_cfg:
            r = self.rng.random()
            if r < 0.05:
                source = np.zeros_like(source)
            elif 0.05 < r < 0.1:
                source[..., 0] = 0
            elif 0.1 < r < 0.15:
                source[..., 1:4] = 0
            elif 0.15 < r < 0.2:
                source[..., 4:] = 0
            elif 0.2 < r < 0.5:
                prompt = ""
        return dict(pixel_values=torch.from_numpy(target), input_ids=prompt, conditioning_pixel_values=torch.from_numpy(source))


def collate(items):
    """training script :700-720: NHWC arrays -> NCHW float tensors, prompts stay a list"""
    return {"pixel_values": torch.stack([i["pixel_values"] for i in items]).permute(0, 3, 1, 2).contiguous().float(),
            "conditioning_pixel_values": torch.stack([i["conditioning_pixel_values"] for i in items]).permute(0, 3, 1, 2).contiguous().float(),
            "prompts": [i["input_ids"] for i in items]}


def main(argv=None):
    """One process per GPU (python -m torch.distributed.run ... -m dreammat_amd.controlnet_train): the loop of
    diffusers_train_controlnet.py:858-915 with the ControlNet gradients all-reduced before clipping."""
    import argparse
    import torch.distributed as dist
    from .prompt import StableDiffusionPromptProcessor
    from .sd import AutoencoderKLEncoder, UNet2DConditionModel
    from .sd.models import arch_for
    from .sd.loading import load_component
    ap = argparse.ArgumentParser()
    ap.add_argument("--pretrained_model_name_or_path", default="stabilityai/stable-diffusion-2-1-base")
    ap.add_argument("--train_data_dir", required=True)
    ap.add_argument("--prompt_file", required=True)
    ap.add_argument("--output_dir", default="controlnet_out")
    ap.add_argument("--resolution", type=int, default=512)
    ap.add_argument("--train_batch_size", type=int, default=4)
    ap.add_argument("--learning_rate", type=float, default=1e-5)
    ap.add_argument("--max_train_steps", type=int, default=1000)
    ap.add_argument("--checkpointing_steps", type=int, default=500)
    ap.add_argument("--use_cfg", action="store_true")
    ap.add_argument("--synthetic", action="store_true", help="random-init nets / pseudo text embeddings (no checkpoints on this box)")
    a = ap.parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))) if torch.cuda.is_available() else torch.device("cpu")
    if dev.type == "cuda":
        torch.cuda.set_device(dev)                    # before the process group: RCCL binds to the current device
    if world > 1:
        dist.init_process_group("nccl" if dev.type == "cuda" else "gloo")
    arch = arch_for(a.pretrained_model_name_or_path)
    torch.manual_seed(0)
    vae, unet = AutoencoderKLEncoder(arch), UNet2DConditionModel(arch)
    if not a.synthetic:
        load_component(vae, a.pretrained_model_name_or_path, "vae")
        load_component(unet, a.pretrained_model_name_or_path, "unet")
    dt = torch.bfloat16 if dev.type == "cuda" else torch.float32
    vae.to(dev, dt); unet.to(dev, dt)
    cn = init_controlnet(unet).to(dev, dt)            # the compute dtype of the frozen nets; fp32 masters live in the trainer
    tr = ControlNetTrainer(vae, unet, cn, lr=a.learning_rate)
    ds = ControlNetRenderDataset(a.train_data_dir, a.prompt_file, a.resolution, a.use_cfg)
    sampler = torch.utils.data.distributed.DistributedSampler(ds, world, rank, shuffle=True, seed=0) if world > 1 else None
    dl = torch.utils.data.DataLoader(ds, batch_size=a.train_batch_size, shuffle=sampler is None, sampler=sampler,
                                     collate_fn=collate, drop_last=True)
    embed = {}
    pp = StableDiffusionPromptProcessor({"prompt": " ", "pretrained_model_name_or_path": a.pretrained_model_name_or_path,
                                         "synthetic": a.synthetic, "use_perp_neg": False})      # ONE text encoder front-end

    def text(prompts):
        todo = [p for p in dict.fromkeys(prompts) if p not in embed]
        if todo:
            for p, e in zip(todo, pp._encode([p or " " for p in todo], keep_encoder=True)):      # (loaded once, not per batch)
                embed[p] = e[None].to(dev)
        return torch.cat([embed[p] for p in prompts])
    epoch = 0
    while tr.global_step < a.max_train_steps:
        if sampler is not None:
            sampler.set_epoch(epoch)                  # a new shuffle every pass
        epoch += 1
        for batch in dl:
            loss = controlnet_training_loss(tr.vae, tr.unet, tr.controlnet, tr.scheduler, batch["pixel_values"].to(dev, dt),
                                            batch["conditioning_pixel_values"].to(dev), text(batch["prompts"]))
            loss.backward()
            tr.apply_gradients(world)
            if rank == 0 and tr.global_step % 10 == 0:
                print(json.dumps({"step": tr.global_step, "loss": float(loss)}), flush=True)
            if rank == 0 and tr.global_step % a.checkpointing_steps == 0:
                os.makedirs(a.output_dir, exist_ok=True)
                torch.save(tr.state_dict(), os.path.join(a.output_dir, f"controlnet_step{tr.global_step}.pt"))
            if tr.global_step >= a.max_train_steps:
                break
    return tr


if __name__ == "__main__":
    main()
