"""`stable-diffusion-dreammat-guidance` plugin
(threestudio/models/guidance/dreammat_guidance.py:43-627, class StableDiffusionLightGuidance).

Same Config keys, same __call__ / update_step contract, same arithmetic:
  encode_images (:284-292) -> compute_grad_sds (:440-497) with compute_without_perpneg (:388-438):
  3 branches (text / negative / null) through ControlNet then UNet -> grad = (1-abar_t)(c e_text +
  u e_uncond + n e_null + s e) -> nan_to_num -> loss_sds = 0.5*||latents - (latents-grad)||^2 / B.
Differences, all forced by SURVEY D4 (the reference only works at B=1 per process):
  * the ControlNet conditioning is repeated for the 3 branches (the reference relies on a [3]+[1]
    broadcast that breaks for B>1);
  * all random draws (t, noise, VAE posterior noise) can be injected (`rng=` dict) for parity tests.
Nets run in bf16 on the MI355X (reference: fp16, `half_precision_weights`), attention in the MFMA kernel.
"""
from dataclasses import dataclass, field
from typing import Any, List, Optional

import os

import torch
import torch.nn.functional as F

import dreammat_amd
from .base import BaseObject
from .config import C
from .sd import (AutoencoderKLEncoder, ControlNetModel, DDIMScheduler, PaddedContext, UNet2DConditionModel,
                 arch_for, load_component)


@dreammat_amd.register("stable-diffusion-dreammat-guidance")
class StableDiffusionLightGuidance(BaseObject):
    @dataclass
    class Config(BaseObject.Config):
        width: int = 512
        height: int = 512
        cache_dir: Optional[str] = None
        pretrained_model_name_or_path: str = "stabilityai/stable-diffusion-2-1-base"
        controlnet_path: Optional[str] = None
        enable_memory_efficient_attention: bool = False
        enable_sequential_cpu_offload: bool = False
        enable_attention_slicing: bool = False
        enable_channels_last_format: bool = False
        half_precision_weights: bool = True
        use_controlnet: bool = False
        condition_scale: float = 1.5
        control_anneal_start_step: Optional[int] = None
        control_anneal_end_scale: Optional[float] = None
        control_types: List = field(default_factory=lambda: ["depth", "canny"])
        condition_scales: List = field(default_factory=lambda: [1.0, 1.0])
        condition_scales_anneal: List = field(default_factory=lambda: [1.0, 1.0])
        p2p_condition_type: str = "p2p"
        canny_lower_bound: int = 50
        canny_upper_bound: int = 100
        min_step_percent: Any = 0.02
        max_step_percent: Any = 0.98
        cond_scale: Any = 1
        uncond_scale: Any = 0
        null_scale: Any = -1
        noise_scale: Any = 0
        perpneg_scale: Any = 0.0
        view_dependent_prompting: bool = True
        grad_clip_val: Optional[float] = None
        grad_normalize: Optional[bool] = False
        # additions: compute dtype on the accelerator and the init seed used when no weights are on disk
        # "float16" (default: what the reference's half_precision_weights means, dreammat_guidance.py:56, 92-94 -- noise
        # prediction 1.3e-3 of fp32 at full size) | "bfloat16" (BASELINE configs[1]'s type: 1e-2 of fp32, ~3 % faster) on the accelerator
        weights_dtype: str = "float16"
        # "16bit" | "fp8": the S >= 1024 self-attention of the frozen nets on the MX-FP8 matrix instruction (BASELINE configs[4])
        attention_precision: str = "16bit"
        synthetic_seed: int = 1234
        # seeded random UNet / VAE / ControlNet when no checkpoint is on disk: allowed only when asked for (bench, tests;
        # the 'tiny*' architectures have no checkpoints at all) -- a 30k-step run on random nets must not start silently
        synthetic: bool = False
        # replay the ControlNet + UNet noise prediction (~1300 kernel launches per step, forward only, frozen weights, static
        # shapes) as ONE captured hipGraph: the host then issues one launch per step for it instead of one per kernel, which
        # is what keeps a rank busy once its share of the views is small (8 GPUs: 1 view = batch 3 per rank)
        # "auto": only when the per-rank batch is small enough for the host to be the bottleneck (see _graph_ok)
        hip_graph: Any = "auto"

    cfg: Config

    def configure(self) -> None:
        self.use_controlnet = self.cfg.use_controlnet
        on_gpu = torch.cuda.is_available()
        self.device = self.device if on_gpu else torch.device("cpu")
        if self.cfg.half_precision_weights and on_gpu:
            self.weights_dtype = getattr(torch, self.cfg.weights_dtype)
        else:
            self.weights_dtype = torch.float32
        if self.cfg.attention_precision not in ("16bit", "fp8"):
            raise ValueError(f"guidance.attention_precision must be '16bit' or 'fp8', got '{self.cfg.attention_precision}'")
        # carried by the Attention modules of THIS guidance's nets (set below), not by a process-global
        self.attention_precision = self.cfg.attention_precision if on_gpu and self.cfg.half_precision_weights else "16bit"
        arch = arch_for(self.cfg.pretrained_model_name_or_path)
        self.arch = arch
        root = self._model_root()
        gen_state = torch.random.get_rng_state()
        torch.manual_seed(self.cfg.synthetic_seed)
        with torch.device(self.device):           # parameters are created (and randomly initialised) on the GPU
            self.vae = AutoencoderKLEncoder(arch)
            self.unet = UNet2DConditionModel(arch)
        self.real_weights = {"vae": load_component(self.vae, root, "vae"),
                             "unet": load_component(self.unet, root, "unet")}
        self.controlnets = []
        if self.use_controlnet:
            # dreammat_guidance.py:99-119: 'light' = the 22-channel geometry- and light-aware ControlNet (cfg.controlnet_path);
            # 'depth' / 'normal' = the public sd15 ControlNets (3-channel condition; they only fit an sd15 base model)
            sources = {"light": (self.cfg.controlnet_path, None),
                       "depth": ("lllyasviel/control_v11f1p_sd15_depth", 3),
                       "normal": ("lllyasviel/control_v11p_sd15_normalbae", 3)}
            for ct in self.cfg.control_types:
                if ct not in sources:
                    raise ValueError(f"unsupported controlnet type '{ct}' (the reference exits here, :107-109)")
                path, cch = sources[ct]
                if path is not None and ct != "light":
                    cand = os.path.join(self.cfg.cache_dir or "", path)
                    path = cand if os.path.isdir(cand) else (path if os.path.isdir(path) else None)
                with torch.device(self.device):
                    cn = ControlNetModel.from_unet(self.unet, cch)
                loaded = load_component(cn, path, "controlnet")
                if not loaded:
                    # zero-convs would make a random-init ControlNet a no-op: give the synthetic one signal
                    for conv in list(cn.controlnet_down_blocks) + [cn.controlnet_mid_block, cn.controlnet_cond_embedding.conv_out]:
                        torch.nn.init.normal_(conv.weight, std=0.02)
                self.real_weights["controlnet" if ct == "light" else "controlnet_" + ct] = loaded
                self.controlnets.append(cn)
        torch.random.set_rng_state(gen_state)
        missing = [k for k, ok in self.real_weights.items() if not ok]
        if missing and not (self.cfg.synthetic or self.cfg.pretrained_model_name_or_path.lower().startswith("tiny")):
            raise FileNotFoundError(
                f"no checkpoint found for {missing} (pretrained_model_name_or_path='{self.cfg.pretrained_model_name_or_path}', "
                f"cache_dir='{self.cfg.cache_dir}', controlnet_path='{getattr(self.cfg, 'controlnet_path', None)}', "
                f"$DREAMMAT_SD_DIR); set guidance.synthetic=true to run on seeded random weights (benchmark / test mode)")
        for m in [self.vae, self.unet] + self.controlnets:
            m.to(device=self.device, dtype=self.weights_dtype).eval()
            for p in m.parameters():
                p.requires_grad_(False)
            if self.cfg.enable_channels_last_format:
                m.to(memory_format=torch.channels_last)
        from .sd import layers as _layers
        for m in [self.unet] + self.controlnets:
            _layers.set_attention_precision(m, self.attention_precision)
        self.scheduler = DDIMScheduler()
        self.num_train_timesteps = self.scheduler.num_train_timesteps
        self.set_min_max_steps()
        self.alphas = self.scheduler.alphas_cumprod.to(self.device)
        self.noise_scale, self.cond_scale, self.uncond_scale, self.null_scale, self.perpneg_scale = 0.0, 1.0, -0.0, -1.0, 0.0
        self.condition_scales = list(self.cfg.condition_scales)

    def _model_root(self):
        import os
        name = self.cfg.pretrained_model_name_or_path
        for cand in (name, os.path.join(self.cfg.cache_dir or "", name),
                     os.path.join(self.cfg.cache_dir or "", "models--" + name.replace("/", "--"))):
            if cand and os.path.isdir(cand):
                return cand
        return os.environ.get("DREAMMAT_SD_DIR")

    # ---------------------------------------------------------------- net wrappers (:205-292)
    def multi_control_forward(self, sample, timestep, ctx, controlnet_cond, conditioning_scale):
        down, mid = None, None
        for i, (image, scale, cn) in enumerate(zip(controlnet_cond, conditioning_scale, self.controlnets)):
            d, m = cn(sample.to(self.weights_dtype), timestep, ctx, image.to(self.weights_dtype), scale)
            if i == 0:
                down, mid = d, m
            else:
                down = [a + b for a, b in zip(down, d)]
                mid = mid + m
        return down, mid

    def forward_unet(self, latents, t, ctx, down=None, mid=None):
        return self.unet(latents.to(self.weights_dtype), t, ctx, down_block_additional_residuals=down,
                         mid_block_additional_residual=mid).to(latents.dtype)

    def encode_images(self, imgs, posterior_noise=None):
        input_dtype = imgs.dtype
        imgs = imgs * 2.0 - 1.0
        B = imgs.shape[0]
        lh, lw = imgs.shape[2] // 8, imgs.shape[3] // 8
        if posterior_noise is None:
            posterior_noise = torch.randn(B, 4, lh, lw, device=imgs.device)
        latents = self.vae.sample(imgs.to(self.weights_dtype), posterior_noise) * self.vae.scaling_factor
        return latents.to(input_dtype)

    def get_latents(self, rgb_BCHW, rgb_as_latents=False, posterior_noise=None):
        if rgb_as_latents:
            return F.interpolate(rgb_BCHW, (64, 64), mode="bilinear", align_corners=False)
        if rgb_BCHW.shape[2] != self.cfg.height:
            rgb_BCHW = F.interpolate(rgb_BCHW, (self.cfg.width, self.cfg.height), mode="bilinear", align_corners=False)
        return self.encode_images(rgb_BCHW, posterior_noise)

    def prepare_image_cond(self, control_type, cond):
        control = cond.permute(0, 3, 1, 2)
        if control_type == "depth":
            control = control.repeat(1, 3, 1, 1)
        if control.shape[2] != self.cfg.height:
            # the reference hard-codes (512,512) here (:530-532); generalised to the configured size
            control = F.interpolate(control, (self.cfg.height, self.cfg.width), mode="bilinear", align_corners=False)
        return control

    # ---------------------------------------------------------------- SDS (:388-497)
    def compute_without_perpneg(self, condition_scales, prompt_utils, latents_noisy, t, elevation, azimuth,
                                camera_distances, image_cond):
        text_embeddings = prompt_utils.get_text_embeddings(elevation, azimuth, camera_distances,
                                                           self.cfg.view_dependent_prompting,
                                                           return_null_text_embeddings=True)
        # the same selection as (bank of distinct embeddings, row ids): the frozen cross-attention K / V projections of the bank
        # are computed once and gathered per step (sd/layers.py PaddedContext)
        bank_ids = None
        if hasattr(prompt_utils, "get_text_embedding_bank") and text_embeddings.is_cuda and not os.environ.get("DREAMMAT_NO_KV_BANK"):
            bank, ids = prompt_utils.get_text_embedding_bank(elevation, azimuth, camera_distances, self.cfg.view_dependent_prompting)
            bank_ids = (self._bank_cast(bank), ids)
        with torch.no_grad():
            if self._graph_ok(latents_noisy):
                noise_pred = self._noise_pred_graphed(latents_noisy, t, text_embeddings, image_cond, condition_scales, bank_ids)
            else:
                noise_pred = self._noise_pred(latents_noisy, t, text_embeddings, image_cond, condition_scales, bank_ids)
        return noise_pred.chunk(3)

    def compute_with_perpneg(self, condition_scales, prompt_utils, latents_noisy, t, elevation, azimuth, camera_distances,
                             image_cond):
        """dreammat_guidance.py:319-386: five branch items per view (text, negative prompt, two Perp-Neg view prompts, empty
        prompt) through ControlNet + UNet; the two view-prompt predictions enter through their components perpendicular to
        (text - uncond), weighted by the prompt processor's negative weights."""
        from .prompt import perpendicular_component
        text_embeddings, neg_w = prompt_utils.get_text_embeddings_perp_neg(elevation, azimuth, camera_distances,
                                                                           self.cfg.view_dependent_prompting,
                                                                           return_null_text_embeddings=True)
        B = latents_noisy.shape[0]
        n_neg = neg_w.shape[-1]
        # The reference stacks the negatives view-major (v0n0, v0n1, v1n0, ...) against latents repeated branch-major
        # (cat([x] * 5)) and reads them back with [i::n_neg]: consistent at its B = 1 only.  Here the negatives are put
        # branch-major (all views' first negative, then all views' second) so that item k of every branch is view k (D4).
        neg = text_embeddings[2 * B:2 * B + n_neg * B]
        neg = neg.view(B, n_neg, *neg.shape[1:]).transpose(0, 1).reshape(n_neg * B, *neg.shape[1:])
        text_embeddings = torch.cat([text_embeddings[:2 * B], neg, text_embeddings[2 * B + n_neg * B:]], dim=0)
        with torch.no_grad():
            noise_pred = self._noise_pred(latents_noisy, t, text_embeddings, image_cond, condition_scales, None,
                                          n_branch=3 + n_neg)
        e_text, e_uncond, e_null = noise_pred[:B], noise_pred[B:2 * B], noise_pred[(2 + n_neg) * B:]
        e_pos = e_text - e_uncond
        accum = 0
        for i in range(n_neg):
            e_i = noise_pred[(2 + i) * B:(3 + i) * B]
            accum = accum + neg_w[:, i].view(-1, 1, 1, 1).to(e_pos.dtype) * perpendicular_component(e_i - e_uncond, e_pos)
        return e_text, e_uncond, e_null, accum

    BANK_CAST_POOL = 4

    def _bank_cast(self, bank):
        """the bank in the nets' dtype, converted once per bank TENSOR.  An entry holds the source bank itself and is matched
        by identity + version: an address can be handed to a new bank after the old one is freed (a re-created prompt
        processor), and a key of (data_ptr, _version) alone then served the OLD prompt's cast."""
        casts = self.__dict__.setdefault("_bank_casts", [])
        for src, ver, dt, cast in casts:
            if src is bank and ver == bank._version and dt == self.weights_dtype:
                return cast
        # every live cast stays alive: a captured hipGraph replays against the address of the one it was captured with.
        # When the pool overflows, everything that was derived from the evicted tensors goes with them: the graphs (they would
        # replay against freed memory), the per-layer K / V^T projections of the banks and the per-net grouped gathers (keyed by
        # the CAST's address, which the allocator may hand to the next cast).
        if len(casts) >= self.BANK_CAST_POOL:
            casts.clear()
            if hasattr(self, "_graphs"):
                self._graphs.clear()
            self._drop_bank_projections()
        cast = bank.to(self.weights_dtype)     # (already in the nets' dtype: .to() returns the tensor itself)
        casts.append((bank, bank._version, self.weights_dtype, cast))
        return cast

    def _drop_bank_projections(self):
        from .sd.layers import Attention
        for net in [self.unet] + list(self.controlnets):
            for m in net.modules():
                if isinstance(m, Attention):
                    m.__dict__.pop("_kv_banks", None)
            pro = net.__dict__.get("_net_prologue")
            if pro is not None:
                pro._kv.clear()

    def _noise_pred(self, latents_noisy, t, text_embeddings, image_cond, condition_scales, bank_ids=None, n_branch=3):
        ctx = PaddedContext(text_embeddings.to(self.weights_dtype), *(bank_ids or (None, None)))
        latent_model_input = torch.cat([latents_noisy] * n_branch, dim=0)
        t3 = torch.cat([t] * n_branch)
        if self.use_controlnet and not all(s == 0 for s in condition_scales):
            # the reference relies on a [3]+[1] broadcast here (B=1 only); the ControlNet tiles the
            # conditioning EMBEDDING over the three branches (branch-major, like torch.cat([x]*3))
            down, mid = self.multi_control_forward(latent_model_input, t3, ctx, image_cond, condition_scales)
            return self.forward_unet(latent_model_input, t3, ctx, down, mid)
        return self.forward_unet(latent_model_input, t3, ctx)

    # ---------------------------------------------------------------- hipGraph replay of the noise prediction
    GRAPH_AUTO_MAX_VIEWS = 2
    def _graph_ok(self, latents):
        want = self.cfg.hip_graph
        if want == "auto":
            want = latents.shape[0] <= self.GRAPH_AUTO_MAX_VIEWS
        nets_frozen = not any(p.requires_grad for p in self.unet.parameters())
        return bool(want) and latents.is_cuda and nets_frozen and not torch.cuda.is_current_stream_capturing()

    def _noise_pred_graphed(self, latents_noisy, t, text_embeddings, image_cond, condition_scales, bank_ids=None):
        """Same arithmetic as `_noise_pred`, captured once per (shapes, conditioning scales) and replayed: inputs are copied
        into the capture's static buffers (the 22-channel condition maps are converted to the nets' dtype by that copy),
        the output is cloned out of the graph's pool."""
        scales = tuple(float(s) for s in condition_scales)
        key = (tuple(latents_noisy.shape), latents_noisy.dtype, tuple(text_embeddings.shape), scales,
               tuple((tuple(c.shape), tuple(c.stride())) for c in image_cond),
               None if bank_ids is None else (bank_ids[0].data_ptr(), bank_ids[0]._version))
        g = self._graphs.get(key) if hasattr(self, "_graphs") else None
        if g is None:
            if not hasattr(self, "_graphs"):
                self._graphs = {}
            st = {"lat": torch.empty_like(latents_noisy), "t": torch.empty_like(t),
                  "emb": torch.empty_like(text_embeddings, dtype=self.weights_dtype),
                  "cond": [torch.empty_strided(c.shape, c.stride(), dtype=self.weights_dtype, device=c.device) for c in image_cond]}
            if bank_ids is not None:        # the bank tensor itself is static (one object for the run); the row ids are an input
                st["bank"], st["ids"] = bank_ids[0], torch.empty_like(bank_ids[1])

            def load():
                st["lat"].copy_(latents_noisy); st["t"].copy_(t); st["emb"].copy_(text_embeddings)
                for d, c in zip(st["cond"], image_cond):
                    d.copy_(c)
                if bank_ids is not None:
                    st["ids"].copy_(bank_ids[1])
            load()
            sb = (st["bank"], st["ids"]) if bank_ids is not None else None
            # eager warm-up on a side stream (first-use initialisation inside the library, allocator warm-up), then capture
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                # twice: the first run builds the per-layer K / V^T bank entries, the second the per-net grouped caches of
                # layers.NetPrologue -- nothing that outlives the capture may be first allocated inside it
                self._noise_pred(st["lat"], st["t"], st["emb"], st["cond"], scales, sb)
                self._noise_pred(st["lat"], st["t"], st["emb"], st["cond"], scales, sb)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            # thread-local capture mode: the RCCL watchdog thread of a multi-rank job polls its events with HIP calls of its own,
            # which a global-mode capture would reject
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                st["out"] = self._noise_pred(st["lat"], st["t"], st["emb"], st["cond"], scales, sb)
            g = self._graphs[key] = (graph, st)
            if len(self._graphs) > 4:       # annealed conditioning scales re-capture; keep the pool bounded
                self._graphs.pop(next(iter(self._graphs)))
        graph, st = g
        st["lat"].copy_(latents_noisy); st["t"].copy_(t); st["emb"].copy_(text_embeddings)
        for d, c in zip(st["cond"], image_cond):
            d.copy_(c)
        if "ids" in st:
            if bank_ids is None or bank_ids[0].data_ptr() != st["bank"].data_ptr():
                raise RuntimeError("the captured noise prediction gathers from a prompt-embedding bank that has been replaced")
            st["ids"].copy_(bank_ids[1])
        graph.replay()
        return st["out"].clone()

    def compute_grad_sds(self, prompt_utils, condition_scales, latents, image_cond, elevation, azimuth,
                         camera_distances, rng=None):
        B = latents.shape[0]
        rng = rng or {}
        t = rng.get("t")
        if t is None:
            t = torch.randint(self.min_step, self.max_step + 1, [B], dtype=torch.long, device=latents.device)
        noise = rng.get("noise")
        if noise is None:
            noise = torch.randn_like(latents)
        latents_noisy = self.scheduler.add_noise(latents, noise, t)
        e_perpneg = None
        if getattr(prompt_utils, "use_perp_neg", False):
            e_text, e_uncond, e_null, e_perpneg = self.compute_with_perpneg(condition_scales, prompt_utils, latents_noisy, t,
                                                                            elevation, azimuth, camera_distances, image_cond)
        else:
            e_text, e_uncond, e_null = self.compute_without_perpneg(condition_scales, prompt_utils, latents_noisy, t,
                                                                    elevation, azimuth, camera_distances, image_cond)
        w = (1 - self.alphas[t]).view(-1, 1, 1, 1)
        grad = w * (self.cond_scale * e_text + self.uncond_scale * e_uncond + self.null_scale * e_null
                    + self.noise_scale * noise)
        if e_perpneg is not None:
            grad = grad + w * self.perpneg_scale * e_perpneg
        ev = {"uncond_m_noise_norm": (e_uncond - noise).norm(), "text_m_noise_norm": (e_text - noise).norm(),
              "text_m_uncond_norm": (e_text - e_uncond).norm(), "text_m_null_norm": (e_text - e_null).norm(),
              "null_m_uncond_norm": (e_null - e_uncond).norm(), "noise_norm": noise.norm(),
              "uncond_norm": e_uncond.norm(), "text_norm": e_text.norm()}
        self._last = {"t": t, "noise": noise, "e_text": e_text, "e_uncond": e_uncond, "e_null": e_null}
        return grad, ev

    def __call__(self, rgb, prompt_utils, elevation, azimuth, camera_distances, env_id=None,
                 rgb_as_latents=False, rng=None, **kwargs):
        batch_size = rgb.shape[0]
        rng = rng or {}
        latents = self.get_latents(rgb.permute(0, 3, 1, 2), rgb_as_latents, rng.get("posterior_noise"))
        image_cond, condition_scales = [], []
        if self.use_controlnet:
            for k, ct in enumerate(self.cfg.control_types):
                src = {"depth": kwargs.get("cond_depth"), "normal": kwargs.get("cond_normal"),
                       "light": kwargs.get("condition_map")}.get(ct, kwargs.get("cond_rgb"))
                image_cond.append(self.prepare_image_cond(ct, src))
                condition_scales.append(self.condition_scales[k])
        else:
            condition_scales = [0]
        grad, ev = self.compute_grad_sds(prompt_utils, condition_scales, latents, image_cond, elevation, azimuth,
                                         camera_distances, rng)
        grad = torch.nan_to_num(grad)
        if self.cfg.grad_clip_val is not None:
            grad = grad.clamp(-self.cfg.grad_clip_val, self.cfg.grad_clip_val)
        if self.cfg.grad_normalize:
            grad = grad / (grad.norm(2) + 1e-8)
        target = (latents - grad).detach()
        loss_sds = 0.5 * F.mse_loss(latents, target, reduction="sum") / batch_size
        out = {"loss_sds": loss_sds, "grad_norm": grad.norm()}
        out.update(ev)
        self._last["latents"] = latents
        self._last["grad"] = grad
        return out

    # ---------------------------------------------------------------- schedules (:603-626)
    def set_min_max_steps(self, min_step_percent=0.02, max_step_percent=0.98):
        self.min_step = int(self.num_train_timesteps * min_step_percent)
        self.max_step = int(self.num_train_timesteps * max_step_percent)

    def update_step(self, epoch: int, global_step: int, on_load_weights: bool = False):
        self.noise_scale = C(self.cfg.noise_scale, epoch, global_step)
        self.cond_scale = C(self.cfg.cond_scale, epoch, global_step)
        self.uncond_scale = C(self.cfg.uncond_scale, epoch, global_step)
        self.null_scale = C(self.cfg.null_scale, epoch, global_step)
        self.perpneg_scale = C(self.cfg.perpneg_scale, epoch, global_step)
        self.set_min_max_steps(min_step_percent=C(self.cfg.min_step_percent, epoch, global_step),
                               max_step_percent=C(self.cfg.max_step_percent, epoch, global_step))
        if (self.use_controlnet and self.cfg.control_anneal_start_step is not None
                and global_step > self.cfg.control_anneal_start_step):
            self.condition_scales = list(self.cfg.condition_scales_anneal)
