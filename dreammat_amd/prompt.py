"""`stable-diffusion-prompt-processor` plugin (threestudio/models/prompt_processors/base.py:187-543 and
stable_diffusion_prompt_processor.py:15-106): CLIP text-encodes the prompt, the negative prompt and
"" once, with the 4 view-dependent variants (", side view" / ", front view" / ", back view" /
", overhead view", base.py:281-312), caches them by md5 (base.py:19-23,364-412) and hands out
[3B,77,D] (text, negative, null) batches (base.py:52-85).

The CLIP weights are not on this box: when `<model dir>/{tokenizer,text_encoder}` cannot be found the
processor emits deterministic pseudo-embeddings seeded by the md5 of each prompt (synthetic benchmark
mode, announced on stdout) -- shapes, caching and view-dependent selection are unchanged.
"""
import hashlib
import os
from dataclasses import dataclass
from typing import Optional

import torch

import dreammat_amd
from .base import BaseObject, barrier, get_rank
from .sd import arch_for


def shift_azimuth_deg(azimuth):
    return (azimuth + 180) % 360 - 180


def hash_prompt(model: str, prompt: str) -> str:
    return hashlib.md5(f"{model}-{prompt}".encode()).hexdigest()


@dataclass
class DirectionConfig:
    name: str
    prompt: callable
    negative_prompt: callable
    condition: callable


@dataclass
class PromptProcessorOutput:
    text_embeddings: torch.Tensor
    uncond_text_embeddings: torch.Tensor
    null_text_embeddings: torch.Tensor
    text_embeddings_vd: torch.Tensor
    uncond_text_embeddings_vd: torch.Tensor
    directions: list
    direction2idx: dict
    use_perp_neg: bool = False
    banks: dict = None          # persistent (bank tensors live in the prompt processor: one object for the whole run)
    perp_neg_f_sb: tuple = (1, 0.5, -0.606)
    perp_neg_f_fsb: tuple = (1, 0.5, +0.967)
    perp_neg_f_fs: tuple = (4, 0.5, -2.426)
    perp_neg_f_sf: tuple = (4, 0.5, -2.426)

    def get_text_embeddings(self, elevation, azimuth, camera_distances, view_dependent_prompting=True,
                            return_null_text_embeddings=False):
        batch_size = elevation.shape[0]
        if view_dependent_prompting:
            direction_idx = torch.zeros_like(elevation, dtype=torch.long)
            for d in self.directions:
                direction_idx[d.condition(elevation, azimuth, camera_distances)] = self.direction2idx[d.name]
            text = self.text_embeddings_vd[direction_idx]
            uncond = self.uncond_text_embeddings_vd[direction_idx]
        else:
            text = self.text_embeddings.expand(batch_size, -1, -1)
            uncond = self.uncond_text_embeddings.expand(batch_size, -1, -1)
        null = self.null_text_embeddings.expand(batch_size, -1, -1)
        if return_null_text_embeddings:
            return torch.cat([text, uncond, null], dim=0)
        return torch.cat([text, uncond], dim=0)

    def get_text_embedding_bank(self, elevation, azimuth, camera_distances, view_dependent_prompting=True):
        """Same selection as get_text_embeddings(..., return_null_text_embeddings=True), returned as (bank, ids) with
        `bank[ids]` equal to its result: bank = the distinct embeddings (view-dependent text | negative | empty prompt: a
        tensor that is built once and never changes), ids [3B].  The frozen nets project the bank once (sd/layers.py)."""
        batch_size = elevation.shape[0]
        if view_dependent_prompting:
            direction_idx = torch.zeros_like(elevation, dtype=torch.long)
            for d in self.directions:
                direction_idx[d.condition(elevation, azimuth, camera_distances)] = self.direction2idx[d.name]
            nd = self.text_embeddings_vd.shape[0]
            banks = self.banks if self.banks is not None else {}
            if "vd" not in banks:
                banks["vd"] = torch.cat([self.text_embeddings_vd, self.uncond_text_embeddings_vd, self.null_text_embeddings], dim=0)
            ids = torch.cat([direction_idx, nd + direction_idx, torch.full_like(direction_idx, 2 * nd)])
            return banks["vd"], ids
        banks = self.banks if self.banks is not None else {}
        if "plain" not in banks:
            banks["plain"] = torch.cat([self.text_embeddings, self.uncond_text_embeddings, self.null_text_embeddings], dim=0)
        z = torch.zeros(batch_size, dtype=torch.long, device=elevation.device)
        return banks["plain"], torch.cat([z, z + 1, z + 2])


    def get_text_embeddings_perp_neg(self, elevation, azimuth, camera_distances, view_dependent_prompting=True,
                                     return_null_text_embeddings=False):
        """prompt_processors/base.py:87-184 (Perp-Neg, arXiv 2304.04968): per view a positive embedding interpolated between
        the front / side / back prompts by |azimuth|, two "negative" view prompts with weights
        -f(r) = -(a exp(-b r) + c), the view's negative-prompt embedding, optionally the empty prompt.
        -> (text_embeddings [B (pos) + B (uncond) + 2B (neg) (+ B null), 77, D], neg_guidance_weights [B, 2])."""
        assert view_dependent_prompting, "Perp-Neg only works with view-dependent prompting"
        batch_size = elevation.shape[0]
        direction_idx = torch.zeros_like(elevation, dtype=torch.long)
        for d in self.directions:
            direction_idx[d.condition(elevation, azimuth, camera_distances)] = self.direction2idx[d.name]
        side, front, back, overhead = (self.text_embeddings_vd[i] for i in range(4))
        decay = lambda abc, r: abc[0] * torch.exp(-abc[1] * r) + abc[2]
        pos, neg, wts, unc = [], [], [], []
        for idx, azi in zip(direction_idx, azimuth):
            azi = shift_azimuth_deg(azi)
            unc.append(self.uncond_text_embeddings_vd[idx])
            if idx.item() == 3:                                   # overhead: no interpolation, dummy negatives with weight 0
                pos.append(overhead)
                neg += [self.uncond_text_embeddings_vd[idx], self.uncond_text_embeddings_vd[idx]]
                wts += [0.0, 0.0]
            elif torch.abs(azi) < 90:                             # front <-> side: r = 1 at the front
                r = 1 - torch.abs(azi) / 90
                pos.append(r * front + (1 - r) * side)
                neg += [front, side]
                wts += [-decay(self.perp_neg_f_fs, r), -decay(self.perp_neg_f_sf, 1 - r)]
            else:                                                 # side <-> back: r = 1 at the side
                r = 2.0 - torch.abs(azi) / 90
                pos.append(r * side + (1 - r) * back)
                neg += [side, front]
                wts += [-decay(self.perp_neg_f_sb, r), -decay(self.perp_neg_f_fsb, r)]
        parts = [torch.stack(pos, dim=0), torch.stack(unc, dim=0), torch.stack(neg, dim=0)]
        if return_null_text_embeddings:
            parts.append(self.null_text_embeddings.expand(batch_size, -1, -1))
        return torch.cat(parts, dim=0), torch.as_tensor(wts, device=elevation.device).reshape(batch_size, 2)


def perpendicular_component(x, y):
    """utils/ops.py:431-441: the component of x [B,C,H,W] perpendicular to y, per batch item"""
    eps = torch.ones_like(x[:, 0, 0, 0]) * 1e-6
    return x - (torch.mul(x, y).sum(dim=[1, 2, 3]) / torch.maximum(torch.mul(y, y).sum(dim=[1, 2, 3]), eps)).view(-1, 1, 1, 1) * y


@dreammat_amd.register("stable-diffusion-prompt-processor")
class StableDiffusionPromptProcessor(BaseObject):
    @dataclass
    class Config(BaseObject.Config):
        prompt: str = "a hamburger"
        prompt_front: Optional[str] = None
        prompt_side: Optional[str] = None
        prompt_back: Optional[str] = None
        prompt_overhead: Optional[str] = None
        negative_prompt: str = ""
        pretrained_model_name_or_path: str = "runwayml/stable-diffusion-v1-5"
        pretrained_model_cache_dir: str = "../../model/SD"
        overhead_threshold: float = 60.0
        front_threshold: float = 45.0
        back_threshold: float = 45.0
        view_dependent_prompt_front: bool = False
        use_cache: bool = True
        spawn: bool = True
        use_perp_neg: bool = False
        # a exp(-b r) + c (prompt_processors/base.py:214-224)
        perp_neg_f_sb: tuple = (1, 0.5, -0.606)
        perp_neg_f_fsb: tuple = (1, 0.5, +0.967)
        perp_neg_f_fs: tuple = (4, 0.5, -2.426)
        perp_neg_f_sf: tuple = (4, 0.5, -2.426)
        cache_dir: str = ".threestudio_cache/text_embeddings"
        # addition: md5-seeded pseudo embeddings when the CLIP text encoder is not on disk.  Off by default: a run with
        # a real model name and no encoder is an error, not a silent stand-in ('tiny*' architectures are synthetic by
        # definition).  Synthetic embeddings are cached under their own suffix and never read back as real ones.
        synthetic: bool = False

    cfg: Config

    def configure(self) -> None:
        if not torch.cuda.is_available():
            self.device = torch.device("cpu")
        c = self.cfg
        if c.view_dependent_prompt_front:
            fmt = lambda v: (lambda s: f"{v} view of {s}")
        else:
            fmt = lambda v: (lambda s: f"{s}, {v} view")
        ident = lambda s: s
        self.directions = [
            DirectionConfig("side", fmt("side"), ident, lambda ele, azi, dis: torch.ones_like(ele, dtype=torch.bool)),
            DirectionConfig("front", fmt("front"), ident,
                            lambda ele, azi, dis: (shift_azimuth_deg(azi) > -c.front_threshold) & (shift_azimuth_deg(azi) < c.front_threshold)),
            DirectionConfig("back", fmt("back"), ident,
                            lambda ele, azi, dis: (shift_azimuth_deg(azi) > 180 - c.back_threshold) | (shift_azimuth_deg(azi) < -180 + c.back_threshold)),
            DirectionConfig("overhead", fmt("overhead"), ident, lambda ele, azi, dis: ele > c.overhead_threshold),
        ]
        self.direction2idx = {d.name: i for i, d in enumerate(self.directions)}
        self.prompt = c.prompt
        self.negative_prompt = c.negative_prompt
        manual = {"front": c.prompt_front, "side": c.prompt_side, "back": c.prompt_back, "overhead": c.prompt_overhead}
        self.prompts_vd = [manual[d.name] or d.prompt(self.prompt) for d in self.directions]
        self.negative_prompts_vd = [d.negative_prompt(self.negative_prompt) for d in self.directions]
        self.embed_dim = arch_for(c.pretrained_model_name_or_path).cross_dim
        self.prepare_text_embeddings()
        self.load_text_embeddings()

    # ---------------------------------------------------------------- encode + cache
    def _model_dir(self):
        name = self.cfg.pretrained_model_name_or_path
        for cand in (name, os.path.join(self.cfg.pretrained_model_cache_dir, name),
                     os.environ.get("DREAMMAT_SD_DIR") or ""):
            if cand and os.path.isdir(os.path.join(cand, "text_encoder")):
                return cand
        return None

    def _encode(self, prompts, keep_encoder=False):
        """keep_encoder: leave tokenizer + CLIPTextModel loaded on the processor for later calls (the ControlNet training launcher
        encodes unseen prompts batch by batch; the SDS path encodes once at start-up and frees the encoder)"""
        root = self._model_dir()
        if root is None:
            if not self._synthetic_allowed():
                raise FileNotFoundError(
                    f"CLIP text encoder for '{self.cfg.pretrained_model_name_or_path}' not found under "
                    f"'{self.cfg.pretrained_model_cache_dir}' or $DREAMMAT_SD_DIR; set prompt_processor.synthetic=true to "
                    f"run on md5-seeded pseudo embeddings (benchmark / test mode)")
            print("[dreammat_amd] CLIP text encoder not found: using md5-seeded pseudo text embeddings (synthetic mode)")
            outs = []
            for p in prompts:
                g = torch.Generator().manual_seed(int(hash_prompt(self.cfg.pretrained_model_name_or_path, p)[:8], 16))
                outs.append(torch.randn(77, self.embed_dim, generator=g))
            return torch.stack(outs)
        tok, enc = self.__dict__.get("_text_encoder", (None, None))
        if enc is None:
            from transformers import AutoTokenizer, CLIPTextModel
            tok = AutoTokenizer.from_pretrained(os.path.join(root, "tokenizer"))
            enc = CLIPTextModel.from_pretrained(os.path.join(root, "text_encoder")).to(self.device)
            if keep_encoder:
                self.__dict__["_text_encoder"] = (tok, enc)
        with torch.no_grad():
            ids = tok(prompts, padding="max_length", max_length=tok.model_max_length, return_tensors="pt")
            emb = enc(ids.input_ids.to(self.device))[0]
        del enc
        return emb.float().cpu()

    def _synthetic_allowed(self):
        return bool(self.cfg.synthetic) or self.cfg.pretrained_model_name_or_path.lower().startswith("tiny")

    def _cache_path(self, p):
        suffix = ".pt" if self._model_dir() is not None else ".synthetic.pt"
        return os.path.join(self.cfg.cache_dir, hash_prompt(self.cfg.pretrained_model_name_or_path, p) + suffix)

    def prepare_text_embeddings(self):
        os.makedirs(self.cfg.cache_dir, exist_ok=True)
        all_prompts = [self.prompt, self.negative_prompt, ""] + self.prompts_vd + self.negative_prompts_vd
        todo = []
        for p in dict.fromkeys(all_prompts):
            path = self._cache_path(p)
            if not (self.cfg.use_cache and os.path.exists(path)):
                todo.append(p)
        if todo and get_rank() == 0:
            emb = self._encode(todo)
            for p, e in zip(todo, emb):
                torch.save(e, self._cache_path(p))
        barrier()   # other ranks wait for rank 0's cache (base.py:416)

    def _load(self, p):
        return torch.load(self._cache_path(p), map_location=self.device, weights_only=True)     # a cache file is data, never code

    def load_text_embeddings(self):
        self.text_embeddings = self._load(self.prompt)[None]
        self.uncond_text_embeddings = self._load(self.negative_prompt)[None]
        self.null_text_embeddings = self._load("")[None]
        self.text_embeddings_vd = torch.stack([self._load(p) for p in self.prompts_vd])
        self.uncond_text_embeddings_vd = torch.stack([self._load(p) for p in self.negative_prompts_vd])

    def __call__(self) -> PromptProcessorOutput:
        return PromptProcessorOutput(self.text_embeddings, self.uncond_text_embeddings, self.null_text_embeddings,
                                     self.text_embeddings_vd, self.uncond_text_embeddings_vd, self.directions,
                                     self.direction2idx, bool(self.cfg.use_perp_neg), self.__dict__.setdefault("_banks", {}),
                                     tuple(self.cfg.perp_neg_f_sb), tuple(self.cfg.perp_neg_f_fsb),
                                     tuple(self.cfg.perp_neg_f_fs), tuple(self.cfg.perp_neg_f_sf))
