"""`dreammat-system` (threestudio/systems/dreammat.py:19-300 on top of systems/base.py:21-394) and the
training loop that replaces Lightning's Trainer for this path (launch.py:172-193).

One process per GPU.  Per optimizer step:
  update_step schedules -> collate (views of this rank) -> renderer -> guidance -> loss ->
  backward (VAE-enc bwd -> antialias bwd -> shade bwd -> MLP bwd -> hash-grid scatter) ->
  ONE all-reduce (RCCL over xGMI) of the flat fp32 gradient buffer -> fused Adam (mean folded in); or, with
  `optimizer.sharded: true`, reduce-scatter -> Adam on this rank's slice -> all-gather of the parameters.
All trainable parameters (hash table + 2 MLP matrices) live in one flat, 16 B-aligned buffer whose
slices back the module parameters, so the collective and the optimizer each touch memory once.
"""
import json
import os
import time
from dataclasses import dataclass, field
from typing import Any, Dict, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

import dreammat_amd
from . import hipops
from .base import BaseModule, Updateable, get_device, get_rank
from .config import C, parse_structured


class FlatParams:
    """Re-homes the given parameters into one flat fp32 buffer (+ matching flat grad buffer)."""

    def __init__(self, params, pad_to=4):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        n_pad = (n + pad_to - 1) // pad_to * pad_to          # 16-byte vectors; 4 x world for the sharded optimizer
        dev = self.params[0].device
        self.flat = torch.zeros(n_pad, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n_pad, device=dev, dtype=torch.float32)
        self.numel = n
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.grad[off:off + k].view_as(p)
            off += k


def allreduce_sum_(flat_grad):
    """The step's only data-path collective: SUM over ranks of the flat gradient (mean is folded into
    the Adam kernel as grad_scale = 1/world)."""
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        return dist.get_world_size()
    return 1


def sync_parameters_(flat):
    """Replica start-up: every rank adopts rank 0's parameters (what Lightning's DDP wrapper does when it wraps the
    module, threestudio_dreammat/launch.py:172-179).  Construction is already seeded rank-independently
    (launch.seed_for_build), so this is a guard against any non-deterministic initialiser, not the mechanism."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=0)
    return flat


def replicas_in_sync(flat):
    """True when every rank holds bit-identical parameters (one 2-float all-reduce; used by tests and by
    DREAMMAT_CHECK_REPLICAS=1 every checkpoint interval)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return True
    c = flat.double()
    sig = torch.stack([c.sum(), (c * torch.arange(1, c.numel() + 1, device=c.device, dtype=c.dtype)).sum()])
    lo, hi = sig.clone(), sig.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool(torch.equal(lo, hi))


class FusedAdam:
    """torch.optim.Adam(lr, betas, eps) semantics (systems/utils.py:34-53; dreammat.yaml:110-115)."""

    def __init__(self, fp: FlatParams, lr=0.01, betas=(0.9, 0.99), eps=1e-15):
        self.fp, self.lr, self.betas, self.eps = fp, lr, tuple(betas), eps
        self.exp_avg = torch.zeros_like(fp.flat)
        self.exp_avg_sq = torch.zeros_like(fp.flat)
        self.step_count = 0

    def step(self, world=1):
        self.step_count += 1
        hipops.adam_step(self.fp.flat, self.fp.grad, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr,
                         self.betas[0], self.betas[1], self.eps, grad_scale=1.0 / world, zero_grad=True)

    def sync_and_step(self):
        """the step's data-path collective + the update: ONE all-reduce (SUM) of the flat gradient, mean folded into Adam"""
        self.step(allreduce_sum_(self.fp.grad))

    def state_dict(self):
        """moments WITHOUT the alignment padding (fp.numel entries): the same file loads under either optimizer at any world
        size, whatever each pads the flat buffer to"""
        n = self.fp.numel
        return {"exp_avg": self.exp_avg[:n], "exp_avg_sq": self.exp_avg_sq[:n], "step": self.step_count}

    def load_state_dict(self, sd):
        _copy_padded(self.exp_avg, sd["exp_avg"], 0); _copy_padded(self.exp_avg_sq, sd["exp_avg_sq"], 0)
        self.step_count = sd["step"]


def _copy_padded(dst, src_full, lo):
    """dst[:] = src_full[lo : lo + len(dst)], zeros where src_full (stored without padding, or with another padding) ends"""
    k = max(0, min(dst.numel(), src_full.numel() - lo))
    dst.zero_()
    if k:
        dst[:k].copy_(src_full[lo:lo + k])


class ShardedFusedAdam:
    """`optimizer.sharded: true` -- SURVEY 8(e)'s preferred exchange: reduce-scatter of the flat gradient, Adam on this
    rank's 1/world slice (moments held for that slice only), all-gather of the updated parameters.  Same bytes on the wire as
    the all-reduce (each of the 7 xGMI links carries 1/8 of the buffer per phase), 1/world of the optimizer's work and state.
    Same result as FusedAdam up to the summation order of the collective (identical at world 2).  `adam_fn` = the update of
    one contiguous slice (the HIP kernel; the CPU tests pass a torch restatement)."""

    def __init__(self, fp: FlatParams, lr=0.01, betas=(0.9, 0.99), eps=1e-15, adam_fn=None):
        on = dist.is_available() and dist.is_initialized()
        self.world, self.rank = (dist.get_world_size(), dist.get_rank()) if on else (1, 0)
        self.collective = on                 # a process group of one rank still goes through RCCL (bench.py DREAMMAT_FORCE_DIST)
        n = fp.flat.numel()
        assert n % (4 * self.world) == 0, "FlatParams(pad_to=4 * world) is required for the sharded optimizer"
        self.fp, self.lr, self.betas, self.eps = fp, lr, tuple(betas), eps
        self.adam_fn = adam_fn or hipops.adam_step
        self.shard = n // self.world
        self.lo = self.rank * self.shard
        self.p_shard = fp.flat[self.lo:self.lo + self.shard].clone()      # this rank's master copy of its slice
        self.g_shard = torch.zeros_like(self.p_shard)
        self.exp_avg = torch.zeros_like(self.p_shard)
        self.exp_avg_sq = torch.zeros_like(self.p_shard)
        self.step_count = 0

    def sync_and_step(self):
        self.step_count += 1
        if self.collective:
            dist.reduce_scatter_tensor(self.g_shard, self.fp.grad, op=dist.ReduceOp.SUM)
        else:
            self.g_shard.copy_(self.fp.grad)
        self.fp.grad.zero_()                                              # autograd accumulates into it next step
        self.adam_fn(self.p_shard, self.g_shard, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr, self.betas[0],
                     self.betas[1], self.eps, grad_scale=1.0 / self.world, zero_grad=True)
        if self.collective:
            dist.all_gather_into_tensor(self.fp.flat, self.p_shard)
        else:
            self.fp.flat.copy_(self.p_shard)

    def _gathered(self, t):
        if not self.collective:
            return t.clone()
        full = torch.empty(self.shard * self.world, device=t.device, dtype=t.dtype)
        dist.all_gather_into_tensor(full, t)
        return full

    def state_dict(self):
        """the FULL moments, gathered and cut to fp.numel entries (same layout as FusedAdam's: a checkpoint resumes under
        either optimizer at ANY world size -- the padding to 4 x world is never stored) -- a collective: every rank calls it"""
        n = self.fp.numel
        return {"exp_avg": self._gathered(self.exp_avg)[:n], "exp_avg_sq": self._gathered(self.exp_avg_sq)[:n],
                "step": self.step_count}

    def load_state_dict(self, sd):
        _copy_padded(self.exp_avg, sd["exp_avg"], self.lo); _copy_padded(self.exp_avg_sq, sd["exp_avg_sq"], self.lo)
        self.step_count = sd["step"]
        self.p_shard.copy_(self.fp.flat[self.lo:self.lo + self.shard])    # the parameters were restored just before


@dreammat_amd.register("dreammat-system")
class DreamMat(nn.Module, Updateable):
    @dataclass
    class Config:
        loggers: dict = field(default_factory=dict)
        loss: dict = field(default_factory=dict)
        optimizer: dict = field(default_factory=dict)
        scheduler: Optional[dict] = None
        weights: Optional[str] = None
        weights_ignore_modules: Optional[list] = None
        cleanup_after_validation_step: bool = False
        cleanup_after_test_step: bool = False
        geometry_type: str = "dreammat-mesh"
        geometry: dict = field(default_factory=dict)
        geometry_convert_from: Optional[str] = None
        geometry_convert_inherit_texture: bool = False
        geometry_convert_override: dict = field(default_factory=dict)
        material_type: str = "dreammat-material"
        material: dict = field(default_factory=dict)
        background_type: str = "solid-color-background"
        background: dict = field(default_factory=dict)
        renderer_type: str = "raytracing-renderer"
        renderer: dict = field(default_factory=dict)
        guidance_type: str = "stable-diffusion-dreammat-guidance"
        guidance: dict = field(default_factory=dict)
        prompt_processor_type: str = "stable-diffusion-prompt-processor"
        prompt_processor: dict = field(default_factory=dict)
        exporter_type: str = "mesh-exporter"
        exporter: dict = field(default_factory=dict)
        # DreamMat.Config (systems/dreammat.py:21-31)
        texture: bool = True
        latent_steps: int = 1000
        save_train_image: bool = True
        save_train_image_iter: int = 1
        init_step: int = 0
        init_width: int = 512
        init_height: int = 512
        test_background_white: Optional[bool] = False

    def __init__(self, cfg, resumed=False, material_kwargs=None):
        super().__init__()
        self.cfg = parse_structured(self.Config, cfg)
        self.device_ = get_device() if torch.cuda.is_available() else torch.device("cpu")
        self.true_global_step = 0
        self.true_current_epoch = 0
        self._save_dir = None
        # systems/base.py:284-295: geometry -> material -> background -> renderer
        self.geometry = dreammat_amd.find(self.cfg.geometry_type)(self.cfg.geometry)
        self.material = dreammat_amd.find(self.cfg.material_type)(self.cfg.material, **(material_kwargs or {}))
        self.background = dreammat_amd.find(self.cfg.background_type)(self.cfg.background)
        self.to(self.device_)
        self.renderer = dreammat_amd.find(self.cfg.renderer_type)(self.cfg.renderer, geometry=self.geometry,
                                                                 material=self.material, background=self.background)
        self.guidance = None
        self.prompt_processor = None

    def C(self, value):
        return C(value, self.true_current_epoch, self.true_global_step)

    def on_fit_start(self):
        """systems/dreammat.py:44-50: prompt processor + guidance are only built for training."""
        self.prompt_processor = dreammat_amd.find(self.cfg.prompt_processor_type)(self.cfg.prompt_processor)
        self.guidance = dreammat_amd.find(self.cfg.guidance_type)(self.cfg.guidance)

    def configure_optimizers(self):
        """systems/base.py:95-106 -> parse_optimizer: Adam over system.parameters()."""
        opt = self.cfg.optimizer or {"name": "Adam", "args": {"lr": 0.01, "betas": [0.9, 0.99], "eps": 1e-15}}
        if opt.get("name", "Adam") != "Adam":
            raise NotImplementedError("only the Adam of dreammat.yaml:110-115 is implemented (fused HIP kernel)")
        sharded = bool(opt.get("sharded", False))
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.flat = FlatParams(list(self.parameters()), pad_to=4 * world if sharded else 4)
        sync_parameters_(self.flat.flat)
        args = opt.get("args", {})
        cls = ShardedFusedAdam if sharded else FusedAdam
        self.optimizer = cls(self.flat, lr=args.get("lr", 0.01), betas=args.get("betas", (0.9, 0.99)), eps=args.get("eps", 1e-15))
        return self.optimizer

    def forward(self, batch):
        return self.renderer(**batch, render_rgb=self.cfg.texture)

    def do_update(self):
        """systems/base.py:174-178 on_train_batch_start -> recursive update_step."""
        for m in (self.geometry, self.material, self.background, self.renderer, self.guidance, self.prompt_processor):
            if isinstance(m, Updateable):
                m.do_update_step(self.true_current_epoch, self.true_global_step)

    def training_step(self, batch, rng=None):
        """systems/dreammat.py:57-86.  Returns (loss, logged scalars dict)."""
        prompt_utils = self.prompt_processor()
        out = self(batch)
        batch = dict(batch)
        batch["cond_normal"] = out.get("comp_normal")
        batch["cond_depth"] = out.get("comp_depth")
        guidance_out = self.guidance(out["comp_rgb"], prompt_utils, **batch, rgb_as_latents=False, rng=rng)
        loss = 0.0
        logs = {}
        for name, value in guidance_out.items():
            logs[f"train/{name}"] = value
            if name.startswith("loss_"):
                loss = loss + value * self.C(self.cfg.loss[name.replace("loss_", "lambda_")])
        for name, value in out.items():
            if name.startswith("loss_"):
                logs[f"train/{name}"] = value
                loss = loss + value * self.C(self.cfg.loss[name.replace("loss_", "lambda_")])
        self._last_out = out
        return loss, logs


def to_device(batch, device):
    return {k: (v.to(device, non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


class Trainer:
    """fit / validate / test loop (launch.py:172-193) with checkpoint/resume
    (every_n_train_steps, dreammat.yaml:125-128) and rank-0 CSV metrics + PNG grids."""

    def __init__(self, system: DreamMat, datamodule, max_steps=30000, trial_dir="outputs/dream_mat/run",
                 val_check_interval=100, checkpoint_every=3999, log_every=1, resume=None, seed=0):
        self.system, self.dm = system, datamodule
        self.seed = seed
        self.max_steps, self.trial_dir = max_steps, trial_dir
        self.val_check_interval, self.checkpoint_every, self.log_every = val_check_interval, checkpoint_every, log_every
        self.resume = resume
        self.rank = get_rank()
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def save_checkpoint(self, path, optimizer_state=None):
        s = self.system
        torch.save({"state_dict": {k: v for k, v in s.state_dict().items()},
                    "optimizer": optimizer_state if optimizer_state is not None else s.optimizer.state_dict(),
                    "global_step": s.true_global_step,
                    "epoch": s.true_current_epoch}, path)

    def load_checkpoint(self, path):
        s = self.system
        ck = torch.load(path, map_location=s.device_)
        # the same rule as `weights:` loading (base.BaseObject.check_state_dict_match): missing keys, or unexpected keys outside
        # the reference-only allow-lists, raise -- a resumed run must not silently keep random parameters
        from .base import BaseModule
        res = s.load_state_dict(ck["state_dict"], strict=False)
        BaseModule.check_state_dict_match(f"checkpoint '{path}' does not match {type(s).__name__}", res, own_keys=set(s.state_dict()))
        s.optimizer.load_state_dict(ck["optimizer"])
        s.true_global_step, s.true_current_epoch = ck["global_step"], ck["epoch"]

    def seed_rank_streams(self):
        """threestudio_dreammat/launch.py:102 `pl.seed_everything(cfg.seed + get_rank())`: the parameters are identical
        on every rank (seed_for_build + sync_parameters_), the per-step draws (timestep t, noise, VAE posterior noise,
        tangent jitter; the view draws have their own per-rank generator in data.py) must NOT be."""
        torch.manual_seed(self.seed + self.rank)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(self.seed + self.rank)

    def train_one_step(self, batch=None, rng=None):
        s = self.system
        s.do_update()
        if batch is None:
            batch = to_device(self.dm.train_dataset.collate(), s.device_)
        loss, logs = s.training_step(batch, rng=rng)
        loss.backward()
        if hasattr(s.optimizer, "sync_and_step"):
            s.optimizer.sync_and_step()           # all-reduce + Adam, or reduce-scatter + sharded Adam + all-gather
        else:
            s.optimizer.step(allreduce_sum_(s.flat.grad))
        s.true_global_step += 1
        return loss, logs

    def fit(self):
        s = self.system
        self.dm.setup("fit")
        s.on_fit_start()
        s.configure_optimizers()
        if self.resume:
            self.load_checkpoint(self.resume)
        self.seed_rank_streams()
        if self.rank == 0:
            os.makedirs(os.path.join(self.trial_dir, "ckpts"), exist_ok=True)
            self.csv = open(os.path.join(self.trial_dir, "metrics.csv"), "a")
        t0 = time.time()
        while s.true_global_step < self.max_steps:
            loss, logs = self.train_one_step()
            step = s.true_global_step
            if self.rank == 0 and step % self.log_every == 0:
                row = {"step": step, "loss": float(loss), "time": time.time() - t0}
                row.update({k: float(v) for k, v in logs.items()})
                self.csv.write(json.dumps(row) + "\n")
                self.csv.flush()
            if self.checkpoint_every and step % self.checkpoint_every == 0:
                opt_state = s.optimizer.state_dict()          # gathers the moments under the sharded optimizer: every rank calls it
                if self.rank == 0:
                    self.save_checkpoint(os.path.join(self.trial_dir, "ckpts", f"step={step}.ckpt"), opt_state)
            if self.rank == 0 and self.val_check_interval and step % self.val_check_interval == 0:
                self.validate()
            if self.checkpoint_every and step % self.checkpoint_every == 0 and os.environ.get("DREAMMAT_CHECK_REPLICAS"):
                assert replicas_in_sync(s.flat.flat), f"replicas diverged by step {step}"
        return s

    @torch.no_grad()
    def validate(self):
        from .saving import save_image_grid
        s = self.system
        self.dm.setup("validate")
        for i in range(len(self.dm.val_dataset)):
            out = s(to_device(self.dm.val_dataset[i], s.device_))
            save_image_grid(os.path.join(self.trial_dir, "save", f"it{s.true_global_step}-{i}.png"),
                            [out["comp_rgb"][0], out["albedo"][0], out["roughness"][0].expand(-1, -1, 3),
                             out["metalness"][0].expand(-1, -1, 3), out["comp_normal"][0]])

    @torch.no_grad()
    def export(self):
        """systems/base.py:309-334 (`--export`): run the configured exporter and write its outputs under
        <trial>/save/it<N>-export/."""
        from .saving import save_obj
        s = self.system
        exporter = dreammat_amd.find(s.cfg.exporter_type)(s.cfg.exporter, geometry=s.geometry, material=s.material,
                                                            background=s.background)
        base = os.path.join(self.trial_dir, "save", f"it{s.true_global_step}-export")
        written = []
        for out in exporter():
            if out.save_type != "obj":
                raise ValueError(f"{out.save_type} export not supported yet.")
            written += save_obj(os.path.join(base, out.save_name), **out.params)
        return written

    @torch.no_grad()
    def test(self):
        """systems/dreammat.py:247-300: per view the five-panel grid `view/<i>.png` (render | normal | albedo | metalness |
        roughness) and the albedo / roughness / metallic / render RGBA PNGs; then `view/eval.gif` (on_test_epoch_end, 30 fps)."""
        from .saving import save_gif, save_image_grid, save_rgba
        s = self.system
        self.dm.setup("test")
        base = os.path.join(self.trial_dir, "save", f"it{s.true_global_step}-test")
        n = len(self.dm.test_dataset)
        for i in range(n):
            out = s(to_device(self.dm.test_dataset[i], s.device_))
            a = out["opacity"][0]
            save_image_grid(os.path.join(base, "view", f"{i}.png"),
                            [out["comp_rgb"][0], out["comp_normal"][0], out["albedo"][0], out["metalness"][0].expand(-1, -1, 3),
                             out["roughness"][0].expand(-1, -1, 3)])
            for name, key in (("albedo", "albedo"), ("roughness", "roughness"), ("metallic", "metalness"), ("render", "comp_rgb")):
                img = out[key][0]
                if img.shape[-1] == 1:
                    img = img.expand(-1, -1, 3)
                save_rgba(os.path.join(base, name, f"{i}.png"), img, a)
        return save_gif(os.path.join(base, "view"), n_frames=n, fps=30)
