"""Plugin base classes: threestudio/utils/base.py:21-118 (Updateable, BaseObject, BaseModule) and
threestudio/utils/misc.py:28-29 (get_device) / :104-120 (barrier, broadcast)."""
import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from .config import parse_structured


def get_rank():
    for k in ("RANK", "LOCAL_RANK", "SLURM_PROCID", "JSM_NAMESPACE_RANK"):
        if k in os.environ:
            return int(os.environ[k])
    return 0


def get_local_rank():
    return int(os.environ.get("LOCAL_RANK", 0))


def get_device():
    """misc.py:28-29 hard-codes cuda:{rank}; one process per GPU here as well."""
    return torch.device(f"cuda:{get_local_rank()}")


def barrier():
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()


class Updateable:
    def do_update_step(self, epoch: int, global_step: int, on_load_weights: bool = False):
        for attr in self.__dir__():
            if attr.startswith("_"):
                continue
            try:
                module = getattr(self, attr)
            except Exception:
                continue
            if isinstance(module, Updateable):
                module.do_update_step(epoch, global_step, on_load_weights=on_load_weights)
        self.update_step(epoch, global_step, on_load_weights=on_load_weights)

    def update_step(self, epoch: int, global_step: int, on_load_weights: bool = False):
        pass


class BaseObject(Updateable):
    @dataclass
    class Config:
        pass

    cfg: Config

    def __init__(self, cfg=None, *args, **kwargs):
        super().__init__()
        self.cfg = parse_structured(self.Config, cfg)
        self.device = get_device()
        self.configure(*args, **kwargs)

    def configure(self, *args, **kwargs):
        pass


class BaseModule(nn.Module, Updateable):
    @dataclass
    class Config:
        weights: Optional[str] = None

    cfg: Config
    # state_dict key prefixes of reference modules that exist in its checkpoints but are never used on the path
    IGNORED_REFERENCE_KEYS = ("metallic_predictor.", "roughness_predictor.", "albedo_predictor.", "inner_light.",
                              "light_pts", "FG_LUT", "sdf_network.", "normal_network.")
    # the reference ALWAYS stores these buffers (dreammat_mesh.py:188-199); this repo registers them only when the mesh has the
    # attribute, so a reference checkpoint of a UV-carrying mesh must still load into a UV-less local mesh
    OPTIONAL_REFERENCE_BUFFERS = ("vtex_buffer", "ttex_buffer")

    @classmethod
    def check_state_dict_match(cls, what, res, own_keys=()):
        """shared by `weights:` loading and Trainer.load_checkpoint: any missing key, or an unexpected key outside the two
        allow-lists, raises (instead of silently leaving parameters at their random initialisation)"""
        skip = cls.IGNORED_REFERENCE_KEYS

        def allowed(k):         # (substring match: the keys carry module prefixes in a whole-system checkpoint)
            return any(s in k for s in skip) or (k.rsplit(".", 1)[-1] in cls.OPTIONAL_REFERENCE_BUFFERS and k not in own_keys)
        unexpected = [k for k in res.unexpected_keys if not allowed(k)]
        if res.missing_keys or unexpected:
            raise RuntimeError(f"{what}: missing {list(res.missing_keys)}, unexpected {unexpected}")
        return [k for k in res.unexpected_keys if allowed(k)]

    def __init__(self, cfg=None, *args, **kwargs):
        super().__init__()
        self.cfg = parse_structured(self.Config, cfg)
        self.device = get_device()
        self.configure(*args, **kwargs)
        if self.cfg.weights is not None:
            # "path:module_name" partial load (base.py:103-112, misc.py:32-62)
            path, module_name = self.cfg.weights.split(":")
            ckpt = torch.load(path, map_location="cpu")
            sd = {k[len(module_name) + 1:]: v for k, v in ckpt["state_dict"].items() if k.startswith(module_name + ".")}
            # The reference is strict here (utils/base.py:109).  Its checkpoints carry keys this repo does not instantiate
            # (never-used predictor heads of dreammat_mesh.py:137-139, the inner-light MLP / probe buffers of
            # dreammat_material.py:400-414): only THOSE may be skipped; any missing key, or an unexpected key outside the
            # allow-list (wrong module prefix, truncated or mismatched checkpoint), raises instead of silently leaving
            # parameters at their random initialisation.
            res = self.load_state_dict(sd, strict=False)
            ignored = self.check_state_dict_match(f"weights '{self.cfg.weights}' do not match {type(self).__name__}", res,
                                                  own_keys=set(self.state_dict()))
            if ignored:
                print(f"[dreammat_amd] weights '{self.cfg.weights}': ignored reference-only keys {ignored}")
            self.do_update_step(ckpt.get("epoch", 0), ckpt.get("global_step", 0), on_load_weights=True)
        self._dummy: torch.Tensor
        self.register_buffer("_dummy", torch.zeros(0).float(), persistent=False)

    def configure(self, *args, **kwargs):
        pass
