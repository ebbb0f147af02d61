"""CLI entry (threestudio_dreammat/launch.py:42-246 without Lightning):
    python -m dreammat_amd.launch --config configs/dreammat.yaml --train system.prompt_processor.prompt="a chair" ...
One process per GPU; under torchrun the RANK/LOCAL_RANK/WORLD_SIZE env is honoured and gradients are
all-reduced over RCCL.  Seeding (reference launch.py:102 + Lightning DDP): the system is BUILT under the
rank-independent cfg.seed (hash-grid table and MLP initialisers draw from the global RNG) and rank 0's parameters are
broadcast in `configure_optimizers`; only then does each rank switch to `cfg.seed + rank` for its per-step draws."""
import argparse
import os

import torch
import torch.distributed as dist

import dreammat_amd
from .base import barrier, get_local_rank, get_rank
from .config import load_config
from .system import Trainer


def seed_for_build(seed):
    """Rank-INDEPENDENT seed for everything that initialises parameters or shared tables."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--gpu", default="0")
    g = ap.add_mutually_exclusive_group(required=True)
    g.add_argument("--train", action="store_true")
    g.add_argument("--validate", action="store_true")
    g.add_argument("--test", action="store_true")
    g.add_argument("--export", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    args, extras = ap.parse_known_args(argv)
    cfg = load_config(args.config, cli_args=extras)
    if cfg.get("_missing"):
        raise ValueError(f"Missing mandatory config values: {cfg['_missing']}")
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
        torch.cuda.set_device(get_local_rank())
        dist.init_process_group("nccl")
    rank = get_rank()
    seed = int(cfg.get("seed", 0))
    seed_for_build(seed)
    dreammat_amd._import_plugins()
    system = dreammat_amd.find(cfg["system_type"])(cfg["system"])
    data_cfg = dict(cfg.get("data", {}))
    data_cfg.setdefault("seed", cfg.get("seed", 0))
    dm = dreammat_amd.find(cfg["data_type"])(mesh=system.geometry.isosurface(), cfg=data_cfg, rank=rank,
                                             device=system.device_, atlas=getattr(system.material, "atlas", None))
    tr = cfg.get("trainer", {})
    ck = cfg.get("checkpoint", {})
    trial_dir = os.path.join(cfg.get("exp_root_dir", "outputs"), cfg.get("name", "dream_mat"), str(cfg.get("tag", "run")))
    trainer = Trainer(system, dm, max_steps=tr.get("max_steps", 30000), trial_dir=trial_dir,
                      val_check_interval=tr.get("val_check_interval", 100),
                      checkpoint_every=ck.get("every_n_train_steps", 3999), resume=cfg.get("resume"), seed=seed)
    if args.train:
        trainer.fit()
        if rank == 0:                 # every rank holds the same model: one writer for the test renders
            trainer.test()
    else:
        system.configure_optimizers()                 # all ranks: it contains the parameter broadcast
        if cfg.get("resume"):
            trainer.load_checkpoint(cfg["resume"])
        if rank == 0:                 # validate / export / test are single-writer jobs
            if args.validate:
                trainer.validate()
            elif args.export:
                for p in trainer.export():
                    print("[dreammat_amd] wrote", p)
            else:
                trainer.test()
    barrier()
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
