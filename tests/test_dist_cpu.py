"""N>1 path on CPU (gloo, world_size 2): per-rank view sharding and the single flat-gradient all-reduce
whose mean is folded into the optimizer step (system.FlatParams / allreduce_sum_)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dreammat_amd.data import RandomCameraDataModule
    from dreammat_amd.system import FlatParams, allreduce_sum_
    torch.manual_seed(0)                                   # same init on every rank
    params = [torch.nn.Parameter(torch.randn(1000)), torch.nn.Parameter(torch.randn(64, 32)), torch.nn.Parameter(torch.randn(5, 64))]
    fp = FlatParams(params)
    assert fp.flat.numel() % 4 == 0 and params[1].data.data_ptr() == fp.flat[1000:].data_ptr()
    dm = RandomCameraDataModule(cfg={"height": 16, "width": 16, "batch_size": 4, "use_fix_views": True, "seed": 0}, rank=rank)
    dm.setup("fit")
    batch = dm.train_dataset.collate()
    # rank-dependent "loss": every parameter gets a gradient that depends on this rank's views
    loss = sum((p * (batch["azimuth"].sum() + i)).sum() for i, p in enumerate(params))
    loss.backward()
    local = fp.grad.clone()
    w = allreduce_sum_(fp.grad)
    assert w == world
    torch.save({"views": batch["view_id"], "env": batch["env_id"], "table": dm.train_dataset.azimuth_degs, "local": local,
                "summed": fp.grad.clone(), "grad_is_view": params[2].grad.data_ptr() == fp.grad[1000 + 2048:].data_ptr()},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_sharding_and_flat_allreduce(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["table"], r1["table"])                 # the 128 fixed views are shared
    assert not torch.equal(r0["views"], r1["views"])             # each rank draws its own views (seed + rank)
    assert r0["grad_is_view"] and r1["grad_is_view"]             # autograd accumulated INTO the flat buffer
    assert torch.allclose(r0["summed"], r0["local"] + r1["local"]) and torch.equal(r0["summed"], r1["summed"])
    # mean-of-ranks Adam == single-process Adam on the averaged gradient
    torch.manual_seed(0)
    p = torch.randn(1000, requires_grad=True)
    ref = torch.optim.Adam([p], lr=0.01, betas=(0.9, 0.99), eps=1e-15)
    p.grad = (r0["summed"][:1000] / 2).clone()
    ref.step()
    g = r0["summed"][:1000] * 0.5                                # what dm_adam_step does with grad_scale = 1/world
    m = 0.1 * g; v = 0.01 * g * g
    torch.manual_seed(0)
    q = torch.randn(1000)
    q = q - (0.01 / (1 - 0.9)) * (m / (v.sqrt() / (1 - 0.99) ** 0.5 + 1e-15))
    assert torch.allclose(q, p.detach(), atol=1e-6)
