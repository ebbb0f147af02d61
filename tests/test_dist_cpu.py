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


# ---- replica start-up and per-rank streams through the launcher's own code path (launch.seed_for_build ->
#      configure_optimizers -> sync_parameters_ -> Trainer.fit -> seed_rank_streams -> train_one_step x N)
class _CpuAdam:
    """torch.optim.Adam semantics with the mean folded in (what dm_adam_step does on the GPU)."""

    def __init__(self, fp):
        self.fp, self.m, self.v, self.t = fp, torch.zeros_like(fp.flat), torch.zeros_like(fp.flat), 0

    def step(self, world=1):
        self.t += 1
        g = self.fp.grad / world
        self.m.mul_(0.9).add_(g, alpha=0.1)
        self.v.mul_(0.99).addcmul_(g, g, value=0.01)
        mh, vh = self.m / (1 - 0.9 ** self.t), self.v / (1 - 0.99 ** self.t)
        self.fp.flat.sub_(0.01 * mh / (vh.sqrt() + 1e-15))
        self.fp.grad.zero_()


def _stub_system(build_seed_offset=0):
    """A CPU stand-in for DreamMat with the REAL initialisers of dreammat_amd/geometry.py (hash table U(-1e-4,1e-4) and
    the bias-free MLP, both drawing from the global RNG) and the real FlatParams / sync_parameters_ / Trainer; only the
    HIP kernels (render, nets, fused Adam) are replaced by a toy loss and a CPU Adam."""
    import torch.nn as nn
    from dreammat_amd.geometry import HashGridEncoding, VanillaMLP
    from dreammat_amd.system import FlatParams, sync_parameters_

    class Stub(nn.Module):
        def __init__(self):
            super().__init__()
            enc = {"otype": "HashGrid", "n_levels": 4, "n_features_per_level": 2, "log2_hashmap_size": 10,
                   "base_resolution": 4, "per_level_scale": 1.5}
            self.encoding = HashGridEncoding(3, enc)
            self.feature_network = VanillaMLP(self.encoding.n_output_dims, 5, {"n_neurons": 16, "n_hidden_layers": 1})
            self.device_ = torch.device("cpu")
            self.true_global_step = self.true_current_epoch = 0
            self.draws = []

        def on_fit_start(self):
            pass

        def do_update(self):
            pass

        def configure_optimizers(self):
            self.flat = FlatParams(list(self.parameters()))
            sync_parameters_(self.flat.flat)
            self.optimizer = _CpuAdam(self.flat)

        def training_step(self, batch, rng=None):
            B = batch["azimuth"].shape[0]
            t = torch.randint(20, 981, [B])                      # guidance.py: the timestep draw
            noise = torch.randn(B, 8)                            # guidance.py: the noise draw (global RNG as well)
            self.draws.append((t.clone(), noise.clone()))
            x = self.feature_network(torch.tanh(self.encoding.encoding.params[:8 * B].view(B, 8)).repeat(1, 1))
            loss = ((x.sum(-1) - noise.sum(-1)) ** 2 * (1 + t / 1000.0) * torch.cos(batch["azimuth"] * 0.01)).mean()
            return loss, {"train/loss_sds": loss.detach()}
    return Stub()


def _replica_worker(rank, world, port, out_dir, divergent_build):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dreammat_amd import launch
    from dreammat_amd.data import RandomCameraDataModule
    from dreammat_amd.system import Trainer, replicas_in_sync
    seed = 3
    # `divergent_build` reproduces round 1's bug (construction under seed + rank): the broadcast alone must repair it
    launch.seed_for_build(seed + (rank if divergent_build else 0))
    system = _stub_system()
    built = torch.cat([p.detach().reshape(-1) for p in system.parameters()]).clone()
    dm = RandomCameraDataModule(cfg={"height": 8, "width": 8, "batch_size": 2, "use_fix_views": True, "seed": seed}, rank=rank)
    trainer = Trainer(system, dm, max_steps=4, trial_dir=os.path.join(out_dir, f"trial{rank}"), val_check_interval=0,
                      checkpoint_every=0, seed=seed)
    system_params_before = []
    orig = trainer.train_one_step

    def spy(*a, **k):
        if not system_params_before:
            system_params_before.append(system.flat.flat.clone())
            assert replicas_in_sync(system.flat.flat)
        return orig(*a, **k)
    trainer.train_one_step = spy
    trainer.fit()
    assert replicas_in_sync(system.flat.flat)
    torch.save({"built": built, "before": system_params_before[0], "after": system.flat.flat.clone(),
                "t": torch.stack([d[0] for d in system.draws]), "noise": torch.stack([d[1] for d in system.draws])},
               os.path.join(out_dir, f"rep{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("divergent_build", [False, True])
def test_replicas_start_identical_stay_identical_and_draw_different_noise(tmp_path, divergent_build):
    """ranks were built under seed + rank and never synchronised."""
    port = _free_port()
    mp.spawn(_replica_worker, args=(2, port, str(tmp_path), divergent_build), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "rep0.pt"), torch.load(tmp_path / "rep1.pt")
    assert torch.equal(r0["built"], r1["built"]) == (not divergent_build)     # rank-independent build seed
    assert torch.equal(r0["before"], r1["before"])                            # (i) identical before step 1 ...
    assert torch.equal(r0["before"][:r0["built"].numel()], r0["built"])       #     ... and they are rank 0's
    assert torch.equal(r0["after"], r1["after"]) and not torch.equal(r0["after"], r0["before"])   # ... and after N
    assert not torch.equal(r0["t"], r1["t"]) and not torch.equal(r0["noise"], r1["noise"])       # (ii) per-rank draws


# ---- optimizer.sharded: reduce-scatter + Adam on the rank's slice + all-gather (SURVEY 8e "prefer") against all-reduce + full Adam
def _cpu_adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, grad_scale=1.0, zero_grad=True):
    """dm_adam_step's contract (csrc/adam.hip) restated in torch: torch.optim.Adam semantics, mean folded in, grad zeroed."""
    g = grad * grad_scale
    exp_avg.mul_(beta1).add_(g, alpha=1 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    mh, vh = exp_avg / (1 - beta1 ** step), exp_avg_sq / (1 - beta2 ** step)
    param.sub_(lr * mh / (vh.sqrt() + eps))
    if zero_grad:
        grad.zero_()


def _sharded_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dreammat_amd import system as sysm

    def make(sharded):
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.randn(1001)), torch.nn.Parameter(torch.randn(7, 5)), torch.nn.Parameter(torch.randn(5))]   # 1041: pads to 1044 (4) / 1048 (4 x world)
        fp = sysm.FlatParams(params, pad_to=4 * world if sharded else 4)
        if sharded:
            opt = sysm.ShardedFusedAdam(fp, lr=0.01, betas=(0.9, 0.99), eps=1e-15, adam_fn=_cpu_adam_step)
        else:
            opt = sysm.FusedAdam(fp, lr=0.01, betas=(0.9, 0.99), eps=1e-15)
            opt.step = lambda w, o=opt: (setattr(o, "step_count", o.step_count + 1),
                                         _cpu_adam_step(fp.flat, fp.grad, o.exp_avg, o.exp_avg_sq, o.step_count, o.lr, *o.betas, o.eps,
                                                        grad_scale=1.0 / w))
        return params, fp, opt

    def run(params, fp, opt, steps, g):
        for _ in range(steps):
            loss = sum((p * torch.randn(p.shape, generator=g)).sum() + (p ** 2).sum() * (rank + 1) for p in params)   # rank-dependent
            loss.backward()
            opt.sync_and_step()
        return fp.flat[:fp.numel].clone()

    ga, gb = torch.Generator().manual_seed(100 + rank), torch.Generator().manual_seed(100 + rank)
    pa, fa, oa = make(False)
    pb, fb, ob = make(True)
    ref = run(pa, fa, oa, 3, ga)
    got = run(pb, fb, ob, 3, gb)
    assert ob.exp_avg.numel() * world == fb.flat.numel() and fb.flat.numel() % (4 * world) == 0      # 1/world of the state
    assert pb[1].data.data_ptr() == fb.flat[1001:].data_ptr()                                    # parameters still live in the flat buffer
    assert float(fb.grad.abs().max()) == 0.0                                                       # zeroed for the next accumulation
    # checkpoint round trip through the gathered state: resume under the sharded optimizer and continue identically
    sd = ob.state_dict()
    assert fa.flat.numel() != fb.flat.numel()                        # the two optimizers pad the buffer differently ...
    assert sd["exp_avg"].numel() == fa.numel == 1041 and torch.allclose(sd["exp_avg"], oa.exp_avg[:fa.numel], atol=1e-7)
    pd, fd, od = make(False)                                         # ... and the same file loads under the un-sharded one
    od.load_state_dict(sd)
    assert torch.equal(od.exp_avg[:fa.numel], sd["exp_avg"]) and float(od.exp_avg[fa.numel:].abs().sum()) == 0 and od.step_count == 3
    oe = make(True)[2]
    oe.load_state_dict(oa.state_dict())                              # and the un-sharded optimizer's file under the sharded one
    assert torch.allclose(oe._gathered(oe.exp_avg)[:fa.numel], oa.exp_avg[:fa.numel])
    pc, fc, oc = make(True)
    fc.flat.copy_(fb.flat)
    oc.load_state_dict(sd)
    cont_b = run(pb, fb, ob, 2, gb)
    gc = torch.Generator().manual_seed(100 + rank); [torch.randn(p.shape, generator=gc) for _ in range(3) for p in pc]   # same stream position
    cont_c = run(pc, fc, oc, 2, gc)
    torch.save({"ref": ref, "got": got, "cont_b": cont_b, "cont_c": cont_c}, os.path.join(out_dir, f"s{rank}.pt"))
    dist.destroy_process_group()


def test_sharded_adam_matches_allreduce_adam_on_two_ranks(tmp_path):
    """system.ShardedFusedAdam over gloo at world size 2: three steps with rank-dependent gradients give the parameters of
    all-reduce + full Adam (the two-term sums are order-independent), every rank ends with the same full parameter buffer,
    and a checkpoint taken from the gathered moments resumes to the same trajectory."""
    port = _free_port()
    mp.spawn(_sharded_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "s0.pt"), torch.load(tmp_path / "s1.pt")
    assert torch.equal(r0["got"], r1["got"]) and torch.equal(r0["ref"], r1["ref"])
    assert torch.allclose(r0["got"], r0["ref"], atol=1e-6, rtol=1e-6)
    assert torch.equal(r0["cont_b"], r0["cont_c"]) and torch.equal(r0["cont_b"], r1["cont_b"])


def _sharded_fit_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dreammat_amd import launch
    from dreammat_amd.data import RandomCameraDataModule
    from dreammat_amd.system import FlatParams, ShardedFusedAdam, Trainer, replicas_in_sync, sync_parameters_

    def build(trial, max_steps, resume=None):
        launch.seed_for_build(5)
        system = _stub_system()

        def configure_optimizers(s=system):
            s.flat = FlatParams(list(s.parameters()), pad_to=4 * world)
            sync_parameters_(s.flat.flat)
            s.optimizer = ShardedFusedAdam(s.flat, lr=0.01, betas=(0.9, 0.99), eps=1e-15, adam_fn=_cpu_adam_step)
        system.configure_optimizers = configure_optimizers
        dm = RandomCameraDataModule(cfg={"height": 8, "width": 8, "batch_size": 2, "use_fix_views": True, "seed": 5}, rank=rank)
        return system, Trainer(system, dm, max_steps=max_steps, trial_dir=os.path.join(out_dir, trial), val_check_interval=0,
                               checkpoint_every=2, seed=5, resume=resume)

    system, trainer = build("run", 4)
    trainer.fit()                                        # checkpoints at steps 2 and 4: the moment gather is a collective
    assert replicas_in_sync(system.flat.flat)
    ck = os.path.join(out_dir, "run", "ckpts", "step=2.ckpt")
    dist.barrier()
    assert os.path.exists(ck) or rank != 0
    saved = torch.load(ck, map_location="cpu")
    assert saved["optimizer"]["exp_avg"].numel() == system.flat.numel and saved["global_step"] == 2
    torch.save({"after4": system.flat.flat.clone()}, os.path.join(out_dir, f"fit{rank}.pt"))
    dist.destroy_process_group()


def test_fit_loop_with_the_sharded_optimizer_checkpoints_without_deadlock(tmp_path):
    """Trainer.fit over gloo at world size 2 with optimizer.sharded semantics: the checkpoint's moment gather runs on EVERY rank
    (rank 0 alone writes the file), replicas stay bit-identical, and the checkpoint holds the full-size moments."""
    port = _free_port()
    mp.spawn(_sharded_fit_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "fit0.pt"), torch.load(tmp_path / "fit1.pt")
    assert torch.equal(r0["after4"], r1["after4"])
    assert sorted(os.listdir(tmp_path / "run" / "ckpts")) == ["step=2.ckpt", "step=4.ckpt"]
