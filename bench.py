#!/usr/bin/env python
"""Benchmark of the SDS material-fitting step (BASELINE.json metric: "SDS steps/sec (512^2, 8 views)").

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = one optimizer step over ALL `--views` views (default 8): render (rasterize -> G-buffer ->
hash-grid field -> split-sum shade -> antialias) -> VAE encode -> ControlNet + UNet (3 branches) ->
SDS gradient -> backward into hash grid + MLP -> all-reduce (RCCL) -> fused Adam.  The 8 views are
sharded over the N ranks (strong scaling: total work per step is fixed), config = BASELINE configs[2]:
50 880-triangle displaced sphere, 512^2, 5 synthetic env maps, SD-2.1-base shaped nets (the reference's
model, dreammat.yaml:60; --sd sd15 selects the SD-1.5 shapes), bf16, seeded random weights and
synthetic condition maps (no checkpoints / Blender on this box).  The per-view camera tensors and condition maps of all
128 x 5 (view, environment) pairs are resident in HBM; each timed step includes its own collate (draw + device gather).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--views", type=int, default=8, help="total views per optimizer step (all ranks)")
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--sd", default="sd21-base", choices=["sd21-base", "sd15", "tiny"])
    ap.add_argument("--mesh", default="sphere:160:160")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-timeout", type=int, default=150)
    ap.add_argument("--env-res", type=int, default=128)
    ap.add_argument("--graph", choices=["on", "off", "auto"], default="auto",
                    help="hipGraph replay of the ControlNet+UNet noise prediction (guidance hip_graph)")
    ap.add_argument("--resident", choices=["auto", "on", "off"], default="auto",
                    help="condition maps / cameras of all (view, env) pairs resident in HBM (data.py); off = per-step H2D")
    ap.add_argument("--raytracing", action="store_true",
                    help="the reference's DEFAULT material branch (use_raytracing: true, 200 + 128 Monte-Carlo directions per pixel with "
                         "occlusion rays) instead of the split-sum branch BASELINE.json's metric is quoted on: not a BASELINE config")
    ap.add_argument("--sharded-adam", action="store_true",
                    help="optimizer.sharded: reduce-scatter + Adam on this rank's slice + all-gather instead of all-reduce + full Adam")
    ap.add_argument("--dtype", choices=["f16", "bf16"], default="f16",
                    help="16-bit type of the nets (guidance.weights_dtype): f16 (default since round 6) = the reference's "
                         "half_precision_weights (dreammat_guidance.py:56) -- noise prediction 1.3e-3 of fp32 -- and BASELINE configs[4]; "
                         "bf16 = the type BASELINE configs[1] names (1e-2 of fp32), reported beside the main line as `bf16_leg`")
    ap.add_argument("--attention", choices=["16bit", "fp8"], default="16bit",
                    help="fp8: the S >= 1024 self-attention of the frozen nets on the MX-FP8 matrix instruction (BASELINE configs[4])")
    ap.add_argument("--cfg5", action="store_true",
                    help="preset = the shape of BASELINE configs[4] in the reference's precision: --res 1024 --views 16 --mesh sphere:320:314 "
                         "--dtype f16 (200 320 triangles).  The fp8 MFMA attention configs[4] names is EXPERIMENTAL (INTEGRATION.md: 1.4-1.6x "
                         "slower than the 16-bit kernel it replaces) and no longer part of the preset: add --attention fp8 for it")
    ap.add_argument("--no-second-leg", "--no-f16-leg", dest="no_second_leg", action="store_true",
                    help="skip the second leg of the default 1-GPU run (the same step with the nets in the OTHER 16-bit type, "
                         "reported as `bf16_leg` / `f16_leg`)")
    ap.add_argument("--no-calibration", action="store_true",
                    help="skip `box_calibration` (a fixed vendor-library bf16 GEMM loop and a 1 GB copy: how fast THIS box is)")
    ap.add_argument("--no-debug-outputs", action="store_true",
                    help="renderer returns only the 5 keys the loss needs (the default writes all 12 keys of RaytraceRender.forward "
                         "every step, as the reference does: raytracing_renderer.py:209-222)")
    ap.add_argument("--dump-kernels", default=None, help="write the per-kernel HIP-event table of the timed region (JSON) here")
    ap.add_argument("--dump-shade", default=None,
                    help="after the clock has stopped, run one more step and save the shade kernels' REAL in-step inputs (G-buffer, "
                         "features, atlas) here as a .pt (tools/r4_shade_probe.py --case replays them)")
    a = ap.parse_args()
    if a.cfg5:
        a.res, a.views, a.mesh, a.dtype = 1024, 16, "sphere:320:314", "f16"
    return a


def synthetic_latlong(seed, h=256, w=512):
    """seeded log-normal sky + one Gaussian sun lobe (map1..5 are missing large blobs)."""
    import math
    g = torch.Generator().manual_seed(1000 + seed)
    sky = torch.exp(torch.randn(h // 8, w // 8, 3, generator=g))
    sky = torch.nn.functional.interpolate(sky.permute(2, 0, 1)[None], (h, w), mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
    d = torch.nn.functional.normalize(torch.randn(3, generator=g), dim=0)
    v = torch.linspace(0, math.pi, h)[:, None].expand(h, w)
    u = torch.linspace(-math.pi, math.pi, w)[None, :].expand(h, w)
    dirs = torch.stack([torch.sin(v) * torch.sin(u), torch.cos(v), -torch.sin(v) * torch.cos(u)], -1)
    lobe = 50.0 * torch.exp(-(1 - (dirs * d).sum(-1)) / 0.01)
    return (sky * 0.5 + lobe[..., None]).contiguous()


def system_config(a, views_per_rank):
    return {
        "geometry": {"shape_init": a.mesh, "shape_init_params": 0.8},
        "material": {"use_raytracing": bool(a.raytracing), "diffuse_sample_num": 200, "specular_sample_num": 128,
                     "environment_scale": 2.0, "env_max_res": a.env_res, "env_min_res": 16,
                     "n_envs": 5,
                     # the reference's own split-sum LUT (tests/golden/assets, copied from load/lights); falls back to the
                     # analytic stand-in when the file is absent -- `config.fg_lut` in the JSON line says which one ran
                     "fg_lut_path": os.path.join(ROOT, "tests", "golden", "assets", "bsdf_256_256.bin")},
        "guidance": {"use_controlnet": True, "control_types": ["light"], "condition_scales": [1.0],
                     "condition_scales_anneal": [0.8], "control_anneal_start_step": 700, "width": a.res,
                     "height": a.res, "pretrained_model_name_or_path": a.sd, "synthetic": True, "cond_scale": 1.05,
                     "weights_dtype": {"bf16": "bfloat16", "f16": "float16"}[a.dtype], "attention_precision": a.attention,
                     "hip_graph": {"on": True, "off": False, "auto": "auto"}[a.graph],
                     "uncond_scale": [0, -1.0, -0.5, 2000], "null_scale": [0, 0.0, -0.5, 2000], "noise_scale": 0.0,
                     "min_step_percent": [500, 0.2, 0.02, 501], "max_step_percent": [500, 0.8, 0.5, 501]},
        "prompt_processor": {"prompt": "a DSLR photo of a ceramic vase", "negative_prompt": "ugly, low resolution",
                             "pretrained_model_name_or_path": a.sd, "synthetic": True,
                             # shared by all ranks of one job (rank 0 writes, the others read after the barrier)
                             "cache_dir": os.path.join("/tmp", f"dm_text_cache_{os.environ.get('MASTER_PORT', os.getpid())}")},
        "loss": {"lambda_sds": 1.0, "lambda_mat_reg": 1.0},
        "optimizer": {"name": "Adam", "args": {"lr": 0.01, "betas": [0.9, 0.99], "eps": 1e-15}, "sharded": bool(a.sharded_adam)},
    }


def _layer_fallbacks():
    """16-bit CUDA calls of the nets that ran on ATen / hipBLASLt instead of the kernels of csrc/ (sd/layers.note_fallback)"""
    from dreammat_amd.sd import layers
    return layers.fallbacks()


def baseline_config_name(a, n_tris):
    """which BASELINE.json `configs` entry the run's shape is (the string used to be hard-coded to configs[2])"""
    if a.raytracing:
        return "custom shape (no BASELINE.json entry: the reference's default Monte-Carlo ray-traced material branch, SURVEY row f-1)"
    if a.res == 512 and a.views == 8 and 40000 <= n_tris <= 60000:
        return "BASELINE configs[2]" if int(os.environ.get("WORLD_SIZE", 1)) == 1 else "BASELINE configs[3] (configs[2] sharded by view)"
    if a.res == 512 and a.views == 4 and n_tris < 12000:
        return "BASELINE configs[1]"
    if a.res == 1024 and a.views == 16 and n_tris >= 190000:
        if a.dtype == "f16" and a.attention == "fp8":
            return "BASELINE configs[4] (f16 nets, MX-FP8 self-attention) on 1 GPU"
        return f"BASELINE configs[4] shape ({a.dtype} nets, {a.attention} attention: the entry names fp16 + fp8)"
    return "custom shape (no BASELINE.json entry)"


def _profile_commit(path):
    """provenance of a committed counter profile: the commit / date tools/pmc_collect.py stamped into it (the GPU box has no
    .git) and a content hash -- a stale counter file is visible in the JSON line"""
    import hashlib
    try:
        raw = open(path, "rb").read()
        d = json.loads(raw)
        return {"collected_at_commit": d.get("collected_at_commit"), "collected_on": d.get("collected_on"),
                "sha256_12": hashlib.sha256(raw).hexdigest()[:12]}
    except Exception:
        return None


def pmc_traffic(kernel_key, batch):
    """HBM traffic per launch of the kernel named by `kernel_key` ("... conv3x3[Cin->Cout@HxW,s1]" at batch `batch`, or
    "... attention_fwd_bf16[Sq=..,Skv=..,h=..,D=..]" at batch 3*`batch`) from the newest committed counter profile
    (profiles/r*_pmc_final.json, written by tools/pmc_r2.sh + tools/pmc_collect.py), if exactly that shape was profiled."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_final.json")))
    if not files:
        return None
    m = re.search(r"conv3x3\[(?:gn\+)?(\d+)->(\d+)@(\d+)x(\d+),s1\]", kernel_key)      # (a `gn+` launch moves the same tensors)
    m2 = re.search(r"attention_fwd_bf16\[Sq=(\d+),Skv=(\d+),h=(\d+),D=(\d+)\]", kernel_key)
    if m:
        cin, cout, h, w = (int(g) for g in m.groups())
        tag = f"conv_{batch}_{h}_{w}_{cin}_{cout}"
        alg = 2.0 * (batch * h * w * (cin + cout) + 9 * cin * cout)
    elif m2:
        sq, skv, heads, d = (int(g) for g in m2.groups())
        tag = f"attn_{3 * batch}_{heads}_{sq}_{skv}_{d}"
        alg = 2.0 * 3 * batch * heads * d * (2 * sq + 2 * skv)
    else:
        return None

    def first(name, counter):       # the profiled process runs one kernel: take the row that has the counter
        for row in ctr.get(name, {}).values():
            if counter in row:
                return row[counter]
        raise KeyError(name)
    try:
        ctr = json.load(open(files[-1]))["counters"]
        fetch, write = first("fetch_" + tag, "FETCH_SIZE"), first("write_" + tag, "WRITE_SIZE")
    except (KeyError, ValueError, OSError):
        return None
    out = {"traffic": (2.0 * fetch + write) * 1024.0, "traffic_unit": "bytes/launch", "algorithmic_bytes": alg,
           "traffic_source": "profiles/" + os.path.basename(files[-1]), "traffic_source_commit": _profile_commit(files[-1])}
    try:
        out["l2_hit_rate"] = first("tcc_" + tag, "TCC_HIT_sum") / max(1.0, first("tcc_" + tag, "TCC_REQ_sum"))
    except KeyError:
        pass
    try:    # clock the kernel actually ran at: GRBM_GUI_ACTIVE (summed over 8 XCDs) / kernel duration
        st = json.load(open(files[-1]))["kernel_stats"].get(tag)
        if st:
            out["shader_clock_ghz_under_load"] = first("grbm_" + tag, "GRBM_GUI_ACTIVE") / 8.0 / (st[0]["avg_us"] * 1e3)
    except (KeyError, ValueError, OSError):
        pass
    return out


def shade_traffic(key, texel):
    """HBM bytes per launch of k_shade_fwd / k_shade_bwd from the committed counter pass on the bench scene's REAL G-buffer
    (tools/r2_probe.py dumps it, tools/_abi_pmc `shadef` replays it under rocprofv3 --pmc; 1.15 M covered pixels)."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_final.json")))
    if not files:
        return None
    try:
        ctr = json.load(open(files[-1]))["counters"]
        case = texel if ("fetch_shade_" + texel) in ctr else "fp32"
        kern = "k_" + key
        fetch = next(v["FETCH_SIZE"] for k, v in ctr["fetch_shade_" + case].items() if kern in k)
        write = next(v["WRITE_SIZE"] for k, v in ctr["write_shade_" + case].items() if kern in k)
    except (KeyError, StopIteration, ValueError, OSError):
        return None
    return {"traffic": (2.0 * fetch + write) * 1024.0, "traffic_unit": "bytes/launch at 1.15 M covered pixels (counter pass, atlas " + case + ")",
            "traffic_source": "profiles/" + os.path.basename(files[-1]), "traffic_source_commit": _profile_commit(files[-1])}


def effective_cores():
    """CPU cores this process may actually use (affinity mask and cgroup quota), capped at 64: os.cpu_count()
    reports the host's cores even inside a small container and oversubscribing torch's thread pool by 10x
    makes the fp32 oracle crawl."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, min(n, 64))


def cpu_baseline(a, system, max_threads=None):
    """The oracle (fp32 torch + C rasterizer) timed on this box's host cores on a BOUNDED sample of the same workload
    (~20-40 s of CPU work), every stage at the REAL resolution (staged timing at cfg2/3): ONE view's render
    fwd+bwd at res^2 on the full 50k-triangle mesh and the full 16-level hash grid, ONE image's VAE-encoder fwd+bwd at
    res^2, ONE branch-item of ControlNet+UNet at (res/8)^2 latents -- multiplied by the step's counts (x views, x 3*views
    branch items).  No extrapolation in resolution (round 2 ran at half resolution and scaled by 4)."""
    import numpy as np
    from oracle import envlight as oenv, field as ofield, raster as oraster, render as orender, sd_nets as osd
    from oracle import camera as ocam
    cores = effective_cores()
    torch.set_num_threads(max_threads or cores)
    H = W = a.res
    mesh = system.geometry.mesh
    md = dict(v_pos=mesh.v_pos.cpu().numpy(), v_nrm=mesh.v_nrm.cpu().numpy(),
              t_pos_idx=mesh.t_pos_idx.cpu().numpy().astype(np.int32))
    md["opp"] = oraster.build_topology(md["t_pos_idx"])
    batch = ocam.camera_batch(torch.tensor([20.0]), torch.tensor([30.0]), torch.tensor([3.5]), torch.tensor([35.0]), H, W)
    batch["env_id"] = torch.tensor([0])
    env = oenv.EnvLight(synthetic_latlong(0, 32, 64), scale=2.0, min_res=8, max_res=16)   # prefilter = init-time work
    lv, tot = ofield.grid_levels()
    geo = system.geometry
    table = geo.encoding.encoding.params.detach().float().cpu().reshape(-1, 2).clone().requires_grad_()
    w1 = geo.feature_network.layers[0].weight.detach().float().cpu().clone().requires_grad_()
    w2 = geo.feature_network.layers[2].weight.detach().float().cpu().clone().requires_grad_()
    fg = system.material.atlas.fg_lut.cpu()
    g = torch.Generator().manual_seed(0)
    t0 = time.time()
    out = orender.render(md, batch, dict(table=table, w1=w1, w2=w2, levels=lv, radius=1.0), [env], fg,
                         torch.rand(1, H, W, generator=g), torch.randn(1, H, W, generator=g))
    (out["comp_rgb"].sum() + out["loss_mat_reg"]).backward()
    t_render = time.time() - t0
    gd = system.guidance
    arch = gd.arch
    sd_vae = {k: v.float().cpu() for k, v in gd.vae.state_dict().items()}
    img = torch.rand(1, 3, H, W, generator=g).requires_grad_()
    t0 = time.time()
    mean, logvar = osd.vae_encode_moments(sd_vae, img * 2 - 1)
    (mean + torch.exp(0.5 * logvar)).sum().backward()
    t_vae = time.time() - t0
    del sd_vae
    sd_u = {k: v.float().cpu() for k, v in gd.unet.state_dict().items()}
    sd_c = {k: v.float().cpu() for k, v in gd.controlnets[0].state_dict().items()}
    lat = torch.randn(1, 4, H // 8, W // 8, generator=g)
    ctx = torch.randn(1, 77, arch.cross_dim, generator=g)
    cond = torch.rand(1, 22, H, W, generator=g)
    tt = torch.tensor([500])
    with torch.no_grad():
        t0 = time.time()
        d, m = osd.controlnet_forward(sd_c, lat, tt, ctx, cond, 1.0, arch.heads, arch.use_linear_projection)
        osd.unet_forward(sd_u, lat, tt, ctx, arch.heads, arch.use_linear_projection, d, m)
        t_nets = time.time() - t0
    step_s = a.views * (t_render + t_vae) + 3 * a.views * t_nets
    return {"value": 1.0 / step_s, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "extrapolated": True,      # a bounded sample (1 view, 1 image, 1 branch item, all at full resolution) x the step's counts
            "sample": f"oracle fp32 on host at the real {H}^2: 1 view render fwd+bwd {t_render:.2f}s, "
                      f"1 VAE-enc fwd+bwd {t_vae:.2f}s, 1 branch-item ControlNet+UNet fwd {t_nets:.2f}s; "
                      f"step = {a.views} x (render+vae) + {3 * a.views} x nets (counts only: nothing is scaled in resolution)",
            "host_cpus": os.cpu_count()}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # test tier only (tests/test_hip_gpu.py): DREAMMAT_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and DREAMMAT_BENCH_BACKEND=gloo
    # carries the collectives, so the world-2 control flow of this file (who takes which step, who waits at which barrier) runs on
    # a 1-GPU box.  RCCL refuses two ranks on one device; the driver's multi-GPU runs use neither switch.
    share_gpu = os.environ.get("DREAMMAT_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("DREAMMAT_BENCH_BACKEND", "nccl")
    dev_index = 0 if share_gpu else local_rank
    if share_gpu:
        os.environ["LOCAL_RANK"] = "0"            # base.get_device() = cuda:{LOCAL_RANK}
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or bool(os.environ.get("DREAMMAT_FORCE_DIST"))   # FORCE_DIST: exercise RCCL at world 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert a.views % world == 0, "views must divide over ranks"
    vpr = a.views // world

    import dreammat_amd
    from dreammat_amd import hipops
    from dreammat_amd.data import RandomCameraDataModule
    from dreammat_amd.system import Trainer, to_device
    dreammat_amd._import_plugins()
    torch.manual_seed(0)          # identical initial parameters on every rank (DDP broadcast equivalent)
    lat = [synthetic_latlong(i) for i in range(5)]
    system = dreammat_amd.find("dreammat-system")(system_config(a, vpr), material_kwargs={"latlongs": lat})
    # all 12 keys of RaytraceRender.forward every step, like the reference (round 4 timed 5 of them; --no-debug-outputs
    # restores that for the A/B: the 7 logging buffers are ~140 MB of scatters)
    system.renderer.debug_outputs = not a.no_debug_outputs
    dm = RandomCameraDataModule(cfg={"height": a.res, "width": a.res, "batch_size": vpr, "use_fix_views": True,
                                     "camera_distance_range": [3.0, 4.0], "fovy_range": [25, 45], "camera_perturb": 0.0,
                                     "center_perturb": 0.0, "up_perturb": 0.0, "elevation_range": [-20, 45],
                                     "azimuth_range": [-180, 180], "condition_source": "synthetic", "seed": 0,
                                     "resident": {"auto": None, "on": True, "off": False}[a.resident]},
                                rank=rank, device=dev)
    dm.setup("fit")
    system.on_fit_start()
    system.configure_optimizers()
    trainer = Trainer(system, dm, max_steps=10 ** 9, seed=0)
    trainer.seed_rank_streams()   # parameters are identical (seed 0 build + rank-0 broadcast); t / noise / jitter are per rank
    # The data path is INSIDE the timed step: collate() draws this step's views / environments and gathers the camera
    # tensors and the 22-channel condition maps from the HBM-resident tables (data.py `resident`; built during warm-up,
    # like the reference decodes its pre-render tree once at start-up).  No step input is prepared ahead of the clock.
    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(max(a.warmup, 1)):
        trainer.train_one_step()
    sync()
    # the `roofline` kernel = the conv shape with the largest total time in one (untimed) fully instrumented step
    hipops.enable_kernel_timing(True, only=("conv3x3",))
    trainer.train_one_step()
    sync()
    hipops.enable_kernel_timing(False)
    _kt0 = hipops.kernel_times()
    roof_key = max(_kt0, key=lambda k: _kt0[k]["avg_ms"] * _kt0[k]["launches"]) if _kt0 else "conv3x3"
    # HIP events INSIDE the timed region only around the launches of the `roofline` kernel (the dominant 3x3 convolution
    # shape, 9 launches per step); the other per-kernel tables (all conv shapes, attention, GEMM, --dump-kernels) come from
    # extra steps after the clock has stopped -- an event pair costs ~25 us of stream time (it drains the queue), 700 of
    # them per step cost 7 ms.
    hipops.enable_kernel_timing(True, only=(roof_key,))
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss, logs = trainer.train_one_step()
    sync()
    elapsed = time.perf_counter() - t0
    kt_live = hipops.kernel_times()
    roof_steps = min(a.steps, 3)
    hipops.enable_kernel_timing(True)
    for i in range(roof_steps):
        trainer.train_one_step()
    sync()
    hipops.enable_kernel_timing(False)
    if use_dist:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kt = hipops.kernel_times()
    # the shade kernels REPLAYED on the last step's G-buffer (back-to-back launches, each between its own HIP events): once with
    # the step's own features and once with per-pixel-random features -- at this point of training the field is nearly constant
    # (roughness 0.525 +- 3e-5, every pixel on one mip pair), which is the kernels' best case
    shade_replay = {}
    if not a.raytracing:
        # (the extra step is COLLECTIVE -- every rank takes it, its all-reduce included; only the replay below is rank 0's alone
        # and contains no collective: a rank-0-only step would leave the other ranks' all-reduce unmatched)
        hipops.SHADE_KEEP["on"] = rank == 0
        trainer.train_one_step()
        sync()
        hipops.SHADE_KEEP["on"] = False
        kept, hipops.SHADE_KEEP["last"] = hipops.SHADE_KEEP["last"], None
        if rank == 0 and kept is not None and kept[0].shape[0] > 0:
            feat0 = kept[0]
            gen = torch.Generator(device=dev).manual_seed(7)
            cases = {"step_features": feat0, "random_features": torch.randn(feat0.shape, device=dev, generator=gen)}
            for nm, ft in cases.items():
                # the same [5, pitch] feature-major storage the field MLP hands the kernels
                Np = (ft.shape[0] + 3) // 4 * 4
                st = torch.empty(ft.shape[1], Np, device=dev)[:, :ft.shape[0]]
                st.copy_(ft.t())
                f = st.t().requires_grad_(True)
                gs = torch.empty(3, Np, device=dev)[:, :ft.shape[0]]
                gs.copy_(torch.randn(3, ft.shape[0], device=dev, generator=gen))
                g = gs.t()                                   # the upstream gradient in the SoA form the scatter's backward hands over
                for it in range(13):
                    if it == 3:
                        torch.cuda.synchronize()
                        hipops.enable_kernel_timing(True, only=("shade_fwd", "shade_bwd"))
                        # hold the stream while the host enqueues the ten iterations: the launches then run back to back and an
                        # event pair brackets its kernel only (an idle stream would add the host's launch latency to every pair)
                        torch.cuda._sleep(int(3e7))
                    col = hipops.shade(f, *kept[1:], want_debug=False)[0]
                    col.backward(g)
                    f.grad = None
                torch.cuda.synchronize()
                hipops.enable_kernel_timing(False)
                shade_replay[nm] = hipops.kernel_times()
    # Second leg (1 GPU, default run only): the SAME step with the nets in the OTHER 16-bit type.  The main line runs IEEE half --
    # the reference's half_precision_weights type, the precision class in which the hand-written stack's noise prediction is within
    # 1.3e-3 of fp32 --, the leg runs bf16 (the type BASELINE configs[1] names; 1e-2 of fp32, a few % faster: the f16 multiplier array
    # draws more power and the chip clocks lower under it).  The nets are cast in place (timing only: parity is
    # tests/test_hip_gpu.py::test_full_size_sd21_unet_controlnet_eps_vs_oracle) and the leg runs after everything the main line reports.
    other = {"f16": "bf16", "bf16": "f16"}[a.dtype]
    second_leg = None
    if world == 1 and a.attention == "16bit" and not a.raytracing and not a.no_second_leg:
        try:
            tdt = {"f16": torch.float16, "bf16": torch.bfloat16}[other]
            g_ = system.guidance
            for m_ in [g_.vae, g_.unet] + list(g_.controlnets):
                m_.to(tdt)
            g_.weights_dtype = tdt
            getattr(g_, "_graphs", {}).clear()
            n2 = max(3, min(a.steps, 8))
            for i in range(2):
                trainer.train_one_step()
            sync()
            t1 = time.perf_counter()
            for i in range(n2):
                loss2, _ = trainer.train_one_step()
            sync()
            e2 = time.perf_counter() - t1
            hipops.enable_kernel_timing(True, only=("attention", roof_key))
            trainer.train_one_step()
            sync()
            hipops.enable_kernel_timing(False)
            kt2 = hipops.kernel_times()
            second_leg = {"dtype": other, "value": n2 / e2, "unit": "steps/s", "ms_per_step": e2 / n2 * 1e3, "steps": n2, "warmup": 2,
                          "final_loss": float(loss2.detach()),
                          "nets": f"the same step, UNet / ControlNet / VAE cast to {other} in place"}
            for nm, pre in (("roofline", roof_key), ("roofline_attention", "attention")):      # the two MFMA rows in this type too
                grp = {k: v for k, v in kt2.items() if k.startswith(pre)}
                if grp:
                    key = max(grp, key=lambda k: grp[k]["avg_ms"] * grp[k]["launches"])
                    tf = grp[key]["work_per_launch"] / (grp[key]["avg_ms"] * 1e-3) / 1e12
                    second_leg[nm] = {"kernel": key, "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0,
                                      "avg_us": grp[key]["avg_ms"] * 1e3, "launches_timed": grp[key]["launches"]}
        except Exception as e:      # reporting only: never allowed to kill the bench line
            second_leg = {"dtype": other, "value": None, "error": repr(e)}
    # How fast is THIS box?  Boxes of the pool differ by 4-5 % on every MFMA kernel at once; two fixed
    # vendor-library workloads that no round's code touches let a reader tell a slow box from a regression: a bf16 8192^3 GEMM
    # loop on hipBLASLt (~30 ms) and a 1 GB device copy.  After the clock has stopped; torch is plumbing here, not the product.
    calib = None
    if rank == 0 and not a.no_calibration:
        try:
            ga = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
            gb = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
            src = torch.empty(1 << 30, device=dev, dtype=torch.uint8).random_(0, 255)
            dst = torch.empty_like(src)
            for _ in range(3):
                torch.matmul(ga, gb); dst.copy_(src)
            torch.cuda.synchronize()
            n_mm, n_cp = 25, 10
            e0, e1, e2_ = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            for _ in range(n_mm):
                torch.matmul(ga, gb)
            e1.record()
            for _ in range(n_cp):
                dst.copy_(src)
            e2_.record()
            torch.cuda.synchronize()
            t_mm, t_cp = e0.elapsed_time(e1) * 1e-3 / n_mm, e1.elapsed_time(e2_) * 1e-3 / n_cp
            calib = {"gemm_bf16_8192_tflops": 2.0 * 8192 ** 3 / t_mm / 1e12, "gemm_loop_ms": t_mm * n_mm * 1e3,
                     "copy_1gb_tbps": 2.0 * (1 << 30) / t_cp / 1e12,
                     "what": "torch.matmul bf16 8192^3 x 25 (hipBLASLt) and a 1 GiB device copy x 10 (read + write counted), HIP events, "
                             "after the timed region -- fixed vendor workloads: divide `value` by these to compare boxes / rounds"}
            del ga, gb, src, dst
        except Exception as e:
            calib = {"error": repr(e)}
    if a.dump_shade:
        if rank == 0:
            hipops.SHADE_DUMP["path"] = a.dump_shade
        trainer.train_one_step()        # collective, like the replay's step above
        sync()
    if a.dump_kernels and rank == 0:
        with open(a.dump_kernels, "w") as fh:
            json.dump({"steps": roof_steps, "kernels": kt}, fh, indent=1)

    if rank == 0:
        ms = elapsed / a.steps * 1e3
        res = {"metric": f"SDS steps/sec ({a.res}^2, {a.views} views)", "value": a.steps / elapsed, "unit": "steps/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": a.dtype + ("+fp8attn" if a.attention == "fp8" else ""), "data": "synthetic",
               "config": {"workload": f"{baseline_config_name(a, system.geometry.mesh.t_pos_idx.shape[0])}: "
                                      f"{system.geometry.mesh.t_pos_idx.shape[0]}-tri displaced sphere, "
                                      f"{a.res}^2, {a.views} views/step, 5 synthetic env maps, {a.sd} UNet+22ch ControlNet "
                                      f"(random init), " + ("Monte-Carlo shading (200 + 128 directions, occlusion rays)" if a.raytracing else "split-sum shading") +
                                      ", hash-grid field 16x2 2^19",
                          "views_per_step": a.views, "views_per_rank": vpr, "resolution": a.res, "sd_arch": a.sd,
                          "timed_region": "collate (draw + HBM gather of cameras and condition maps) + render + VAE/ControlNet/UNet + "
                                          "SDS + backward + all-reduce + Adam; renderer outputs: "
                                          + ("the 5 keys the loss needs" if a.no_debug_outputs else "all 12 keys of the reference's render dict"),
                          "fg_lut": "reference bsdf_256_256.bin" if system.material.real_fg_lut else "analytic stand-in (file absent)",
                          "atlas_texel": system.material.atlas.texel,
                          "noise_pred_hip_graph": bool(getattr(system.guidance, "_graphs", None)),
                          "process_group": (dist.get_backend() if use_dist else None),
                          "peak_hbm_gb": torch.cuda.max_memory_allocated() / 1e9,
                          "parallelism": (f"dp{world} (views sharded, reduce-scatter + sharded Adam + all-gather of {system.flat.numel * 4 / 1e6:.1f} MB fp32)"
                                          if a.sharded_adam else
                                          f"dp{world} (views sharded, 1 all-reduce of {system.flat.numel * 4 / 1e6:.1f} MB fp32 grads)"),
                          "left_hand_written_kernels": sorted(f"{k} {d}" for (k, d) in _layer_fallbacks()),
                          "final_loss": float(loss.detach())}}
        # ---- rooflines from HIP events around the launches: `roofline` (conv) live in the timed region, the others on the
        # extra steps that follow it (same workload, same streams)
        def mfma_entry(name, group, n_steps=roof_steps, where="extra steps after the timed region"):
            # dense MFMA peak of the launch's operand type (MI355X_MICROARCH.md): 2.5 PF/s bf16 / f16, 5 PF/s for the MX-FP8 products
            peak = lambda k: 5000.0 if "fp8" in k else 2500.0
            key = max(group, key=lambda k: group[k]["avg_ms"] * group[k]["launches"])
            r = group[key]
            tf = r["work_per_launch"] / (r["avg_ms"] * 1e-3) / 1e12
            tot_fl = sum(v["work_per_launch"] * v["launches"] for v in group.values())
            tot_ms = sum(v["avg_ms"] * v["launches"] for v in group.values())
            tot_peak_fl = sum(v["avg_ms"] * 1e-3 * v["launches"] * peak(k) * 1e12 for k, v in group.items())
            return {"kernel": name + " " + key, "bound": "mfma", "achieved": tf, "peak": peak(key), "unit": "TFLOP/s",
                    "frac": tf / peak(key), "traffic": None, "launches_timed": r["launches"], "avg_us": r["avg_ms"] * 1e3,
                    "all_shapes": {"TFLOP/s": tot_fl / (tot_ms * 1e-3) / 1e12, "frac": tot_fl / tot_peak_fl,
                                   "ms_per_step": tot_ms / n_steps,
                                   "launches_per_step": sum(v["launches"] for v in group.values()) / n_steps},
                    "events": where}
        attn = {k: v for k, v in kt.items() if k.startswith("attention")}
        conv = {k: v for k, v in kt.items() if k.startswith("conv3x3")}
        if roof_key in kt_live:    # the dominant shape's row: measured inside the timed region (same per-step footing)
            conv[roof_key] = dict(kt_live[roof_key], launches=kt_live[roof_key]["launches"] * roof_steps / a.steps)
        gemm = {k: v for k, v in kt.items() if k.startswith("gemm")}
        if conv:      # the dominant hand-written kernel of the step by time
            res["roofline"] = mfma_entry("k_conv3x3_halo / k_conv3x3_dma<9 taps> (persistent LDS-DMA implicit GEMM; halo-patch kernel where "
                                         "the shape has enough patches, `gn+` = GroupNorm apply + SiLU folded into it)", conv, roof_steps,
                                         "dominant shape: timed region; all_shapes: extra steps after it")
            if roof_key in kt_live:
                res["roofline"]["launches_timed"] = kt_live[roof_key]["launches"]
            # a `gn+` row's time includes the folded GroupNorm apply pass (its FLOPs are not counted): the SAME shape without it (the
            # data-gradient launches of the VAE encoder's backward) beside it
            if roof_key.startswith("conv3x3[gn+"):
                plain = roof_key.replace("conv3x3[gn+", "conv3x3[")
                if plain in kt:
                    rp = kt[plain]
                    tfp = rp["work_per_launch"] / (rp["avg_ms"] * 1e-3) / 1e12
                    res["roofline"]["same_shape_without_groupnorm"] = {"kernel": plain, "avg_us": rp["avg_ms"] * 1e3, "achieved": tfp, "frac": tfp / 2500.0,
                                                                       "launches_timed": rp["launches"]}
        if attn:      # north_star target: >= 50 % MFMA
            res["roofline_attention"] = mfma_entry(("k_attn_fwd_fp8 (S >= 1024) + " if a.attention == "fp8" else "") + "k_attn_fwd_w128 / k_attn_fwd_w64 / k_attn_fwd_v3 (dispatch: " + hipops.attention_variant() + ")", attn)
            if "roofline" not in res:
                res["roofline"] = res["roofline_attention"]
        if gemm:      # Linear / 1x1 layers + GEGLU on the 1-tap instantiation of the conv kernel
            res["roofline_gemm"] = mfma_entry("k_conv3x3_dma<1 tap> (fused Linear/GEGLU/residual GEMM)", gemm)
        for nm in ("roofline", "roofline_attention"):
            tr = pmc_traffic(res[nm]["kernel"], vpr) if nm in res else None
            if tr:
                res[nm].update(tr)
        res["roofline_note"] = ("traffic: HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB from `rocprofv3 --pmc` on the "
                                "same kernel and shape (conv, attention) through the C ABI driver tools/abi_pmc.cpp (tools/pmc_r2.sh; FETCH_SIZE "
                                "doubled per MI355X_MICROARCH.md); not collected live because rocprofv3 --pmc segfaults "
                                "under python+torch in this image (profiles/r01_pmc_attempt_segfault.log)")
        for nm, key in (("roofline_shade_fwd", "shade_fwd"), ("roofline_shade_bwd", "shade_bwd")):
            # the forward inside the step also writes the 7 logging buffers of the reference's render dict (debug_outputs: 17 floats
            # per covered pixel on top of SURVEY 8d's 56 B): its row counts the bytes that launch moves; the replay rows below run
            # the kernel without them, on the 56 B / 76 B definition
            step_key = key + "+dbg" if key + "+dbg" in kt else key
            if step_key in kt:
                r = kt[step_key]
                per_px = (56.0 if key == "shade_fwd" else 76.0)
                n_px = r["work_per_launch"] / per_px
                bytes_launch = r["work_per_launch"] + (68.0 * n_px if step_key.endswith("+dbg") else 0.0)
                gbs = bytes_launch / (r["avg_ms"] * 1e-3) / 1e9
                res[nm] = {"kernel": "k_" + key + (" (+ 7 logging outputs)" if step_key.endswith("+dbg") else ""), "bound": "hbm",
                           "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "traffic": None,
                           "avg_us": r["avg_ms"] * 1e3, "covered_pixels": n_px, "algorithmic_bytes": bytes_launch,
                           # the SAME launch on SURVEY 8d's numerator (56 B / 76 B per covered pixel: what the north_star target of
                           # 0.40 is defined on) -- `frac` above counts the 68 B of logging outputs this launch also writes
                           ("frac_56B" if key == "shade_fwd" else "frac_76B"): r["work_per_launch"] / (r["avg_ms"] * 1e-3) / 8e12,
                           "north_star_row": "replay_random_features.frac (8d numerator, per-pixel-random material)"}
                res[nm].update(shade_traffic(key, system.material.atlas.texel) or {})
                res[nm]["measured"] = "inside full steps (cold caches: 70 ms and several GB after the kernel's previous run)"
                for case, ktr in shade_replay.items():      # replayed back-to-back on the step's G-buffer
                    if key in ktr:
                        rr = ktr[key]
                        g2 = rr["work_per_launch"] / (rr["avg_ms"] * 1e-3) / 1e9
                        res[nm]["replay_" + case] = {"avg_us": rr["avg_ms"] * 1e3, "achieved": g2, "frac": g2 / 8000.0,
                                                     "launches_timed": rr["launches"]}
        # SURVEY 8d "report, do not gate" rows: rasterize (+ interpolate, fused into the G-buffer pass), antialias, hash grid --
        # algorithmic bytes per launch / HIP-event time per launch, from the same extra steps as the shade rows
        def hbm_row(kernel, keys):
            rows = [kt[k] for k in keys if k in kt]
            if not rows:
                return None
            t = sum(r["avg_ms"] * r["launches"] for r in rows) * 1e-3
            w = sum(r["work_per_launch"] * r["launches"] for r in rows)
            n = sum(r["launches"] for r in rows)
            return {"kernel": kernel, "bound": "hbm", "achieved": w / t / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": w / t / 8e12,
                    "traffic": None, "avg_us": t / n * 1e6, "launches_timed": n, "algorithmic_bytes": w / n, "gated": False}
        for nm, kernel, keys in (("roofline_rasterize", "k_vertex_transform + k_bin + k_raster_fine", ["rasterize"]),
                                 ("roofline_antialias", "k_aa_plan + k_aa_apply<C> + k_aa_grad<C>",
                                  [k for k in kt if k.startswith("antialias")]),
                                 ("roofline_hashgrid_fwd", "k_hashgrid<false>", ["hashgrid_fwd"]),
                                 ("roofline_hashgrid_bwd", "k_hashgrid_bwd_lds (dense levels) + k_hg_bin / k_hg_acc (hashed levels; k_hg_sum only when pass 2 is split)", [k for k in kt if k.startswith("hashgrid_bwd")])):
            row = hbm_row(kernel, keys)
            if row:
                res[nm] = row
        # noise-prediction parity of every precision class the line can run in (full-size SD-2.1 UNet + ControlNet against the fp32
        # oracle; written by tests/test_hip_gpu.py::test_full_size_sd21_unet_controlnet_eps_vs_oracle on a GPU box, committed)
        for par in ("r06_full_size_eps_parity.json", "r05_full_size_eps_parity.json"):
            try:
                pj = json.load(open(os.path.join(ROOT, "profiles", par)))
                res["noise_pred_rel_fp32"] = dict({k: pj[k] for k in pj if k.endswith("_rel_max") or k.endswith("_rel_mean")},
                                                  this_line=a.dtype + ("+fp8attn" if a.attention == "fp8" else ""),
                                                  source="profiles/" + par + " (tests/test_hip_gpu.py::"
                                                  "test_full_size_sd21_unet_controlnet_eps_vs_oracle)")
                break
            except (OSError, KeyError, ValueError):
                continue
        if second_leg is not None:
            res[other + "_leg"] = second_leg
        if calib is not None:
            res["box_calibration"] = calib
        if world == 1 and not a.no_cpu_baseline:
            import signal

            def _alarm(signum, frame):
                raise TimeoutError(f"cpu baseline exceeded {a.cpu_baseline_timeout}s on {effective_cores()} cores")
            try:   # the baseline is reporting only: bounded, and never allowed to kill the bench line
                signal.signal(signal.SIGALRM, _alarm)
                signal.alarm(a.cpu_baseline_timeout)
                res["cpu_baseline"] = cpu_baseline(a, system)
            except BaseException as e:
                res["cpu_baseline"] = {"value": None, "unit": "steps/s", "cores": effective_cores(), "kind": "port",
                                       "sample": "not completed", "error": repr(e)}
            finally:
                signal.alarm(0)
    # the JSON line must be the LAST thing on stdout: RCCL printf()s "Librccl path : ..." into the C stdio buffer, which a pipe
    # only flushes at exit (after python's own print) -- tear the process group down and flush C stdio first
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)            # every rank: whatever sits in C stdio goes out now ...
    if use_dist:
        dist.barrier()                        # ... before rank 0 prints the line
    if rank == 0:
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
