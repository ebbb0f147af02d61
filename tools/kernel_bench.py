"""Per-kernel timing on the MI355X with HIP events (torch.cuda.Event on the current stream, which is
the stream every dm_* call is enqueued on).  Prints one JSON line per kernel: algorithmic bytes/flops,
average launch time, achieved GB/s or TFLOP/s.  Usage: python tools/kernel_bench.py [--iters N]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreammat_amd import _lib, envlight as penv, hipops, mesh as pmesh   # noqa: E402


def timeit(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3   # seconds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--skip-attn", action="store_true")
    ap.add_argument("--conv-tiles", action="store_true", help="time every conv tile variant per shape")
    ap.add_argument("--only-conv", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from tests import util
    B, H, W = a.views, a.res, a.res
    m = pmesh.displaced_sphere(160, 160)
    batch = util.make_views(B, H, W, seed=0)
    v = m.v_pos.to(dev); tri = m.t_pos_idx.to(dev).int().contiguous(); vn = m.v_nrm.to(dev)
    mvp = batch["mvp_mtx"].to(dev)
    res = []

    def rec(name, t, bytes_=None, flops=None, **kw):
        r = {"kernel": name, "ms": t * 1e3}
        if bytes_ is not None:
            r.update(alg_MB=bytes_ / 1e6, GBps=bytes_ / t / 1e9, frac_hbm_8TBs=bytes_ / t / 8e12)
        if flops is not None:
            r.update(alg_GF=flops / 1e9, TFLOPs=flops / t / 1e12, frac_mfma_2p5PF=flops / t / 2.5e15)
        r.update(kw)
        res.append(r)
        print(json.dumps(r), flush=True)

    def conv_section():
        from dreammat_amd.sd import layers
        for (Bq, Cin, Cout, Hh, Ww) in [(24, 320, 320, 64, 64), (24, 640, 640, 32, 32), (24, 1280, 1280, 16, 16),
                                        (24, 960, 320, 64, 64), (8, 128, 128, 512, 512), (8, 256, 256, 256, 256),
                                        (8, 512, 512, 64, 64)]:
            x = torch.randn(Bq, Hh, Ww, Cin, device=dev, dtype=torch.bfloat16)
            w = torch.randn(Cout, 9 * Cin, device=dev, dtype=torch.bfloat16) * 0.02
            b = torch.zeros(Cout, device=dev, dtype=torch.bfloat16)
            for tile in (["default"] + (["256", "512", "320", "640"] if a.conv_tiles else [])):
                if tile == "default":
                    os.environ.pop("DREAMMAT_CONV_TILE", None)
                else:
                    os.environ["DREAMMAT_CONV_TILE"] = tile
                rec(f"conv3x3 B{Bq} {Cin}->{Cout} @{Hh}x{Ww}" + ("" if tile == "default" else f" tile{tile}"),
                    timeit(lambda: hipops.conv3x3_nhwc(x, w, b), max(3, a.iters // 3), 2),
                    flops=2.0 * Bq * Hh * Ww * Cout * 9 * Cin)
            os.environ.pop("DREAMMAT_CONV_TILE", None)

    if a.only_conv:
        conv_section()
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/kernel_bench_conv.json", "w") as fh:
            json.dump(res, fh, indent=1)
        return

    pos = hipops.vertex_transform(v, mvp)
    ctx = hipops.RasterContext(dev)
    rast = ctx.rasterize(pos, tri, H, W, check_overflow=True)
    cov = float((rast[..., 3] > 0).float().mean())
    Nv, Nf, Ppix = v.shape[0], tri.shape[0], B * H * W
    rec("rasterize(bin+fine)", timeit(lambda: ctx.rasterize(pos, tri, H, W), a.iters),
        bytes_=B * Nv * 16 + Nf * 12 + Ppix * 16, coverage=cov)
    opp = hipops.build_topology(tri)
    rec("antialias_plan", timeit(lambda: hipops.antialias_plan(pos, tri, opp, rast), a.iters), bytes_=Ppix * (16 + 8))
    plan = hipops.antialias_plan(pos, tri, opp, rast)
    col = torch.rand(B, H, W, 3, device=dev)
    rec("antialias_apply_c3", timeit(lambda: hipops.antialias(col, plan), a.iters), bytes_=Ppix * (12 + 8 + 12))
    rays = batch["rays_d"].to(dev)
    ju, jn = torch.rand(B, H, W, device=dev), torch.randn(B, H, W, device=dev)
    gb = hipops.gbuffer_compact(rast, tri, v, vn, rays, ju, jn, 0.05)
    N = gb.n
    rec("gbuffer_compact(+sync)", timeit(lambda: hipops.gbuffer_compact(rast, tri, v, vn, rays, ju, jn, 0.05), a.iters),
        bytes_=Ppix * (16 + 12 + 8) + N * (4 + 48), N=N)
    # hash grid
    spec = hipops.GridSpec()
    table = ((torch.rand(spec.n_params, device=dev) * 2 - 1) * 1e-4).requires_grad_()
    pts2 = torch.cat([gb.pos, gb.pos_jitter], dim=1).t()
    M = pts2.shape[0]
    rec("hashgrid_fwd", timeit(lambda: hipops.hashgrid_encode(pts2, table.detach(), spec, 1.0), a.iters),
        bytes_=M * (12 + 128), gather_MB=M * 16 * 8 * 8 / 1e6, M=M)
    enc = hipops.hashgrid_encode(pts2, table, spec, 1.0)
    dy = torch.randn_like(enc)
    dt = torch.zeros_like(table)
    L = _lib.lib()

    def hg_bwd():
        _lib.check(L.dm_hashgrid_bwd(pts2.data_ptr(), pts2.stride(0), pts2.stride(1), None, M, dy.data_ptr(),
                                     dy.stride(0), dy.stride(1), spec.n_levels, spec.c_scale, spec.c_res, spec.c_size,
                                     spec.c_offset, 1.0, dt.data_ptr(), hipops._stream()))
    rec("hashgrid_bwd(atomics)", timeit(hg_bwd, a.iters), bytes_=M * (12 + 128), atomics_M=M * 16 * 8 * 2 / 1e6)
    # shade
    lat = [util.synthetic_latlong(i, 256, 512) for i in range(5)]
    atlas = penv.EnvAtlas(lat, scale=2.0, min_res=16, max_res=128, fg_lut=penv.approx_fg_lut(), device=dev)
    mat = _lib.MatCfgStruct(0.0, 0.9, 0.1, 0.95)
    feat = torch.randn(5, N, device=dev).t()
    env_of_view = torch.randint(0, 5, (B,), dtype=torch.int32, device=dev)
    rec("shade_fwd(SoA,no dbg)", timeit(lambda: hipops.shade(feat, gb.nrm.t(), gb.view.t(), gb.pix_idx, gb.n_dev,
                                                             env_of_view, atlas, mat, H * W, False), a.iters),
        bytes_=N * 56, N=N)
    rec("shade_fwd(SoA,+dbg)", timeit(lambda: hipops.shade(feat, gb.nrm.t(), gb.view.t(), gb.pix_idx, gb.n_dev,
                                                           env_of_view, atlas, mat, H * W, True), a.iters),
        bytes_=N * (56 + 68), N=N)
    dcol = torch.randn(3, N, device=dev).t()
    dfe = torch.empty(5, N, device=dev)

    def sh_bwd():
        import ctypes
        _lib.check(L.dm_shade_bwd(ctypes.byref(atlas.struct), ctypes.byref(mat), gb.nrm.data_ptr(), 1, gb.nrm.stride(0),
                                  gb.view.data_ptr(), 1, gb.view.stride(0), feat.data_ptr(), 1, N, gb.pix_idx.data_ptr(),
                                  env_of_view.data_ptr(), gb.n_dev.data_ptr(), N, H * W, B, dcol.data_ptr(), 1, N,
                                  dfe.data_ptr(), 1, N, hipops._stream()))
    rec("shade_bwd(SoA)", timeit(sh_bwd, a.iters), bytes_=N * 76, N=N)
    # adam
    n = spec.n_params + 2368
    n = (n + 3) // 4 * 4
    p, g, m1, m2 = (torch.zeros(n, device=dev) for _ in range(4))
    rec("adam_step", timeit(lambda: hipops.adam_step(p, g, m1, m2, 1, 0.01, 0.9, 0.99, 1e-15, 1.0, True), a.iters),
        bytes_=n * 4 * 8)
    if not a.skip_attn:
        for (Bq, h, Sq, Skv, D) in [(24, 5, 4096, 4096, 64), (24, 10, 1024, 1024, 64), (24, 20, 256, 256, 64),
                                    (24, 20, 64, 64, 64), (24, 5, 4096, 77, 64), (24, 8, 4096, 4096, 40),
                                    (24, 8, 1024, 1024, 80), (24, 8, 256, 256, 160)]:
            C = h * D
            q = torch.randn(Bq, Sq, C, device=dev, dtype=torch.bfloat16)
            k = torch.randn(Bq, Skv, C, device=dev, dtype=torch.bfloat16)
            pad = (Skv + 7) // 8 * 8
            vt = torch.randn(Bq, C, pad, device=dev, dtype=torch.bfloat16)
            rec(f"attention B{Bq} h{h} Sq{Sq} Skv{Skv} D{D}", timeit(lambda: hipops.attention(q, k, vt, h), a.iters),
                flops=4.0 * Bq * Sq * Skv * C)
    if not a.skip_attn:
        conv_section()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/kernel_bench.json", "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
