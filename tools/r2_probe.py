"""Round-2 A/B probe on the MI355X (one process, interleaved rounds, HIP events on the launch stream):
  * attention: every kernel variant (dm_attention_select) x the UNet's shapes, TF/s + max error vs an fp32 reference;
  * shade: every atlas texel format x FG pair table on/off on the bench scene's REAL G-buffer (8 views @512^2 of the
    50 880-triangle sphere, 5 environments), forward + backward, GB/s of the algorithmic 56 / 76 B per pixel; the cases
    are also dumped to $DM_SHADE_CASE_DIR (default /tmp)/shade_case_<fmt>.bin for the counter passes of tools/_abi_pmc (`shadef`).
Usage: python tools/r2_probe.py [--rounds 5] [--iters 10] [--skip-attn] [--skip-shade]      -> gpurun_out/r2_probe.json
"""
import argparse
import ctypes
import json
import os
import struct
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreammat_amd import _lib, envlight as penv, hipops, mesh as pmesh   # noqa: E402
from tests import util                                                  # noqa: E402

dev = torch.device("cuda:0")
OUT = os.path.join(ROOT, "gpurun_out")


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def ab(fns, rounds, iters):
    """fns: {name: callable}; returns {name: {"median_s", "min_s"}} from `rounds` interleaved rounds."""
    for f in fns.values():
        f(); f()
    torch.cuda.synchronize()
    t = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            t[k].append(timed(f, iters))
    return {k: {"median_s": float(np.median(v)), "min_s": float(np.min(v))} for k, v in t.items()}


def attention_section(a, res):
    variants = a.variants.split(",")
    shapes = [(24, 5, 4096, 4096, 64), (24, 10, 1024, 1024, 64), (24, 20, 256, 256, 64), (24, 20, 64, 64, 64),
              (24, 5, 4096, 77, 64), (24, 10, 1024, 77, 64), (3, 5, 4096, 4096, 64), (48, 5, 16384, 16384, 64)]
    if a.quick:
        shapes = shapes[:2]
    for (B, h, Sq, Skv, D) in shapes:
        if Sq >= 16384:
            B = 4                                   # cfg5's sequence length at a batch that fits the probe's time
        C = h * D
        torch.manual_seed(0)
        q = torch.randn(B, Sq, C, device=dev, dtype=torch.bfloat16)
        k = torch.randn(B, Skv, C, device=dev, dtype=torch.bfloat16)
        pad = (Skv + 7) // 8 * 8
        vt = torch.zeros(B, C, pad, device=dev, dtype=torch.bfloat16)
        vt[:, :, :Skv] = torch.randn(B, C, Skv, device=dev, dtype=torch.bfloat16)
        # fp32 reference on one (batch, head) slice pair
        qf = q[:1].float().view(1, Sq, h, D).transpose(1, 2)[:, :2]
        kf = k[:1].float().view(1, Skv, h, D).transpose(1, 2)[:, :2]
        vf = vt[:1, :, :Skv].float().view(1, h, D, Skv)[:, :2].transpose(-1, -2)
        if Sq * Skv <= 4096 * 4096:
            ref = torch.softmax(qf @ kf.transpose(-1, -2) * D ** -0.5, -1) @ vf          # [1,2,Sq,D]
        else:
            ref = None
        fns, errs = {}, {}
        for v in variants:
            hipops.attention_select(v)
            out = hipops.attention(q, k, vt, h)
            if ref is not None:
                o = out[:1].float().view(1, Sq, h, D).transpose(1, 2)[:, :2]
                errs[v] = float((o - ref).abs().max())

            def run(v=v):
                hipops.attention_select(v)
                hipops.attention(q, k, vt, h)
            fns[v] = run
        tm = ab(fns, a.rounds, a.iters if Sq * Skv < 16384 * 16384 else max(2, a.iters // 4))
        hipops.attention_select(None)
        flops = 4.0 * B * Sq * Skv * C
        for v in variants:
            r = {"op": "attention", "variant": v, "B": B, "heads": h, "Sq": Sq, "Skv": Skv, "D": D,
                 "us_median": tm[v]["median_s"] * 1e6, "us_min": tm[v]["min_s"] * 1e6,
                 "TFLOPs_median": flops / tm[v]["median_s"] / 1e12, "frac_2p5PF": flops / tm[v]["median_s"] / 2.5e15,
                 "max_abs_err_vs_fp32": errs.get(v)}
            res.append(r)
            print(json.dumps(r), flush=True)


def dump_shade_case(path, atlas, nrm, view, feat, dcol, pix, env_of_view, HW, pairs):
    s = atlas.struct
    spec_b = atlas.spec_packed.contiguous().cpu().numpy().tobytes()
    diff_b = atlas.diff_packed.contiguous().cpu().numpy().tobytes()
    N = pix.shape[0]
    hd = [0x444d5348, N, env_of_view.shape[0], HW, s.n_mips, s.diff_res, s.lut_res, s.texel_format, s.spec_env_stride,
          s.diff_env_stride, len(spec_b), len(diff_b), 1 if pairs else 0, 0, 0, 0]
    with open(path, "wb") as fh:
        fh.write(struct.pack("<16q", *hd))
        fh.write(struct.pack("<8q", *[s.mip_off[i] for i in range(8)]))
        fh.write(struct.pack("<8i", *[s.mip_res[i] for i in range(8)]))
        for t in (nrm, view, feat, dcol):
            fh.write(t.contiguous().cpu().numpy().astype(np.float32).tobytes())
        fh.write(pix.cpu().numpy().astype(np.int32).tobytes())
        fh.write(env_of_view.cpu().numpy().astype(np.int32).tobytes())
        fh.write(spec_b)
        fh.write(diff_b)
        fh.write(atlas.fg_lut.contiguous().cpu().numpy().astype(np.float32).tobytes())
        if pairs:
            fh.write(atlas.fg_pairs.contiguous().cpu().numpy().astype(np.float32).tobytes())


def shade_section(a, res):
    B, H, W = 8, 512, 512
    m = pmesh.displaced_sphere(160, 160)
    batch = util.make_views(B, H, W, seed=0)
    v = m.v_pos.to(dev); tri = m.t_pos_idx.to(dev).int().contiguous(); vn = m.v_nrm.to(dev)
    pos = hipops.vertex_transform(v, batch["mvp_mtx"].to(dev))
    rast = hipops.RasterContext(dev).rasterize(pos, tri, H, W)
    gb = hipops.gbuffer_compact(rast, tri, v, vn, batch["rays_d"].to(dev), torch.rand(B, H, W, device=dev),
                                torch.randn(B, H, W, device=dev), 0.05)
    N = gb.n
    lat = [util.synthetic_latlong(i, 256, 512) for i in range(5)]
    fg = penv.approx_fg_lut()
    mat = _lib.MatCfgStruct(0.0, 0.9, 0.1, 0.95)
    torch.manual_seed(0)
    feat = torch.randn(5, N, device=dev)
    dcol = torch.randn(3, N, device=dev)
    dfe = torch.empty(5, N, device=dev)
    env_of_view = torch.tensor([3, 0, 4, 1, 2, 0, 3, 1], dtype=torch.int32, device=dev)
    L = _lib.lib()
    atlases = {}
    for texel in ("fp32", "rgb18e8", "fp16"):
        atlases[texel] = penv.EnvAtlas(lat, scale=2.0, min_res=16, max_res=128, fg_lut=fg, device=dev, texel=texel)
    cases = {}
    for texel, at in atlases.items():
        for pairs in (True, False):
            st = type(at.struct).from_buffer_copy(at.struct)
            if not pairs:
                st.fg_pairs = None
            cases[f"{texel}{'+pairs' if pairs else ''}"] = (at, st, pairs)
    cases = dict(sorted(cases.items(), key=lambda kv: kv[0] != "fp32"))          # "fp32" (round-1 configuration) first
    # bench-like features: the hash-grid field is a smooth function of position, so roughness (= the mip level) and albedo
    # vary slowly across a wave; `feat` above (white noise) is the incoherent worst case
    w = torch.randn(5, 3, device=dev) * 2.5
    feat_smooth = (1.5 * torch.sin(w @ gb.pos + torch.rand(5, 1, device=dev) * 6.28)).contiguous()
    tiny = penv.EnvAtlas([l[::64, ::64].contiguous() for l in lat], scale=2.0, min_res=1, max_res=2, fg_lut=fg, device=dev,
                         texel="rgb18e8")            # every gather of a wave lands in one or two cache lines: the no-divergence floor
    cases["rgb18e8+pairs tiny-atlas"] = (tiny, type(tiny.struct).from_buffer_copy(tiny.struct), True)
    for fname, ft in (("noise", feat), ("smooth", feat_smooth)):
        fns_f, fns_b, outs = {}, {}, {}
        for name, (at, st, pairs) in cases.items():
            out = torch.empty(3, N, device=dev)

            def fwd(at=at, st=st, out=out, ft=ft):
                _lib.check(L.dm_shade_fwd(ctypes.byref(st), ctypes.byref(mat), gb.nrm.data_ptr(), 1, gb.nrm.stride(0),
                                          gb.view.data_ptr(), 1, gb.view.stride(0), ft.data_ptr(), 1, N, gb.pix_idx.data_ptr(),
                                          env_of_view.data_ptr(), gb.n_dev.data_ptr(), N, H * W, B, out.data_ptr(), 1, N,
                                          None, None, None, None, None, None, None, hipops._stream()))
                return out

            def bwd(at=at, st=st, ft=ft):
                _lib.check(L.dm_shade_bwd(ctypes.byref(st), ctypes.byref(mat), gb.nrm.data_ptr(), 1, gb.nrm.stride(0),
                                          gb.view.data_ptr(), 1, gb.view.stride(0), ft.data_ptr(), 1, N, gb.pix_idx.data_ptr(),
                                          env_of_view.data_ptr(), gb.n_dev.data_ptr(), N, H * W, B, dcol.data_ptr(), 1, N,
                                          dfe.data_ptr(), 1, N, hipops._stream()))
            fns_f[name], fns_b[name] = fwd, bwd
            outs[name] = fwd().clone()
        base = outs["fp32"]
        tf, tb = ab(fns_f, a.rounds, a.iters), ab(fns_b, a.rounds, a.iters)
        for name in cases:
            r = {"op": "shade", "features": fname, "case": name, "N": N, "fwd_us": tf[name]["median_s"] * 1e6,
                 "bwd_us": tb[name]["median_s"] * 1e6, "fwd_GBps": 56.0 * N / tf[name]["median_s"] / 1e9,
                 "bwd_GBps": 76.0 * N / tb[name]["median_s"] / 1e9, "fwd_frac_8TBs": 56.0 * N / tf[name]["median_s"] / 8e12,
                 "bwd_frac_8TBs": 76.0 * N / tb[name]["median_s"] / 8e12,
                 "max_abs_diff_vs_fp32_atlas": float((outs[name] - base).abs().max())}
            res.append(r)
            print(json.dumps(r), flush=True)
    for texel in ("fp32", "rgb18e8"):
        at = atlases[texel]
        dump_shade_case(os.path.join(os.environ.get("DM_SHADE_CASE_DIR", "/tmp"), f"shade_case_{texel}.bin"), at, gb.nrm, gb.view,
                        feat_smooth, dcol, gb.pix_idx[:N], env_of_view, H * W, texel != "fp32")


def hashgrid_section(a, res):
    """backward of the full 16-level grid on the bench scene's 2 x N sample points: atomic route vs binned route."""
    B, H, W = 8, 512, 512
    m = pmesh.displaced_sphere(160, 160)
    batch = util.make_views(B, H, W, seed=0)
    v = m.v_pos.to(dev); tri = m.t_pos_idx.to(dev).int().contiguous(); vn = m.v_nrm.to(dev)
    pos = hipops.vertex_transform(v, batch["mvp_mtx"].to(dev))
    rast = hipops.RasterContext(dev).rasterize(pos, tri, H, W)
    gb = hipops.gbuffer_compact(rast, tri, v, vn, batch["rays_d"].to(dev), torch.rand(B, H, W, device=dev),
                                torch.randn(B, H, W, device=dev), 0.05)
    pts2 = torch.cat([gb.pos, gb.pos_jitter], dim=1).t()
    M = pts2.shape[0]
    spec = hipops.GridSpec()
    dt = torch.zeros(spec.n_params, device=dev)
    dy = torch.randn(2 * spec.n_levels, M, device=dev)
    L = _lib.lib()
    ws_bytes = int(L.dm_hashgrid_bwd_workspace_bytes(M, spec.n_levels, spec.c_res, spec.c_size))
    ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=dev)

    def atomic():
        _lib.check(L.dm_hashgrid_bwd(pts2.data_ptr(), pts2.stride(0), pts2.stride(1), None, M, dy.data_ptr(), 1, M, spec.n_levels,
                                     spec.c_scale, spec.c_res, spec.c_size, spec.c_offset, 1.0, dt.data_ptr(), hipops._stream()))

    def binned():
        _lib.check(L.dm_hashgrid_bwd_binned(pts2.data_ptr(), pts2.stride(0), pts2.stride(1), None, M, dy.data_ptr(), 1, M,
                                            spec.n_levels, spec.c_scale, spec.c_res, spec.c_size, spec.c_offset, 1.0,
                                            dt.data_ptr(), ws.data_ptr(), ws.numel(), hipops._stream()))
    tm = ab({"atomic": atomic, "binned": binned}, a.rounds, max(2, a.iters // 3))
    for k, t in tm.items():
        r = {"op": "hashgrid_bwd", "route": k, "M": M, "ms_median": t["median_s"] * 1e3, "workspace_GB": ws_bytes / 1e9,
             "corner_updates_M": M * 16 * 8 / 1e6}
        res.append(r)
        print(json.dumps(r), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--skip-attn", action="store_true")
    ap.add_argument("--skip-shade", action="store_true")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--skip-hashgrid", action="store_true")
    ap.add_argument("--variants", default="auto,w64,v3l,staged", help="attention kernels to compare (dm_attention_select names)")
    ap.add_argument("--out", default="r2_probe.json")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    res = []
    if not a.skip_hashgrid:
        hashgrid_section(a, res)
    if not a.skip_shade:
        shade_section(a, res)
    if not a.skip_attn:
        attention_section(a, res)
    with open(os.path.join(OUT, a.out), "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
