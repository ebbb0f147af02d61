"""Round-4 shade probe on the MI355X (one process, interleaved rounds, HIP events on the launch stream).

What it separates, on the bench scene's REAL G-buffer (8 views @512^2 of the 50 880-triangle sphere, 5 environments @128):
  * G-buffer row order: "row" (rounds 1-3, the reference's x[selector] order) vs "tile" (8 x 8 pixel blocks per wave);
  * the work distribution of the shade kernels: round 3's grid-stride loop (a library built from the round-3 source,
    dreammat_amd/csrc/_obj/r3shade/libdreammat_hip.so, if present) vs round 4's balanced per-XCD batches, at 2 / 3 resident
    workgroups per CU;
  * fixed cost vs streaming rate: the same kernels on the first N/8, N/4, N/2, N rows.
The two libraries run the same per-pixel arithmetic (under -ffast-math, so equal to rounding: asserted at 1e-5).
Usage: python tools/r4_shade_probe.py [--rounds 5] [--iters 20]      -> gpurun_out/r4_shade_probe.jsonl
"""
import argparse
import ctypes
import json
import os
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
    for f in fns.values():
        f(); f()
    torch.cuda.synchronize()
    t = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            t[k].append(timed(f, iters))
    return {k: float(np.median(v)) for k, v in t.items()}


def load_r3():
    p = os.path.join(ROOT, "dreammat_amd", "csrc", "_obj", "r3shade", "libdreammat_hip.so")
    if not os.path.exists(p):
        return None
    L = ctypes.CDLL(p)
    for nm in ("dm_shade_fwd", "dm_shade_bwd"):
        res, args = _lib._SIGS[nm]
        getattr(L, nm).restype, getattr(L, nm).argtypes = res, args
    return L


def case_section(a):
    """replay of the bench's REAL in-step shade inputs (bench.py --dump-shade): what regime is the step in?"""
    c = torch.load(a.case)
    N = c["feat"].shape[0]
    soa = lambda t: t.t().contiguous().to(dev)                  # [N,C] -> [C,N] storage
    feat, nrm, view = soa(c["feat"].float()), soa(c["nrm"].float()), soa(c["view"].float())
    pix, env_of_view = c["pix_idx"].to(dev), c["env_of_view"].to(dev)
    n_dev = torch.tensor([N], dtype=torch.int32, device=dev)
    spec, diff, fg = c["spec_packed"].to(dev), c["diff_packed"].to(dev), c["fg_lut"].to(dev)
    pairs = penv.fg_pair_table(fg).contiguous()
    st = _lib.EnvAtlasStruct()
    f = c["atlas_fields"]
    st.spec, st.diff, st.fg_lut, st.fg_pairs = spec.data_ptr(), diff.data_ptr(), fg.data_ptr(), pairs.data_ptr()
    st.spec_env_stride, st.diff_env_stride = f["spec_env_stride"], f["diff_env_stride"]
    for i in range(8):
        st.mip_off[i], st.mip_res[i] = f["mip_off"][i], f["mip_res"][i]
    st.n_mips, st.diff_res, st.lut_res = f["n_mips"], f["diff_res"], f["lut_res"]
    st.min_rough_mip, st.max_rough_mip, st.texel_format = f["min_rough_mip"], f["max_rough_mip"], f["texel_format"]
    mat = _lib.MatCfgStruct(*c["mat"])
    rough = torch.sigmoid(feat[4]) * (c["mat"][3] - c["mat"][2]) + c["mat"][2]
    n2 = st.n_mips - 2
    level = torch.where(rough < 0.5, (rough.clamp(0.08, 0.5) - 0.08) / 0.42 * n2, (rough.clamp(0.5, 1.0) - 0.5) / 0.5 + n2)
    l0 = level.floor().clamp(0, st.n_mips - 1).long()
    w = l0.reshape(-1)[: N // 64 * 64].reshape(-1, 64)
    stats = {"N": N, "feature_std_per_channel": [float(x) for x in feat.std(dim=1)], "roughness_mean": float(rough.mean()),
             "roughness_std": float(rough.std()), "mip_l0_histogram": [int((l0 == k).sum()) for k in range(st.n_mips)],
             "waves_with_mixed_l0_fraction": float((w.min(1).values != w.max(1).values).float().mean())}
    print(json.dumps({"op": "bench_case_stats", **stats}), flush=True)
    dcol = torch.randn(3, N, device=dev)
    out = torch.zeros(3, N, device=dev); dfe = torch.zeros(5, N, device=dev)
    HW, B = c["HW"], env_of_view.numel()
    fns_f, fns_b = {}, {}
    for libname, L, wg in (("r3", load_r3(), None), ("r4", _lib.lib(), "1"), ("r4", _lib.lib(), "2"), ("r4", _lib.lib(), "3"), ("r4", _lib.lib(), "4")):
        if L is None:
            continue
        key = f"{libname}{'/wg' + wg if wg else ''}"

        def fwd(L=L, wg=wg):
            os.environ["DREAMMAT_SHADE_WGPCU"] = wg if wg else ""
            if not wg:
                os.environ.pop("DREAMMAT_SHADE_WGPCU", None)
            _lib.check(L.dm_shade_fwd(ctypes.byref(st), ctypes.byref(mat), nrm.data_ptr(), 1, N, view.data_ptr(), 1, N, feat.data_ptr(), 1, N,
                                      pix.data_ptr(), env_of_view.data_ptr(), n_dev.data_ptr(), N, HW, B, out.data_ptr(), 1, N,
                                      None, None, None, None, None, None, None, hipops._stream()))

        def bwd(L=L, wg=wg):
            os.environ["DREAMMAT_SHADE_WGPCU"] = wg if wg else ""
            if not wg:
                os.environ.pop("DREAMMAT_SHADE_WGPCU", None)
            _lib.check(L.dm_shade_bwd(ctypes.byref(st), ctypes.byref(mat), nrm.data_ptr(), 1, N, view.data_ptr(), 1, N, feat.data_ptr(), 1, N,
                                      pix.data_ptr(), env_of_view.data_ptr(), n_dev.data_ptr(), N, HW, B, dcol.data_ptr(), 1, N,
                                      dfe.data_ptr(), 1, N, hipops._stream()))
        fns_f[key], fns_b[key] = fwd, bwd
    tf, tb = ab(fns_f, a.rounds, a.iters), ab(fns_b, a.rounds, a.iters)
    rows = []
    for key in fns_f:
        r = {"op": "shade_bench_case", "case": key, "N": N, "fwd_us": tf[key] * 1e6, "bwd_us": tb[key] * 1e6,
             "fwd_frac_8TBs": 56.0 * N / tf[key] / 8e12, "bwd_frac_8TBs": 76.0 * N / tb[key] / 8e12}
        rows.append(r)
        print(json.dumps(r), flush=True)
    # the same launches ONE AT A TIME (HIP events around a single launch, as bench.py's in-step timing does), hot (the previous
    # launch left everything in L2 / MALL) and cold (512 MB written in between: L2 and the 256 MB MALL hold none of the inputs,
    # the atlas or the LUT -- the state the kernel finds inside a real step, 80 ms and several GB of traffic after its last run)
    flush = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device=dev)

    def single(fn, cold, n=12):
        ts = []
        for _ in range(n):
            if cold:
                flush.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        return float(np.median(ts))
    for key in fns_f:
        r = {"op": "shade_bench_case_single_launch", "case": key, "fwd_hot_us": single(fns_f[key], False), "fwd_cold_us": single(fns_f[key], True),
             "bwd_hot_us": single(fns_b[key], False), "bwd_cold_us": single(fns_b[key], True)}
        rows.append(r)
        print(json.dumps(r), flush=True)
    os.environ.pop("DREAMMAT_SHADE_WGPCU", None)
    with open(os.path.join(OUT, "r4_shade_bench_case.jsonl"), "w") as fh:
        fh.write(json.dumps({"op": "bench_case_stats", **stats}) + "\n")
        for r in rows:
            fh.write(json.dumps(r) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--case", default=None, help="a bench.py --dump-shade file: time the kernels on the step's real inputs only")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    if a.case:
        return case_section(a)
    B, H, W = 8, 512, 512
    m = pmesh.displaced_sphere(160, 160)
    batch = util.make_views(B, H, W, seed=0)
    v = m.v_pos.to(dev); tri = m.t_pos_idx.to(dev).int().contiguous(); vn = m.v_nrm.to(dev)
    pos = hipops.vertex_transform(v, batch["mvp_mtx"].to(dev))
    rast = hipops.RasterContext(dev).rasterize(pos, tri, H, W)
    torch.manual_seed(0)
    ju, jn = torch.rand(B, H, W, device=dev), torch.randn(B, H, W, device=dev)
    gbs = {o: hipops.gbuffer_compact(rast, tri, v, vn, batch["rays_d"].to(dev), ju, jn, 0.05, order=o) for o in ("row", "tile")}
    N = gbs["row"].n
    lat = [util.synthetic_latlong(i, 256, 512) for i in range(5)]
    fg = penv.approx_fg_lut()
    mat = _lib.MatCfgStruct(0.0, 0.9, 0.1, 0.95)
    at = penv.EnvAtlas(lat, scale=2.0, min_res=16, max_res=128, fg_lut=fg, device=dev, texel="rgb18e8")
    st = at.struct
    env_of_view = torch.tensor([3, 0, 4, 1, 2, 0, 3, 1], dtype=torch.int32, device=dev)
    Lnew, Lr3 = _lib.lib(), load_r3()
    wmat = torch.randn(5, 3, device=dev) * 2.5
    ph = torch.rand(5, 1, device=dev) * 6.28
    noise_dense = torch.randn(5, B * H * W, device=dev)
    dcol_dense = torch.randn(3, B * H * W, device=dev)
    rows = []

    def emit(r):
        rows.append(r)
        print(json.dumps(r), flush=True)

    outs = {}
    for fname in ("smooth", "noise"):
        fns_f, fns_b = {}, {}
        for order, gb in gbs.items():
            pix = gb.pix_idx.long()
            feat = ((1.5 * torch.sin(wmat @ gb.pos + ph)) if fname == "smooth" else noise_dense[:, pix]).contiguous()
            dcol = dcol_dense[:, pix].contiguous()
            for libname, L, wg in (("r3", Lr3, None), ("r4", Lnew, None), ("r4", Lnew, "1"), ("r4", Lnew, "2"), ("r4", Lnew, "3"),
                                   ("r4", Lnew, "4")):
                if L is None or (wg and order == "row"):
                    continue
                key = f"{libname}{'/wg' + wg if wg else ''} {order}"
                out = torch.zeros(3, N, device=dev)
                dfe = torch.zeros(5, N, device=dev)

                def fwd(L=L, gb=gb, feat=feat, out=out, wg=wg, n_dev=gb.n_dev, n=N):
                    if wg:
                        os.environ["DREAMMAT_SHADE_WGPCU"] = wg
                    else:
                        os.environ.pop("DREAMMAT_SHADE_WGPCU", None)
                    _lib.check(L.dm_shade_fwd(ctypes.byref(st), ctypes.byref(mat), gb.nrm.data_ptr(), 1, gb.nrm.stride(0),
                                              gb.view.data_ptr(), 1, gb.view.stride(0), feat.data_ptr(), 1, feat.stride(0),
                                              gb.pix_idx.data_ptr(), env_of_view.data_ptr(), n_dev.data_ptr(), n, H * W, B,
                                              out.data_ptr(), 1, out.stride(0), None, None, None, None, None, None, None,
                                              hipops._stream()))
                    return out

                def bwd(L=L, gb=gb, feat=feat, dcol=dcol, dfe=dfe, wg=wg, n_dev=gb.n_dev, n=N):
                    if wg:
                        os.environ["DREAMMAT_SHADE_WGPCU"] = wg
                    else:
                        os.environ.pop("DREAMMAT_SHADE_WGPCU", None)
                    _lib.check(L.dm_shade_bwd(ctypes.byref(st), ctypes.byref(mat), gb.nrm.data_ptr(), 1, gb.nrm.stride(0),
                                              gb.view.data_ptr(), 1, gb.view.stride(0), feat.data_ptr(), 1, feat.stride(0),
                                              gb.pix_idx.data_ptr(), env_of_view.data_ptr(), n_dev.data_ptr(), n, H * W, B,
                                              dcol.data_ptr(), 1, dcol.stride(0), dfe.data_ptr(), 1, dfe.stride(0),
                                              hipops._stream()))
                    return dfe
                fns_f[key], fns_b[key] = fwd, bwd
                # results as dense per-pixel images, so that the two row orders compare
                dense = torch.zeros(3, B * H * W, device=dev); dense[:, pix] = fwd().clone()
                dense_b = torch.zeros(5, B * H * W, device=dev); dense_b[:, pix] = bwd().clone()
                outs[(fname, key)] = (dense, dense_b)
        base = outs[(fname, next(iter(fns_f)))]
        diffs = {}
        for key in fns_f:
            d = outs[(fname, key)]
            diffs[key] = (float((d[0] - base[0]).abs().max()), float((d[1] - base[1]).abs().max() / base[1].abs().max()))
            assert diffs[key][0] < 1e-5 and diffs[key][1] < 1e-5, f"{fname} {key}: results differ from {next(iter(fns_f))}: {diffs[key]}"
        tf, tb = ab(fns_f, a.rounds, a.iters), ab(fns_b, a.rounds, a.iters)
        for key in fns_f:
            emit({"op": "shade", "features": fname, "case": key, "N": N, "fwd_us": tf[key] * 1e6, "bwd_us": tb[key] * 1e6,
                  "fwd_frac_8TBs": 56.0 * N / tf[key] / 8e12, "bwd_frac_8TBs": 76.0 * N / tb[key] / 8e12, "max_diff_vs_first_fwd_abs_bwd_rel": diffs[key]})
    # fixed cost vs streaming rate: the round-4 kernels on a prefix of the tile-ordered rows
    gb = gbs["tile"]
    feat = (1.5 * torch.sin(wmat @ gb.pos + ph)).contiguous()
    dcol = dcol_dense[:, gb.pix_idx.long()].contiguous()
    out = torch.zeros(3, N, device=dev); dfe = torch.zeros(5, N, device=dev)
    for frac in (0.125, 0.25, 0.5, 1.0):
        n = int(N * frac)
        n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
        for libname, L in (("r3", Lr3), ("r4", Lnew)):
            if L is None:
                continue

            def fwd():
                os.environ.pop("DREAMMAT_SHADE_WGPCU", None)
                _lib.check(L.dm_shade_fwd(ctypes.byref(st), ctypes.byref(mat), gb.nrm.data_ptr(), 1, gb.nrm.stride(0),
                                          gb.view.data_ptr(), 1, gb.view.stride(0), feat.data_ptr(), 1, feat.stride(0),
                                          gb.pix_idx.data_ptr(), env_of_view.data_ptr(), n_dev.data_ptr(), n, H * W, B,
                                          out.data_ptr(), 1, out.stride(0), None, None, None, None, None, None, None, hipops._stream()))

            def bwd():
                _lib.check(L.dm_shade_bwd(ctypes.byref(st), ctypes.byref(mat), gb.nrm.data_ptr(), 1, gb.nrm.stride(0),
                                          gb.view.data_ptr(), 1, gb.view.stride(0), feat.data_ptr(), 1, feat.stride(0),
                                          gb.pix_idx.data_ptr(), env_of_view.data_ptr(), n_dev.data_ptr(), n, H * W, B,
                                          dcol.data_ptr(), 1, dcol.stride(0), dfe.data_ptr(), 1, dfe.stride(0), hipops._stream()))
            t = ab({"f": fwd, "b": bwd}, a.rounds, a.iters)
            emit({"op": "shade_prefix", "lib": libname, "order": "tile", "n": n, "fwd_us": t["f"] * 1e6, "bwd_us": t["b"] * 1e6,
                  "fwd_frac_8TBs": 56.0 * n / t["f"] / 8e12, "bwd_frac_8TBs": 76.0 * n / t["b"] / 8e12})
    os.environ.pop("DREAMMAT_SHADE_WGPCU", None)
    # the tile-ordered case for the counter passes (tools/_abi_pmc shadef, tools/pmc_r2.sh)
    from tools.r2_probe import dump_shade_case
    dump_shade_case(os.path.join(os.environ.get("DM_SHADE_CASE_DIR", "/tmp"), "shade_case_rgb18e8.bin"), at, gb.nrm, gb.view, feat, dcol,
                    gb.pix_idx[:N], env_of_view, H * W, True)
    with open(os.path.join(OUT, "r4_shade_probe.jsonl"), "w") as fh:
        for r in rows:
            fh.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
