"""How far is the product's cube-seam rule from nvdiffrast's documented one?  (CPU only, oracle only.)

The oracle (and the product's atlas borders) replace a bilinear tap that leaves its cube face by the nearest texel along the
tap's direction (oracle/envlight.py SEAM_MODE "nearest").  nvdiffrast filters across faces: an edge-crossing tap reads the
adjacent face's edge texel with the same index along the shared edge, a corner tap is dropped and the other three weights are
renormalised (SEAM_MODE "edge_wrap").  This renders BASELINE configs[1] (apple.obj, 4 views @512^2, the real HDR probe at scale
2.0, the real FG LUT, seeded random field) with the oracle under both rules and reports the difference of every renderer output.
Usage: python tools/seam_delta.py [--res 512] [--views 4]      -> profiles/r04_seam_delta_cfg2.json
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import envlight as oenv, field as ofield, raster as oraster, render as orender   # noqa: E402
from dreammat_amd import envlight as penv, mesh as pmesh                                    # noqa: E402  (mesh loader + LUT reader only)
from tests import util                                                                      # noqa: E402

ASSETS = os.path.join(ROOT, "tests", "golden", "assets")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--views", type=int, default=4)
    a = ap.parse_args()
    torch.manual_seed(0)
    m = pmesh.load_obj(os.path.join(ASSETS, "apple.obj")) if hasattr(pmesh, "load_obj") else None
    from dreammat_amd.geometry import DreamMatMesh
    geom = DreamMatMesh({"shape_init": "mesh:" + os.path.join(ASSETS, "apple.obj"), "shape_init_params": 0.7,
                         "shape_init_mesh_up": "+y", "shape_init_mesh_front": "+z"})
    md = dict(v_pos=geom.v_buffer.cpu().numpy(), v_nrm=geom.vnrm_buffer.cpu().numpy(),
              t_pos_idx=geom.t_buffer.cpu().numpy().astype(np.int32))
    md["opp"] = oraster.build_topology(md["t_pos_idx"])
    lv, tot = ofield.grid_levels()
    table = (torch.rand(tot, 2) * 2 - 1)
    w1, w2 = torch.randn(64, 32) * 0.3, torch.randn(5, 64) * 0.3
    gz = np.load(os.path.join(ROOT, "tests", "golden", "cfg2_env.npz"))
    env = oenv.EnvLight.__new__(oenv.EnvLight)
    env.specular = [torch.from_numpy(gz[f"spec{i}"]) for i in range(4)]
    env.diffuse = torch.from_numpy(gz["diffuse"])
    env.base = torch.from_numpy(gz["base"])
    fg = penv.load_fg_lut(os.path.join(ASSETS, "bsdf_256_256.bin"))
    B, H, W = a.views, a.res, a.res
    batch = util.make_views(B, H, W, seed=0)
    batch["env_id"] = torch.zeros(B, dtype=torch.long)
    g = torch.Generator().manual_seed(11)
    ju, jn = torch.rand(B, H, W, generator=g), torch.randn(B, H, W, generator=g)
    outs = {}
    for mode in ("nearest", "edge_wrap"):
        oenv.SEAM_MODE = mode
        with torch.no_grad():
            outs[mode] = orender.render(md, batch, dict(table=table, w1=w1, w2=w2, levels=lv, radius=1.0), [env], fg, ju, jn)
    oenv.SEAM_MODE = "nearest"
    cov = outs["nearest"]["opacity"] > 0
    res = {"config": f"BASELINE configs[1]: apple.obj, {B} views @{H}^2, mud_road_puresky_1k.hdr x 2.0, real FG LUT",
           "covered_pixels": int(cov.sum()), "outputs": {}}
    for k in ("comp_rgb", "specular_light", "diffuse_light"):
        d = (outs["nearest"][k] - outs["edge_wrap"][k]).abs()
        per_px = d.max(dim=-1).values
        res["outputs"][k] = {"max_abs_delta": float(d.max()), "mean_abs_delta_over_covered": float(d.sum() / (3 * cov.sum())),
                             "pixels_above_1e-3": int((per_px > 1e-3).sum()), "pixels_above_1e-4": int((per_px > 1e-4).sum()),
                             "pixels_changed": int((per_px > 0).sum())}
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r04_seam_delta_cfg2.json"), "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
