"""Golden fixture for BASELINE configs[1] (cfg2): the ORACLE's envlight restatement (oracle/envlight.py: lat-long -> cube,
2x2 mip chain, GGX-prefiltered specular mips, cosine-convolved diffuse cube) run on the reference's own HDR probe
(tests/golden/assets/mud_road_puresky_1k.hdr) with the reference's settings (environment_scale 2.0, dreammat.yaml:85;
envlight's upstream resolutions 16..128).  The prefilter is O(res^4) on the CPU (~4 min at 128 on 8 cores), too slow for
a test, so its output is stored once: tests/golden/cfg2_env.npz.  Run from the repo root:
    python tests/golden/make_cfg2_env.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import envlight as oenv  # noqa: E402

if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    lat = torch.from_numpy(oenv.load_hdr(os.path.join(here, "assets", "mud_road_puresky_1k.hdr")))
    env = oenv.EnvLight(lat, scale=2.0, min_res=16, max_res=128)
    out = {f"spec{i}": m.numpy().astype(np.float32) for i, m in enumerate(env.specular)}
    out["diffuse"] = env.diffuse.numpy().astype(np.float32)
    out["base"] = env.base.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(here, "cfg2_env.npz"), **out)
    print({k: v.shape for k, v in out.items()})
