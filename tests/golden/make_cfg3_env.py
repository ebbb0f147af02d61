"""Golden fixture for the BENCH configuration (BASELINE configs[2], "cfg3"): the ORACLE's envlight restatement run on the
first of the five seeded synthetic lat-long probes bench.py uses (bench.synthetic_latlong(0): 256 x 512 log-normal sky + one
sun lobe) at the bench's settings (environment_scale 2.0, cube resolutions 16..128).  It pins the product's
GPU-side prefilter at the bench configuration; the render-parity test then hands the oracle the product's own (fp32,
unpacked) cubes for the other four probes, so that only ONE O(res^4) CPU prefilter (~4 min) has to be stored.
Run from the repo root:   python tests/golden/make_cfg3_env.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import envlight as oenv  # noqa: E402
import bench  # noqa: E402

if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    env = oenv.EnvLight(bench.synthetic_latlong(0), scale=2.0, min_res=16, max_res=128)
    out = {f"spec{i}": m.numpy().astype(np.float32) for i, m in enumerate(env.specular)}
    out["diffuse"] = env.diffuse.numpy().astype(np.float32)
    out["base"] = env.base.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(here, "cfg3_env0.npz"), **out)
    print({k: v.shape for k, v in out.items()})
