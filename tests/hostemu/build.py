"""Builds tests/hostemu/_build/libdm_hostemu.so (TEST INFRASTRUCTURE; host-only run of the product's
__host__ __device__ core headers)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libdm_hostemu.so")


def build(force=False):
    src = os.path.join(HERE, "hostemu.hip")
    deps = [src] + [os.path.join(HERE, "..", "..", "dreammat_amd", "csrc", h)
                    for h in ("raster_core.h", "shade_core.h", "bvh_core.h", "grid_core.h", "mc_shade_core.h", "dm_common.h")]
    os.makedirs(OUT, exist_ok=True)
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared",
                           "-ffp-contract=off", src, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(True))
