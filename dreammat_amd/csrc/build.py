"""Builds dreammat_amd/libdreammat_hip.so for gfx950 with hipcc (in-tree, cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB = os.path.join(PKG, "libdreammat_hip.so")
OBJ = os.path.join(HERE, "_obj")

COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
SOURCES = {
    # the rasterizer's bit-exactness contract needs one rounding per written operation
    "raster.hip": ["-ffp-contract=off"],
    # shading tolerates approximate div/exp/rcp (1e-3 rel budget, results stay within 1e-5 of the oracle)
    "shade.hip": ["-munsafe-fp-atomics", "-ffast-math"],
    "hashgrid.hip": ["-munsafe-fp-atomics"],
    "field_mlp.hip": ["-munsafe-fp-atomics"],
    # MFMA results stay in VGPRs: the softmax consumes every S element with VALU ops, and the AGPR form cost
    # 127 v_accvgpr_read/write per KV tile
    "attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "attn_w64.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "attn_w128.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "attn_bwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "attn_fp8.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "conv.hip": [],
    "conv_small.hip": [],
    "conv_wgrad.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "groupnorm.hip": [],
    "transformer.hip": [],
    "adam.hip": [],
    "bvh.hip": [],
    "mc_shade.hip": [],
    "host.cpp": [],
}
# the net kernels are compiled a second time with -DDM_F16 (dm_elem.h): IEEE-half instantiations, exported with f16 in their names
F16_SOURCES = ["conv.hip", "conv_small.hip", "groupnorm.hip", "transformer.hip", "attention.hip", "attn_w64.hip", "attn_w128.hip"]
HEADERS = ["dm_common.h", "dm_elem.h", "attn_common.h", "raster_core.h", "shade_core.h", "bvh_core.h", "grid_core.h", "mc_shade_core.h"]


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hdr_time = max(os.path.getmtime(os.path.join(HERE, h)) for h in HEADERS)
    objs, jobs = [], []
    for src, extra in SOURCES.items():
        s = os.path.join(HERE, src)
        o = os.path.join(OBJ, src + ".o")
        if force or _newer(s, o) or hdr_time > os.path.getmtime(o):
            jobs.append([hipcc] + COMMON + extra + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", s, "-o", o])
        objs.append(o)
        if src in F16_SOURCES:
            o16 = os.path.join(OBJ, src + ".f16.o")
            if force or _newer(s, o16) or hdr_time > os.path.getmtime(o16):
                jobs.append([hipcc] + COMMON + extra + ["-DDM_F16", "-c", s, "-o", o16])
            objs.append(o16)
    rebuilt = bool(jobs)
    if jobs:
        # the translation units are independent: compile them side by side (a full build is ~17 files of 5-40 s each)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 4) // 2))) as pool:
            list(pool.map(run, jobs))
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
