"""Build the oracle's C restatement (oracle/raster_ref.c) into oracle/_build/.  Test infra only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libdm_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "raster_ref.c")
    os.makedirs(OUT, exist_ok=True)
    if (not force) and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(src):
        return LIB
    cmd = ["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
           "-msse2", "-mfpmath=sse", src, "-o", LIB, "-lm"]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
