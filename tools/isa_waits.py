"""Static look at a kernel's main loop: compiles one csrc/*.hip to gfx950 assembly with the flags of csrc/build.py and prints, for
the basic block with the most MFMAs of every kernel whose mangled name contains `pattern`, the instruction mix and the order of
LDS reads (r = ds_read, t = ds_read_b64_tr_b16), LDS writes (w), DMA / global loads (G), MFMAs (M), barriers and every s_waitcnt.
No GPU needed.  Two findings of round 3 came from exactly this view: an alias-induced `s_waitcnt vmcnt(0)` in front of transposing
reads that drained the LDS-DMA ring (profiles/r03_experiments/attn_w64_natural_v.json), and `lgkmcnt(0)` in front of every MFMA of
the attention backward.
usage: python tools/isa_waits.py dreammat_amd/csrc/attn_bwd.hip k_attn_bwd_dkvILi64E"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreammat_amd.csrc import build as hip_build  # noqa: E402


def assembly(src, defines=()):
    """defines: e.g. ("-DDM_F16",) for the IEEE-half instantiation of a net kernel (csrc/dm_elem.h)"""
    extra = hip_build.SOURCES.get(os.path.basename(src), []) + list(defines)
    out = os.path.join(tempfile.mkdtemp(), "k.s")
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + hip_build.COMMON + extra + ["-S", "--cuda-device-only", src, "-o", out]
    subprocess.check_call([c for c in cmd if c != "-fPIC"], stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def kernels(lines, pattern):
    for i, l in enumerate(lines):
        name = l.split(":")[0]
        if l.startswith("_Z") and l.rstrip().split(";")[0].rstrip().endswith(":") and pattern in name:
            end = next(j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end"))
            yield name, [x.split(";")[0].strip() for x in lines[i + 1:end] if x.split(";")[0].strip()]


def blocks(body):
    cur = ("entry", [])
    for l in body:
        if re.match(r"^\.LBB\d+_\d+:", l):
            yield cur
            cur = (l, [])
        elif not l.startswith("."):
            cur[1].append(l)
    yield cur


def tag(i):
    op = i.split()[0]
    if op.startswith("s_waitcnt"):
        return "W(" + i.split(None, 1)[1].replace(" ", "") + ")"
    if op.startswith("s_barrier"):
        return "BAR"
    if op.startswith("ds_read_b64_tr"):
        return "t"
    if op.startswith("ds_read"):
        return "r"
    if op.startswith("ds_write"):
        return "w"
    if op.startswith("buffer_load") or op.startswith("global_load"):
        return "G"
    if op.startswith("v_mfma"):
        return "M"
    return None


def main():
    src, pattern = sys.argv[1], sys.argv[2]
    for name, body in kernels(assembly(src), pattern):
        label, ins = max(blocks(body), key=lambda b: sum(x.startswith("v_mfma") for x in b[1]))
        mix = collections.Counter()
        for i in ins:
            op = i.split()[0]
            mix["mfma" if op.startswith("v_mfma") else "exp" if op.startswith("v_exp") else "accvgpr" if op.startswith("v_accvgpr")
                else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
                else "vmem" if op.startswith(("buffer_", "global_", "scratch_")) else "other"] += 1
        seq, out = [t for t in map(tag, ins) if t], []
        for t in seq:
            if out and out[-1][0] == t:
                out[-1][1] += 1
            else:
                out.append([t, 1])
        print(f"{name}\n  block {label} {len(ins)} instructions {dict(mix)}  scratch: {sum('scratch_' in i for i in ins)}")
        print("  " + " ".join(t if n == 1 else f"{t}x{n}" for t, n in out))


if __name__ == "__main__":
    main()
