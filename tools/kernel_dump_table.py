#!/usr/bin/env python3
"""Per-key table of a `bench.py --dump-kernels FILE` dump (HIP-event timers of hipops._Timed): ms per step, launches per step,
average us and -- for the MFMA families -- TFLOP/s and the fraction of the dense 16-bit peak.  Largest first.

    python tools/kernel_dump_table.py gpurun_out/kernels.json [prefix ...]
"""
import json
import sys


def main():
    d = json.load(open(sys.argv[1]))
    prefixes = tuple(sys.argv[2:])
    steps = d["steps"]
    rows = []
    for k, v in d["kernels"].items():
        if prefixes and not k.startswith(prefixes):
            continue
        ms_step = v["avg_ms"] * v["launches"] / steps
        mfma = k.startswith(("conv3x3", "gemm", "attention"))
        tf = v["work_per_launch"] / (v["avg_ms"] * 1e-3) / 1e12 if mfma and v["avg_ms"] > 0 else None
        rows.append((ms_step, k, v["launches"] / steps, v["avg_ms"] * 1e3, tf))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"# {len(rows)} keys, {tot:.2f} ms/step in total ({steps} steps)")
    print(f"{'ms/step':>8} {'n/step':>7} {'avg us':>8} {'TF/s':>7} {'frac':>5}  key")
    for ms_step, k, n, us, tf in rows:
        peak = 5000.0 if "fp8" in k else 2500.0
        print(f"{ms_step:8.3f} {n:7.1f} {us:8.1f} " + (f"{tf:7.0f} {tf / peak:5.2f}" if tf else f"{'':7} {'':5}") + f"  {k}")


if __name__ == "__main__":
    main()
