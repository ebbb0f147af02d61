"""Merge the per-pass counter summaries of tools/pmc_r2.sh (gpurun_out/pmc_r2/*.json) into ONE committed file:
usage: pmc_collect.py gpurun_out/pmc_r2 profiles/r02_pmc_final.json
-> {"counters": {pass_name: {kernel: {counter: mean per dispatch}}}, "kernel_stats": {name: [rows of *_kernel_stats.csv]}}"""
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
out = {"counters": {}, "kernel_stats": {}, "note": "rocprofv3 --pmc passes (own runs, --kernel-trace only) on tools/_abi_pmc; "
       "FETCH_SIZE / WRITE_SIZE in KiB; GRBM_GUI_ACTIVE is summed over the 8 XCDs"}
for f in sorted(glob.glob(os.path.join(src, "*.json"))):
    try:
        d = json.load(open(f))
    except ValueError:
        continue
    out["counters"][os.path.basename(f)[:-5]] = {k: {c: v["mean"] for c, v in ctr.items()} for k, ctr in d.items()}
for f in sorted(glob.glob(os.path.join(src, "stats_*.csv"))):
    rows = list(csv.DictReader(open(f)))
    out["kernel_stats"][os.path.basename(f)[6:-4]] = [
        {"name": r["Name"][:120], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3} for r in rows[:4]]
# provenance: the tree the counters were taken on (pmc_collect runs where .git is) -- bench.py copies it into the JSON line so
# that a stale counter file is visible
import subprocess
import time
try:
    out["collected_at_commit"] = subprocess.run(["git", "log", "-1", "--format=%h %cs"], capture_output=True, text=True,
                                                 cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
except Exception:
    out["collected_at_commit"] = None
out["collected_on"] = time.strftime("%Y-%m-%d")
json.dump(out, open(dst, "w"), indent=1)
print(dst, len(out["counters"]), "passes")
