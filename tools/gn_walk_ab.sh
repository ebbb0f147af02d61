#!/bin/bash
# GroupNorm statistics pass walking the tensor backwards (default) vs forwards (DREAMMAT_GN_REVERSE=0): per-kernel totals of one
# bench run each under rocprofv3 --kernel-trace --stats, same box.  -> stdout
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
for pass in 1 2; do
for m in 1 0; do
  rm -rf /tmp/gnab_$m
  (cd /tmp && DREAMMAT_GN_REVERSE=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gnab_$m -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-f16-leg > /tmp/gnab_$m.log 2>&1 < /dev/null)
  f=$(find /tmp/gnab_$m -name "*kernel_stats.csv" | head -1)
  echo "reverse=$m pass $pass: $(grep '^{"metric' /tmp/gnab_$m.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2),'ms/step')")"
  python3 - "$f" <<'PY'
import csv,sys
tot={}
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    for k in ("k_gn_stats<0>","k_gn_stats<1>","k_gn_apply<0, false>","k_gn_apply<0, true>","k_gn_apply<1, false>"):
        if k in n: tot[k]=tot.get(k,0)+float(r["TotalDurationNs"])/1e6
print("   ", {k: round(v,2) for k,v in sorted(tot.items())}, "sum", round(sum(tot.values()),2), "ms over the run")
PY
done
done
