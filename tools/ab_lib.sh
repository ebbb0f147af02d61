#!/bin/bash
# A/B of two builds of the library on one box (separate processes, alternating): tools/ab_lib.sh <base.so> [probe args]
base=$1; shift
mkdir -p gpurun_out
: > gpurun_out/ab_lib.jsonl
for i in 1 2; do
  for which in base new; do
    if [ $which = base ]; then export DREAMMAT_LIB=$base; else unset DREAMMAT_LIB; fi
    python tools/r2_probe.py --skip-shade --skip-hashgrid --variants w64 "$@" --out ab_tmp.json 2>/dev/null | grep '^{' | sed "s/^{/{\"lib\": \"$which\", \"pass\": $i, /" >> gpurun_out/ab_lib.jsonl
  done
done
python - <<'P'
import json
for ln in open('gpurun_out/ab_lib.jsonl'):
    r = json.loads(ln)
    print(r["lib"], r["pass"], r["B"], r["heads"], r["Sq"], r["Skv"], "%.1f TF/s  frac %.3f  err %s" % (r["TFLOPs_median"], r["frac_2p5PF"], r["max_abs_err_vs_fp32"]))
P
