#!/bin/bash
# step time of this tree against an older copy of it (git archive <rev> into _ab_old/, built there) on ONE box, alternating:
#   tools/ab_step.sh [bench args]   -> gpurun_out/ab_step.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
: > $R/gpurun_out/ab_step.txt
for pass in $(seq 1 ${AB_PASSES:-3}); do
  for which in old new; do
    if [ $which = old ]; then d=$R/_ab_old; extra="${AB_OLD_ARGS:-}"; else d=$R; extra="${AB_NEW_ARGS:-}"; fi      # AB_NEW_ARGS / AB_OLD_ARGS: flags only one tree knows
    (cd $d && timeout 200 python bench.py --steps 15 --warmup 3 --no-cpu-baseline $extra "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$which', $pass, 'steps/s %.3f  ms %.2f  conv %.3f attn %.3f  shade %.3f / %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_attention']['frac'], d['roofline_shade_fwd']['frac'], d['roofline_shade_bwd']['frac']))") >> $R/gpurun_out/ab_step.txt
  done
done
cat $R/gpurun_out/ab_step.txt
