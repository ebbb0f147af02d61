#!/bin/bash
# round 6, first look: where a tile of the two dominant conv shapes spends its cycles, and the box's baseline step
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for s in "8 512 512 128 128" "8 256 256 256 256" "8 128 128 512 512"; do
  echo "== $s"; tools/_abi_pmc conv $s 10
  for m in 1 4; do echo "-- timeline $m"; DREAMMAT_CONV_TIMELINE=$m tools/_abi_pmc conv $s 2 2>&1 | grep -A9 "wg 0" | tail -10 | cut -c1-900; done
done
echo "== gemm 98304 320 320"; tools/_abi_pmc gemm 98304 320 320 0 10 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-f16-leg 2>/dev/null | grep '^{"metric' | cut -c1-400
