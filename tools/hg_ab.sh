#!/bin/bash
# k_hg_bin / k_hg_acc / k_hashgrid_bwd_lds per-kernel times (rocprofv3 --kernel-trace --stats) of tools/hashgrid_bwd_time.py under library variants
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/dreammat_amd/csrc/_obj
export TMPDIR=/tmp
for v in main "$@"; do
  rm -rf /tmp/hgprof_$v
  if [ $v = main ]; then L=""; else L="DREAMMAT_LIB=$O/$v/libdreammat_hip.so"; fi
  (cd /tmp && env $L PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hgprof_$v -- python $R/tools/hashgrid_bwd_time.py > /dev/null 2>&1)
  f=$(find /tmp/hgprof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep -E "k_hg_|k_hashgrid" $f | awk -F, '{printf "%-60s calls %s avg_us %.1f\n", substr($1,1,60), $2, $4/1000}'
done
