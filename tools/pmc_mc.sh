#!/bin/bash
# Counter passes on the Monte-Carlo shading forward kernel (row f-1) through the C ABI driver: tools/_abi_pmc mc.
# -> gpurun_out/pmc_mc/   (summaries; copy what is to be judged into profiles/)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_mc; mkdir -p $OUT
ARGS="mc 100000 160 160 grid 3"
$R/tools/_abi_pmc $ARGS > $OUT/plain.log 2>&1; cat $OUT/plain.log
$R/tools/_abi_pmc mc 100000 160 160 bvh 2 >> $OUT/plain.log 2>&1; tail -1 $OUT/plain.log
rm -rf /tmp/st_mc; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_mc -- $R/tools/_abi_pmc $ARGS > $OUT/stats.log 2>&1 < /dev/null
f=$(find /tmp/st_mc -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
           "SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  name=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pmc_mc_$name
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_mc_$name -- $R/tools/_abi_pmc $ARGS > $OUT/$name.log 2>&1 < /dev/null
  csv=$(find /tmp/pmc_mc_$name -name "*counter_collection.csv" 2>/dev/null | head -1)
  [ -n "$csv" ] && python3 $R/tools/pmc_summarize.py "$csv" > $OUT/$name.json
done
ls $OUT
