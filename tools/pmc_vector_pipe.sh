#!/bin/bash
# Round 5, "the vector pipe": SQ counters of the GEGLU projection (M = 98304, K = 320, N = 2560) on the C ABI driver, this tree against
# the tree in _ab_old/ (git archive of the commit before the epilogue change, built there).  Counters in their own passes with
# --kernel-trace only.  -> gpurun_out/pmc_vp/{new,old}_{sq,sq2}.json
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_vp; mkdir -p $OUT
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
SQ2="SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM"
for which in new old; do
  if [ $which = new ]; then bin=$R/tools/_abi_pmc; else bin=$R/_ab_old/tools/_abi_pmc; fi
  [ -x $bin ] || continue
  for pass in sq sq2; do
    if [ $pass = sq ]; then ctr=$SQ; else ctr=$SQ2; fi
    rm -rf /tmp/pmc_vp_${which}_$pass
    DREAMMAT_GEMM_TILE=512 timeout 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_vp_${which}_$pass -- $bin gemm 98304 320 2560 1 5 > $OUT/${which}_$pass.log 2>&1 < /dev/null
    csv=$(find /tmp/pmc_vp_${which}_$pass -name "*counter_collection.csv" 2>/dev/null | head -1)
    if [ -n "$csv" ]; then python3 $R/tools/pmc_summarize.py "$csv" > $OUT/${which}_$pass.json; else echo "no counter csv" >> $OUT/${which}_$pass.log; fi
  done
  DREAMMAT_GEMM_TILE=512 $bin gemm 98304 320 2560 1 10 > $OUT/${which}_time.txt 2>&1
done
cat $OUT/new_time.txt $OUT/old_time.txt
