#!/bin/bash
# Hardware counters of the conv kernel on the C ABI (no python in the profiled process).  Usage: tools/pmc_conv.sh
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_conv; mkdir -p $OUT
run() {
  name=$1; ctr=$2; shift 2
  rm -rf /tmp/pmc_$name
  timeout 150 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$name -- $R/tools/_conv_pmc "$@" > $OUT/$name.log 2>&1 < /dev/null
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_summarize.py "$f" > $OUT/$name.json; else echo "no counter csv" >> $OUT/$name.log; fi
}
for shape in "8 512 512 128 128" "8 256 256 256 256" "24 64 64 320 320"; do
  tag=$(echo $shape | tr ' ' '_')
  run fetch_$tag "FETCH_SIZE" $shape 3
  run write_$tag "WRITE_SIZE" $shape 3
  run tcc_$tag "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" $shape 3
  run sq_$tag "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" $shape 3
done
ls -la $OUT | head -40
