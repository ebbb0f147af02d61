#!/bin/bash
# Hardware counters + clean kernel-trace statistics of the hot kernels on the C ABI (tools/abi_pmc.cpp; no python in
# the profiled process: `rocprofv3 --pmc` segfaults under python + torch in this image).  Results: gpurun_out/pmc_abi/.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_abi; mkdir -p $OUT
run_pmc() {   # name counters args...
  name=$1; ctr=$2; shift 2
  rm -rf /tmp/pmc_$name
  timeout 150 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$name -- $R/tools/_abi_pmc "$@" > $OUT/$name.log 2>&1 < /dev/null
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_summarize.py "$f" > $OUT/$name.json; else echo "no counter csv" >> $OUT/$name.log; fi
}
run_stats() { # name args...   (no counters: undisturbed durations)
  name=$1; shift
  rm -rf /tmp/st_$name
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$name -- $R/tools/_abi_pmc "$@" > $OUT/stats_$name.log 2>&1 < /dev/null
  f=$(find /tmp/st_$name -name "*kernel_stats.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" $OUT/stats_$name.csv
}
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
all_passes() { # tag args...
  tag=$1; shift
  run_stats $tag "$@"
  run_pmc fetch_$tag "FETCH_SIZE" "$@"
  run_pmc write_$tag "WRITE_SIZE" "$@"
  run_pmc tcc_$tag "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "$@"
  run_pmc sq_$tag "$SQ" "$@"
}
all_passes 8_512_512_128_128 conv 8 512 512 128 128 5
all_passes 8_256_256_256_256 conv 8 256 256 256 256 5
all_passes 24_64_64_320_320 conv 24 64 64 320 320 5
all_passes attn_24_5_4096_4096_64 attn 24 5 4096 4096 64 5
all_passes shade_1143565_5 shade 1143565 5 5
ls $OUT | wc -l
