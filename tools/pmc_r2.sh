#!/bin/bash
# Round-2 counter passes on the C ABI driver (no python in the profiled process): attention variants at S = 4096 and the
# shade kernels on the bench scene's REAL G-buffer dumped by tools/r2_probe.py (/tmp/shade_case_<fmt>.bin on the GPU box).
# Counters in their own passes with --kernel-trace only (never combined with other trace domains).  -> gpurun_out/pmc_r2/
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_r2; mkdir -p $OUT
run_pmc() {   # name counters args...
  name=$1; ctr=$2; shift 2
  rm -rf /tmp/pmc_$name
  timeout 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$name -- $R/tools/_abi_pmc "$@" > $OUT/$name.log 2>&1 < /dev/null
  local csv=$(find /tmp/pmc_$name -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$csv" ]; then python3 $R/tools/pmc_summarize.py "$csv" > $OUT/$name.json; else echo "no counter csv" >> $OUT/$name.log; fi
}
run_stats() { # name args...
  name=$1; shift
  rm -rf /tmp/st_$name
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$name -- $R/tools/_abi_pmc "$@" > $OUT/stats_$name.log 2>&1 < /dev/null
  local csv=$(find /tmp/st_$name -name "*kernel_stats.csv" 2>/dev/null | head -1)
  [ -n "$csv" ] && cp "$csv" $OUT/stats_$name.csv
}
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
SQ2="SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM"
rocprofv3 -L > $OUT/counters_available.txt 2>&1
SECTIONS=${PMC_SECTIONS:-attn shade}
if [[ " $SECTIONS " == *" conv "* ]]; then
  # the dominant conv shapes: 8 x 128->128 @512^2 (VAE, tile 640) and 24 x 320->320 @64^2 (UNet, tile 320)
  # (round 6: "-" = the dispatcher's own choice -- the halo-patch kernel for the first and third shape; a forced tile turns it off)
  for case in "- 8 512 512 128 128" "- 8 128 128 512 512" "320 24 64 64 320 320" "256 24 32 32 640 640"; do
    set -- $case; if [ "$1" != "-" ]; then export DREAMMAT_CONV_TILE=$1; fi; shift 1
    nm=$(echo "$@" | tr ' ' '_')              # conv_<B>_<H>_<W>_<Cin>_<Cout>: the tag bench.py's pmc_traffic() looks up
    run_stats conv_$nm conv "$@" 10
    run_pmc sq_conv_$nm "$SQ" conv "$@" 5
    run_pmc sq2_conv_$nm "$SQ2" conv "$@" 5
    run_pmc grbm_conv_$nm "GRBM_GUI_ACTIVE GRBM_COUNT" conv "$@" 5
    run_pmc fetch_conv_$nm "FETCH_SIZE" conv "$@" 5
    run_pmc write_conv_$nm "WRITE_SIZE" conv "$@" 5
    run_pmc tcc_conv_$nm "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" conv "$@" 5
    run_pmc tcp_conv_$nm "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" conv "$@" 5
    unset DREAMMAT_CONV_TILE
  done
fi
if [[ " $SECTIONS " == *" attn "* ]]; then
for v in ${ATTN_VARIANTS:-v3p v3l v3}; do
  run_stats attn_$v attn 24 5 4096 4096 64 5 $v
  run_pmc sq_attn_$v "$SQ" attn 24 5 4096 4096 64 5 $v
  run_pmc sq2_attn_$v "$SQ2" attn 24 5 4096 4096 64 5 $v
  run_pmc grbm_attn_$v "GRBM_GUI_ACTIVE GRBM_COUNT" attn 24 5 4096 4096 64 5 $v
done
v=${ATTN_MAIN:-v3}
run_pmc fetch_attn_24_5_4096_4096_64 "FETCH_SIZE" attn 24 5 4096 4096 64 5 $v
run_pmc write_attn_24_5_4096_4096_64 "WRITE_SIZE" attn 24 5 4096 4096 64 5 $v
fi
[[ " $SECTIONS " == *" shade "* ]] && for c in ${SHADE_CASES:-fp32 rgb18e8}; do
  f=${DM_SHADE_CASE_DIR:-/tmp}/shade_case_$c.bin
  [ -f $f ] || continue
  run_stats shade_$c shadef $f 10
  run_pmc sq_shade_$c "$SQ" shadef $f 5
  run_pmc sq2_shade_$c "$SQ2" shadef $f 5
  run_pmc fetch_shade_$c "FETCH_SIZE" shadef $f 5
  run_pmc write_shade_$c "WRITE_SIZE" shadef $f 5
  run_pmc tcc_shade_$c "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" shadef $f 5
  run_pmc ta_shade_$c "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" shadef $f 5
  run_pmc tcp_shade_$c "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" shadef $f 5
done
ls $OUT | wc -l
