R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
for v in 4 2; do
  rm -rf /tmp/prof_v_$v
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v_$v -- python $R/bench.py --views $v --steps 4 --warmup 2 --no-cpu-baseline --no-second-leg --no-calibration > /dev/null 2>&1 < /dev/null)
  f=$(find /tmp/prof_v_$v -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 tools/step_window.py "$f" 3 6 > gpurun_out/step_kernels_${v}views.csv
  head -1 gpurun_out/step_kernels_${v}views.csv | cut -c40-160
done
