R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
for v in 8 1; do
  rm -rf /tmp/prof_final_$v
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final_$v -- python $R/bench.py --views $v --steps 4 --warmup 2 --no-cpu-baseline --no-second-leg > $R/gpurun_out/final_rocprof_$v.log 2>&1 < /dev/null)
  f=$(find /tmp/prof_final_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/final_kernel_stats_${v}views.csv
  f=$(find /tmp/prof_final_$v -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 tools/step_window.py "$f" 3 6 > gpurun_out/final_step_kernels_${v}views.csv
  head -1 gpurun_out/final_step_kernels_${v}views.csv | cut -c40-160
done
for v in 8 4 2 1; do timeout 200 python bench.py --views $v --steps 6 --warmup 2 --no-cpu-baseline --no-second-leg 2>/dev/null < /dev/null | grep '^{"metric' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'views': $v, 'steps_per_s': d['value'], 'ms_per_step': d['ms_per_step']}))"; done > gpurun_out/final_views_table.jsonl
cat gpurun_out/final_views_table.jsonl
timeout 300 python bench.py --no-cpu-baseline --no-second-leg --no-calibration --dump-kernels /tmp/kernels.json > /dev/null 2>&1 < /dev/null && python tools/kernel_dump_table.py /tmp/kernels.json > gpurun_out/final_kernel_event_table.txt
PYTHONPATH=$R timeout 200 python tools/small_conv_time.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/final_small_conv_time.txt
for m in hip blas hip blas; do DREAMMAT_VAE_ATTENTION=$m timeout 300 python bench.py --no-cpu-baseline --no-second-leg --no-calibration 2>/dev/null < /dev/null | grep '^{"metric' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', round(d['value'],3), round(d['ms_per_step'],2))"; done > gpurun_out/final_vae_attention_ab.txt
cat gpurun_out/final_vae_attention_ab.txt
