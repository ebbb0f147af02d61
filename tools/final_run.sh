#!/bin/bash
# Round-end validation on the GPU box: full GPU test suite, the default bench line, rocprofv3 kernel-trace stats of
# the same bench command, conv hardware counters, per-kernel micro-benchmarks.  Writes only small files to gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q > gpurun_out/final_pytest_full.log 2>&1
rc=$?
tail -5 gpurun_out/final_pytest_full.log > gpurun_out/final_pytest.log
tail -2 gpurun_out/final_pytest.log
if [ $rc -ne 0 ]; then echo "GPU TESTS FAILED (rc=$rc): skipping the measurement passes"; grep -E "^(FAILED|ERROR)" gpurun_out/final_pytest_full.log | head; exit 1; fi
timeout 400 python bench.py > gpurun_out/final_bench.log 2>&1 < /dev/null
grep '^{"metric' gpurun_out/final_bench.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_final
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/final_rocprof.log 2>&1 < /dev/null
f=$(find /tmp/prof_final -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $R/gpurun_out/final_kernel_stats.csv
cd $R
bash tools/pmc_abi.sh > /dev/null 2>&1
if [ "${SKIP_KERNEL_BENCH:-0}" != "1" ]; then timeout 200 python tools/kernel_bench.py --iters 10 > gpurun_out/final_kernel_bench.log 2>&1 < /dev/null; fi
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
