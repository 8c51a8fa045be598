#!/bin/bash
# one GPU-box visit: parity tests + smoke (+ optional extra command), logs under gpurun_out/
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 900 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -60 gpurun_out/pytest_gpu.log
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
tail -5 gpurun_out/smoke.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/bench.log
tail -5 gpurun_out/bench.log
if [ "${PROFILE:-0}" = "1" ]; then
  export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof.log 2>&1
  echo "rocprof exit $?" >> gpurun_out/prof.log
  find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -20
fi
