#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for skip in 0 64 128 192; do
  for wv in 4; do
  rm -rf gpurun_out/p2
  STG_DEBUG_SKIP=$skip STG_WGRAD_WAVES=$wv rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p2 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  echo "skip=$skip waves=$wv: $(find gpurun_out/p2 -name '*kernel_stats.csv' | head -1 | xargs grep -h 'txp_wgrad\|model_bwd\|model_fwd' | cut -d, -f1,4 | tr '\n' ' ')"
  done
done
