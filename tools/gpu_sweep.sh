#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for fw in 1 2 4; do for bw in 1 2 4; do
  rm -rf gpurun_out/p3
  STG_FWD_WAVES=$fw STG_BWD_WAVES=$bw rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p3 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  echo "fwd_waves=$fw bwd_waves=$bw: $(find gpurun_out/p3 -name '*kernel_stats.csv' | head -1 | xargs grep -h 'txp_wgrad\|model_bwd\|model_fwd' | sed 's/void stg:://; s/(stg::[A-Za-z]*)//' | cut -d, -f1,4 | tr '\n' ' ')"
done; done
