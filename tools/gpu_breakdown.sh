#!/bin/bash
# timing-only phase breakdown of the fused kernels (results are wrong under STG_DEBUG_SKIP)
mkdir -p gpurun_out
for skip in 0 1 2 3 4 7 16 32 48; do
  echo "== STG_DEBUG_SKIP=$skip" >> gpurun_out/breakdown.log
  STG_DEBUG_SKIP=$skip timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('value %.0f  ms/step %.3f  bwd_ms %.3f  fwd_ms %.3f  adj %.3f ms  agg %.3f ms' % (d['value'], d['ms_per_step'], r['launch_ms'], r['fwd_kernel']['launch_ms'], d['kernels']['adj_build']['ms'], d['kernels']['spatial_agg_fwd']['ms']))" >> gpurun_out/breakdown.log 2>&1
done
cat gpurun_out/breakdown.log
