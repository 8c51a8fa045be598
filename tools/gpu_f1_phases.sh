#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for skip in 0 32; do
  rm -rf gpurun_out/p3
  STG_DEBUG_SKIP=$skip rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p3 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  echo "skip=$skip $(python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/p3/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'model_fwd' in r['Name']: print('model_fwd %.0f us' % (float(r['AverageNs']) / 1e3))
PY
)"
done
