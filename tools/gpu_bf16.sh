#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
STG_BWD_BF16=1 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4
for m in 0 1; do
  rm -rf gpurun_out/p3
  STG_BWD_BF16=$m rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p3 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bf16_bench.log 2>&1
  echo "bf16=$m $(grep -o '"value": [0-9.]*' gpurun_out/bf16_bench.log | head -1) $(python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/p3/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'txp_bwd_wave' in r['Name']: print('txp_bwd_wave %.0f us' % (float(r['AverageNs']) / 1e3))
PY
)"
done
