#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for w in 4 8; do
  rm -rf gpurun_out/p3
  STG_TXP_WPB=$w rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p3 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  echo "wpb=$w $(python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/p3/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'txp_fwd_wave' in r['Name'] or 'txp_bwd_wave' in r['Name']: print(r['Name'][35:62], '%.0f us' % (float(r['AverageNs']) / 1e3), end=' | ')
PY
)"
done
