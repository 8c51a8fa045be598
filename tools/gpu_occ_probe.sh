#!/bin/bash
# F2 time per pedestrian across the 8 -> 4 waves/CU LDS boundary (V = 22 | 24)
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 20 22 24 26 28; do
  rm -rf gpurun_out/p6
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p6 -- python3 bench.py --peds $v --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  echo "V=$v $(python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/p6/*/*kernel_stats.csv')[0]
out = []
for r in csv.DictReader(open(f)):
    n = r['Name']
    if any(k in n for k in ('txp_', 'model_bwd', 'model_fwd')):
        n = n.replace('void stg::', '').replace('stg::', '').replace('(anonymous namespace)::', '').split('(')[0].replace('_kernel', '')
        out.append('%s %.0f' % (n, float(r['AverageNs']) / 1e3))
print(' | '.join(sorted(out)))
PY
)"
done
