#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/p7
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p7 -- python3 bench.py --peds 128 --batch 4096 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/v128.log 2>&1
echo "$*: $(grep -o '"value": [0-9.]*' gpurun_out/v128.log | head -1) $(python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/p7/*/*kernel_stats.csv')[0]
out = []
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'stg' in n:
        n = n.replace('void stg::', '').replace('stg::', '').replace('(anonymous namespace)::', '').split('(')[0].replace('_kernel', '')
        out.append('%s %.0f' % (n, float(r['AverageNs']) / 1e3))
print(' | '.join(out))
PY
)"
