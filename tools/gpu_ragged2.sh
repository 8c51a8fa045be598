#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2
for walk in 0 1; do
  rm -rf gpurun_out/p4
  STG_WALK=$walk rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p4 -- python3 bench.py --batch 2048 --ragged shuffled --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/rag.log 2>&1
  echo "walk=$walk $(grep -o '"value": [0-9.]*' gpurun_out/rag.log | head -1) $(python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/p4/*/*kernel_stats.csv')[0]
out = []
for r in csv.DictReader(open(f)):
    n = r['Name']
    if any(k in n for k in ('txp_', 'model_bwd', 'model_fwd', 'scene_order')):
        n = n.replace('void stg::', '').replace('stg::', '').replace('(anonymous namespace)::', '').split('(')[0].replace('_kernel', '')
        out.append('%s %.0f' % (n, float(r['AverageNs']) / 1e3))
print(' | '.join(out))
PY
)"
done
