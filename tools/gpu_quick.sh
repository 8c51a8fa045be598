#!/bin/bash
# tests + one profiled bench; prints per-kernel average times
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4
rm -rf gpurun_out/p3
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p3 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/sweep_bench.log 2>&1
echo "$*: $(grep -o '"value": [0-9.]*' gpurun_out/sweep_bench.log | head -1) $(python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/p3/*/*kernel_stats.csv')[0]
out = []
for r in csv.DictReader(open(f)):
    n = r['Name']
    if any(k in n for k in ('txp_', 'model_bwd', 'model_fwd', 'reduce_slabs', 'nll_', 'bn_fold')):
        n = n.replace('void stg::', '').replace('stg::', '').replace('(anonymous namespace)::', '').split('(')[0].replace('_kernel', '')
        out.append('%s %.0f' % (n, float(r['AverageNs']) / 1e3))
print(' | '.join(out))
PY
)"
