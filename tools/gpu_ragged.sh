#!/bin/bash
# ragged (eth/train-shaped) batches: shuffled vs sorted by crowd size, 512 and 2048 scenes, + per-kernel times
mkdir -p gpurun_out; export TMPDIR=/tmp; : > gpurun_out/ragged.log
for cfg in "512 shuffled" "512 sorted" "2048 shuffled" "2048 sorted"; do
  set -- $cfg
  timeout -k 10 300 python3 bench.py --batch $1 --ragged $2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('N=$1 $2', round(d['value']), 'windows/s', round(d['ms_per_step'],3), 'ms/step', 'bwd_ms', round(r['launch_ms'],3), 'fwd_ms', round(r['fwd_kernel']['launch_ms'],3), 'e2e_tflops', round(d['end_to_end']['algorithmic_tflops'],1))" >> gpurun_out/ragged.log || exit 1
done
cat gpurun_out/ragged.log
rm -rf gpurun_out/p4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p4 -- python3 bench.py --batch 2048 --ragged shuffled --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
find gpurun_out/p4 -name '*kernel_stats.csv' | head -1 | xargs cut -d, -f1,2,4 | cut -c1-150 | head -12
