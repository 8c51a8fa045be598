#!/bin/bash
# tests + one profiled bench; prints per-kernel average times
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4
rm -rf gpurun_out/p3
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p3 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/sweep_bench.log 2>&1
echo "$*: $(grep -o '"value": [0-9.]*' gpurun_out/sweep_bench.log | head -1) $(find gpurun_out/p3 -name '*kernel_stats.csv' | head -1 | xargs grep -h 'txp_\|model_bwd\|model_fwd\|reduce_slabs' | sed 's/void stg:://; s/(stg::[A-Za-z]*)//; s/_kernel//; s/(anonymous namespace):://' | cut -d, -f1,4 | sed 's/\.[0-9]*$//' | tr '\n' ' ')"
