#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  rm -rf gpurun_out/p3
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p3 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/sweep_bench.log 2>&1
  echo "$*: $(grep -o '"value": [0-9.]*' gpurun_out/sweep_bench.log | head -1) $(find gpurun_out/p3 -name '*kernel_stats.csv' | head -1 | xargs grep -h 'txp_\|model_bwd\|model_fwd\|reduce_slabs' | sed 's/void stg:://; s/(stg::[A-Za-z]*)//; s/_kernel//' | cut -d, -f1,4 | sed 's/\.[0-9]*$//' | tr '\n' ' ')"
}
run STG_TXP_WPB=2
run STG_TXP_WPB=1
run STG_TXP_WPB=4
run STG_TXP_WPB=2 STG_FWD_WAVES=2
run STG_TXP_WPB=2 STG_BWD_WAVES=1
run STG_TXP_WPB=2 STG_BWD_WAVES=4
run STG_NO_WAVE_PATH=1
