#!/bin/bash
# throughput over the crowd size V (SURVEY 8d sweep + cfg5 V=128/N=4096); one bench line per V
mkdir -p gpurun_out; : > gpurun_out/vsweep.log
for cfg in "4 2048" "8 2048" "16 2048" "32 2048" "64 2048" "128 4096"; do
  set -- $cfg
  timeout -k 10 300 python3 bench.py --peds $1 --batch $2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('V=$1 N=$2', round(d['value']), 'windows/s', round(d['ms_per_step'],3), 'ms/step', 'bwd_ms', round(r['launch_ms'],3), 'fwd_ms', round(r['fwd_kernel']['launch_ms'],3), 'e2e_tflops', round(d['end_to_end']['algorithmic_tflops'],1), 'e2e_gbs', round(d['end_to_end']['algorithmic_gbs'],1))" >> gpurun_out/vsweep.log || exit 1
done
cat gpurun_out/vsweep.log
