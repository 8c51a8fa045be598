#!/bin/bash
# Small batches (run through gpurun from the repo root): default path choice (workgroup-per-scene kernels up to 512
# scenes) against the wave-per-scene kernels (--wave-path), then rocprofv3 kernel tables of both at 128 scenes -- the
# wave-path table is the latency of ONE scene on a lone wave (profiles/r02_small_batch.log).
for args in "--batch 128" "--batch 128 --wave-path" "--batch 256" "--batch 256 --wave-path" "--batch 512" "--batch 512 --wave-path" "--batch 1024"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --repeats 8 --steps 20 $args 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$args:', '%.3f M/s  %.3f ms/step' % (d['value']/1e6, d['ms_per_step']))" ; done
TAG=p128 tools/gpu.sh prof --batch 128 --wave-path --no-extras
TAG=p128wg tools/gpu.sh prof --batch 128 --no-extras
