#!/bin/bash
# eth/train x 512 (BASELINE configs[1]) on the wave-per-scene path (default there: V = 57 > 40) against the
# workgroup-per-scene kernels with 2 / 4 / 8 waves per scene
for args in "" "--wg-path" "--wg-path --wg-waves 2" "--wg-path --wg-waves 4" "--wg-path --wg-waves 8"; do
  timeout -k 10 300 python bench.py --dataset eth-train --batch 512 --no-cpu-baseline --no-extras --repeats 8 --steps 20 $args 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$args:', '%.3f M/s  %.3f ms/step' % (d['value']/1e6, d['ms_per_step']))" ; done
