#!/bin/bash
# A/B on ONE GPU box: the default (bf16 matrix pipe, exact three-piece operands) against STG_OPT_F32_MFMA (fp32 MFMA
# kernels), same bench command, rocprofv3 kernel table of each.  Output: gpurun_out/ab_mfma.log
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
{
for opt in "" "--f32-mfma"; do
  echo "== bench.py $opt"
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras $opt 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['timing']; print('%.3f M scene-windows/s  %.4f ms/step (p10 %.4f, p90 %.4f)' % (d['value']/1e6, d['ms_per_step'], t['ms_per_step_p10'], t['ms_per_step_p90']))"
  TAG=ab tools/gpu.sh prof --no-extras $opt 2>&1 | grep -E "txp_|stgcn_|reduce_"
done
} > gpurun_out/ab_mfma.log 2>&1
cat gpurun_out/ab_mfma.log
