#!/bin/bash
# SQ counters of every kernel of one training step (eager launches), two passes
export TMPDIR=/tmp; mkdir -p gpurun_out; : > gpurun_out/pmc_step.log
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  rm -rf gpurun_out/p5
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/p5 -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-graph "$@" > /dev/null 2>&1
  f=$(find gpurun_out/p5 -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY' >> gpurun_out/pmc_step.log
import csv,sys,collections,re
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    m = re.search(r"stg::(?:\(anonymous namespace\)::)?(\w+(?:<\d+>)?)", r['Kernel_Name'])
    if m and any(k in m.group(1) for k in ('txp_', 'model_')):
        d[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in d.items():
    print(k, {c: round(sum(x[len(x)//2:])/len(x[len(x)//2:])) for c,x in v.items()})
PY
done
cat gpurun_out/pmc_step.log
