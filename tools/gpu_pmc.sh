export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 -L 2>/dev/null | grep -oE "(SQ|TCC|GRBM|TCP)_[A-Z0-9_]+" | sort -u > gpurun_out/counters.txt
wc -l gpurun_out/counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  rm -rf gpurun_out/p5
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/p5 -- python3 tools/time_fwd.py > /dev/null 2>&1
  f=$(find gpurun_out/p5 -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r['Kernel_Name']
    if 'txp_fwd_wave' in k or 'model_fwd' in k:
        d[k[:50]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in d.items():
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
done
