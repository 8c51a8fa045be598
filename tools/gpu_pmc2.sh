export TMPDIR=/tmp
mkdir -p gpurun_out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT"; do
  rm -rf gpurun_out/p6
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/p6 -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-graph > /dev/null 2>&1
  f=$(find gpurun_out/p6 -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void stg::','').split('(')[0]
    if any(x in k for x in ('model_bwd','model_fwd','txp_')):
        d[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in d.items():
    print(k, {c: round(sum(x[len(x)//2:])/len(x[len(x)//2:])) for c,x in v.items()})
PY
done
