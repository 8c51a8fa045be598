export TMPDIR=/tmp
rm -rf gpurun_out/p4; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p4 -- python3 tools/time_fwd.py 2>&1 | grep -v amdgpu
find gpurun_out/p4 -name '*kernel_trace.csv' | head -1 | xargs python3 -c "
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(list)
for r in rows:
    d[r['Kernel_Name'][:60]].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in d.items():
    if 'stg' in k: print(k, len(v), 'first-half avg %.1f us  second-half avg %.1f us' % (sum(v[:len(v)//2])/max(1,len(v)//2)/1e3, sum(v[len(v)//2:])/max(1,len(v)-len(v)//2)/1e3))
"
