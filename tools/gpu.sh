#!/bin/bash
# ONE entry point for GPU-box visits (run through gpurun from the repo root):
#   tools/gpu.sh test                      parity tests (-m gpu) + smoke
#   tools/gpu.sh bench [bench.py args]     one bench line
#   tools/gpu.sh prof  [bench.py args]     rocprofv3 --kernel-trace --stats of a short bench run; per-kernel table
#   tools/gpu.sh pmc <counters> [args]     one rocprofv3 --pmc pass (space-separated counters, quoted) over a bench run
#   tools/gpu.sh sweep                     crowd-size / batch-size sweep lines
# logs land under gpurun_out/<tag>.*; TAG=<name> names them (default: the sub-command)
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
cmd=$1; shift
tag=${TAG:-$cmd}
kernel_table() {      # $1 = rocprof output dir
python3 - "$1" <<'PY'
import csv, glob, sys
fs = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
if not fs:
    print("no kernel_stats.csv under", sys.argv[1]); sys.exit(0)
rows = []
for r in csv.DictReader(open(fs[0])):
    n = r['Name'].replace('void stg::', '').replace('stg::', '').replace('(anonymous namespace)::', '').split('(')[0]
    rows.append((float(r['TotalDurationNs']), n, int(r['Calls']), float(r['AverageNs']) / 1e3))
rows.sort(reverse=True)
for tot, n, calls, avg in rows[:16]:
    print('%-48s calls %5d  avg %8.1f us' % (n[:48], calls, avg))
PY
}
case $cmd in
test)
  timeout -k 10 900 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > gpurun_out/$tag.pytest.log 2>&1
  echo "pytest exit $?" >> gpurun_out/$tag.pytest.log; tail -25 gpurun_out/$tag.pytest.log
  timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/$tag.smoke.log 2>&1
  echo "smoke exit $?" >> gpurun_out/$tag.smoke.log; tail -3 gpurun_out/$tag.smoke.log ;;
bench)
  timeout -k 10 900 python bench.py "$@" > gpurun_out/$tag.json.log 2>&1; echo "bench exit $?" >> gpurun_out/$tag.json.log
  tail -3 gpurun_out/$tag.json.log ;;
prof)
  rm -rf gpurun_out/$tag.prof
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag.prof -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/$tag.prof.log 2>&1
  echo "rocprof exit $?" >> gpurun_out/$tag.prof.log
  grep -o '"value": [0-9.e+]*\|"ms_per_step": [0-9.e+]*' gpurun_out/$tag.prof.log | head -2 | tr '\n' ' '; echo
  kernel_table gpurun_out/$tag.prof | tee gpurun_out/$tag.kernels.txt
  find gpurun_out/$tag.prof -name '*kernel_stats.csv' | head -1 | xargs -r -I{} cp {} gpurun_out/$tag.kernel_stats.csv
  find gpurun_out/$tag.prof -type f ! -name '*kernel_stats.csv' -delete ;;
pmc)
  ctr=$1; shift
  rm -rf gpurun_out/$tag.pmc
  timeout -k 10 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/$tag.pmc -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/$tag.pmc.log 2>&1
  echo "rocprof exit $?" >> gpurun_out/$tag.pmc.log
  python3 tools/pmc_sum.py gpurun_out/$tag.pmc | tee gpurun_out/$tag.counters.txt
  rm -rf gpurun_out/$tag.pmc ;;
sweep)
  for args in "--peds 4" "--peds 8" "--peds 16" "--peds 32" "--peds 64" "--peds 128 --batch 4096" "--batch 512" "--batch 128" "--ragged shuffled"; do
    timeout -k 10 600 python bench.py --no-cpu-baseline --no-extras --repeats 8 --steps 20 $args 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$args:', '%.3f M/s  %.3f ms/step' % (d['value']/1e6, d['ms_per_step']))" ; done | tee gpurun_out/$tag.sweep.log ;;
ksweep)
  # per-kernel device times (us) of a list of configurations: tools/gpu.sh ksweep "--peds 64" "--dataset eth-train --batch 512" ...
  for args in "$@"; do
    timeout -k 10 600 python bench.py --no-cpu-baseline --kernels-only --repeats 8 --steps 20 $args 2>gpurun_out/$tag.err | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); ks=d['roofline']['kernels']
print('%-36s %.3f M/s  %.3f ms/step |' % ('$args', d['value']/1e6, d['ms_per_step']), '  '.join('%s %.1f' % (k['kernel'].split(' ')[0].replace('_kernel','').replace('txp_','').replace('stgcn_',''), k['launch_ms']*1e3) for k in ks))" ; done | tee gpurun_out/$tag.ksweep.log ;;
*) echo "usage: tools/gpu.sh test|bench|prof|pmc|sweep|ksweep ..."; exit 2 ;;
esac
