#!/bin/bash
# HBM traffic of every kernel of one training step (rocprofv3 PMC, separate passes for FETCH_SIZE / WRITE_SIZE)
export TMPDIR=/tmp; mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pm_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pm_$c -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json, re
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pm_%s/*/*counter_collection.csv" % c)[0]
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in d.items():
        m = re.search(r"stg::(?:\(anonymous namespace\)::)?(\w+)", k)
        if m:
            v = v[len(v) // 2:]              # steady state
            out[m.group(1)][c + "_KB_per_launch"] = sum(v) / len(v)
for k, v in out.items():
    print(k, {a: round(b, 1) for a, b in v.items()})
# corrected HBM bytes per launch: FETCH_SIZE doubled (gfx950, wide coalesced reads; MI355X_MICROARCH.md), WRITE_SIZE as is
for k, v in out.items():
    v["hbm_bytes_per_launch"] = int(1024 * (2 * v.get("FETCH_SIZE_KB_per_launch", 0) + v.get("WRITE_SIZE_KB_per_launch", 0)))
json.dump(out, open("gpurun_out/traffic_raw.json", "w"), indent=1)
PY
# stats run for the same configuration (graph replay), kept as the round's profile
rm -rf gpurun_out/prof_final
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_final_bench.log 2>&1
find gpurun_out/prof_final -name '*kernel_stats.csv' | head -1 | xargs cut -c1-120 | head -14
