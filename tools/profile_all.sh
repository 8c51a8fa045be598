#!/bin/bash
# Regenerates EVERY file under profiles/ for the current HEAD in one GPU-box visit (run through gpurun from the repo
# root; the results land in gpurun_out/profiles_r03/ -- copy them to profiles/ afterwards):
#   r03_bench.json.log                 default bench line (BASELINE north-star workload) incl. roofline / cpu_baseline
#   r03_bench_eth_train_512.json.log   BASELINE configs[1]: real eth/train windows, batch 512 (+ device-resident epochs)
#   r03_bench_bf16.json.log            BASELINE configs[2]: bf16 storage
#   r03_bench_all_train_2048[_bf16].json.log   BASELINE configs[2]: the five train sets concatenated, batch 2048, fp32 / bf16 storage
#   r03_bench_v64 / r03_bench_v128_4096.json.log   team kernels: V = 64 x 2048, BASELINE configs[4] (V = 128 x 4096)
#   r03_team_<cfg>_{kernel_stats.csv,pmc_traffic.json,sq_counters.log}   the same three profiles for eth/train x 512 and V = 128 x 4096
#   r03_bench_gloo2.json.log           two ranks on one GPU over gloo (rehearsal of the multi-rank path)
#   r03_sweep.log                      crowd-size / batch-size sweep
#   r03_kernel_stats.csv               rocprofv3 --kernel-trace --stats of the default bench command
#   r03_pmc_traffic.json               HBM bytes per launch per kernel: --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE
#                                      passes, FETCH_SIZE doubled (gfx950 wide-load correction, MI355X_MICROARCH.md)
#   r03_sq_counters.log                SQ wave / wait / MFMA / LDS counters per kernel (two --pmc passes)
set -o pipefail
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
OUT=gpurun_out/profiles_r03
rm -rf $OUT; mkdir -p $OUT
run() { timeout -k 10 600 "$@"; }
# kernel stats / HBM traffic (separate PMC passes over eager steps, counters are per dispatch) / SQ counters of a configuration
profile_cfg() {    # $1 = file prefix, rest = bench.py arguments
  pre=$1; shift
  rm -rf $OUT/ks
  run rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -- python3 bench.py --no-cpu-baseline --no-extras "$@" > /dev/null 2>&1
  find $OUT/ks -name '*kernel_stats.csv' | head -1 | xargs -r -I{} cp {} $OUT/${pre}kernel_stats.csv; rm -rf $OUT/ks
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/pm_$c
    run rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pm_$c -- python3 bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras --no-graph "$@" > /dev/null 2>&1
  done
  python3 - "$OUT" "${pre}pmc_traffic.json" <<'PY'
import csv, glob, collections, json, re, subprocess, sys
out_dir, name = sys.argv[1], sys.argv[2]
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(out_dir + "/pm_%s/**/*counter_collection.csv" % c, recursive=True)
    if not fs:
        continue
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] == c:
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in d.items():
        m = re.search(r"stg::(?:\(anonymous namespace\)::)?(\w+)", k)
        if m:
            v = v[len(v) // 2:]              # steady state
            out[m.group(1)][c + "_KB_per_launch"] = sum(v) / len(v)
for k, v in out.items():
    # corrected HBM bytes per launch: FETCH_SIZE doubled (gfx950, wide coalesced reads), WRITE_SIZE as is
    v["hbm_bytes_per_launch"] = int(1024 * (2 * v.get("FETCH_SIZE_KB_per_launch", 0) + v.get("WRITE_SIZE_KB_per_launch", 0)))
sys.path.insert(0, ".")
import bench
out["_meta"] = {"kernel_source_sha1": bench.kernel_source_hash(),
                "note": "sha1 over csrc/*.hip, *.hpp and include/*.h at measurement time: bench.py quotes this file as "
                        "roofline.traffic only while the sources still hash to it"}
json.dump(out, open(out_dir + "/" + name, "w"), indent=1, sort_keys=True)
print("traffic kernels:", sorted(out))
PY
  rm -rf $OUT/pm_FETCH_SIZE $OUT/pm_WRITE_SIZE
  : > $OUT/${pre}sq_counters.log
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
    rm -rf $OUT/p5
    run rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p5 -- python3 bench.py --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras --no-graph "$@" > /dev/null 2>&1
    python3 tools/pmc_sum.py $OUT/p5 | grep -E "txp_|stgcn_|nll_|reduce_|model_" >> $OUT/${pre}sq_counters.log
    rm -rf $OUT/p5
  done
}
profile_cfg r03_
# the bench lines quote roofline.traffic from profiles/r03_pmc_traffic.json while it matches the kernel sources: this visit's file first
cp $OUT/r03_pmc_traffic.json profiles/r03_pmc_traffic.json
run python bench.py > $OUT/r03_bench.json.log 2>/dev/null; echo "bench $?"
run python bench.py --dataset eth-train --batch 512 --no-cpu-baseline > $OUT/r03_bench_eth_train_512.json.log 2>/dev/null; echo "eth $?"
run python bench.py --dtype bf16 --no-cpu-baseline > $OUT/r03_bench_bf16.json.log 2>/dev/null; echo "bf16 $?"
run python bench.py --dataset all-train --batch 2048 --no-cpu-baseline > $OUT/r03_bench_all_train_2048.json.log 2>/dev/null; echo "all-train $?"
run python bench.py --dataset all-train --batch 2048 --dtype bf16 --no-cpu-baseline > $OUT/r03_bench_all_train_2048_bf16.json.log 2>/dev/null; echo "all-train bf16 $?"
run python bench.py --peds 64 --no-cpu-baseline --kernels-only > $OUT/r03_bench_v64.json.log 2>/dev/null; echo "v64 $?"
run python bench.py --peds 128 --batch 4096 --no-cpu-baseline --kernels-only > $OUT/r03_bench_v128_4096.json.log 2>/dev/null; echo "v128 $?"
STG_DIST_BACKEND=gloo run python bench.py --gpus 2 --no-cpu-baseline --no-extras --steps 20 --repeats 10 > $OUT/r03_bench_gloo2.json.log 2>/dev/null; echo "gloo2 $?"
TAG=profiles_sweep tools/gpu.sh sweep > /dev/null; cp gpurun_out/profiles_sweep.sweep.log $OUT/r03_sweep.log
profile_cfg r03_team_eth512_ --dataset eth-train --batch 512
profile_cfg r03_team_v128_4096_ --peds 128 --batch 4096
ls -la $OUT
