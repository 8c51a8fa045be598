"""Sum rocprofv3 --pmc counter values per kernel name (one pass directory)."""
import csv
import glob
import sys
from collections import defaultdict

fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not fs:
    print("no counter_collection.csv under", sys.argv[1])
    sys.exit(0)
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for r in csv.DictReader(open(fs[0])):
    n = r["Kernel_Name"].replace("void stg::", "").replace("stg::", "").replace("(anonymous namespace)::", "").split("(")[0]
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[n].add(r["Dispatch_Id"])
for n in sorted(acc):
    c = max(1, len(calls[n]))
    print(n, "calls", c, {k: round(v / c, 1) for k, v in sorted(acc[n].items())})
