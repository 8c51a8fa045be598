#!/usr/bin/env python3
"""Static instruction statistics per kernel of a hipcc -save-temps .s file (no GPU needed):
   tools/isa_stats.py <file.s> [name-filter ...]
prints, per kernel whose mangled name contains a filter: instruction count, MFMAs, LDS ops, flat / scratch accesses,
s_barrier, full vmcnt(0) waits.  flat_* on what should be LDS or global memory and scratch_* (spills) are the red flags."""
import collections
import re
import sys

path, filters = sys.argv[1], sys.argv[2:]
lines = open(path).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z[\w]+:", l)]
for k, (i, name) in enumerate(starts):
    if filters and not any(f in name for f in filters):
        continue
    end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
    c = collections.Counter()
    vm0 = 0
    for l in lines[i:end]:
        m = re.match(r"\s+([a-z_0-9]+)", l)
        if m:
            c[m.group(1)] += 1
            if m.group(1) == "s_waitcnt" and "vmcnt(0)" in l:
                vm0 += 1
    pre = lambda p: sum(v for op, v in c.items() if op.startswith(p))
    print("%-78s insts %6d mfma %4d ds %5d flat %4d scratch %4d global_ld %4d global_st %4d s_barrier %3d vmcnt(0) %3d"
          % (name[:78], sum(c.values()), pre("v_mfma"), pre("ds_"), pre("flat_"), pre("scratch_"), pre("global_load"),
             pre("global_store"), c["s_barrier"], vm0))
