#!/bin/bash
# usage (GPU box): scripts/pmc_extra.sh <tag> [bench flags]  -- one extra counter pass for the instruction / scalar
# caches and LDS of the dominant comparison kernel; prints per-launch means.  BIOEM_HIP_LIBRARY is honoured.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmcx_$tag
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py --no-cpu-baseline $@ --steps 1 --warmup 0"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $O/ic -- $B > $O/ic.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE --output-format csv -d $O/dc -- $B > $O/dc.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys, statistics, collections
for name in ("ic", "dc"):
    fs = glob.glob(sys.argv[1] + "/" + name + "/*/*counter_collection.csv") + glob.glob(sys.argv[1] + "/" + name + "/*counter_collection.csv")
    if not fs:
        print(name, "no counter file"); continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "k_compare" in r["Kernel_Name"]]
    tot = collections.Counter()
    for r in rows:
        tot[r["Kernel_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    k = max(tot, key=tot.get)
    vals = collections.defaultdict(list)
    for r in rows:
        if r["Kernel_Name"] == k:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(name, k[:90])
    for c, v in sorted(vals.items()):
        print("   %-32s %16.0f per launch (%d launches)" % (c, statistics.mean(v), len(v)))
PY
rm -rf $O/ic $O/dc
