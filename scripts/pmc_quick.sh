#!/bin/bash
# usage (GPU box): scripts/pmc_quick.sh <tag> [bench flags] -- the SQ issue / activity passes of ONE bench shape and a
# per-comparison table (quad-cycle units for the *_CYCLES / ACTIVE / WAIT counters).  BIOEM_HIP_LIBRARY is honoured;
# PMC_KERNEL=<substring> picks another kernel than the dominant k_compare_* one (e.g. k_project_bands).
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmcq_$tag
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py --no-cpu-baseline $@ --steps 1 --warmup 0"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sqa -- $B > $O/sqa.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sqb -- $B > $O/sqb.log 2>&1
python3 - $O "$tag" <<'PY'
import csv, glob, os, sys, statistics, collections
KF = os.environ.get("PMC_KERNEL", "k_compare")
allv = {}
kname, nl, dur = None, 0, 0
for name in ("sqa", "sqb"):
    fs = glob.glob(sys.argv[1] + "/" + name + "/*/*counter_collection.csv") + glob.glob(sys.argv[1] + "/" + name + "/*counter_collection.csv")
    if not fs:
        print(name, "no counter file"); continue
    rows = [r for r in csv.DictReader(open(fs[0])) if KF in r["Kernel_Name"]]
    tot = collections.Counter()
    for r in rows:
        tot[r["Kernel_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    k = max(tot, key=tot.get)
    kname = k
    vals = collections.defaultdict(list)
    d = {}
    for r in rows:
        if r["Kernel_Name"] == k:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
            d[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size"]), int(r["Workgroup_Size"]))
    for c, v in vals.items():
        allv[c] = statistics.mean(v)
    dur = statistics.mean(x[0] for x in d.values()) / 1e6
    grid, wg = list(d.values())[0][1:]
w = allv.get("SQ_WAVES", 1)
print(sys.argv[2], kname[:100])
print("   launch %.3f ms under PMC, %d waves per launch, clock %.0f MHz" % (dur, w, allv.get("GRBM_GUI_ACTIVE", 0) / 8 / (dur / 1e3) / 1e6))
for c in sorted(allv):
    print("   %-24s %12.1f per wave" % (c, allv[c] / w))
if "SQ_WAVE_CYCLES" in allv:
    wc = allv["SQ_WAVE_CYCLES"]
    for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
        if c in allv:
            print("   %-24s %5.1f %% of wave cycles" % (c, 100 * allv[c] / wc))
    print("   VALU issue utilisation %.3f of 1024 SIMDs x clock / 2" % (allv["SQ_INSTS_VALU"] / (dur / 1e3) / (1024 * (allv.get("GRBM_GUI_ACTIVE", 0) / 8 / (dur / 1e3)) / 2)))
PY
rm -rf $O/sqa $O/sqb
