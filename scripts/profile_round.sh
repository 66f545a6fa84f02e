#!/bin/bash
# usage (on the GPU box): scripts/profile_round.sh <tag>
# three rocprofv3 runs of the default bench workload: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in
# SEPARATE counter passes (MI355X_MICROARCH.md).  Output under gpurun_out/prof_<tag>/; summarise afterwards with
#   scripts/pmc_summary.py <tag> gpurun_out/prof_<tag>/fetch gpurun_out/prof_<tag>/write
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$tag
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/write.log 2>&1
rc=$?
find $O -name "*_agent_info.csv" -delete
ls -R $O | head -40
exit $rc
