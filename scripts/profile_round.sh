#!/bin/bash
# usage (on the GPU box): scripts/profile_round.sh <tag> [bench.py flags of the workload, default = BASELINE config 2]
# rocprofv3 runs of ONE bench workload, every counter group in its OWN pass (MI355X_MICROARCH.md: 8 SQ slots, FETCH_SIZE
# and WRITE_SIZE do not fit one pass, never --pmc together with a trace domain other than --kernel-trace):
#   stats  kernel trace + stats (un-perturbed launch durations)
#   sqa    SQ issue counters   : SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES + GRBM_GUI_ACTIVE
#   sqb    SQ activity counters: SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
#   tc     cache counters      : TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
#   fetch  FETCH_SIZE          write  WRITE_SIZE
# Output under gpurun_out/prof_<tag>/; summarise (here or in the container) with
#   python3 scripts/pmc_summary.py <tag> gpurun_out/prof_<tag>
tag=$1
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$tag
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
echo "$@" > $O/bench_flags.txt
B="python3 $R/bench.py --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 2 --warmup 1 > $O/bench_under_rocprof.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sqa -- $B --steps 1 --warmup 0 > $O/sqa.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sqb -- $B --steps 1 --warmup 0 > $O/sqb.log 2>&1 &&
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tc -- $B --steps 1 --warmup 0 > $O/tc.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --steps 1 --warmup 0 > $O/fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --steps 1 --warmup 0 > $O/write.log 2>&1
rc=$?
find $O -name "*_agent_info.csv" -delete
# the per-dispatch traces of the counter passes are large; keep the counter files and the stats
find $O/sqa $O/sqb $O/tc $O/fetch $O/write -name "*_kernel_trace.csv" -delete 2>/dev/null
find $O/stats -name "*_kernel_trace.csv" -delete 2>/dev/null
# counter files: only the rows of the comparison kernels travel back (gpurun merges at most 64 MiB)
for f in $(find $O -name "*_counter_collection.csv"); do
  (head -1 $f; grep k_compare $f) > $f.tmp && mv $f.tmp $f
done
ls -R $O | head -60
exit $rc
