#!/bin/bash
# the few-particle regime, A/B on one box: VAR=value pairs given as arguments are toggled against the default build,
# e.g. scripts/small_jobs_ab.sh BIOEM_R2C=dft, or BIOEM_HIP_LIBRARY=<another build>.  Every workload runs default, variant,
# default, variant (the first run of a workload on a box is 2-3 % slower than a repeat, whichever build it is).
run() { python bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f M/s %.3f ms' % (d['value']/1e6, d['ms_per_step']))"; }
WL=${WORKLOADS:-"--particles 20 --orientations 2304|--particles 10 --orientations 4608|--pixels 128 --particles 10 --orientations 576 --envelopes 4|--pixels 128 --particles 10 --orientations 4608 --envelopes 4|--particles 100 --orientations 2304|--particles 1000 --orientations 1152"}
IFS='|' read -ra W <<< "$WL"
for w in "${W[@]}"; do
  for rep in 1 2; do
    echo "[$w] default: $(run $w)"
    for kv in "$@"; do echo "[$w] $kv: $(env $kv bash -c "$(declare -f run); run $w")"; done
  done
done
