#!/bin/bash
# usage: scripts/sweep_env.sh VAR "v1 v2 ..." [bench args]  -- bench.py once per value of an environment knob
var=$1; vals=$2; shift 2
for v in $vals; do
  env $var=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v value %.3fM  kernel_ms %.3f  frac %.3f' % (d['value']/1e6, d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done
