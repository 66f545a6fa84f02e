#!/bin/bash
# usage: scripts/bench_variants.sh "<bench args 1>" "<bench args 2>" ...   -- one bench.py line per argument set
for args in "$@"; do
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$args] value %.3fM  kernel_ms %.3f  ms_per_step %.1f kernel %s' % (d['value']/1e6, d['roofline']['avg_launch_ms'], d['ms_per_step'], d['roofline']['kernel']))"
done
