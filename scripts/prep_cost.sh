#!/bin/bash
# how a pass divides between the per-orientation preparation (projection, r2c, CTF product, Parseval sums) and the
# comparisons: the same 2 304 orientations x 5 CTFs with 1 .. 200 particles; the 1-particle pass is almost all preparation
for p in 1 5 20 50 100 200; do
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --particles $p --orientations 2304 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('particles $p: %.2f M/s  %.3f ms/pass  launches %d avg %.3f ms' % (d['value']/1e6, d['ms_per_step'], d['roofline']['launches'], d['roofline']['avg_launch_ms']))"
done
