#!/bin/bash
# the few-particle regime (BASELINE config 1: 10 particles): whole-job rate with the batch grown to ~320 000 pairs per
# launch against the fixed 64-orientation batch (BIOEM_FIXED_BATCH=1)
for w in "--particles 20 --orientations 2304" "--particles 10 --orientations 4608" "--pixels 128 --particles 10 --orientations 576 --envelopes 4" "--pixels 128 --particles 10 --orientations 4608 --envelopes 4" "--particles 100 --orientations 2304"; do for mode in adaptive fixed; do
  if [ $mode = fixed ]; then export BIOEM_FIXED_BATCH=1; else unset BIOEM_FIXED_BATCH; fi
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline $w 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$w] $mode %.2f M/s  %.3f ms/pass  launches %d  %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['launches'], d['roofline']['kernel']))"
done; done
