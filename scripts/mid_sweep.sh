#!/bin/bash
# 27/31-row windows over the common box sizes: k_compare_fastm (default) against the former choice (k_compare_wide2
# small variants where they applied: BIOEM_MID_WIDE2=1), one bench line each
for n in ${SIZES:-128 160 192 200 224 240 256 320}; do for d in 13 15; do for mode in fastm wide2; do
  if [ $mode = wide2 ]; then export BIOEM_MID_WIDE2=1; else unset BIOEM_MID_WIDE2; fi
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --orientations 288 --pixels $n --max-displacement $d 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=$n +-$d $mode %.2f M/s  %s' % (d['value']/1e6, d['roofline']['kernel']))"
done; done; done
