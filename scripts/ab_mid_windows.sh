#!/bin/bash
# usage (GPU box): scripts/ab_mid_windows.sh -- 27/31-row windows at mixed-radix image sizes, register-FFT length <= 16 (default)
# against the longest one (BIOEM_WIDE_R32=1)
for cfg in "180 15" "180 12" "150 15" "150 12" "100 15" "100 12" "250 15" "250 12" "300 15" "300 12" "90 15" "60 12" "210 15" "270 12"; do set -- $cfg
for e in "" "BIOEM_WIDE_R32=1"; do
env $e python bench.py --pixels $1 --max-displacement $2 --orientations 288 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 +-$2 $e', round(d['value']/1e6,2), d['roofline']['kernel'])"
done; done
