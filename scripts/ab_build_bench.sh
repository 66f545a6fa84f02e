#!/bin/bash
# usage: [PIXELS="128 256"] [BENCH_ARGS="--max-displacement 40 ..."] scripts/ab_build_bench.sh "<EXTRA flags A>" "<EXTRA flags B>" ...
# builds each variant ON THE BOX (same device for every arm) and benches it (bench.py, no CPU baseline) at every image
# size in $PIXELS, twice
PIXELS=${PIXELS:-224}
for extra in "$@"; do
  rm -f bioem_amd/lib/libbioem_hip.so
  make -s -C bioem_amd/csrc EXTRA="$extra" all >/dev/null 2>&1 || { echo "build failed: $extra"; continue; }
  for n in $PIXELS; do
    for i in 1 2; do
      python bench.py --steps 2 --warmup 1 --no-cpu-baseline --pixels $n $BENCH_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('VARIANT [$extra] N=$n value %.3fM  kernel_ms %.3f  %s' % (d['value']/1e6, d['roofline']['avg_launch_ms'], d['roofline']['kernel']))"
    done
  done
done
