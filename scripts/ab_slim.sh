#!/bin/bash
# Same-box A/B of experiment libraries abl/<variant>.so (scripts/slim_build.sh): for every bench shape in SHAPES (one
# per line; default: the tutorial production set) a parity check of the shape against the CPU oracle (small counts) and
# two short bench runs per variant.
#   VARIANTS="base new" [SHAPES=$'--max-displacement 40 --envelopes 4 --defocus 8\n--max-displacement 13'] scripts/ab_slim.sh
SHAPES=${SHAPES:-"--max-displacement 40 --envelopes 4 --defocus 8"}
REPS=${REPS:-2}
while IFS= read -r w; do
  [ -z "$w" ] && continue
  for v in $VARIANTS; do
    BIOEM_HIP_LIBRARY=abl/$v.so python scripts/parity_shape.py $w 2>&1 | tail -1 | sed "s/^/$v [$w] /"
    for i in $(seq $REPS); do
      BIOEM_HIP_LIBRARY=abl/$v.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --orientations ${ORIENT:-288} $w 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v [$w] %.2f M/s  kernel %.3f ms  %s' % (d['value']/1e6, d['roofline']['avg_launch_ms'], d['roofline']['kernel']))"
    done
  done
done <<< "$SHAPES"
